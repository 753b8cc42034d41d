#!/bin/bash
# round 5, second call: the suite on the staged host copies + k_stft_band_4k2, n_fft 4096 rates (shipped vs variants in
# audioflux_amd/lib/variants), the legacy one-clip protocol and the host-pointer batch path.
#   gpurun --timeout 1500 -- 'bash tools/gpu_call5b.sh r05b [nosuite] [variant ...]'
set -u
TAG=${1:-r05b}; shift
NOSUITE=${1:-}; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
if [ "$NOSUITE" != "nosuite" ]; then
  rm -f $OUT/parity.jsonl
  (time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 1000 python -m pytest tests -q -m gpu -x) > $OUT/pytest.log 2>&1
  echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
  grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -40
  tail -n 30 $OUT/pytest.log | grep -aE "Error|assert" | head
fi
V=$PWD/audioflux_amd/lib/variants
for r in 1 2; do
  for n in shipped "$@"; do
    L=""; [ "$n" != shipped ] && L="AFX_LIB=$V/libafx_$n.so"
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1)" | tee -a $OUT/nfft4096.txt
  done
done
echo "[shipped hop 900] $(timeout -k 10 120 python tools/bench_nfft.py 12 900 2>&1 | tail -n 1)" | tee -a $OUT/nfft4096.txt
echo "[shipped n_fft 2048] $(timeout -k 10 120 python tools/bench_nfft.py 11 512 2>&1 | tail -n 1)" | tee -a $OUT/nfft4096.txt
for s in 0 1; do
  E=""; [ $s = 1 ] && E="AFX_NO_STAGING=1"
  echo "[legacy $E] $(env $E timeout -k 10 240 python tools/legacy_bench.py 1000 2>&1 | tail -n 1 | cut -c1-600)" | tee -a $OUT/legacy.txt
done
for s in 0 1; do
  E=""; [ $s = 1 ] && E="AFX_NO_STAGING=1"
  echo "--- hostabi $E" | tee -a $OUT/hostabi.txt
  env $E timeout -k 10 240 python tools/bench_hostabi.py 2>&1 | tail -n 5 | tee -a $OUT/hostabi.txt
done
