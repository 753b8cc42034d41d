#!/bin/bash
# round 5: scalar wave index (readfirstlane) in the wave kernels against the previous build: suite + cepstrogram / n_fft 4096 / 1024 / stft / cfg 4
set -u
TAG=${1:-r05k}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
V=$PWD/audioflux_amd/lib/variants
(time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 1200 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -20
for r in 1 2; do
  for n in shipped prev; do
    L=""; [ "$n" != shipped ] && L="AFX_LIB=$V/libafx_$n.so"
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1)" | tee -a $OUT/other.txt
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 10 256 2>&1 | tail -n 1)" | tee -a $OUT/other.txt
    env $L timeout -k 10 120 python tools/bench_cepstrogram.py 2>&1 | tail -n 2 | sed "s/^/[$n] /" | tee -a $OUT/other.txt
    env $L timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 3 | sed "s/^/[$n] /" | tee -a $OUT/other.txt
  done
done
