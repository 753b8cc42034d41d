#!/bin/bash
# round 5: in-place register images (shift_rows_inplace / rows_shift_fetch) against the previous build (variants/libafx_prev.so)
set -u
TAG=${1:-r05i}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 600 python -m pytest tests/test_bft_gpu.py tests/test_fullsize_gpu.py tests/test_spectrogram_gpu.py tests/test_xxcc_gpu.py tests/test_batch_gpu.py tests/test_realaudio_gpu.py -q -m gpu -x 2>&1 | tail -n 4 | tee $OUT/pytest_tail.txt
bash tools/gpu_ab_lib.sh $TAG 3 prev
for r in 1 2 3; do
  for n in shipped prev; do
    L=""; [ "$n" != shipped ] && L="AFX_LIB=$V/libafx_$n.so"
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1)" | tee -a $OUT/nfft.txt
  done
done
echo "[shipped hop 900] $(timeout -k 10 120 python tools/bench_nfft.py 12 900 2>&1 | tail -n 1)" | tee -a $OUT/nfft.txt
echo "[shipped 2048/512] $(timeout -k 10 120 python tools/bench_nfft.py 11 512 2>&1 | tail -n 1)" | tee -a $OUT/nfft.txt
echo "[prev 2048/512] $(AFX_LIB=$V/libafx_prev.so timeout -k 10 120 python tools/bench_nfft.py 11 512 2>&1 | tail -n 1)" | tee -a $OUT/nfft.txt
