#!/bin/bash
# STFT wave kernel: 12 waves per workgroup at n_fft 2048, the n_fft 4096 instantiation in 4-wave workgroups (AFX_STFT4K=1)
set -u
TAG=${1:-r05w}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
for r in 1 2 3; do
  echo "shipped:"; timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 2
  echo "stft12:"; AFX_LIB=$V/libafx_stft12.so timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 2
  echo "stft12 + 4k:"; AFX_STFT4K=1 AFX_LIB=$V/libafx_stft12.so timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 2
done | tee $OUT/stft.txt
AFX_STFT4K=1 AFX_LIB=$V/libafx_stft12.so timeout -k 10 600 python -m pytest tests/test_stft_gpu.py tests/test_reassign_gpu.py -q -m gpu -x 2>&1 | tail -n 3 | tee $OUT/pytest_tail.txt
