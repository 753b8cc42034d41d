#!/bin/bash
# STFT object on the bank kernels' transforms at n_fft 512 / 1024 (afxk_stft512 / afxk_stft1k): parity files + rates against the previous library
set -u
TAG=${1:-r05aa}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_stft_gpu.py tests/test_bft_gpu.py tests/test_reassign_gpu.py tests/test_spectrogram_gpu.py -q -m gpu -x 2>&1 | tail -n 12 | tee $OUT/pytest_tail.txt
for r in 1 2; do
  echo "shipped:"; timeout -k 10 200 python tools/bench_stft_sizes.py 2>&1 | grep "n_fft"
  echo "prev:"; AFX_LIB=$V/libafx_prev.so timeout -k 10 200 python tools/bench_stft_sizes.py 2>&1 | grep "n_fft"
done | tee $OUT/stft_sizes.txt
