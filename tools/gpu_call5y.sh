#!/bin/bash
# n_fft 4096 STFT on the bank kernel's transform (afxk_stft4k): parity files, then rates against the previous library
set -u
TAG=${1:-r05y}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_stft_gpu.py tests/test_bft_gpu.py tests/test_reassign_gpu.py tests/test_spectrogram_gpu.py -q -m gpu -x 2>&1 | tail -n 15 | tee $OUT/pytest_tail.txt
for r in 1 2 3; do
  echo "shipped:"; timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 2
  echo "prev:"; AFX_LIB=$V/libafx_prev.so timeout -k 10 120 python tools/bench_stft.py 2>&1 | tail -n 2
done | tee $OUT/stft.txt
