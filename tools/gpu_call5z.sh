#!/bin/bash
# n_fft 512 bank kernel (k_stft_band_512): parity files, then rates against the previous library (size-generic kernel there)
set -u
TAG=${1:-r05z1}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_bft_gpu.py tests/test_batch_gpu.py tests/test_spectrogram_gpu.py tests/test_realaudio_gpu.py -q -m gpu -x 2>&1 | tail -n 12 | tee $OUT/pytest_tail.txt
for r in 1 2 3; do
  echo "shipped: $(timeout -k 10 120 python tools/bench_nfft.py 9 128 2>&1 | tail -n 1)"
  echo "prev:    $(AFX_LIB=$V/libafx_prev.so timeout -k 10 120 python tools/bench_nfft.py 9 128 2>&1 | tail -n 1)"
done | tee $OUT/nfft512.txt
echo "shipped hop 160: $(timeout -k 10 120 python tools/bench_nfft.py 9 160 2>&1 | tail -n 1)" | tee -a $OUT/nfft512.txt
echo "shipped complex: $(timeout -k 10 120 python tools/bench_complex.py 9 2>&1 | tail -n 1)" | tee -a $OUT/nfft512.txt
