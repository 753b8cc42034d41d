#!/bin/bash
# n_fft 4096 complex results on k_stft_band_4k2: parity tests + rate (tools/bench_complex.py 12; round 4: 91 M frames/s)
set -u
TAG=${1:-r05p}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
timeout -k 10 900 python -m pytest tests/test_bft_gpu.py tests/test_spectrogram_gpu.py tests/test_realaudio_gpu.py tests/test_zz_corun_gpu.py tests/dropin -q -m gpu -x 2>&1 | tail -n 4 | tee $OUT/pytest_tail.txt
for r in 1 2 3; do timeout -k 10 120 python tools/bench_complex.py 12 2>&1 | tail -n 1 | tee -a $OUT/complex.txt; done
timeout -k 10 120 python tools/bench_complex.py 11 2>&1 | tail -n 1 | tee -a $OUT/complex.txt
timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1 | tee -a $OUT/complex.txt
