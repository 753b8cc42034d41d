#!/bin/bash
# n_fft 1024 kernel with hand-issued LDS traffic + complex results with the imaginary parts' row parked in LDS (12 waves):
# BFT parity files, then interleaved rates against the previous library and the variants under audioflux_amd/lib/variants
set -u
TAG=${1:-r05u}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_bft_gpu.py tests/test_batch_gpu.py tests/test_fullsize_gpu.py tests/test_reassign_gpu.py -q -m gpu -x 2>&1 | tail -n 5 | tee $OUT/pytest_tail.txt
run() { local label=$1; shift; echo -n "$label: "; env "$@" 2>&1 | tail -n 1; }
for r in 1 2 3; do
  for n in shipped prev k1b2; do
    L=$V/libafx_$n.so; [ $n = shipped ] && L=
    run "$n" AFX_LIB=$L timeout -k 10 120 python tools/bench_nfft.py 10 256 | tee -a $OUT/nfft1024.txt
  done
  for n in shipped prev cplx8; do
    L=$V/libafx_$n.so; [ $n = shipped ] && L=
    run "$n" AFX_LIB=$L timeout -k 10 120 python tools/bench_complex.py 11 | tee -a $OUT/complex.txt
  done
done
run "shipped 1k complex" timeout -k 10 120 python tools/bench_complex.py 10 | tee -a $OUT/complex.txt
run "prev 1k complex" AFX_LIB=$V/libafx_prev.so timeout -k 10 120 python tools/bench_complex.py 10 | tee -a $OUT/complex.txt
run "shipped headline" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy --steps 20 --warmup 5 | tee -a $OUT/headline.txt
