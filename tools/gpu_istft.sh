#!/bin/bash
# iSTFT: parity tests + rates (tools/bench_next.py lines) under the shipped library and named variants
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/istft_$1.txt; shift
V=$PWD/audioflux_amd/lib/variants
(timeout 600 python -m pytest tests/test_stft_gpu.py tests/test_batch_gpu.py -q -m gpu -x 2>&1 | tail -n 15) | tee -a $OUT
(timeout 300 python tools/bench_next.py 2>&1 | grep -a "^stft\|^istft" | sed 's/^/shipped: /') | tee -a $OUT
for n in "$@"; do (AFX_LIB=$V/libafx_$n.so timeout 300 python tools/bench_next.py 2>&1 | grep -a "^stft\|^istft" | sed "s/^/$n: /") | tee -a $OUT; done
