#!/bin/bash
# tools/bench_next.py (STFT, spectrogram objects incl. STFT-chroma / log-chroma, PWT) under the shipped library and named variants
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/next_$1.txt; shift
V=$PWD/audioflux_amd/lib/variants
(timeout 300 python tools/bench_next.py 2>&1 | grep -a "frames/s\|chunks/s" | sed 's/^/shipped: /') | tee -a $OUT
for n in "$@"; do (AFX_LIB=$V/libafx_$n.so timeout 300 python tools/bench_next.py 2>&1 | grep -a "chroma\|^stft" | sed "s/^/$n: /") | tee -a $OUT; done
timeout 300 python tools/bench_stft_sizes.py 2>&1 | tail -8 | tee -a $OUT
