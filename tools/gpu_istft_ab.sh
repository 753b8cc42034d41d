#!/bin/bash
# inverse STFT at n_fft 2048 (tools/bench_next.py's 128-clip call) under the shipped library and named variants, interleaved
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/istft_ab_$1.txt; R=$2; shift; shift
V=$PWD/audioflux_amd/lib/variants
for i in $(seq $R); do
  (timeout 200 python tools/bench_next.py 2>&1 | grep -a "^istft" | cut -c1-110 | sed 's/^/shipped: /') | tee -a $OUT
  for n in "$@"; do (AFX_LIB=$V/libafx_$n.so timeout 200 python tools/bench_next.py 2>&1 | grep -a "^istft" | cut -c1-110 | sed "s/^/$n: /") | tee -a $OUT; done
done
