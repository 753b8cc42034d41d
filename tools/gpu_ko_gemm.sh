#!/bin/bash
# knock-out table of k_gemm_bank_bf16x3 (tools/bench_gemm_bank.py under every named variant, shipped first and last)
#   gpurun -- 'bash tools/gpu_ko_gemm.sh <tag> name1 name2 ...'
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kog_$TAG.txt
V=$PWD/audioflux_amd/lib/variants
(timeout 200 python tools/bench_gemm_bank.py 250 40 1 2>&1 | grep -a "^gemm_bank" | sed 's/^/shipped: /') | tee -a $OUT
for n in "$@"; do (AFX_LIB=$V/libafx_$n.so timeout 200 python tools/bench_gemm_bank.py 250 40 0 2>&1 | grep -a "^gemm_bank" | sed "s/^/$n: /") | tee -a $OUT; done
(timeout 200 python tools/bench_gemm_bank.py 250 40 0 2>&1 | grep -a "^gemm_bank" | sed 's/^/shipped: /') | tee -a $OUT
