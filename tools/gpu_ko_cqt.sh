#!/bin/bash
# measurement builds of k_cqt_pyramid against the shipped library on BASELINE cfg 5: step time + shader clock (tools/ko_cqt.py),
# shipped first and last.  Variants: audioflux_amd/lib/variants/libafx_<name>.so (tools/build_variant.sh), e.g. the knock-out
# builds kocqt<mask> (afx_cqt_f16.hip AFX_KO_CQT) or the cache-policy builds cqrow<aux>.
#   gpurun -- 'bash tools/gpu_ko_cqt.sh r06a kocqt1 kocqt2 ...'
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 120 python tools/ko_cqt.py shipped 2>&1 | tail -n 1 | tee -a $OUT/ko_cqt.txt
for m in "$@"; do AFX_LIB=$V/libafx_$m.so timeout -k 10 120 python tools/ko_cqt.py $m 2>&1 | tail -n 1 | tee -a $OUT/ko_cqt.txt; done
timeout -k 10 120 python tools/ko_cqt.py shipped 2>&1 | tail -n 1 | tee -a $OUT/ko_cqt.txt
