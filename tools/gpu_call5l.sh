#!/bin/bash
# wave-priority masks re-measured on the round-5 kernels (variants libafx_m<mask>.so: both k_stft_mel_v2 and k_stft_band_4k2)
set -u
TAG=${1:-r05l}; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
bash tools/gpu_ab_lib.sh $TAG 2 "$@"
for r in 1 2; do
  for n in shipped "$@"; do
    L=""; [ "$n" != shipped ] && L="AFX_LIB=$V/libafx_$n.so"
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1)" | tee -a $OUT/nfft.txt
  done
done
