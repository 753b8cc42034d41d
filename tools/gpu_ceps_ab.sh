#!/bin/bash
# One GPU-box call: A/B of compile-time variants of the cepstrogram wave kernels
# (audioflux_amd/lib/variants/libafx_c*.so, built with -DAFX_CEPS_*), interleaved rounds.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ceps_ab
rm -rf $OUT; mkdir -p $OUT
for rnd in 1 2; do
  for v in default c1 c2 c3 c4 c5; do
    if [ $v = default ]; then unset AFX_LIB; else export AFX_LIB=$GRAFT_REPO_ROOT/audioflux_amd/lib/variants/libafx_$v.so; fi
    timeout 200 python tools/bench_cepstrogram.py 2>&1 | grep cepstrogram | cut -c1-75 | sed "s/^/$v r$rnd: /" | tee -a $OUT/bench.txt
  done
done
