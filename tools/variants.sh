#!/bin/bash
# builds experiment variants of the library (compile-time switches AFX_V of afx_melfused.hip)
# into gpurun_variants/ for within-probe A/B runs: tools/variants.sh 0 1 2 4 ...
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/audioflux_amd/lib/variants
for v in "$@"; do
  make -s -C $ROOT/audioflux_amd/csrc -j8 BUILD=$ROOT/build/v$v TARGET=$ROOT/audioflux_amd/lib/variants/libafx_v$v.so EXTRA=-DAFX_V=$v 2>&1 | grep -E "error" || true
done
ls -la $ROOT/audioflux_amd/lib/variants
