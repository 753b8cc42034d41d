#!/bin/bash
# round 4 headline A/B: shipped vs measurement builds (audioflux_amd/lib/variants/libafx_<name>.so), interleaved, sustained figures
# usage: bash tools/gpu_headline_ab4.sh <rounds> name1 name2 ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/headline4
R=${1:-2}; shift
one() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.4f kernel_ms %.4f sustained_ms %.4f check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['sustained_ms'],d['oracle_check']['clip0_max_rel_err']))"
}
V=$PWD/audioflux_amd/lib/variants
for i in $(seq $R); do
  one shipped AFX_X=0
  for n in "$@"; do one $n AFX_LIB=$V/libafx_$n.so; done
done
