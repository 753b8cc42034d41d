#!/bin/bash
# A/B of measurement builds on one bench configuration: bash tools/gpu_ab_cfg.sh <config> <rounds> name1 name2 ...
cd "$GRAFT_REPO_ROOT"
C=$1; R=$2; shift; shift
one() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --config $C --no-cpu-baseline --steps ${AFX_AB_STEPS:-20} --warmup 5 ${AFX_AB_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.4f kernel_ms %.4f sustained_ms %.4f check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['sustained_ms'],(d.get('oracle_check') or {}).get('clip0_max_rel_err')))"; }
V=$PWD/audioflux_amd/lib/variants
for i in $(seq $R); do one shipped AFX_X=0; for n in "$@"; do one $n AFX_LIB=$V/libafx_$n.so; done; done
