#!/bin/bash
# interleaved A/B of the shipped library against variants on BASELINE cfg 2 (bench.py, oracle check on):
#   gpurun -- 'bash tools/gpu_ab_lib.sh <tag> <rounds> name1 name2 ...'     (audioflux_amd/lib/variants/libafx_<name>.so)
set -u
TAG=$1; R=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
one() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.4f kernel_ms %.4f sustained_ms %.4f check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['sustained_ms'],d['oracle_check']['clip0_max_rel_err']))"
}
for i in $(seq $R); do
  one shipped AFX_X=0 | tee -a $OUT/ab.txt
  for n in "$@"; do one $n AFX_LIB=$V/libafx_$n.so | tee -a $OUT/ab.txt; done
done
