#!/bin/bash
# knock-out study of the headline kernel (measurement builds libafx_ko<mask>.so, afx_melfused2.hip AFX_KO): step time of
# BASELINE cfg 2 with one class of LDS traffic / arithmetic removed at a time.   gpurun -- 'bash tools/gpu_knockout.sh r05e 1 2 4 ...'
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
one() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: ms/step %.4f kernel_ms %.4f sustained_ms %.4f'%(d['ms_per_step'],r['kernel_ms'],r['sustained_ms']))"
}
one shipped AFX_X=0 | tee -a $OUT/knockout.txt
for m in "$@"; do one ko$m AFX_LIB=$V/libafx_ko$m.so | tee -a $OUT/knockout.txt; done
one shipped AFX_X=0 | tee -a $OUT/knockout.txt
