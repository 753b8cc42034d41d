#!/bin/bash
# cfg 4 A/B (shipped against variants/libafx_prev.so) + the CWT parity file
set -u
TAG=${1:-r05t}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 600 python -m pytest tests/test_cwt_gpu.py tests/test_pwt_gpu.py -q -m gpu -x 2>&1 | tail -n 3 | tee $OUT/pytest_tail.txt
one4() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label: value %.5g ms/step %.3f check %s'%(d['value'],d['ms_per_step'],d['oracle_check']['clip0_max_rel_err']))"
}
for r in 1 2 3; do
  one4 shipped AFX_X=0 | tee -a $OUT/cfg4_ab.txt
  one4 prev AFX_LIB=$V/libafx_prev.so | tee -a $OUT/cfg4_ab.txt
done
