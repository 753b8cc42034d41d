#!/bin/bash
# cfg 4: two-level four-step twiddles in the narrow-band kernels against the previous build; CWT / PWT / WSST / synsq parity tests
set -u
TAG=${1:-r05n}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
V=$PWD/audioflux_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_cwt_gpu.py tests/test_pwt_gpu.py tests/test_wsst_gpu.py tests/test_synsq_gpu.py tests/test_fullsize_gpu.py tests/test_realaudio_gpu.py tests/test_batch_gpu.py -q -m gpu -x 2>&1 | tail -n 4 | tee $OUT/pytest_tail.txt
one4() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.3f check %s'%(d['value'],d['ms_per_step'],d['oracle_check']['clip0_max_rel_err']))"
}
for r in 1 2 3; do
  one4 shipped AFX_X=0 | tee -a $OUT/cfg4_ab.txt
  one4 prev AFX_LIB=$V/libafx_prev.so | tee -a $OUT/cfg4_ab.txt
done
