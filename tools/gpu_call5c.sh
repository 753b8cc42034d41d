#!/bin/bash
# round 5, third call: full suite; cfg 4 with the time-domain kernel's image low words out of LDS (shipped) against the
# previous kernel (variants/libafx_oldtd.so), interleaved; phases of the one-clip legacy call (variants/libafx_exp.so)
#   gpurun --timeout 1800 -- 'bash tools/gpu_call5c.sh r05c [nosuite]'
set -u
TAG=${1:-r05c}; NOSUITE=${2:-}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
V=$PWD/audioflux_amd/lib/variants
if [ "$NOSUITE" != "nosuite" ]; then
  rm -f $OUT/parity.jsonl
  (time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 1200 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
  echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
  grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -40
fi
one4() { # label env
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.3f kernel_ms %.3f sustained %.5g check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r.get('sustained_value',0),d['oracle_check']['clip0_max_rel_err']))"
}
for r in 1 2; do
  one4 shipped AFX_X=0 | tee -a $OUT/cfg4_ab.txt
  one4 oldtd AFX_LIB=$V/libafx_oldtd.so | tee -a $OUT/cfg4_ab.txt
done
echo "--- legacy phases (instrumented build)" | tee -a $OUT/legacy.txt
AFX_LIB=$V/libafx_exp.so timeout -k 10 240 python tools/legacy_bench.py 1000 2>&1 | cut -c1-400 | tail -n 8 | tee -a $OUT/legacy.txt
echo "--- legacy phases, AFX_NO_STAGING=1" | tee -a $OUT/legacy.txt
AFX_NO_STAGING=1 AFX_LIB=$V/libafx_exp.so timeout -k 10 240 python tools/legacy_bench.py 1000 2>&1 | cut -c1-400 | tail -n 8 | tee -a $OUT/legacy.txt
