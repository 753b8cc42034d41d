#!/bin/bash
# One GPU-box call: validates the narrow-band CWT plan against the two-pass path and the
# reference (every CWT consumer), times BASELINE cfg 4 with the plan on (default) / off, and
# traces the kernels.   usage (from the repo root on the box): bash tools/gpu_cwt_nb.sh
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cwt_nb
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cwt_gpu.py tests/test_pwt_gpu.py tests/test_wsst_gpu.py tests/test_synsq_gpu.py \
    tests/test_fullsize_gpu.py -x -q -m gpu -k "cwt or pwt or wsst or synsq" > $OUT/pytest.log 2>&1
echo "pytest CWT consumers (default plan) rc=$? $(tail -n 1 $OUT/pytest.log)" | tee -a $OUT/status.txt
run() { echo "$1: $(env $1 timeout 300 python bench.py --config 4 --clips 5 --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-sustained 2>&1 | tail -n 1 | cut -c1-160)" | tee -a $OUT/bench.txt; }
for rnd in 1 2; do
  run "AFX_CWT_NARROW_MAX=0 AFX_CWT_GROUP=1"
  run "AFX_CWT_NARROW_MAX=0"
  run "AFX_CWT_NARROW_MAX=8"
  run "AFX_CWT_NARROW_MAX=16"
  run "AFX_DEFAULT=1"
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- \
      python $GRAFT_REPO_ROOT/bench.py --config 4 --clips 5 --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-sustained > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/prof_summary.py $(find $OUT -name '*.db' | sort) > $OUT/summary.txt 2>&1
tail -n 1 $OUT/trace.log >> $OUT/summary.txt
find $OUT -name '*.db' -delete
grep -v "at::\|rocclr" $OUT/summary.txt | cut -c1-170
