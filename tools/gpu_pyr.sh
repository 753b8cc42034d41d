#!/bin/bash
# one GPU-box call for the one-launch CQT ladder (k_cqt_pyramid): CQT parity tests under a short timeout first, the
# device comparison with the per-octave path (bitwise), then bench cfg 5 both ways and a kernel trace
cd "$GRAFT_REPO_ROOT"; T=${1:-p}; D=gpurun_out/pyr_$T; mkdir -p $D
export TMPDIR=/tmp
(timeout -k 10 400 python -m pytest tests/test_cqt_gpu.py -q -m gpu -x 2>&1 | tail -n 15) > $D/pytest_cqt.txt; cat $D/pytest_cqt.txt
(timeout -k 10 400 python -m pytest tests/test_cqt_gpu.py -q -m gpu -k ladder 2>&1 | tail -n 5) > $D/compare.txt; cat $D/compare.txt
for v in 1 0; do
  AFX_CQT_PYRAMID=$v timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $D/bench5_pyr$v.json 2> $D/bench5_pyr$v.err
  python - <<P
import json
try:
    d=json.loads(open("$D/bench5_pyr$v.json").read().strip().splitlines()[-1])
    print("pyramid=$v cfg5", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("check"))
except Exception as e: print("pyramid=$v bench failed", e); print(open("$D/bench5_pyr$v.err").read()[-1500:])
P
done
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$D/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-check > $GRAFT_REPO_ROOT/$D/trace_bench.log 2>&1
cd "$GRAFT_REPO_ROOT"; python tools/prof_summary.py $(find $D/trace -name '*.db' | sort) > $D/trace_summary.txt 2>&1; find $D -name '*.db' -delete
head -n 14 $D/trace_summary.txt | cut -c1-180
