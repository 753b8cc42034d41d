#!/bin/bash
# one GPU-box call: whole -m gpu suite, the default bench line, bitwise determinism of the batched CWT, the co-run probe
cd "$GRAFT_REPO_ROOT"; T=${1:-q}; D=gpurun_out/quick_$T; mkdir -p $D
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -n 15) > $D/pytest.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_default.json 2> $D/bench_default.err
timeout 200 python tools/gpu_td_det2.py > $D/det2.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/gpu_concurrency2.py full 96 2>&1 | grep -a RESULT > $D/conc.txt
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/gpu_concurrency2.py cqt 96 2>&1 | grep -a RESULT >> $D/conc.txt
cat $D/pytest.txt; grep -a "^rep" $D/det2.txt; cat $D/conc.txt
python - <<P
import json
d=json.loads(open("$D/bench_default.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d.get("secondary",{}).items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("frac"), v.get("oracle_check"))
P
