#!/bin/bash
# The evidence call of a round (round 4 on): the whole -m gpu suite on the cleaned tree (new: real audio / hard clips, cfg-2 second
# corpus, cfg-4 ring), the driver's bench line (with `secondary`), PMC traffic + kernel traces of the SHIPPED
# kernels for cfg 2 / 4 / 5.   gpurun --timeout 1800 -- 'bash tools/gpu_call4.sh r04a'
set -u
TAG=${1:-r04a}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
(time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 900 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -40
python tools/parity_table.py $OUT/parity.jsonl > $OUT/parity_table.md 2>&1
timeout -k 10 300 bash tools/prof.sh ev_$TAG > /dev/null 2>&1
cp gpurun_out/prof_ev_$TAG/summary.txt $OUT/rocprofv3_bench_cfg2_summary.txt 2>/dev/null
timeout -k 10 200 python tools/prof_traffic.py 2 > $OUT/traffic_cfg2.log 2>&1
timeout -k 10 200 python tools/prof_traffic.py 5 --clips 125 > $OUT/traffic_cfg5.log 2>&1
timeout -k 10 200 python tools/prof_traffic.py 4 --clips 20 --steps 1 > $OUT/traffic_cfg4.log 2>&1
cp gpurun_out/r04_bench_cfg*_pmc.json $OUT/ 2>/dev/null
cp gpurun_out/r04_bench_cfg*_pmc.json profiles/ 2>/dev/null   # the bench line below quotes THIS build's traffic
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default rc=$?" | tee -a $OUT/status.txt
for c in 5 4; do
  timeout -k 10 200 bash tools/prof_cmd.sh ev_${TAG}_cfg$c "" python bench.py --config $c --clips $([ $c = 4 ] && echo 40 || echo 125) --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
  cp gpurun_out/prof_ev_${TAG}_cfg$c/summary.txt $OUT/rocprofv3_bench_cfg${c}_trace.txt 2>/dev/null
done
timeout -k 10 120 python tools/bench_hostabi.py > $OUT/hostabi.txt 2>&1
cat $OUT/status.txt
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("cfg2 value %.4g ms/step %.4f frac %.4f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    for k, v in d.get("secondary", {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "frac", "sustained_frac", "traffic_over_algorithmic", "oracle_check", "error")})
except Exception as e:
    print("bench line:", e); print(open("$OUT/bench_default.err").read()[-2000:])
PY
tail -n 25 $OUT/parity_table.md
