#!/bin/bash
# quick A/B of the time-domain CWT kernel: CWT parity tests, bench cfg 4, isolated kernel times (PMC pass serialises)
set -u
TAG=${1:-q}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout -k 10 400 python -m pytest tests/test_realaudio_gpu.py tests/test_cwt_gpu.py tests/test_pwt_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "cwt or pwt or cfg4") > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
grep -aE "^FAILED|^ERROR|^E  " $OUT/pytest.log | head -20
timeout -k 10 200 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_cfg4_td.json 2> $OUT/bench_cfg4_td.err
timeout -k 10 200 bash tools/prof_cmd.sh ev_${TAG}_pmc "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" python bench.py --config 4 --clips 20 --steps 1 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
python - <<PY
import json
d = json.loads(open("$OUT/bench_cfg4_td.json").read().strip().splitlines()[-1])
print("td value %.5g %s ms/step %.3f frac %.4f oracle %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["oracle_check"]))
PY
head -12 gpurun_out/prof_ev_${TAG}_pmc/summary.txt | cut -c1-130
