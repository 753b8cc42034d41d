#!/bin/bash
# The evidence call of round 5: the whole -m gpu suite, the driver's bench line, PMC traffic of cfg 2 / 4 / 5, and -- new --
# WARM kernel traces: every configuration's trace and compute-counter passes run `bench.py --clock-warmup 0.5 --steps S` with
# S >= 200 for cfg 2 / cfg 5 (cfg 4: 8 steps of 219 group launches), so the trace average and the counters' clock are those
# of the loaded chip the bench line quotes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_call5.sh r05a [nosuite]'
set -u
TAG=${1:-r05a}
NOSUITE=${2:-}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
if [ -z "$NOSUITE" ]; then
  rm -f $OUT/parity.jsonl
  (time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 1000 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
  echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
  grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -40
  python tools/parity_table.py $OUT/parity.jsonl > $OUT/parity_table.md 2>&1
fi
# ---- PMC traffic (clock-independent: cold 3-step passes are fine)
timeout -k 10 200 python tools/prof_traffic.py 2 > $OUT/traffic_cfg2.log 2>&1
timeout -k 10 200 python tools/prof_traffic.py 5 --clips 125 > $OUT/traffic_cfg5.log 2>&1
timeout -k 10 200 python tools/prof_traffic.py 4 --clips 20 --steps 1 > $OUT/traffic_cfg4.log 2>&1
cp gpurun_out/r05_bench_cfg*_pmc.json $OUT/ 2>/dev/null
cp gpurun_out/r05_bench_cfg*_pmc.json profiles/ 2>/dev/null   # the bench line below quotes THIS build's traffic
# ---- warm traces + compute counters
COMMON="--no-cpu-baseline --no-sustained --no-check --no-secondary --no-legacy --clock-warmup 0.5"
SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;GRBM_GUI_ACTIVE"
timeout -k 10 400 bash tools/prof_cmd.sh ev_${TAG}_cfg5 "$SETS" python bench.py --config 5 --steps 200 --warmup 5 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_ev_${TAG}_cfg5/summary.txt $OUT/rocprofv3_bench_cfg5_trace.txt 2>/dev/null
timeout -k 10 400 bash tools/prof_cmd.sh ev_${TAG}_cfg2 "$SETS" python bench.py --config 2 --steps 200 --warmup 5 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_ev_${TAG}_cfg2/summary.txt $OUT/rocprofv3_bench_cfg2_summary.txt 2>/dev/null
timeout -k 10 400 bash tools/prof_cmd.sh ev_${TAG}_cfg4 "" python bench.py --config 4 --clips 100 --steps 8 --warmup 1 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_ev_${TAG}_cfg4/summary.txt $OUT/rocprofv3_bench_cfg4_trace.txt 2>/dev/null
python tools/prof_compute.py $OUT/rocprofv3_bench_cfg5_trace.txt k_cqt_pyramid 5 1292000 > $OUT/r05_bench_cfg5_compute.json 2> $OUT/compute5.err
python tools/prof_compute.py $OUT/rocprofv3_bench_cfg2_summary.txt k_stft_mel_v2 2 934000 > $OUT/r05_bench_cfg2_compute.json 2> $OUT/compute2.err
for c in 2 5; do [ -s $OUT/r05_bench_cfg${c}_compute.json ] && cp $OUT/r05_bench_cfg${c}_compute.json profiles/; done
# ---- the driver's line (quotes the traffic / compute files written above)
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default rc=$?" | tee -a $OUT/status.txt
cat $OUT/status.txt
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("cfg2 value %.4g ms/step %.4f frac %.4f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    for k, v in d.get("secondary", {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "frac", "sustained_frac", "traffic_over_algorithmic", "oracle_check", "error")})
    print("legacy", d.get("legacy"))
except Exception as e:
    print("bench line:", e); print(open("$OUT/bench_default.err").read()[-2000:])
PY
grep -E "k_cqt_pyramid|k_stft_mel_v2" $OUT/rocprofv3_bench_cfg5_trace.txt $OUT/rocprofv3_bench_cfg2_summary.txt | head -12
tail -n 12 $OUT/parity_table.md 2>/dev/null
