#!/bin/bash
# Round 3, third GPU call: the time-domain CWT kernel (afx_cwt_td.hip) -- parity first, then A/B against the two-pass
# plan (AFX_CWT_NO_TD=1), trace and PMC traffic of cfg 4.   gpurun --timeout 1200 -- 'bash tools/gpu_call3.sh r03c'
set -u
TAG=${1:-r03c}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
(time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 600 python -m pytest tests/test_realaudio_gpu.py tests/test_cwt_gpu.py tests/test_pwt_gpu.py tests/test_wsst_gpu.py tests/test_synsq_gpu.py tests/test_fullsize_gpu.py tests/test_batch_gpu.py tests/dropin -q -m gpu) > $OUT/pytest.log 2>&1
RC=$?
echo "pytest (cwt family + real audio + fullsize) rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
grep -aE "^FAILED|^ERROR|^E  " $OUT/pytest.log | head -40
python tools/parity_table.py $OUT/parity.jsonl > $OUT/parity_table.md 2>&1
if [ $RC -eq 124 ]; then echo "TIMEOUT in tests -- stopping"; exit 1; fi
for v in td; do

  timeout -k 10 200 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_cfg4_$v.json 2> $OUT/bench_cfg4_$v.err
  echo "bench cfg4 $v rc=$?" | tee -a $OUT/status.txt
done

timeout -k 10 200 bash tools/prof_cmd.sh ev_${TAG}_cfg4 "" python bench.py --config 4 --clips 40 --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
cp gpurun_out/prof_ev_${TAG}_cfg4/summary.txt $OUT/rocprofv3_bench_cfg4_trace.txt 2>/dev/null
timeout -k 10 200 python tools/prof_traffic.py 4 --clips 20 --steps 1 > $OUT/traffic_cfg4.log 2>&1
cp gpurun_out/r03_bench_cfg4_pmc.json $OUT/ 2>/dev/null
timeout -k 10 200 bash tools/prof_cmd.sh ev_${TAG}_cfg4pmc "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python bench.py --config 4 --clips 20 --steps 1 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
cp gpurun_out/prof_ev_${TAG}_cfg4pmc/summary.txt $OUT/rocprofv3_bench_cfg4_mfma_pmc.txt 2>/dev/null
cat $OUT/status.txt
python - <<PY
import json
for v in ("td",):
    try:
        d = json.loads(open("$OUT/bench_cfg4_%s.json" % v).read().strip().splitlines()[-1])
        print(v, "value %.5g %s ms/step %.3f frac %.4f sustained_frac %.4f oracle %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["sustained_frac"] or 0, d["oracle_check"]))
    except Exception as e:
        print(v, "no line:", e); print(open("$OUT/bench_cfg4_%s.err" % v).read()[-1500:])
PY
head -14 $OUT/rocprofv3_bench_cfg4_trace.txt | cut -c1-150
grep -A12 "counters" $OUT/rocprofv3_bench_cfg4_mfma_pmc.txt | grep k_cwt_td | cut -c1-170
python - <<PY
import json
try:
    d = json.load(open("$OUT/r03_bench_cfg4_pmc.json"))
    tot = (2 * d["fetch_kib_per_step"] + d["write_kib_per_step"]) * 1024
    print("cfg4 traffic per chunk MB", tot / (d["clips"] * 7) / 1e6, "x algorithmic", tot / (d["clips"] * 7) / (65536 * 676))
    for k, v in d["kernels"].items(): print("   ", k[:50], v)
except Exception as e:
    print("pmc:", e)
PY
