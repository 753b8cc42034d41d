#!/bin/bash
# round 6: one parameterised GPU call.   gpurun -- 'bash tools/gpu_call6.sh <tag> <step> [<step> ...]'
# steps: evidence | suite | tests:<pytest -k expr or file list> | bench | mfcc | ceps | phases | cwtphases | cfg5 | cfg4 | cfg2 (each under its own timeout; outputs in gpurun_out/call_<tag>/)
set -u
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r06
V=$PWD/audioflux_amd/lib/variants
line() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("cfg%s value %.5g ms/step %.4f frac %.4f kernel_ms %.4f clock %s" % (d["config"]["workload"][-2:-1], d["value"], d["ms_per_step"], r["frac"], r["kernel_ms"], r.get("clock_mhz_this_run")))
    for k, v in d.get("secondary", {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "frac", "traffic_over_algorithmic", "oracle_check", "error")})
except Exception as e:
    print("bench line:", e)
P
}
for STEP in "$@"; do
  case $STEP in
    suite) (time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 1200 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
           grep -aE "[0-9]+ passed|failed" $OUT/pytest.log | tail -n 1; grep -aE "^FAILED|^ERROR" $OUT/pytest.log | head -20 ;;
    tests:*) timeout -k 10 900 python -m pytest -q -m gpu -x ${STEP#tests:} > $OUT/pytest_sel.log 2>&1; tail -n 15 $OUT/pytest_sel.log | cut -c1-400 ;;
    bench) timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; line $OUT/bench_default.json ;;
    cfg2|cfg4|cfg5) c=${STEP#cfg}; timeout -k 10 300 python bench.py --config $c --steps $([ $c = 4 ] && echo 3 || echo 50) --warmup 3 --no-cpu-baseline --no-secondary --no-legacy > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; line $OUT/bench_cfg$c.json ;;
    evidence)  # the round's evidence set: PMC traffic, WARM traces (>= 200 dispatches) + compute counters of cfg 2 / 5 / 4, cfg 4's occupancy
      R=$AFX_ROUND
      timeout -k 10 200 python tools/prof_traffic.py 2 > $OUT/traffic_cfg2.log 2>&1
      timeout -k 10 200 python tools/prof_traffic.py 5 --clips 125 > $OUT/traffic_cfg5.log 2>&1
      timeout -k 10 200 python tools/prof_traffic.py 4 --clips 20 --steps 1 > $OUT/traffic_cfg4.log 2>&1
      cp gpurun_out/${R}_bench_cfg*_pmc.json $OUT/ 2>/dev/null; cp gpurun_out/${R}_bench_cfg*_pmc.json profiles/ 2>/dev/null
      COMMON="--no-cpu-baseline --no-sustained --no-check --no-secondary --no-legacy --clock-warmup 0.5"
      SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;GRBM_GUI_ACTIVE"
      timeout -k 10 400 bash tools/prof_cmd.sh ev_${TAG}_cfg5 "$SETS" python bench.py --config 5 --steps 200 --warmup 5 $COMMON > /dev/null 2>&1
      cp gpurun_out/prof_ev_${TAG}_cfg5/summary.txt $OUT/rocprofv3_bench_cfg5_trace.txt 2>/dev/null
      timeout -k 10 400 bash tools/prof_cmd.sh ev_${TAG}_cfg2 "$SETS" python bench.py --config 2 --steps 200 --warmup 5 $COMMON > /dev/null 2>&1
      cp gpurun_out/prof_ev_${TAG}_cfg2/summary.txt $OUT/rocprofv3_bench_cfg2_summary.txt 2>/dev/null
      PROF_DB_HOOK="tools/step_occupancy.py" timeout -k 10 600 bash tools/prof_cmd.sh ev_${TAG}_cfg4 "$SETS" python bench.py --config 4 --clips 100 --steps 8 --warmup 1 $COMMON > /dev/null 2>&1
      cp gpurun_out/prof_ev_${TAG}_cfg4/summary.txt $OUT/rocprofv3_bench_cfg4_trace.txt 2>/dev/null
      cp gpurun_out/prof_ev_${TAG}_cfg4/hook.txt $OUT/cfg4_occupancy.json 2>/dev/null
      python tools/prof_compute.py $OUT/rocprofv3_bench_cfg5_trace.txt k_cqt_pyramid 5 1292000 > $OUT/${R}_bench_cfg5_compute.json 2> $OUT/compute5.err
      python tools/prof_compute.py $OUT/rocprofv3_bench_cfg2_summary.txt k_stft_mel_v2 2 934000 > $OUT/${R}_bench_cfg2_compute.json 2> $OUT/compute2.err
      python tools/cfg4_compute.py $OUT/rocprofv3_bench_cfg4_trace.txt $OUT/cfg4_occupancy.json > $OUT/${R}_bench_cfg4_compute.json 2> $OUT/compute4.err
      for c in 2 4 5; do [ -s $OUT/${R}_bench_cfg${c}_compute.json ] && cp $OUT/${R}_bench_cfg${c}_compute.json profiles/; done
      grep -E "^k_cqt_pyramid|^k_stft_mel_v2|^k_cwt" $OUT/rocprofv3_bench_cfg*.txt | cut -c1-150 | head -24
      python -c "import json; o=json.load(open('$OUT/cfg4_occupancy.json')); print({k: o[k] for k in ('span_ms','union_busy_ms','sum_of_durations_ms','sum_over_union','idle_share_of_span')})" ;;
    ceps) timeout -k 10 300 python tools/bench_cepstrogram.py 2>&1 | tee $OUT/cepstrogram.txt ;;
    mfcc) AFX_BENCH_NUMS=${AFX_BENCH_NUMS:-128,80,40} timeout -k 10 400 python tools/bench_mfcc_sizes.py 2>&1 | tee $OUT/mfcc_sizes.txt ;;
    cwtphases) AFX_LIB=$V/libafx_exp.so timeout -k 10 200 python tools/cwt_phases.py 20 3 2>&1 | tail -n 12 | tee $OUT/cwt_phases.txt ;;
    phases) AFX_LIB=$V/libafx_exp.so timeout -k 10 200 python tools/pyr_phases.py 125 10 2>&1 | tail -n 16 | tee $OUT/pyr_phases.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
done
