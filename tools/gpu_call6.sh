#!/bin/bash
# round 6: one parameterised GPU call.   gpurun -- 'bash tools/gpu_call6.sh <tag> <step> [<step> ...]'
# steps: suite | tests:<pytest -k expr or file list> | bench | mfcc | phases | cfg5 | cfg4 | cfg2 (each under its own timeout; outputs in gpurun_out/call_<tag>/)
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
    ceps) timeout -k 10 300 python tools/bench_cepstrogram.py 2>&1 | tee $OUT/cepstrogram.txt ;;
    mfcc) AFX_BENCH_NUMS=${AFX_BENCH_NUMS:-128,80,40} timeout -k 10 400 python tools/bench_mfcc_sizes.py 2>&1 | tee $OUT/mfcc_sizes.txt ;;
    phases) AFX_LIB=$V/libafx_exp.so timeout -k 10 200 python tools/pyr_phases.py 125 10 2>&1 | tail -n 16 | tee $OUT/pyr_phases.txt ;;
    *) echo "unknown step $STEP" ;;
  esac
done
