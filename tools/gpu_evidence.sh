#!/bin/bash
# One GPU-box call: the round's evidence set -> gpurun_out/evidence_<tag>/ (summaries are then copied
# to profiles/ by hand).  Every step runs under its own timeout (rocprofv3 --pmc serialises
# dispatches: cfg 4 is profiled on 20 clips = 140 chunks, never on the full 7000).
#   bash tools/gpu_evidence.sh <tag> [parts]     parts: any of  tests bench prof dense rates  (default: all)
set -u
TAG=${1:-r}
PARTS=${2:-"tests bench prof dense rates"}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  rm -f $OUT/parity.jsonl
  (time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout 600 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
  echo "pytest -m gpu rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
  python tools/parity_table.py $OUT/parity.jsonl > $OUT/parity_table.md 2>&1
fi
if has bench; then
  for c in 2 5 4; do timeout 300 python bench.py --config $c > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; done
fi
if has prof; then
  timeout 300 bash tools/prof.sh ev_$TAG > /dev/null 2>&1
  cp gpurun_out/prof_ev_$TAG/summary.txt $OUT/rocprofv3_bench_cfg2_summary.txt 2>/dev/null
  timeout 200 python tools/prof_traffic.py 2 > $OUT/traffic_cfg2.log 2>&1
  timeout 200 python tools/prof_traffic.py 5 --clips 125 > $OUT/traffic_cfg5.log 2>&1
  timeout 200 python tools/prof_traffic.py 4 --clips 20 --steps 1 > $OUT/traffic_cfg4.log 2>&1
  cp gpurun_out/r03_bench_cfg*_pmc.json $OUT/ 2>/dev/null
  for c in 5 4; do
    timeout 200 bash tools/prof_cmd.sh ev_${TAG}_cfg$c "" python bench.py --config $c --clips $([ $c = 4 ] && echo 40 || echo 125) --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
    cp gpurun_out/prof_ev_${TAG}_cfg$c/summary.txt $OUT/rocprofv3_bench_cfg${c}_trace.txt 2>/dev/null
  done
fi
if has dense; then
  timeout 300 bash tools/prof_cmd.sh ev_${TAG}_dense "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS;GRBM_GUI_ACTIVE FETCH_SIZE;WRITE_SIZE" python tools/bench_dense.py 1000 3 > /dev/null 2>&1
  cp gpurun_out/prof_ev_${TAG}_dense/summary.txt $OUT/rocprofv3_dense_gemm.txt 2>/dev/null
  timeout 120 python tools/bench_dense.py > $OUT/dense_rate.txt 2>&1
fi
if has rates; then
  rm -f $OUT/rates.txt
  for t in bench_cepstrogram.py bench_split.py bench_stft.py "bench_nfft.py 10 256" "bench_nfft.py 12 1024" bench_complex.py bench_cwt_small.py bench_next.py; do
    echo "== tools/$t" >> $OUT/rates.txt
    timeout 120 python tools/$t 2>&1 | grep -vE "^\s*$|Warning|warn|amdgpu.ids" | tail -n 14 >> $OUT/rates.txt
  done
  
fi
ls $OUT; cat $OUT/status.txt 2>/dev/null
