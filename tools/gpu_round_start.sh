#!/bin/bash
# First GPU call of a round: (1) the whole `pytest -m gpu` suite with the parity log, (2) the three bench lines,
# (3) A/B of the candidates that were written without hardware access at the end of round 2 and ship switched off:
#     AFX_CQT_CHROMA_V2=1 (k_cqt_chroma_v2: host-built bin lists, wave-uniform class walk) -- parity tests of the
#     CQT first, then the cfg-5 step with and without it, and its kernel trace; AFX_CQT_FUSED=1 (k_cqt_all_f16:
#     seven octaves + chroma in one launch) the same way.
# -> gpurun_out/round_start_<tag>/ ; every step under its own timeout.
#   gpurun --timeout -k 10 1500 -- 'bash tools/gpu_round_start.sh r03'
set -u
TAG=${1:-r}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_start_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# what the f16 CQT kernels assume of the hardware (per-dword bounds checks of raw buffer accesses, MFMA layout)
(timeout -k 10 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_oob.hip -o /tmp/buffer_oob && timeout -k 10 60 /tmp/buffer_oob) > $OUT/buffer_oob.txt 2>&1
echo "buffer_oob rc=$? $(grep -c ' ok$' $OUT/buffer_oob.txt)/3 ok" | tee $OUT/status.txt
rm -f $OUT/parity.jsonl
(time AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 900 python -m pytest tests -q -m gpu -x) > $OUT/pytest.log 2>&1
RC=$?
echo "pytest -m gpu rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee -a $OUT/status.txt
if [ $RC -ne 0 ]; then  # a failure or a hang in the shipped kernels: do not spend GPU time on anything else
  tail -n 40 $OUT/pytest.log
  exit 1
fi
python tools/parity_table.py $OUT/parity.jsonl > $OUT/parity_table.md 2>&1
# tests written without hardware access (skipped unless asked for): launch splitting beyond 2^32 threads
(AFX_TEST_UNVERIFIED=1 timeout -k 10 300 python -m pytest tests/test_stft_gpu.py -q -m gpu -x -k beyond_2_32) > $OUT/pytest_unverified.log 2>&1
echo "unverified tests rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_unverified.log | tail -n 1)" | tee -a $OUT/status.txt
# a step that runs into its timeout (rc 124) means a hung kernel: every later step would hang too -- stop
guard() { local rc=$?; if [ $rc -eq 124 ]; then echo "TIMEOUT in: $1 -- stopping" | tee -a $OUT/status.txt; cat $OUT/status.txt; exit 1; fi; }
for c in 2 5 4; do timeout -k 10 300 python bench.py --config $c > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; guard "bench cfg $c"; done
# candidates
(AFX_CQT_CHROMA_V2=1 timeout -k 10 300 python -m pytest tests/test_cqt_gpu.py tests/test_batch_gpu.py tests/dropin -q -m gpu -x) > $OUT/pytest_chroma_v2.log 2>&1
RC=$?
echo "chroma v2 tests rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_chroma_v2.log | tail -n 1)" | tee -a $OUT/status.txt
if [ $RC -ne 0 ]; then tail -n 30 $OUT/pytest_chroma_v2.log; cat $OUT/status.txt; exit 0; fi
for v in 0 1; do
  if [ $v = 1 ]; then export AFX_CQT_CHROMA_V2=1; else unset AFX_CQT_CHROMA_V2; fi
  timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5_chroma_v2_$v.json 2> $OUT/bench_cfg5_chroma_v2_$v.err
  guard "bench cfg 5, chroma v2 = $v"
  AFX_CQT_OVERLAP=0 timeout -k 10 200 bash tools/prof_cmd.sh rs_${TAG}_chroma$v "" python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
  cp gpurun_out/prof_rs_${TAG}_chroma$v/summary.txt $OUT/trace_cfg5_chroma_v2_$v.txt 2>/dev/null
done
unset AFX_CQT_CHROMA_V2
# AFX_CQT_FUSED=1 (k_cqt_all_f16: seven octaves + chroma in one launch): smallest parity test first, short timeout
(AFX_CQT_FUSED=1 timeout -k 10 120 python -m pytest tests/test_cqt_gpu.py -q -m gpu -x) > $OUT/pytest_fused.log 2>&1
RC=$?
echo "fused tests (cqt) rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_fused.log | tail -n 1)" | tee -a $OUT/status.txt
if [ $RC -eq 0 ]; then
  (AFX_CQT_FUSED=1 timeout -k 10 300 python -m pytest tests/test_batch_gpu.py tests/dropin -q -m gpu -x) >> $OUT/pytest_fused.log 2>&1
  RC=$?
  echo "fused tests (batch, dropin) rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_fused.log | tail -n 1)" | tee -a $OUT/status.txt
fi
if [ $RC -eq 0 ]; then
  AFX_CQT_FUSED=1 timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5_fused.json 2> $OUT/bench_cfg5_fused.err
  guard "bench cfg 5 fused"
  AFX_CQT_FUSED=1 AFX_CQT_CHUNK=125 timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5_fused_onepass.json 2> $OUT/bench_cfg5_fused_onepass.err
  # decimations of pass p + 1 on the side stream under the launch of pass p: its own parity run first
  (AFX_CQT_FUSED=2 AFX_CQT_CHUNK=2 timeout -k 10 200 python -m pytest tests/test_batch_gpu.py -q -m gpu -x -k cqt) > $OUT/pytest_fused2.log 2>&1
  RC2=$?
  echo "fused mode 2 tests rc=$RC2 $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_fused2.log | tail -n 1)" | tee -a $OUT/status.txt
  if [ $RC2 -eq 0 ]; then
    for ch in 63 32 16; do
      AFX_CQT_FUSED=2 AFX_CQT_CHUNK=$ch timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_cfg5_fused2_chunk$ch.json 2> $OUT/bench_cfg5_fused2_chunk$ch.err
    done
  fi
  AFX_CQT_FUSED=1 timeout -k 10 200 bash tools/prof_cmd.sh rs_${TAG}_fused "" python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
  cp gpurun_out/prof_rs_${TAG}_fused/summary.txt $OUT/trace_cfg5_fused.txt 2>/dev/null
else
  tail -n 30 $OUT/pytest_fused.log
  # a candidate that ran into its timeout has most likely hung the device: nothing after it would run
  if [ $RC -eq 124 ]; then echo "TIMEOUT in the fused CQT tests -- stopping" | tee -a $OUT/status.txt; cat $OUT/status.txt; exit 0; fi
fi
# AFX_GEMM_BF16=1 (k_gemm_nt128_bf16x3: dense filter-bank GEMM on three bf16 words per operand)
(AFX_GEMM_BF16=1 timeout -k 10 200 python -m pytest tests/test_bft_gpu.py tests/test_spectrogram_gpu.py -q -m gpu -x -k "dense or gammatone or chroma") > $OUT/pytest_gemm_bf16.log 2>&1
RC=$?
echo "bf16 GEMM tests rc=$RC $(grep -aE '[0-9]+ passed|failed' $OUT/pytest_gemm_bf16.log | tail -n 1)" | tee -a $OUT/status.txt
if [ $RC -eq 0 ]; then
  timeout -k 10 120 python tools/bench_dense.py > $OUT/dense_f32.txt 2>&1; guard "bench_dense f32"
  AFX_GEMM_BF16=1 timeout -k 10 120 python tools/bench_dense.py > $OUT/dense_bf16.txt 2>&1; guard "bench_dense bf16"
  tail -n 4 $OUT/dense_f32.txt $OUT/dense_bf16.txt
else
  tail -n 25 $OUT/pytest_gemm_bf16.log
fi
cat $OUT/status.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_cfg*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.4g %s ms/step %.4f frac %.4f" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"]))
    except Exception as e:
        print(f.split("/")[-1], "no line:", e)
PY
grep -h "k_cqt_chroma" $OUT/trace_cfg5_chroma_v2_*.txt | cut -c1-120
