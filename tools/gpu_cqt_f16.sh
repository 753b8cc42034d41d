#!/bin/bash
# One GPU-box call for the f16 CQT octave kernel: parity tests of everything built on the CQT, cfg-5 bench with
# the f16 and the f32 kernels (AFX_CQT_F32=1), kernel trace of the f16 run.  -> gpurun_out/cqt_f16_<tag>/
set -u
TAG=${1:-a}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cqt_f16_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
(AFX_PARITY_LOG=$PWD/$OUT/parity.jsonl timeout -k 10 400 python -m pytest tests/test_cqt_gpu.py tests/test_spectrogram_gpu.py tests/test_fullsize_gpu.py tests/dropin -q -m gpu -x) > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(grep -aE '[0-9]+ passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
grep -a "cqt\|chroma\|cqcc\|cqhc" $OUT/parity.jsonl | python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin]
rows.sort(key=lambda r:-r['measured']/r['bar'])
for r in rows[:12]: print('%-70s %.2e / %.0e' % (r['what'][:70], r['measured'], r['bar']))
" > $OUT/parity_top.txt 2>&1
timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_f16.json 2> $OUT/bench_f16.err
AFX_CQT_F32=1 timeout -k 10 200 python bench.py --config 5 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
timeout -k 10 200 bash tools/prof_cmd.sh cqtf16_$TAG "" python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
cp gpurun_out/prof_cqtf16_$TAG/summary.txt $OUT/trace.txt 2>/dev/null
cat $OUT/status.txt; tail -n 5 $OUT/pytest.log; cat $OUT/parity_top.txt
python - <<PY
import json
for n in ("f16","f32"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, "value %.4g ms/step %.4f sustained %.4g check %s" % (d["value"], d["ms_per_step"], d["roofline"]["sustained_value"], d["oracle_check"]))
    except Exception as e:
        print(n, "failed", e); print(open("$OUT/bench_%s.err"%n).read()[-1500:])
PY
head -n 16 $OUT/trace.txt | cut -c1-150
