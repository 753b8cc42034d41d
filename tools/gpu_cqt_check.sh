#!/bin/bash
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/cqtchk; mkdir -p $D; export TMPDIR=/tmp
timeout 120 tools/micro/pk_forms_corun > $D/pk_forms.txt 2>&1
for i in 1 2; do timeout 200 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 value %.0f ms %.4f frac %.4f check %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('oracle_check')))"; done | tee $D/bench.txt
timeout 200 bash tools/prof_cmd.sh cqtchk "" python bench.py --config 5 --clips 125 --steps 3 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
cp gpurun_out/prof_cqtchk/summary.txt $D/trace.txt 2>/dev/null
(timeout 300 python -m pytest tests/test_cqt_gpu.py tests/test_realaudio_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "cqt or chroma" 2>&1 | tail -n 3) | tee $D/tests.txt
tail -n 4 $D/pk_forms.txt; head -n 14 $D/trace.txt
