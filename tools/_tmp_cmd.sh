cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cqt_gpu.py tests/dropin -q -m gpu -x 2>&1 | tail -n 3
timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap value %.4g ms/step %.4f sustained %.4g check %s' % (d['value'], d['ms_per_step'], d['roofline']['sustained_value'], d['oracle_check']))"
AFX_CQT_OVERLAP=0 bash tools/prof_cmd.sh cqtf16_serial "" python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --no-check --clock-warmup 0 > /dev/null 2>&1
cut -c1-130 gpurun_out/prof_cqtf16_serial/summary.txt | grep -v "at::\|rocclr" | head -14
