#!/bin/bash
# headline A/B: parity tests of the fused path, then bench cfg 2 three times (sustained numbers)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/headline
(timeout -k 10 300 python -m pytest tests/test_bft_gpu.py tests/test_xxcc_gpu.py tests/test_batch_gpu.py tests/test_fullsize_gpu.py tests/test_realaudio_gpu.py -q -m gpu -k "cfg2 or mel or bft or xxcc or mfcc" 2>&1 | tail -3)
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 value %.5g ms/step %.4f kernel_ms %.4f sustained_ms %.4f frac %.4f check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['sustained_ms'],r['frac'],d['oracle_check']['clip0_max_rel_err']))"; done
