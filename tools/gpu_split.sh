#!/bin/bash
# One GPU-box call: parity of the split band plans + rates with / without them
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/split
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_bft_gpu.py tests/test_spectrogram_gpu.py tests/test_xxcc_gpu.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest bft + spectrogram + xxcc rc=$? $(tail -n 1 $OUT/pytest.log)" | tee -a $OUT/status.txt
grep -E "^E .*(Error|assert)|FAILED" $OUT/pytest.log | head -20
timeout 300 python tools/bench_split.py 2>&1 | grep "plan kind" | sed 's/^/split:    /' | tee -a $OUT/bench.txt
AFX_NO_SPLIT=1 timeout 300 python tools/bench_split.py 2>&1 | grep "plan kind" | sed 's/^/no split: /' | tee -a $OUT/bench.txt
