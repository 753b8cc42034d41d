#!/bin/bash
# One GPU-box call: cepstrogram parity (wave kernels for n_fft 2048 / 4096 + the size-generic
# kernel) and rates with / without the wave kernels.   usage: bash tools/gpu_ceps.sh
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ceps
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cepstrogram_gpu.py tests/test_batch_gpu.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest cepstrogram + batch rc=$? $(tail -n 1 $OUT/pytest.log)" | tee -a $OUT/status.txt
grep -E "^E .*(Error|assert)|FAILED" $OUT/pytest.log | head -30
for rnd in 1 2; do
  timeout 300 python tools/bench_cepstrogram.py 2>&1 | grep cepstrogram | sed 's/^/wave kernels: /' | tee -a $OUT/bench.txt
  AFX_NO_FUSED=1 timeout 300 python tools/bench_cepstrogram.py 2>&1 | grep cepstrogram | sed 's/^/AFX_NO_FUSED: /' | tee -a $OUT/bench.txt
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_cepstrogram.py > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/prof_summary.py $(find $OUT -name '*.db' | sort) > $OUT/summary.txt 2>&1
find $OUT -name '*.db' -delete
grep -v "at::\|rocclr" $OUT/summary.txt | cut -c1-170
