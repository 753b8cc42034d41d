#!/bin/bash
# One GPU-box call: bench.py with the one-frame-per-wave (default) and two-frames-per-wave
# (AFX_PAIR=1) fused kernels, interleaved, + parity of the pair kernel.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pair
rm -rf $OUT; mkdir -p $OUT
for rnd in 1 2 3; do
  for P in 0 1; do
    if [ $P = 1 ]; then export AFX_PAIR=1; else unset AFX_PAIR; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('PAIR=$P round $rnd: %.1f M frames/s, step %.3f ms, fused kernel %.3f ms, cepstra %.3f ms' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['second_kernel']['kernel_ms']))" | tee -a $OUT/bench.txt
  done
done
AFX_PAIR=1 timeout 600 python -m pytest tests/test_bft_gpu.py tests/test_xxcc_gpu.py tests/test_spectrogram_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "not cwt and not cqt" > $OUT/pytest_pair.log 2>&1
echo "pytest AFX_PAIR=1 rc=$? $(tail -n 1 $OUT/pytest_pair.log)" | tee -a $OUT/bench.txt
