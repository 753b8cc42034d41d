#!/bin/bash
# one GPU-box call for the dense filter-bank route (gammatone-128 at the headline shape: afxk_stft2k -> k_gemm_bank_bf16x3):
# parity tests of everything that runs through it, interleaved A/B against named library variants
# (audioflux_amd/lib/variants/libafx_<name>.so), kernel trace + matrix-pipe counters of the shipped library.
#   gpurun -- 'bash tools/gpu_dense.sh <tag> <rounds> [variant ...]'
set -u
TAG=$1; R=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/dense_$TAG; mkdir -p $OUT
V=$PWD/audioflux_amd/lib/variants
(timeout 900 python -m pytest tests/test_bft_gpu.py tests/test_stft_gpu.py tests/test_spectrogram_gpu.py tests/test_batch_gpu.py tests/test_reassign_gpu.py \
   -q -m gpu -x 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -n 15) | tee $OUT/pytest.txt
for i in $(seq $R); do
  timeout 300 python tools/bench_dense.py 1000 20 2>&1 | grep -a "^dense" | sed 's/^/shipped: /' | tee -a $OUT/ab.txt
  for n in "$@"; do AFX_LIB=$V/libafx_$n.so timeout 300 python tools/bench_dense.py 1000 20 2>&1 | grep -a "^dense" | sed "s/^/$n: /" | tee -a $OUT/ab.txt; done
done
bash tools/prof_cmd.sh dense_$TAG "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES;GRBM_GUI_ACTIVE FETCH_SIZE;WRITE_SIZE" \
  python tools/bench_dense.py 1000 10 > $OUT/prof.txt 2>&1
tail -n 60 $OUT/prof.txt
