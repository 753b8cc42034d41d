#!/bin/bash
# PMC of the cfg-4 launches: LDS conflicts and unit occupancy per kernel
set -u
TAG=${1:-r05m}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-sustained --no-check --no-secondary --no-legacy --clock-warmup 0.3"
SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;GRBM_GUI_ACTIVE"
timeout -k 10 500 bash tools/prof_cmd.sh ${TAG}_cfg4 "$SETS" python bench.py --config 4 --clips 40 --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_cfg4/summary.txt $OUT/cfg4_pmc_summary.txt
grep -E "k_cwt|SQ_|GRBM" $OUT/cfg4_pmc_summary.txt | cut -c1-220 | head -120
