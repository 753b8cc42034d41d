#!/bin/bash
# Profiles `python bench.py` on the GPU box: one kernel-trace pass (timings) and separate
# --pmc passes (counters are never combined with trace domains other than kernel-trace).
# usage: tools/prof.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-legacy $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/trace_bench.log 2>&1
i=0
for CTRS in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc$i -o pmc$i -- python bench.py $ARGS > $OUT/pmc${i}_bench.log 2>&1
done
python tools/prof_summary.py $(find $OUT -name '*.db' | sort) > $OUT/summary.txt 2>&1
grep -h '"metric"' $OUT/trace_bench.log >> $OUT/summary.txt
# keep the merged payload small: databases stay on the box
find $OUT -name '*.db' -delete
cat $OUT/summary.txt
