#!/bin/bash
# one GPU-box call: the co-run probes (tools/micro/mfma_corun, tools/gpu_concurrency*.py) with 8 HIP hardware queues
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/conc; T=${1:-x}; O=gpurun_out/conc/conc_$T.txt; : > $O
r() { echo "## $*" >> $O; timeout 200 env "$@" >> $O 2>&1; }
r X=1 tools/micro/mfma_corun
export GPU_MAX_HW_QUEUES=8
for c in full seq cqt mel fwdonly; do r X=1 python tools/gpu_concurrency2.py $c 96; done
r X=1 python tools/gpu_concurrency.py 96 co
unset GPU_MAX_HW_QUEUES
(timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -n 5) >> $O
grep -a "^RESULT\|^##\|Error\|error\|^summary\|passed\|failed" $O; grep -a "^victim" $O | grep -a "form:" | grep -a "65,536 B" | grep -a "MFMAs" | cut -c1-250
