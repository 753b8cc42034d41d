#!/bin/bash
# dense route (tools/bench_dense.py) against the size of its spectrum scratch (AFX_SCRATCH_MB: clips per chunk), twice
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/dense_chunks_$1.txt; shift
for r in 1 2; do for mb in "$@"; do
  (AFX_SCRATCH_MB=$mb timeout 200 python tools/bench_dense.py 1000 20 2>&1 | grep -a "^dense" | cut -c1-90 | sed "s/^/scratch $mb MB: /") | tee -a $OUT
done; done
