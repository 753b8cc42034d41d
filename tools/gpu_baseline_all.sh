#!/bin/bash
# One GPU-box call (~4 GPU-minutes): the whole evidence set in one go -- GPU test suite, the bench
# line, every secondary rate quoted in README.md, and the rocprofv3 summary of bench.py.  Meant as the
# first call of a round: bash tools/gpu_baseline_all.sh <tag>   -> gpurun_out/baseline_<tag>/
set -u
TAG=${1:-r}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/baseline_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(time timeout -k 10 600 python -m pytest tests -q -m gpu) > $OUT/pytest.log 2>&1
echo "pytest -m gpu rc=$? $(grep -E 'passed|failed' $OUT/pytest.log | tail -n 1)" | tee $OUT/status.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json
for t in bench_cepstrogram.py bench_split.py bench_stft.py \
         "bench_nfft.py 10 256" "bench_nfft.py 12 1024" bench_complex.py bench_cwt_small.py bench_next.py; do
  echo "== tools/$t" >> $OUT/rates.txt
  timeout -k 10 300 python tools/$t 2>&1 | grep -vE "^\s*$|Warning|warn" | tail -n 14 >> $OUT/rates.txt
done
cat $OUT/rates.txt | cut -c1-200
bash tools/prof.sh base_$TAG > /dev/null 2>&1
cp gpurun_out/prof_base_$TAG/summary.txt $OUT/rocprofv3_bench_summary.txt 2>/dev/null
head -n 12 $OUT/rocprofv3_bench_summary.txt | cut -c1-170
