#!/bin/bash
# round 5, fourth call: (1) kernel trace of cfg 4 with the dispatch intervals of k_cwt_td against the FFT-path launches
# (tools/td_overlap.py); GPU_MAX_HW_QUEUES 4 (default) vs 8; (2) headline / n_fft 4096 with the inline-asm pads removed
# (measurement builds, tools/exp/nonop_build.sh); (3) the legacy one-clip call with small calls spread over all CUs
set -u
TAG=${1:-r05d}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG
mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=r05
V=$PWD/audioflux_amd/lib/variants
COMMON="--no-cpu-baseline --no-sustained --no-check --no-secondary --no-legacy --clock-warmup 0.3"
PROF_DB_HOOK=tools/td_overlap.py timeout -k 10 300 bash tools/prof_cmd.sh ${TAG}_cfg4 "" python bench.py --config 4 --clips 100 --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_cfg4/hook.txt $OUT/td_overlap.txt; cp gpurun_out/prof_${TAG}_cfg4/summary.txt $OUT/cfg4_trace.txt
cat $OUT/td_overlap.txt
AFX_LIB=$V/libafx_oldtd.so PROF_DB_HOOK=tools/td_overlap.py timeout -k 10 300 bash tools/prof_cmd.sh ${TAG}_cfg4old "" python bench.py --config 4 --clips 100 --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_cfg4old/hook.txt $OUT/td_overlap_oldtd.txt
cat $OUT/td_overlap_oldtd.txt
one4() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.3f'%(d['value'],d['ms_per_step']))"
}
one4 "shipped hwq default" AFX_X=0 | tee -a $OUT/cfg4_ab.txt
one4 "shipped hwq 8" GPU_MAX_HW_QUEUES=8 | tee -a $OUT/cfg4_ab.txt
one4 "shipped hwq 2" GPU_MAX_HW_QUEUES=2 | tee -a $OUT/cfg4_ab.txt
one2() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-legacy --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label: value %.5g ms/step %.4f kernel_ms %.4f sustained_ms %.4f check %s'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['sustained_ms'],d['oracle_check']['clip0_max_rel_err']))"
}
for r in 1 2 3; do
  one2 shipped AFX_X=0 | tee -a $OUT/headline_ab.txt
  one2 nonop AFX_LIB=$V/libafx_nonop.so | tee -a $OUT/headline_ab.txt
done
for r in 1 2; do
  for n in shipped nonop4k; do
    L=""; [ "$n" != shipped ] && L="AFX_LIB=$V/libafx_$n.so"
    echo "[$n] $(env $L timeout -k 10 120 python tools/bench_nfft.py 12 1024 2>&1 | tail -n 1)" | tee -a $OUT/nfft4096.txt
  done
done
echo "[legacy shipped] $(timeout -k 10 240 python tools/legacy_bench.py 1000 2>&1 | tail -n 1 | cut -c300-420)" | tee -a $OUT/legacy.txt
echo "[legacy no staging] $(AFX_NO_STAGING=1 timeout -k 10 240 python tools/legacy_bench.py 1000 2>&1 | tail -n 1 | cut -c300-420)" | tee -a $OUT/legacy.txt
timeout -k 10 300 python -m pytest tests/test_bft_gpu.py tests/test_spectrogram_gpu.py tests/test_xxcc_gpu.py -q -m gpu -x 2>&1 | tail -n 3
