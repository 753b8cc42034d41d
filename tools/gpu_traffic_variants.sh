#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/prof_traffic.py) of one bench configuration for the shipped
# library and for variant builds (audioflux_amd/lib/variants/libafx_<name>.so): which class of accesses the bytes belong to.
#   gpurun -- 'bash tools/gpu_traffic_variants.sh r06d 5 kocqt1 kocqt2 kocqt4'
set -u
TAG=$1; CFG=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/call_$TAG; mkdir -p $OUT
export TMPDIR=/tmp AFX_ROUND=tmp
V=$PWD/audioflux_amd/lib/variants
one() { local label=$1; shift
  env "$@" timeout -k 10 240 python tools/prof_traffic.py $CFG > $OUT/traffic_$label.log 2>&1
  python - <<P | tee -a $OUT/traffic_variants.txt
import json
try:
    d = json.load(open("gpurun_out/tmp_bench_cfg$CFG" + "_pmc.json"))
    print("$label: 2 x FETCH %.3f GB  WRITE %.3f GB per step" % (2 * d["fetch_kib_per_step"] * 1024 / 1e9, d["write_kib_per_step"] * 1024 / 1e9))
except Exception as e:
    print("$label: failed", e)
P
  rm -f gpurun_out/tmp_bench_cfg${CFG}_pmc.json
}
one shipped AFX_X=0
for n in "$@"; do one $n AFX_LIB=$V/libafx_$n.so; done
