#!/bin/bash
# cfg 4 at several sizes / chunk groups per call, shipped library against variants: bash tools/gpu_cfg4_group.sh "<clips list>" "<group list>" [variant ...]
cd $GRAFT_REPO_ROOT
CL=${1:-1000}; GR=${2:-32}; shift; shift
for c in $CL; do for g in $GR; do for l in shipped "$@"; do
  L=""; [ $l != shipped ] && L=$PWD/audioflux_amd/lib/variants/libafx_$l.so
  echo "clips $c GROUP $g lib $l: $(AFX_LIB=$L AFX_CFG4_GROUP=$g timeout 300 python bench.py --config 4 --clips $c --no-cpu-baseline --steps 5 --warmup 2 --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.5g ms/step %.3f'%(d['value'],d['ms_per_step']))")"
done; done; done
