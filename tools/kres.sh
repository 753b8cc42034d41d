#!/bin/bash
# kernel resource usage (VGPR/SGPR/LDS/scratch) of one .hip file: tools/kres.sh afx_cwt [filter]
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$ROOT/include -I$ROOT/audioflux_amd/csrc/hip -I$ROOT/audioflux_amd/csrc/host \
  ${KRES_EXTRA:-} --save-temps=obj -c $ROOT/audioflux_amd/csrc/hip/$1.hip -o $T/k.o 2>&1 | grep -E "error|warning" 
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name):" $T/*gfx950*.s | paste - - - - - | sed 's/\s\+/ /g' | grep "${2:-.}"
cp $T/*gfx950*.s /tmp/isa/$1.s 2>/dev/null
rm -rf $T
