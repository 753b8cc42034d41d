#!/bin/bash
# measurement / A-B build of the library with extra compiler flags: only the named .hip / .c units are recompiled, the
# rest is taken from the shipped build's objects.  -> audioflux_amd/lib/variants/libafx_<name>.so (AFX_LIB=... selects it)
#   tools/build_variant.sh kocqt4 -DAFX_KO_CQT=4 afx_cqt_f16
set -eu
NAME=$1; EXTRA=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/build/var_$NAME
mkdir -p $B $ROOT/audioflux_amd/lib/variants
cp -p $ROOT/build/csrc/*.o $B/
for u in "$@"; do rm -f $B/hip_$u.o $B/host_$u.o; done
# (objects copied with -p keep their time stamps: newer than the sources unless a source was edited since the shipped build)
make -s -C $ROOT/audioflux_amd/csrc BUILD=$B TARGET=$ROOT/audioflux_amd/lib/variants/libafx_$NAME.so EXTRA="$EXTRA" > $B/make.log 2>&1 || { tail -n 20 $B/make.log; exit 1; }
echo "built libafx_$NAME.so"
