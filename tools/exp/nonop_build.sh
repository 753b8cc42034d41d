#!/bin/bash
# MEASUREMENT build (never shipped): a variant library whose <unit>.hip device code has the `s_nop 0` next to inline-asm
# statements removed.  The compiler pads every inline-asm VALU result with one wait state before an adjacent use (it
# assumes a dst_sel forwarding hazard it cannot rule out); the packed-f32 statements of afx_asm.h write whole 64-bit
# results, for which the hardware interlocks.  This tells what the pads cost, i.e. the most any source-level
# re-arrangement of those statements could win.
#   tools/exp/nonop_build.sh afx_melfused2 nonop     -> audioflux_amd/lib/variants/libafx_nonop.so
set -e
UNIT=$1; NAME=${2:-nonop}
ROOT=$(cd $(dirname $0)/../.. && pwd)
B=$ROOT/build/v_$NAME; mkdir -p $B $ROOT/audioflux_amd/lib/variants
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-slp-vectorize -I$ROOT/include -I$ROOT/audioflux_amd/csrc/hip -I$ROOT/audioflux_amd/csrc/host"
SRC=$ROOT/audioflux_amd/csrc/hip/$UNIT.hip
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $SRC -o $B/$UNIT.dev.s
python3 - $B/$UNIT.dev.s $B/$UNIT.dev2.s <<'PY'
import sys, re
L = open(sys.argv[1]).read().split('\n')
def code(i, step):
    i += step
    while 0 <= i < len(L) and (not L[i].strip() or L[i].strip().startswith(';') and not L[i].strip().startswith(';;#ASM')):
        i += step
    return L[i].strip() if 0 <= i < len(L) else ''
out, dropped = [], 0
for i, l in enumerate(L):
    if l.strip() == 's_nop 0' and (code(i, -1).startswith(';;#ASMEND') or code(i, 1).startswith(';;#ASMSTART')):
        dropped += 1
        continue
    out.append(l)
open(sys.argv[2], 'w').write('\n'.join(out))
print('dropped', dropped, 's_nop 0')
PY
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/$UNIT.dev2.s -o $B/$UNIT.dev2.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/$UNIT.hsaco $B/$UNIT.dev2.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/$UNIT.hsaco -output=$B/$UNIT.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -c $SRC -Xclang -fcuda-include-gpubinary -Xclang $B/$UNIT.hipfb -o $B/hip_$UNIT.o
OBJS=$(ls $ROOT/build/csrc/*.o | grep -v "hip_$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/audioflux_amd/lib/variants/libafx_$NAME.so $OBJS $B/hip_$UNIT.o -lm
ls -la $ROOT/audioflux_amd/lib/variants/libafx_$NAME.so
