"""The shipped device code must not contain a packed-f32 instruction with op_sel[0] = 0 and op_sel[1] = 1.

Round 3 (csrc/hip/afx_asm.h, profiles/r03_pk_add_opsel.txt): on gfx950 a v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 whose
low result lane takes src0's low half and src1's HIGH half returns wrong values in lanes 48-63 while another wave of the
CU streams v_mfma + ds_read_b128 -- exact alone, wrong beside the time-domain CWT kernel or the CQT f16 kernels.  The
hand-written helpers order their operands accordingly and the library is built without the SLP vectoriser; this test
disassembles every code object of the built library and fails on any instruction that breaks the rule, wherever it
came from."""
import os
import re
import shutil
import subprocess

import pytest

import audioflux_amd._lib as _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32)\b([^\n;]*)")
SEL = re.compile(r"\bop_sel:\[([01]),([01])")


def vulnerable(line):
    m = PK.search(line)
    if not m:
        return False
    s = SEL.search(m.group(2))
    return bool(s) and s.group(1) == "0" and s.group(2) == "1"


def test_the_rule_matcher():
    assert vulnerable("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
    assert vulnerable("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")
    assert vulnerable("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert not vulnerable("v_pk_fma_f32 v[0:1], v[2:3], 1.0, v[6:7] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")
    assert not vulnerable("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")
    assert not vulnerable("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
    assert not vulnerable("v_pk_add_f32 v[0:1], v[2:3], v[4:5]")


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_no_packed_f32_instruction_breaks_the_operand_select_rule(tmp_path):
    lib = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f)
    assert len(objs) >= 15, objs  # one code object per .hip file
    packed, bad = 0, []
    for f in objs:
        dis = subprocess.run([OBJDUMP, "-d", str(tmp_path / f)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        sym = "?"
        for line in dis.splitlines():
            if line.endswith(">:"):
                sym = line.split("<")[-1][:-2]
            if "v_pk_" in line:
                packed += 1
                if vulnerable(line):
                    bad.append((f, sym[:80], line.strip()[:120]))
    assert packed > 20000, packed  # the FFT kernels are made of these
    assert not bad, f"{len(bad)} instructions break the rule, e.g. {bad[:5]}"


# kernels whose frame loops must not touch scratch memory (demangled-name fragment -> the instantiations meant)
NO_SCRATCH = {
    "k_stft_band_4k2": "n_fft 4096, every instantiation incl. the spectrum stores (round 5: replaces k_stft_band_4k's 68 B per lane; the STFT epilogue "
                       "spilled 300-440 B per lane before its stores went through scalar-register bases)",
    "k_stft_band_1k": "n_fft 1024, every instantiation (round 5: the (32, 32) / (72, 32) variants spilled 56-64 B per lane before the band sums were pinned)",
    "k_stft_band_512": "n_fft 512, every instantiation",
    "k_stft_mel_v2": "n_fft 2048, every instantiation (round 5: the complex ones spilled 264 B before the imaginary row moved to LDS)",
    "k_cepstrogram_w4096": "cepstrogram n_fft 4096 (round 5: 52 B per lane before the wave index became scalar)",
    "k_cepstrogram_w2048": "cepstrogram n_fft 2048",
    "k_cqt_pyramid": None,  # (reported, not asserted: 160 B of loop invariants by design)
}


def test_hot_kernels_do_not_spill(tmp_path):
    """no scratch_* instruction in the fused STFT kernels of any size (the headline instantiation the
    bench times, k_stft_mel_v2<48, 16, 4, false, 1, false>, named apart) and the cepstrogram wave kernels"""
    lib = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f)
    seen, bad = {}, []
    for f in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--demangle", str(tmp_path / f)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        sym = "?"
        for line in dis.splitlines():
            if line.endswith(">:"):
                sym = line.split("<", 1)[-1][:-2]
            if "scratch_" in line:
                seen[sym] = seen.get(sym, 0) + 1
    headline = [s for s in seen if "k_stft_mel_v2<48, 16, 4, false, 1, false" in s]
    assert not headline, headline
    for frag, why in NO_SCRATCH.items():
        if why is None:
            continue
        hit = {s: n for s, n in seen.items() if frag + "<" in s or frag + "(" in s}
        if hit:
            bad.append((frag, why, sorted(hit.items())[:3]))
    assert not bad, bad


# ---- loads issued by hand inside asm statements (afx_asm.h rows_shift_fetch / rows_fetch_all, the LOAD_*_B128 macros): the compiler
# believes their destinations are defined AT the statement, the hand-written s_waitcnt vmcnt comes statements later.  Nothing may touch
# a destination register while its load is in flight (a copy, spill or coalesce there would read stale registers).
REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def inflight_hazards(lines):
    """linear scan of one kernel's disassembly with gfx9's in-order vmcnt: (instruction, registers) pairs that read or write the
    destination of a vector-memory load that no s_waitcnt has covered yet.  Branch targets are not followed (conservative for
    the straight-line frame loops this is meant for: a back edge re-enters with whatever is outstanding, as the hardware does)."""
    fifo, bad = [], []  # outstanding VM operations in issue order: set of destination registers (empty for stores)
    for raw in lines:
        ins = raw.split("//")[0].strip()
        if not ins or ins.endswith(":"):
            continue
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
            continue
        if op.startswith(("s_endpgm", "s_branch", "s_setpc")):  # the next line is not this one's successor
            fifo = []
            continue
        is_load = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
        # (a load's own destination is exempt: loads return in order, so a second load into a register in flight -- the other
        #  arm of an if / else in this linear scan, or a real overwrite -- is served after the first; its ADDRESS operands count)
        used = _regs(ins.split(None, 1)[1].split(",", 1)[1]) if is_load and "," in ins else _regs(ins)
        busy = set().union(*fifo) if fifo else set()
        if used & busy:
            bad.append((ins[:100], sorted(used & busy)))
        if is_load:
            dst = ins.split(None, 1)[1].split(",")[0]
            fifo.append(_regs(dst))
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
            fifo.append(set())
    return bad


def test_the_inflight_scanner():
    ok = ["global_load_dwordx2 v[12:13], v[40:41], off", "v_mov_b64 v[0:1], v[2:3]", "s_waitcnt vmcnt(0)", "v_pk_mul_f32 v[4:5], v[12:13], v[6:7]"]
    assert not inflight_hazards(ok)
    assert inflight_hazards(["global_load_dwordx2 v[12:13], v[40:41], off", "v_mov_b32 v3, v13", "s_waitcnt vmcnt(0)"])
    # in-order counter: vmcnt(1) covers the first of two loads only
    two = ["global_load_dword v1, v[8:9], off", "global_load_dword v2, v[8:9], off offset:4", "s_waitcnt vmcnt(1)", "v_add_f32 v3, v1, v1", "v_add_f32 v3, v2, v2"]
    assert [b[1] for b in inflight_hazards(two)] == [[2]]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_no_instruction_touches_a_register_whose_hand_issued_load_is_in_flight(tmp_path):
    """the n_fft 4096 kernels (rows_shift_fetch / rows_fetch_all: 4 + 16 loads in one asm statement, waited for by hand before the row
    stores), the kernels with the cepstrum block's hand-waited loads (k_stft_mel_v2) and the dense-bank product's rings of loads in
    flight around its loop (k_gemm_bank_bf16x3: LOAD_B128_SLOT, waited for by count): every instantiation in the shipped library"""
    lib = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f)
    checked, bad = 0, []
    for f in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--demangle", str(tmp_path / f)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        sym, body = None, []
        for line in dis.splitlines() + ["<end>:"]:
            if line.endswith(">:"):
                if sym and ("k_stft_band_4k2<" in sym or "k_stft_mel_v2<" in sym or "k_gemm_bank_bf16x3" in sym):
                    checked += 1
                    h = inflight_hazards(body)
                    if h:
                        bad.append((sym[:90], h[:3]))
                sym, body = line.split("<", 1)[-1][:-2], []
            elif sym:
                body.append(re.sub(r"^\s*[0-9a-fA-F]+:\s+", "", line) if re.match(r"^\s*[0-9a-fA-F]+:\s", line) else line)
    assert checked >= 21, checked
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_the_gemm_rings_stay_untouched_until_their_final_wait(tmp_path):
    """k_gemm_bank_bf16x3 leaves its loop with loads in flight whose values nobody uses (behind the last k-step): to the compiler
    their destination registers are free, and an epilogue whose first instructions it scheduled ahead of the hand-written
    `s_waitcnt vmcnt(0)` returned wrong values on the device (the linear scanner above starts afresh behind a branch and cannot see
    it).  The pins behind that wait keep the rings alive: between the epilogue's label and the wait no instruction may write a
    register any global load of the kernel targets."""
    lib = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f)
    found = 0
    for f in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--demangle", str(tmp_path / f)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        sym, body = None, []
        for line in dis.splitlines() + ["<end>:"]:
            if line.endswith(">:"):
                if sym and "k_gemm_bank_bf16x3" in sym:
                    found += 1
                    ins = [re.sub(r"^\s*[0-9a-fA-F]+:\s+", "", l).split("//")[0].strip() for l in body]
                    ring = set()
                    for i in ins:
                        if i.startswith("global_load_dwordx4"):
                            ring |= _regs(i.split(None, 1)[1].split(",")[0])
                    assert len(ring) >= 80, len(ring)  # 2 x 4 A quads + 2 x 6 bank fragments of four registers
                    waits = [n for n, i in enumerate(ins) if i.startswith("s_waitcnt") and "vmcnt(0)" in i]
                    assert waits, "no final drain of the rings"
                    last = waits[-1]
                    start = last
                    while start > 0 and not ins[start - 1].startswith(("s_cbranch", "s_branch", "s_barrier")):
                        start -= 1
                    early = []
                    for i in ins[start:last]:
                        if not i or i.startswith("s_"):
                            continue
                        dst = i.split(None, 1)[1].split(",")[0] if " " in i else ""
                        if _regs(dst) & ring:
                            early.append(i[:90])
                    assert not early, early[:5]
                sym, body = line.split("<", 1)[-1][:-2], []
            elif sym:
                body.append(line)
    assert found == 1, found

