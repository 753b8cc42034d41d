"""Objects on different streams must not change each other's results.  (Named to run last: these tests probe a
hardware behaviour with child processes and micro kernels; the parity suite proper is recorded before them.)

Round 3: the packed-f32 butterflies (afx_asm.h pk_add_mi / pk_add_pi, then v_pk_add_f32 with an op_sel half swap)
returned wrong sums in lanes 48-63 whenever a kernel that streams v_mfma + ds_read_b128 (the time-domain CWT kernel,
the CQT f16 octave kernels) really ran beside them -- between two objects on two streams, or inside one batched CWT
call with its time-domain scales on a side stream (profiles/r03_pk_add_opsel.txt).  The probes run in a child
process with 8 HIP hardware queues, so that the streams do not share (and serialise on) one queue."""
import os
import shutil
import subprocess
import sys

import pytest

from tests.conftest import parity_log

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _child(args, timeout=300):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    res = subprocess.run(args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return res.returncode, res.stdout.decode(errors="replace")


@pytest.mark.parametrize("partner", ["full", "seq", "cqt", "mel"])
def test_fft_path_cwt_is_bitwise_stable_beside_other_kernels(partner):
    """an FFT-path-only CWT object (48 low scales) on one stream, `partner` on another: every output bit equals the
    solo run's (partner full: an 84-scale CWT object incl. its time-domain kernel; seq: time-domain-only object then
    FFT-only object on one stream; cqt: the CQT + chroma call; mel: the fused mel + MFCC call)"""
    rc, out = _child([sys.executable, os.path.join("tools", "gpu_concurrency2.py"), partner, "48"])
    assert rc == 0, out[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("RESULT")]
    assert lines and lines[-1].rstrip().endswith("wrong elements 0"), out[-2000:]


def test_packed_adds_of_afx_asm_are_exact_beside_mfma_and_lds_traffic(tmp_path):
    """tools/micro/mfma_corun.hip: dft16 / cmul / the quarter-turn adds as shipped, beside back-to-back
    v_mfma + ds_read_b128 partners, compared bitwise with their solo runs (needs hipcc on the box)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    exe = str(tmp_path / "mfma_corun")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "audioflux_amd", "csrc", "hip"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "micro", "mfma_corun.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    rc, out = _child([exe])
    assert rc == 0, out[-2000:]
    shipped = [ln for ln in out.splitlines() if ln.startswith("victim [shipped:")]
    assert len(shipped) >= 40
    bad = [ln for ln in shipped if "]: 0 wrong words" not in ln]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("victim", ["mel", "stft", "ceps", "cqt", "cwt", "spec"])
def test_every_family_is_bitwise_stable_beside_matrix_core_kernels(victim):
    """tools/gpu_concurrency3.py: the family's batched device call on one stream, a full CWT object / the CQT + chroma
    call on another; every output bit equals the solo run's (before the operand-select rule of afx_asm.h was enforced
    on compiler-generated code too, the STFT and cepstrogram wave kernels failed this beside the CWT)"""
    rc, out = _child([sys.executable, os.path.join("tools", "gpu_concurrency3.py"), victim])
    assert rc == 0, out[-2000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("RESULT")]
    assert len(lines) == 2 and all(ln.split("wrong elements ")[1].startswith("0 ") for ln in lines), out[-2000:]


def test_operand_select_rule_on_the_device(tmp_path):
    """tools/micro/pk_forms_corun.hip: every packed-f32 form that keeps the rule of afx_asm.h (not op_sel[0] = 0 with
    op_sel[1] = 1) is exact beside the v_mfma + ds_read_b128 partner -- what tests/test_isa_forms.py relies on"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    exe = str(tmp_path / "pk_forms_corun")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tools", "micro", "pk_forms_corun.hip"),
                           "-o", exe], stderr=subprocess.DEVNULL)
    rc, out = _child([exe])
    assert rc == 0, out[-2000:]
    import re
    rows = [ln for ln in out.splitlines() if ln.startswith("v_pk_")]
    assert len(rows) >= 26
    for ln in rows:
        m = re.search(r"op_sel:\[([01]),([01])", ln)
        keeps_rule = not (m and m.group(1) == "0" and m.group(2) == "1")
        if keeps_rule:
            assert re.search(r"\)\s+0 wrong words", ln), ln
    # INFORMATIONAL (never fails the suite): do the forms that BREAK the rule go wrong on THIS box?  The claim of
    # profiles/r03_pk_add_opsel.txt rests on the boxes of one lease; every run of the suite adds a data point -- fail
    # counts per form, the device's name / firmware as rocm-smi reports them -- to the parity log and to stdout.
    breaking = []
    for ln in rows:
        m = re.search(r"op_sel:\[([01]),([01])", ln)
        if m and m.group(1) == "0" and m.group(2) == "1":
            w = re.search(r"\)\s+(\d+) wrong words", ln)
            breaking.append((ln.split("  ")[0].strip()[:90], int(w.group(1)) if w else -1))
    ident = {}
    try:
        smi = subprocess.run(["rocm-smi", "--showproductname", "--showfwinfo", "--showserial", "--json"], capture_output=True, text=True,
                             timeout=60).stdout
        import json
        card = next(iter(json.loads(smi).values()))
        ident = {k: card[k] for k in card if any(t in k.lower() for t in ("series", "sku", "serial", "mec", "smc", "vbios", "sdma"))}
    except Exception as e:  # (no rocm-smi, or another output format: the fail counts are still logged)
        ident = {"rocm-smi": repr(e)[:80]}
    print("operand-select erratum probe on this box:", breaking, ident)
    parity_log("operand-select erratum: wrong words of the rule-BREAKING packed-f32 forms beside v_mfma + ds_read_b128 (informational)",
               float(sum(max(n, 0) for _, n in breaking)), 1e300, "informational", {"forms": breaking, "device": ident})
