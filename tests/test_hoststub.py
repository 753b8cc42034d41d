"""CPU-only: the C host objects (audioflux_amd/csrc/host/*.c) built with AddressSanitizer + UBSan against a stand-in
device layer (tests/hoststub/gen_stub.py: "device" memory is host memory, kernels do no arithmetic, the CQT launchers
touch every range they are handed).

  * tests/hoststub/driver_cqt.c drives every CQT entry point -- passes (AFX_CQT_CHUNK), the f32 / f16 glue
    (AFX_CQT_F32), chroma with a changing class count, free -- with leak detection on;
  * the whole `-m gpu` suite then runs against the sanitized library (AFX_HOSTSTUB=1: parity assertions are skipped,
    tests that need torch end with ImportError): every constructor and every first compute call of every test case
    goes through the host code under the sanitizers.  Results are meaningless there; a sanitizer report is a failure.
  * tests/hoststub/cqt_functional.c replaces the CQT launchers by double-precision loops that do what afx_device.h
    says the kernels do: the CQT host code then has to reproduce the reference's golden vectors on every launch path.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "hoststub")
INC = [f"-I{ROOT}/include", f"-I{ROOT}/audioflux_amd/csrc/hip", f"-I{ROOT}/audioflux_amd/csrc/host"]
SAN = ["-std=c99", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off"]


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or _asan_runtime() is None,
                                reason="needs gcc with the AddressSanitizer runtime")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("hoststub"))
    stub = os.path.join(tmp, "stub.c")
    subprocess.run([sys.executable, os.path.join(HERE, "gen_stub.py"),
                    os.path.join(ROOT, "audioflux_amd", "csrc", "hip", "afx_device.h"), stub], check=True)
    host = sorted(os.path.join(ROOT, "audioflux_amd", "csrc", "host", f)
                  for f in os.listdir(os.path.join(ROOT, "audioflux_amd", "csrc", "host")) if f.endswith(".c"))
    lib = os.path.join(tmp, "libafx_stub.so")
    r = subprocess.run(["gcc", *SAN, "-shared", "-fPIC", *INC, *host, stub, "-lm", "-o", lib],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exes = {}
    for name in ("driver_cqt", "driver_batch"):
        exes[name] = os.path.join(tmp, name)
        r = subprocess.run(["gcc", *SAN, *INC, os.path.join(HERE, name + ".c"), lib, f"-Wl,-rpath,{tmp}", "-lm", "-o",
                            exes[name]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return tmp, lib, exes


def _run_driver(exe, env, knobs):
    e = dict(os.environ)
    for k in knobs:
        e.pop(k, None)
    e.update(kv.split("=") for kv in env.split())
    e["ASAN_OPTIONS"] = "detect_leaks=1"
    r = subprocess.run([exe], capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "OK" in r.stdout, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-3000:]


@pytest.mark.parametrize("env", ["", "AFX_CQT_CHUNK=2", "AFX_CQT_CHUNK=3", "AFX_CQT_F32=1", "AFX_NO_FUSED=1"])
def test_cqt_host_logic_is_clean_under_sanitizers(built, env):
    _run_driver(built[2]["driver_cqt"], env,
                ("AFX_CQT_CHUNK", "AFX_CQT_F32", "AFX_NO_FUSED"))


@pytest.mark.parametrize("env", ["", "AFX_SCRATCH_MB=1", "AFX_NO_FUSED=1", "AFX_CWT_GROUP=2", "AFX_CWT_GROUP=1",
                                 "AFX_CWT_NARROW_MAX=0"])
def test_device_pointer_entry_points_are_clean_under_sanitizers(built, env):
    """tests/hoststub/driver_batch.c: the ...BatchDevice calls of include/afx_batch.h (which the Python GPU tests reach
    through torch tensors) with exactly-sized buffers: mel + MFCC in one call, dense-bank route in several chunks,
    temporal features, complex results, STFT / inverse STFT, spectrogram object, cepstrogram, reassignment, CWT at
    2^12 and 2^16 (chunk groups, chains, narrow-band plan on / off, padded), PWT, WSST"""
    _run_driver(built[2]["driver_batch"], env,
                ("AFX_SCRATCH_MB", "AFX_NO_FUSED", "AFX_CWT_GROUP", "AFX_CWT_NARROW_MAX"))


def test_every_gpu_test_case_drives_clean_host_code(built):
    tmp, lib, _ = built
    e = dict(os.environ)
    e.update(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0", AFX_HOSTSTUB="1", AFX_LIB=lib,
             AFX_HIP_RUNTIME="system", UBSAN_OPTIONS="print_stacktrace=1")
    e.pop("AFX_PARITY_LOG", None)
    log = os.path.join(tmp, "suite.log")
    with open(log, "w") as f:
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-s", "-p", "no:cacheprovider",
                            "--ignore=tests/dropin", "--ignore=tests/test_dist_cpu.py", "--ignore=tests/test_dist_gpu.py",
                            "--ignore=tests/test_hoststub.py"],
                           stdout=f, stderr=subprocess.STDOUT, env=e, cwd=ROOT, timeout=1500)
    out = open(log, errors="replace").read()
    assert r.returncode in (0, 1), f"the suite did not run to its end (rc {r.returncode}):\n{out[-3000:]}"
    bad = [ln for ln in out.splitlines() if "AddressSanitizer" in ln or "runtime error" in ln]
    assert not bad, "\n".join(bad[:20])
    import re
    m = re.search(r"(\d+) failed, (\d+) passed", out) or re.search(r"(\d+) passed", out)
    assert m, out[-2000:]
    ran = sum(int(g) for g in m.groups())
    assert ran >= 150, f"only {ran} tests reached the library"


@pytest.mark.parametrize("seed", [1, 35])
def test_constructor_arguments_fuzz(built, seed):
    """tests/hoststub/fuzz_ctor.py: random valid / borderline / invalid constructor arguments for every object through
    raw ctypes against the sanitized host code: a handle and status 0, or a refusal -- never a crash or a sanitizer
    report.  (It found: the default hop fftLength / 4 = 0 at fftLength 2 dividing by zero in the frame count, as in the
    reference; xxccObj_new building a [num, num] DCT for any num; seed 35: bftObj_new(radix2Exp 1, isReassign 1), whose
    reassignment object silently falls back to 2^12 -- as in the reference -- and wrote 15905 x 2049 results into
    planes sized for fftLength 2.  AFX_FUZZ_VERBOSE=1 prints every argument drawn.)"""
    tmp, lib, _ = built
    e = dict(os.environ)
    e.update(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0", AFX_LIB=lib, AFX_FUZZ_SEED=str(seed),
             UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_ctor.py"), "40"], capture_output=True, text=True, env=e,
                       timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "constructed" in out and "\nOK" in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]


@pytest.mark.parametrize("seed", [1])
def test_compute_call_edges_fuzz(built, seed):
    """tests/hoststub/fuzz_calls.py: zero / one-sample / shorter-than-a-frame inputs, batch 0, overlapping clip strides,
    NULL outputs, padding and streaming switches on ordinary BFT / STFT / CQT / XXCC objects"""
    tmp, lib, _ = built
    e = dict(os.environ)
    e.update(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0", AFX_LIB=lib, AFX_FUZZ_SEED=str(seed),
             UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_calls.py"), "40"], capture_output=True, text=True, env=e,
                       timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "accepted" in out and "\nOK" in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]


@pytest.fixture(scope="module")
def functional(tmp_path_factory):
    """the host objects + tests/hoststub/cqt_functional.c (CQT launchers that COMPUTE, in double-precision loops, what
    afx_device.h says the kernels do) + the stand-in for the rest of the device layer, under ASan / UBSan"""
    tmp = str(tmp_path_factory.mktemp("functional"))
    stub = os.path.join(tmp, "stub.c")
    subprocess.run([sys.executable, os.path.join(HERE, "gen_stub.py"),
                    os.path.join(ROOT, "audioflux_amd", "csrc", "hip", "afx_device.h"), stub, "--functional-cqt"], check=True)
    host = sorted(os.path.join(ROOT, "audioflux_amd", "csrc", "host", f)
                  for f in os.listdir(os.path.join(ROOT, "audioflux_amd", "csrc", "host")) if f.endswith(".c"))
    lib = os.path.join(tmp, "libafx_functional.so")
    r = subprocess.run(["gcc", *SAN, "-shared", "-fPIC", *INC, *host, stub, os.path.join(HERE, "cqt_functional.c"), "-lm",
                        "-o", lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the same without the sanitizers, for processes that cannot run under an LD_PRELOADed ASan runtime (the
    # reference's Python wrapper imports matplotlib: C++ exceptions before libstdc++ is loaded)
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-ffp-contract=off", "-shared", "-fPIC", *INC, *host, stub,
                        os.path.join(HERE, "cqt_functional.c"), "-lm", "-o", lib.replace(".so", "_plain.so")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return lib


@pytest.mark.parametrize("env,path", [("", "octave_f16"), ("AFX_CQT_CHUNK=2", "octave_f16"), ("AFX_CQT_F32=1", "octave_f32"),
                                      ("AFX_NO_FUSED=1", "octave_f32")])
def test_cqt_host_glue_meets_the_golden_vectors_with_functional_launchers(functional, env, path):
    """tests/hoststub/functional_cqt.py: the reference's golden CQT / chroma vectors through the C host code with the
    kernels replaced by their contracts.  Every launch path's arguments (f16 image words and column multipliers,
    float32 image, spectral kernels, passes with a row stride that is not a multiple of four) must reproduce the
    reference to 1e-5.  `path` = the launcher the default 84-bin plan must have used."""
    e = dict(os.environ)
    for k in ("AFX_CQT_CHUNK", "AFX_CQT_F32", "AFX_NO_FUSED"):
        e.pop(k, None)
    if env:
        e.update(kv.split("=") for kv in env.split())
    e.update(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0", AFX_LIB=functional, UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "functional_cqt.py"), "c84_32k_area", "c84_44k_none_noscale",
                        "c48_16k_area", "c72_24bpo_hop200", "stream:c84_32k_area", "stream:c48_16k_area@192"],
                       capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "\nOK" in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    import re
    m = re.search(r"c84_32k_area launches: octave_f16 (\d+) octave_f32 (\d+) chroma (\d+)", out)
    assert m, out[-2000:]
    n = dict(zip(("octave_f16", "octave_f32", "chroma"), map(int, m.groups())))
    assert n[path] > 0 and all(n[k] == 0 for k in ("octave_f16", "octave_f32") if k != path), n


@pytest.mark.parametrize("env", ["", "AFX_CQT_F32=1"])
def test_reference_wrapper_drives_the_cqt_host_code_on_the_cpu(functional, tmp_path, env):
    """tests/dropin/flows.py cqt_functional: the reference's own unmodified Python wrapper (set_fft_lib) on the host
    objects + functional CQT launchers, against the stock library through the same wrapper: CQT, chroma, frequency
    table, frame count.  (The GPU version of this, every flow on the real kernels, is tests/dropin/test_dropin.py.)"""
    import numpy as np
    dropin = os.path.join(ROOT, "tests", "dropin")
    sys.path.insert(0, dropin)
    import flows
    if not (os.path.exists(flows.STOCK) and (os.path.exists(flows.WRAPPER_ZIP) or os.path.isdir("/root/reference/python/audioflux"))):
        pytest.skip("needs oracle/_ref (make -C oracle)")
    e = dict(os.environ)
    for k in ("AFX_CQT_CHUNK", "AFX_CQT_F32", "AFX_NO_FUSED"):
        e.pop(k, None)
    if env:
        e.update(kv.split("=") for kv in env.split())
    e.update(AFX_LIB=functional.replace(".so", "_plain.so"))
    out = str(tmp_path / "flows.npz")
    r = subprocess.run([sys.executable, os.path.join(dropin, "flows.py"), str(tmp_path / "pkg"), out, "cqt_functional"],
                       capture_output=True, text=True, env=e, timeout=900, cwd=str(tmp_path))
    log = r.stdout + r.stderr
    assert r.returncode == 0 and "flows done" in log, log[-3000:]
    d = np.load(out)
    assert np.array_equal(d["stock/T"], d["mi355x/T"]) and np.array_equal(d["stock/fre"], d["mi355x/fre"])
    for k in ("cqt", "chroma", "core_cqt_abs", "core_chroma"):
        want, got = d[f"stock/{k}"], d[f"mi355x/{k}"]
        assert got.shape == want.shape and got.dtype == want.dtype, k
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= 1e-5, (k, err)


CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang (UBSan integer checks)")
def test_size_arithmetic_at_288_gb_scale(tmp_path):
    """tests/hoststub/driver_scale.c: every device-pointer entry point at the BASELINE sizes, 20x, and batch sizes whose
    element counts pass 2^31 / 2^32, against the stand-in device layer in DRY mode (address ranges without memory,
    launchers do nothing), built with clang -fsanitize=undefined,integer: no wrapped product, truncating conversion or
    sign change on the way to an allocation size or a pointer offset.  (The GPU tests run at one device's share.)"""
    tmp = str(tmp_path)
    stub = os.path.join(tmp, "stub.c")
    subprocess.run([sys.executable, os.path.join(HERE, "gen_stub.py"),
                    os.path.join(ROOT, "audioflux_amd", "csrc", "hip", "afx_device.h"), stub], check=True)
    host = sorted(os.path.join(ROOT, "audioflux_amd", "csrc", "host", f)
                  for f in os.listdir(os.path.join(ROOT, "audioflux_amd", "csrc", "host")) if f.endswith(".c"))
    exe = os.path.join(tmp, "driver_scale")
    r = subprocess.run([CLANG, "-std=gnu11", "-g", "-O1", "-DAFX_STUB_DRY", "-fsanitize=undefined,integer", "-fno-omit-frame-pointer",
                        "-ffp-contract=off", *INC, *host, stub, os.path.join(HERE, "driver_scale.c"), "-lm", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "\nOK" in r.stdout, out[-3000:]
    assert "runtime error" not in out, "\n".join(ln for ln in out.splitlines() if "runtime error" in ln)[:3000]


HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def launch_audit(tmp_path_factory):
    """the product's REAL host side -- C objects and the launchers of csrc/hip/*.hip, compiled --cuda-host-only -- with
    UBSan + clang's integer checks, linked against tests/hoststub/fake_hip.cpp (a HIP runtime that checks every
    launch configuration instead of launching) and tests/hoststub/driver_scale.c"""
    if not (os.path.exists(CLANG) and os.path.exists(HIPCC)):
        pytest.skip("needs hipcc / clang")
    tmp = str(tmp_path_factory.mktemp("audit"))
    san = ["-g", "-O1", "-fPIC", "-fsanitize=undefined,integer", "-fno-omit-frame-pointer"]
    exes, objs = _build_with_fake_hip(tmp, san, ("driver_scale", "driver_batch", "driver_cqt"))
    # the same objects as a shared library for the ctypes fuzzers (UBSan runtime as a shared object beside it)
    rt = os.path.dirname(subprocess.run([CLANG, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"], capture_output=True,
                                        text=True).stdout.strip())
    exes["lib"] = os.path.join(tmp, "libafx_audit.so")
    r = subprocess.run([CLANG + "++", "-shared", *san, "-shared-libsan", *objs, os.path.join(tmp, "fatbins.o"), "-lm", "-lpthread", "-ldl",
                        f"-Wl,-rpath,{rt}", "-o", exes["lib"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exes


@pytest.mark.parametrize("env,expect", [("", "k_stft_mel_v2"), ("AFX_NO_FUSED=1", "k_stft_generic"),
                                        ("AFX_CQT_F32=1", "k_cqt_octave_mfma"), ("AFX_CQT_CHUNK=7", "k_cqt_chroma"),
                                        ("AFX_CWT_NARROW_MAX=0", "k_cwt_inv_rows512"), ("AFX_SCRATCH_MB=64", "bf16x3")])
def test_launch_audit(launch_audit, env, expect):
    """every device-pointer entry point at the BASELINE sizes (must succeed), 20x, and far beyond the device's memory,
    through the real launchers: no launch configuration outside the HIP limits (block size, grid dimensions, dynamic
    LDS vs the raised attribute and the 160 KB of a CU), no UBSan / integer-check report in the launch arithmetic, and
    the kernel the switch is about was reached.  Launches
    with 2^32 or more threads in one dimension are rejected by the stand-in runtime as HIP rejects them; they may only
    come from batches whose buffers exceed the device (the size-generic STFT / cepstrogram / inverse-STFT launchers
    split such batches into launches of whole clips -- found here: 4 477 clips of 30 s at n_fft 512 / hop 128 used to
    fail)."""
    e = dict(os.environ)
    for k in ("AFX_NO_FUSED", "AFX_CQT_F32", "AFX_CWT_NARROW_MAX", "AFX_SCRATCH_MB", "AFX_CQT_CHUNK"):
        e.pop(k, None)
    if env:
        e.update(kv.split("=") for kv in env.split())
    e.update(AFX_QUIET="1", AFX_AUDIT_EXPECT=expect)
    r = subprocess.run([launch_audit["driver_scale"]], capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "\nOK" in r.stdout, out[-3000:]
    assert "runtime error:" not in out and "VIOLATION" not in out, "\n".join(
        ln for ln in out.splitlines() if "runtime error:" in ln or "VIOLATION" in ln)[:3000]
    # what HIP would reject comes only from the sizes beyond the device (the last entries of the size tables)
    rejected = [ln for ln in r.stdout.splitlines() if "refused (-3)" in ln]
    assert all(" at 2000000:" in ln or " at 400000:" in ln or " at 3000000:" in ln for ln in rejected), rejected


@pytest.mark.parametrize("driver,env", [("driver_batch", ""), ("driver_batch", "AFX_SCRATCH_MB=1"), ("driver_batch", "AFX_NO_FUSED=1"),
                                        ("driver_batch", "AFX_CWT_GROUP=1"), ("driver_batch", "AFX_CWT_GROUP=2"),
                                        ("driver_batch", "AFX_CWT_NARROW_MAX=0"), ("driver_cqt", ""), ("driver_cqt", "AFX_CQT_F32=1"),
                                        ("driver_cqt", "AFX_NO_FUSED=1"), ("driver_cqt", "AFX_CQT_CHUNK=2")])
def test_launch_audit_of_the_small_configurations(launch_audit, driver, env):
    """the configurations of tests/hoststub/driver_batch.c / driver_cqt.c (mel / gammatone / segment-plan / temporal banks
    at n_fft 512 ... 4096 -- every fused kernel family, real and complex results, with and without cepstra; STFT and inverse,
    spectrogram, reassignment at the same four sizes; cepstrogram at two sizes, CWT at 2^12 and 2^16 padded and not, PWT,
    WSST; CQT plans of 84 and 48 bins, short clips, odd strides, 12 and 6 chroma classes) through the real launchers
    and the checking HIP stand-in.  For the CQT kernels the stand-in also decodes the argument lists and keeps the
    happens-before relation of the streams: a launch that reads or writes a range another stream's launch writes,
    without an event or synchronisation between them, is a "FAKEHIP RACE" (the side-stream decimations of the default
    path); the same for the four-step CWT kernels, whose scale
    lists it reads from the retained uploads (forward batch -> narrow-band scales on a side stream + two-pass chunk
    groups alternating over the chain streams, each with its own intermediate)."""
    e = dict(os.environ)
    for k in ("AFX_NO_FUSED", "AFX_CQT_F32", "AFX_CWT_GROUP", "AFX_SCRATCH_MB", "AFX_CQT_CHUNK", "FAKEHIP_DROP_WAIT"):
        e.pop(k, None)
    if env:
        e.update(kv.split("=") for kv in env.split())
    e.update(AFX_QUIET="1", FAKEHIP_ORDER="1")
    r = subprocess.run([launch_audit[driver]], capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "OK" in r.stdout, out[-3000:]
    bad = ("runtime error:", "VIOLATION", "rejected as HIP", "RACE", "OVERRUN")
    assert not any(b in out for b in bad), "\n".join(ln for ln in out.splitlines() if any(b in ln for b in bad))[:3000]


@pytest.mark.parametrize("env,pair", [("AFX_CQT_PYRAMID=0", "k_cqt_decimate (write)  <->  k_cqt_octave_f16 (read)")])
def test_the_stream_order_check_sees_a_lost_wait(launch_audit, env, pair):
    """the detector's own test: with the second hipStreamWaitEvent of the run ignored (FAKEHIP_DROP_WAIT=2: "the octave
    product waits for the decimation that produced its input") the same driver must end in a race report (the
    per-octave schedule: the one-launch ladder, the default since round 4, has no stream edges to lose)"""
    e = dict(os.environ)
    for k in ("AFX_CQT_F32", "AFX_CQT_CHUNK", "AFX_NO_FUSED"):
        e.pop(k, None)
    if env:
        e.update(kv.split("=") for kv in env.split())
    e.update(AFX_QUIET="1", FAKEHIP_ORDER="1", FAKEHIP_DROP_WAIT="2")
    r = subprocess.run([launch_audit["driver_cqt"]], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 1 and "FAKEHIP RACE " + pair in r.stderr, (r.stdout + r.stderr)[-3000:]


def test_the_stream_order_check_sees_a_lost_wait_in_the_cwt_schedule(launch_audit):
    """as above for the CWT chains: one of the first dozen wait edges of tests/hoststub/driver_batch.c orders a chain's
    row pass behind the forward transform -- without it the run must report that pair"""
    e = dict(os.environ)
    for k in ("AFX_CWT_CHAINS", "AFX_CWT_GROUP", "AFX_CWT_OVERLAP", "AFX_CWT_NARROW_MAX", "AFX_NO_FUSED"):
        e.pop(k, None)
    seen = []
    for k in range(1, 13):
        e.update(AFX_QUIET="1", FAKEHIP_ORDER="1", FAKEHIP_DROP_WAIT=str(k))
        r = subprocess.run([launch_audit["driver_batch"]], capture_output=True, text=True, env=e, timeout=900)
        seen += [ln for ln in r.stderr.splitlines() if "FAKEHIP RACE k_cwt_fwd_rows (write)  <->  k_cwt_inv_" in ln]
        if seen:
            break
    assert seen, "no dropped wait edge produced a forward -> inverse race report"


@pytest.mark.parametrize("script,seed", [("fuzz_ctor.py", 5), ("fuzz_calls.py", 5)])
def test_fuzzers_through_the_real_launchers(launch_audit, script, seed):
    """the constructor / compute-call fuzzers against the launch-audit build: odd sizes, empty inputs, hops longer than
    a frame ... must not produce an empty grid, an oversized block, an LDS request beyond its attribute or an integer
    report in a launcher (40 seeds of each were run when this was written: none)"""
    e = dict(os.environ)
    e.update(AFX_QUIET="1", AFX_LIB=launch_audit["lib"], AFX_FUZZ_SEED=str(seed))
    r = subprocess.run([sys.executable, os.path.join(HERE, script), "25"], capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "\nOK" in out, out[-3000:]
    assert "runtime error:" not in out and "VIOLATION" not in out and "rejected as HIP" not in out, "\n".join(
        ln for ln in out.splitlines() if "runtime error:" in ln or "VIOLATION" in ln or "rejected" in ln)[:3000]


def _build_with_fake_hip(tmp, san, drivers):
    """host C objects + the real launchers (--cuda-host-only) + tests/hoststub/fake_hip.cpp + one executable per driver"""
    from concurrent.futures import ThreadPoolExecutor
    import re
    hipdir, hostdir = os.path.join(ROOT, "audioflux_amd", "csrc", "hip"), os.path.join(ROOT, "audioflux_amd", "csrc", "host")
    jobs = [[HIPCC, "--cuda-host-only", "--offload-arch=gfx950", *san, *INC, "-c", os.path.join(hipdir, f), "-o",
             os.path.join(tmp, f[:-4] + "_hip.o")] for f in sorted(os.listdir(hipdir)) if f.endswith(".hip")]
    jobs += [[CLANG, "-std=gnu11", *san, "-ffp-contract=off", *INC, "-c", os.path.join(hostdir, f), "-o",
              os.path.join(tmp, f[:-2] + "_c.o")] for f in sorted(os.listdir(hostdir)) if f.endswith(".c")]
    jobs += [[CLANG, "-std=gnu11", *san, *INC, "-c", os.path.join(HERE, d + ".c"), "-o", os.path.join(tmp, d + ".drv")] for d in drivers]
    jobs.append([CLANG + "++", "-std=c++17", *san, "-I/opt/rocm/include", *INC, "-c", os.path.join(HERE, "fake_hip.cpp"), "-o",
                 os.path.join(tmp, "fake_hip.o")])
    with ThreadPoolExecutor(8) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            assert r.returncode == 0, r.stderr[-3000:]
    objs = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".o"))
    syms = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True).stdout
    with open(os.path.join(tmp, "fatbins.c"), "w") as f:
        for name in sorted(set(re.findall(r"__hip_fatbin_[0-9a-f]+", syms))):
            f.write(f"const char {name}[16] = {{0}};\n")
    r = subprocess.run([CLANG, "-fPIC", "-c", os.path.join(tmp, "fatbins.c"), "-o", os.path.join(tmp, "fatbins.o")], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exes = {}
    for d in drivers:
        exes[d] = os.path.join(tmp, "audit_" + d)
        r = subprocess.run([CLANG + "++", *san, *objs, os.path.join(tmp, d + ".drv"), os.path.join(tmp, "fatbins.o"), "-lm", "-lpthread",
                            "-ldl", "-o", exes[d]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return exes, objs


@pytest.mark.skipif(not (os.path.exists(CLANG) and os.path.exists(HIPCC)), reason="needs hipcc / clang")
def test_concurrent_objects_are_race_free_through_the_real_launchers(tmp_path):
    """tests/hoststub/driver_threads.c (six threads building, using and freeing BFT and CQT objects) with the real
    launchers under clang's ThreadSanitizer: the launcher-level statics (per-device attribute flags, lazily built
    tables) as well as the host objects"""
    exes, _ = _build_with_fake_hip(str(tmp_path), ["-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer"], ("driver_threads",))
    for staging in ("", "1"):  # AFX_STAGING=1: host-pointer copies through the per-stream pinned slabs (afx_runtime.hip: a table shared by the threads)
        e = dict(os.environ, AFX_QUIET="1")
        e.pop("AFX_STAGING", None)
        if staging:
            e["AFX_STAGING"] = staging
        r = subprocess.run([exes["driver_threads"]], capture_output=True, text=True, env=e, timeout=600)
        out = r.stdout + r.stderr
        assert r.returncode == 0 and "OK" in r.stdout, out[-3000:]
        assert "ThreadSanitizer" not in out, out[-3000:]


def _tsan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_tsan_runtime() is None, reason="needs gcc with the ThreadSanitizer runtime")
def test_concurrent_objects_are_race_free(tmp_path):
    """tests/hoststub/driver_threads.c under ThreadSanitizer: six threads build, use and free Kaiser-window BFT objects
    and CQT objects (whose resampler table is a Kaiser window with another beta) at the same time; every thread also
    compares its window with a single-threaded one.  (afx_window.c once kept the beta in a temporarily overwritten
    global: TSan reports that version.)"""
    tmp = str(tmp_path)
    stub = os.path.join(tmp, "stub.c")
    subprocess.run([sys.executable, os.path.join(HERE, "gen_stub.py"),
                    os.path.join(ROOT, "audioflux_amd", "csrc", "hip", "afx_device.h"), stub], check=True)
    host = sorted(os.path.join(ROOT, "audioflux_amd", "csrc", "host", f)
                  for f in os.listdir(os.path.join(ROOT, "audioflux_amd", "csrc", "host")) if f.endswith(".c"))
    exe = os.path.join(tmp, "driver_threads")
    r = subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer", "-ffp-contract=off",
                        *INC, *host, stub, os.path.join(HERE, "driver_threads.c"), "-lm", "-lpthread", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "OK" in r.stdout, out[-3000:]
    assert "ThreadSanitizer" not in out, out[-3000:]

