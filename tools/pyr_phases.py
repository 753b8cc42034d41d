#!/usr/bin/env python3
"""GPU: where the steps of k_cqt_pyramid go -- AFX_CQT_PYR_TIMING=1 runs the instrumented instantiation; shader cycles
per wave role and phase, averaged over the workgroups and divided by the steps of a run.  The switch exists in measurement
builds only: tools/build_variant.sh exp -DAFX_EXPERIMENTS afx_cqt afx_cqt_f16 afx_melfused2, then
AFX_LIB=audioflux_amd/lib/variants/libafx_exp.so python tools/pyr_phases.py [clips] [steps]   (gpu_call6.sh phases does both)"""
import ctypes as C, os, sys, time
os.environ["AFX_CQT_PYR_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audioflux_amd as af
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 125
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1323000
o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
x = 0.1 * torch.randn((clips, n), device="cuda")
T = o.cal_time_length(n)
re = torch.empty((clips, T, 84), device="cuda"); im = torch.empty_like(re)
lib = af.get_lib()
lib.afx_cqt_pyramid_timing.restype = C.c_int
buf = np.zeros((256, 11, 8), np.uint64)
for _ in range(3):
    o.cqt_device(x, re, im)
torch.cuda.synchronize()
lib.afx_cqt_pyramid_timing(C.c_void_p(o._obj.value if hasattr(o._obj, "value") else o._obj), buf.ctypes.data_as(C.c_void_p))
t0 = time.perf_counter()
for _ in range(reps):
    o.cqt_device(x, re, im)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
got = lib.afx_cqt_pyramid_timing(C.c_void_p(o._obj.value if hasattr(o._obj, "value") else o._obj), buf.ctypes.data_as(C.c_void_p))
if not got:
    sys.exit("no timing recorded: this library was built without -DAFX_EXPERIMENTS (see the header of this file); "
             "the shipped build has no instrumented ladder")
nT = (T + 31) // 32
wgs = min(256, clips * max(1, min(256 // clips, nT // 48)))
used = buf[:wgs].astype(np.float64)
print(f"{clips} clips, {ms:.3f} ms per call (instrumented), {wgs} workgroups; cycles per step and wave (mean over workgroups)")
cpc = max(1, min(256 // clips, nT // 48)); tpc = -(-nT // cpc); steps = (tpc + 31) * reps * max(1, -(-clips * cpc // 256))
names_c = ["wait-window", "convert", "fetch-issue", "k-loop", "wait-prefetch", "resample", "barrier", "store+vm-wait"]
names_p = ["issue-loads", "vm-wait", "wait-input", "lds-stage", "taps", "stores/other", "barrier", "-"]
for w in range(8):
    m = used[:, w, :].mean(axis=0) / steps
    nm = names_c if w < 7 else names_p
    role = ["level 0 (multiply)", "level 1", "level 2", "level 3", "level 0 (prepare)", "level 6 (early)", "level 5 (early)", "level 4 (early)"][w]
    print(f"wave {w} ({role}): total {m.sum():8.0f} | " +
          "  ".join(f"{names_c[i]} {m[i]:.0f}" for i in range(8)))
tot = used.sum(axis=2).mean() / steps
print(f"cycles per step ~{tot:.0f}; at {ms*1e-3/ (steps/reps) * 1e9:.0f} ns per step -> clock ~{tot / (ms*1e-3/(steps/reps)) / 1e6:.0f} MHz")
# ---- round 6: the launch as a whole -- kernel cycles per workgroup against the cycles inside the step loop, and the step series of workgroup 0
kc = buf[:wgs, 8, 0].astype(np.float64) / reps
inloop = used[:, 0, :].sum(axis=1) / reps
print(f"kernel cycles per workgroup and launch: mean {kc.mean():.0f} min {kc.min():.0f} max {kc.max():.0f}; inside the step loop (wave 0): mean {inloop.mean():.0f}")
t0k, t1k = buf[:wgs, 8, 1].astype(np.float64), buf[:wgs, 8, 2].astype(np.float64)
print(f"last launch: workgroup starts spread over {t0k.max() - t0k.min():.0f} cycles, ends over {t1k.max() - t1k.min():.0f}; first start -> last end {t1k.max() - t0k.min():.0f}")
ser = np.array([buf[i >> 4, 9 + ((i >> 3) & 1), i & 7] for i in range(tpc + 31)], dtype=np.float64)
d = np.diff(ser)
print("step durations of workgroup 0 (cycles): first 12", d[:12].astype(int).tolist(), "| middle (median)", int(np.median(d[20:-30])), "| last 24", d[-24:].astype(int).tolist())
print(f"prologue (kernel start -> end of step 0): {ser[0] - t0k[0]:.0f} cycles; last step end -> kernel end: {t1k[0] - ser[-1]:.0f}")
