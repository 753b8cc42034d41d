#!/usr/bin/env python3
"""GPU: where a workgroup of the narrow-band CWT kernels (k_cwt_inv_cols256_nb<R>, cfg 4) spends its cycles -- s_memtime stamps of
thread 0 at the phase boundaries, summed over the workgroups (measurement builds only: tools/build_variant.sh exp -DAFX_EXPERIMENTS
afx_cwt ...; AFX_LIB=audioflux_amd/lib/variants/libafx_exp.so python tools/cwt_phases.py [clips] [steps]).  The time-domain
kernels run beside them on the second stream, as in the bench."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audioflux_amd as af
import bench

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = af.get_lib()
if not hasattr(lib, "afx_cwt_nb_phases"):
    sys.exit("this library was built without -DAFX_EXPERIMENTS: no phase clocks")
w = bench.Cfg4(torch, af, torch.device("cuda:0"), 0, clips)
buf = np.zeros((5, 8), np.uint64)
for i in range(2):
    w.step(i)
torch.cuda.synchronize()
lib.afx_cwt_nb_phases(buf.ctypes.data_as(C.c_void_p))  # (clears)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    w.step(2 + i)
e1.record()
torch.cuda.synchronize()
lib.afx_cwt_nb_phases(buf.ctypes.data_as(C.c_void_p))
print(f"cfg 4, {clips} clips, {e0.elapsed_time(e1) / steps:.2f} ms per step (instrumented build)")
names = ["loads issued -> arrived", "stage products, barrier", "twiddles + R-term sums", "barrier", "column transform, stores issued",
         "stores acknowledged"]
print("%-40s" % "cycles per workgroup (thread 0)" + "".join("%12s" % f"R = {1 << lr}" for lr in range(1, 5)))
for ph in range(6):
    print("%-40s" % names[ph] + "".join("%12.0f" % (buf[lr, ph] / max(1, buf[lr, 7])) for lr in range(1, 5)))
print("%-40s" % "total" + "".join("%12.0f" % (buf[lr, :6].sum() / max(1, buf[lr, 7])) for lr in range(1, 5)))
print("%-40s" % "workgroups" + "".join("%12d" % buf[lr, 7] for lr in range(1, 5)))
