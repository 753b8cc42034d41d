"""k_gemm_bank_bf16x3 alone at the dense route's shape: A = [frames, 1025] float32 rows at a pitch of 1028 (power-spectrum-like
values), bank = 128 x 1025, through afxk_gemm_bank_prepare / afxk_gemm_nt_bank of the library AFX_LIB names (knock-out variants:
tools/build_variant.sh kog<mask> -DAFX_KO_GEMM=<mask> afx_gemm_bf16).  Prints microseconds per launch, the f32-equivalent
and bf16 FLOP rates and the matrix-pipe floor (24 MFMAs x 32 cycles per k-step and wave).
    python tools/bench_gemm_bank.py [clips = 250] [launches = 40] [check = 1]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audioflux_amd as af

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = af.get_lib()
vp, ll = C.c_void_p, C.c_longlong
lib.afxk_gemm_bank_prepare.restype = C.c_int
lib.afxk_gemm_bank_prepare.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), vp]
lib.afxk_gemm_nt_bank.restype = C.c_int
lib.afxk_gemm_nt_bank.argtypes = [vp, ll, vp, C.c_int, C.c_int, vp, ll, ll, C.c_int, C.c_float, vp]
M, N, K, P = clips * 934, 128, 1025, 1028
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.zeros((M, P), device="cuda")
A[:, :K] = torch.randn((M, K), device="cuda", generator=g) ** 2 * 10.0 ** (10 * torch.rand((M, K), device="cuda", generator=g) - 5)
B = torch.zeros((N, P), device="cuda")
B[:, :K] = torch.rand((N, K), device="cuda", generator=g)
Cm = torch.empty((M, N), device="cuda")
img = vp()
stream = torch.cuda.current_stream().cuda_stream
assert lib.afxk_gemm_bank_prepare(B.data_ptr(), P, N, K, C.byref(img), stream) == 0


def run():
    st = lib.afxk_gemm_nt_bank(A.data_ptr(), P, img, N, K, Cm.data_ptr(), N, M, 0, 0.0, stream)
    assert st == 0, (st, af.last_error())


t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
for _ in range(60):  # clock warm-up
    run()
torch.cuda.synchronize()
t0.record()
for _ in range(n):
    run()
t1.record()
torch.cuda.synchronize()
us = t0.elapsed_time(t1) * 1e3 / n
flop = 2.0 * M * N * K
line = f"gemm_bank {M} x {N} x {K}: {us:.1f} us per launch, {flop / us / 1e6:.1f} TF/s f32-equivalent = {6 * flop / us / 1e6:.0f} TF/s bf16"
if check:
    rows = torch.arange(0, M, max(1, M // 512), device="cuda")
    want = A[rows, :K].double() @ B[:, :K].double().T
    err = ((Cm[rows].double() - want).abs() / want.abs().clamp_min(1e-300)).max().item()
    line += f", elementwise error {err:.2e}"
print(line, flush=True)
