#!/usr/bin/env python3
"""GPU: the one-launch CQT ladder (k_cqt_pyramid) against the per-octave launches (AFX_CQT_PYRAMID=0) -- the same
arithmetic in the same order, so every output word must match; two child processes (the switch is read once per
process), several shapes: many short clips, few long ones, an odd clip stride, runs that end mid-tile."""
import hashlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(3, 61000, 61005), (40, 200000, 200000), (7, 1323000, 1323000), (1, 500000, 500000), (300, 33000, 33000), (2, 128 * 32 * 5 - 1, 128 * 32 * 5 + 3)]


def child():
    sys.path.insert(0, ROOT)
    import torch
    import audioflux_amd as af
    o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
    for batch, n, stride in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(batch + n)
        x = 0.1 * torch.randn((batch, stride), generator=g, device="cuda")
        x[:, n // 2:] *= 1e-3  # a level step: the tile exponents change along the clip
        T = o.cal_time_length(n)
        re = torch.zeros((batch, T, 84), device="cuda"); im = torch.zeros_like(re); ch = torch.zeros((batch, T, 12), device="cuda")
        for rep in range(2):
            o.cqt_chroma_device(x[:, :n], out_real=re, out_imag=im, out=ch)  # (a view: the clip stride stays `stride`)
            torch.cuda.synchronize()
            h = hashlib.sha256(re.cpu().numpy().tobytes() + im.cpu().numpy().tobytes() + ch.cpu().numpy().tobytes()).hexdigest()[:16]
            print(f"SHAPE {batch}x{n}/{stride} rep {rep} {h} finite {bool(torch.isfinite(re).all())} peak {float(re.abs().max()):.4f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
        sys.exit(0)
    outs = {}
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, AFX_CQT_PYRAMID=v))
        outs[v] = [l for l in r.stdout.splitlines() if l.startswith("SHAPE")]
        if r.returncode:
            print(r.stderr[-2000:])
    bad = 0
    for a, b in zip(outs["1"], outs["0"]):
        same = a == b
        bad += not same
        print(("same     " if same else "DIFFERENT"), a, "|", b.split()[-5])
    print("RESULT", "bitwise equal" if not bad and len(outs["1"]) == 2 * len(SHAPES) == len(outs["0"]) else f"{bad} differ / missing")
