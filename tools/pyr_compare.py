#!/usr/bin/env python3
"""GPU: the one-launch CQT ladder (k_cqt_pyramid) against the per-octave launches (AFX_CQT_PYRAMID=0) and itself.
The octave products are the same arithmetic; the level signals come from the matrix-core resampler instead of the
float32 filter chain, so the two agree to rounding (bars: 5e-6 of the tensor peak, 2e-5 of any frame's peak after a
-60 dB level step), and two runs of the ladder agree bit for bit.  Two child processes (the switch is read once per
process); shapes: many short clips, few long ones, an odd clip stride, runs that end mid-tile."""
import hashlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(3, 61000, 61005), (40, 200000, 200000), (7, 1323000, 1323000), (1, 500000, 500000), (300, 33000, 33000), (2, 128 * 32 * 5 - 1, 128 * 32 * 5 + 3)]


def child(out):
    sys.path.insert(0, ROOT)
    import torch
    import audioflux_amd as af
    o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
    res = {}
    for si, (batch, n, stride) in enumerate(SHAPES):
        g = torch.Generator(device="cuda").manual_seed(batch + n)
        x = 0.1 * torch.randn((batch, stride), generator=g, device="cuda")
        x[:, n // 2:] *= 1e-3  # a level step: the tile exponents change along the clip
        T = o.cal_time_length(n)
        re = torch.zeros((batch, T, 84), device="cuda"); im = torch.zeros_like(re); ch = torch.zeros((batch, T, 12), device="cuda")
        hs = []
        for rep in range(2):
            o.cqt_chroma_device(x[:, :n], out_real=re, out_imag=im, out=ch)  # (a view: the clip stride stays `stride`)
            torch.cuda.synchronize()
            hs.append(hashlib.sha256(re.cpu().numpy().tobytes() + im.cpu().numpy().tobytes() + ch.cpu().numpy().tobytes()).hexdigest()[:16])
        print(f"SHAPE {batch}x{n}/{stride} {hs[0]} repeat {'same' if hs[0] == hs[1] else 'DIFFERENT'} finite {bool(torch.isfinite(re).all())}", flush=True)
        keep = slice(0, min(batch, 3))
        res[f"re{si}"] = re[keep].cpu().numpy(); res[f"im{si}"] = im[keep].cpu().numpy(); res[f"ch{si}"] = ch[keep].cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[2])
        sys.exit(0)
    tmp = tempfile.mkdtemp()
    ok = True
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "child", os.path.join(tmp, f"p{v}.npz")], capture_output=True, text=True,
                           env=dict(os.environ, AFX_CQT_PYRAMID=v))
        print(f"-- AFX_CQT_PYRAMID={v}")
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("SHAPE")))
        ok &= r.returncode == 0 and "DIFFERENT" not in r.stdout and "finite False" not in r.stdout
        if r.returncode:
            print(r.stderr[-2000:])
    a, b = np.load(os.path.join(tmp, "p1.npz")), np.load(os.path.join(tmp, "p0.npz"))
    for si, shape in enumerate(SHAPES):
        qa, qb = a[f"re{si}"] + 1j * a[f"im{si}"], b[f"re{si}"] + 1j * b[f"im{si}"]
        peak = np.abs(qa - qb).max() / np.abs(qb).max()
        pf = (np.abs(qa - qb).max(axis=2) / np.maximum(np.abs(qb).max(axis=2), 1e-30)).max()
        chd = np.abs(a[f"ch{si}"] - b[f"ch{si}"]).max()
        good = peak <= 5e-6 and pf <= 2e-5 and chd <= 2e-5
        ok &= good
        print(f"shape {shape}: ladder vs per-octave: tensor-peak {peak:.2e}, worst frame {pf:.2e}, chroma {chd:.2e} {'ok' if good else 'TOO FAR'}")
    print("RESULT", "ok" if ok else "FAILED")
