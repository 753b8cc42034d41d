#!/usr/bin/env python
"""HBM traffic of the synchrosqueezed transform at BASELINE cfg 4's chunk geometry (84 morlet scales, 2^16-sample chunks,
reflect padded: L = 2^17), with and without the derivative transform's time-domain plan (run on the GPU box):

    python tools/wsst_traffic.py [--chunks N]     -> gpurun_out/r04_wsst_traffic.json

One wsstObj_wsstBatchDevice call over N chunks = the CWT, the derivative CWT (wsst_algorithm.c:242-246) and the
squeeze pass.  FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes as tools/prof_traffic.py collects them; bytes =
2 x FETCH_SIZE + WRITE_SIZE (KiB).  Algorithmic bytes per chunk: the chunk in, the squeezed tensor out (re, im)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
NUM, R = 84, 16


def worker(chunks):
    import torch
    import audioflux_amd as af
    o = af.WSST(num=NUM, radix2_exp=R, samplate=44100, low_fre=32.703, bin_per_octave=12,
                wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = 0.1 * torch.randn((chunks, 1 << R), device="cuda", generator=g)
    o.wsst_device(x)  # warm-up: plans, scratch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        o.wsst_device(x)
    torch.cuda.synchronize()
    print(json.dumps({"wsst_ms_per_call": (time.perf_counter() - t0) * 500.0, "chunks": chunks}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=32)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a.chunks)
    import subprocess
    from prof_traffic import kname, run_pass
    out = {"workload": f"WSST, {NUM} morlet scales, {a.chunks} chunks of 2^{R} samples (reflect padded), 3 calls",
           "algorithmic_bytes_per_chunk": (1 << R) * 4 + NUM * (1 << R) * 8}
    for label, env in (("time_domain_derivative", {}), ("two_pass_derivative", {"AFX_CWT_TD_DET": "0"})):
        tmp = os.path.join(ROOT, "gpurun_out", "prof_wsst_" + label)
        args = ["--worker", "--chunks", str(a.chunks)]
        fetch = run_pass("FETCH_SIZE", os.path.join(tmp, "fetch"), args, "tools/wsst_traffic.py", env)
        write = run_pass("WRITE_SIZE", os.path.join(tmp, "write"), args, "tools/wsst_traffic.py", env)
        calls = 3
        per = {}
        for n, c, v in fetch:
            per.setdefault(kname(n), [0, 0.0, 0.0])
            per[kname(n)][0] = c
            per[kname(n)][1] += v
        for n, c, v in write:
            per.setdefault(kname(n), [c, 0.0, 0.0])[2] += v
        tot = sum(2 * f + w for _, f, w in per.values()) * 1024.0 / (calls * a.chunks)
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--chunks", str(a.chunks)],
                             env=dict(os.environ, **env), stdout=subprocess.PIPE, text=True)
        ms = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])["wsst_ms_per_call"]
        out[label] = {"bytes_per_chunk": tot, "over_algorithmic": tot / out["algorithmic_bytes_per_chunk"], "ms_per_call": ms,
                      "kernels": {k: {"launches": c, "fetch_kib": f, "write_kib": w} for k, (c, f, w) in sorted(per.items())}}
    path = os.path.join(ROOT, "gpurun_out", "r04_wsst_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "kernels"}) for k, v in out.items()}))


if __name__ == "__main__":
    main()
