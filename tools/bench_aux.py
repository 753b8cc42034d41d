"""Secondary workloads of BASELINE.json (configs 4 and 5) through the batched device
entry points: CWT morlet 84 scales on 2^16-sample chunks (padded, L = 2^17) and CQT 84 bins
+ chroma on 30 s @ 44.1 kHz clips.  Prints one JSON line per workload with the rate and the
fraction of the HBM roofline on SURVEY 8d's algorithmic bytes."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import audioflux_amd as af

HBM_PEAK = 8.0e12


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e-3


def cwt(args):
    r, num, chunks = 16, 84, args.cwt_chunks
    o = af.CWT(num=num, radix2_exp=r, samplate=44100, low_fre=32.703, bin_per_octave=12,
               wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE,
               is_padding=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = 0.1 * torch.randn((chunks, 1 << r), device="cuda", generator=g)
    re = torch.empty((chunks, num, 1 << r), device="cuda")
    im = torch.empty_like(re)
    sec = timed(lambda: o.cwt_device(x, re, im), args.steps, args.warmup)
    samples = chunks * (1 << r)
    alg = samples * (4 + 8 * num)
    return {"workload": "cfg4 CWT morlet 84 scales, 2^16-sample chunks, padded (L=2^17)",
            "chunks": chunks, "ms": sec * 1e3, "chunks_per_s": chunks / sec,
            "samples_per_s": samples / sec, "alg_GBps": alg / sec / 1e9, "frac_hbm": alg / sec / HBM_PEAK}


def cqt(args):
    clips, n, num = args.cqt_clips, 1323000, 84
    o = af.CQT(num=num, samplate=44100, low_fre=32.703, bin_per_octave=12,
               normal_type=af.SpectralFilterBankNormalType.AREA)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = 0.1 * torch.randn((clips, n), device="cuda", generator=g)
    t = o.cal_time_length(n)
    re = torch.empty((clips, t, num), device="cuda")
    im = torch.empty_like(re)
    ch = torch.empty((clips, t, 12), device="cuda")

    def step():
        o.cqt_device(x, re, im)
        o.chroma_device(re, im, out=ch)
    sec = timed(step, args.steps, args.warmup)
    frames = clips * t
    alg = frames * (4 * 128 + 8 * num + 4 * 12)
    return {"workload": "cfg5 CQT 84 bins (12/oct) + chroma, 30 s @ 44.1 kHz clips",
            "clips": clips, "frames": frames, "ms": sec * 1e3, "frames_per_s": frames / sec,
            "alg_GBps": alg / sec / 1e9, "frac_hbm": alg / sec / HBM_PEAK}


def _cpu_cwt(args):
    """worker: reference CWT on `n` chunks; returns (chunks, seconds)"""
    seed, n = args
    from oracle import ref
    r = ref.RefCWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                   wavelet_type=int(af.WaveletContinueType.MORLET),
                   scale_type=int(af.SpectralFilterBankScaleType.OCTAVE), is_padding=1)
    x = (0.1 * np.random.default_rng(seed).standard_normal((n + 1, 65536))).astype(np.float32)
    r.cwt(x[0])
    t0 = time.perf_counter()
    for i in range(n):
        r.cwt(x[1 + i])
    return n, time.perf_counter() - t0


def _cpu_cqt(args):
    """worker: reference CQT + chroma on `n` clips of 30 s @ 44.1 kHz; returns (frames, seconds)"""
    seed, n = args
    from oracle import ref
    r = ref.RefCQT(num=84, samplate=44100, min_fre=32.703, bin_per_octave=12, normal_type=1)
    x = (0.1 * np.random.default_rng(seed).standard_normal((n + 1, 1323000))).astype(np.float32)
    r.chroma(*r.cqt(x[0]))  # (a shorter warm-up clip makes the reference corrupt its heap)
    t0 = time.perf_counter()
    frames = 0
    for i in range(n):
        re, im = r.cqt(x[1 + i])
        r.chroma(re, im)
        frames += re.shape[0]
    return frames, time.perf_counter() - t0


def cpu_baseline(worker, unit, per):
    """compiled reference (oracle/_ref) on the host cores: (A) one process as shipped (OpenMP
    default), (B) P single-FFT-thread processes; best aggregate reported with the count used"""
    import multiprocessing as mp
    from oracle import ref
    if not ref.available():
        return None
    ncpu = len(os.sched_getaffinity(0))
    try:
        q, per_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            ncpu = max(1, min(ncpu, int(int(q) / int(per_))))
    except (OSError, ValueError):
        pass
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        u, t = pool.map_async(worker, [(100, per)]).get(timeout=120)[0]
    best = {"value": u / t, "cores": ncpu, "how": "A: 1 process, default OpenMP"}
    os.environ["OMP_NUM_THREADS"] = "2"
    p = min(ncpu, 16)
    with ctx.Pool(p) as pool:
        res = pool.map_async(worker, [(200 + i, per) for i in range(p)]).get(timeout=180)
    os.environ.pop("OMP_NUM_THREADS")
    rate = sum(r[0] for r in res) / max(r[1] for r in res)
    if rate > best["value"]:
        best = {"value": rate, "cores": p, "how": f"B: {p} processes x 1 FFT thread"}
    return {"value": best["value"], "unit": unit, "cores": best["cores"], "kind": "reference",
            "sample": f"{best['how']}, {per} unit(s) per process; visible cpus {ncpu}"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cwt-chunks", type=int, default=16)
    ap.add_argument("--cqt-clips", type=int, default=16)
    ap.add_argument("--only", default="")
    ap.add_argument("--cpu-baseline", action="store_true")
    args = ap.parse_args()
    for name, fn, worker, unit, per in (("cwt", cwt, _cpu_cwt, "chunks/s", 2), ("cqt", cqt, _cpu_cqt, "frames/s", 1)):
        if args.only and args.only != name:
            continue
        out = fn(args)
        if args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(worker, unit, per)
        print(json.dumps(out), flush=True)
