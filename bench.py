#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: audio frames/sec for batched STFT -> mel-128 -> MFCC-13
(n_fft = 2048, hop = 512) on synthetic 16 kHz mono clips (--config 2, the default and the line the
driver records), plus the two other measured configurations of BASELINE.json in the same JSON
shape: --config 4 (CWT morlet, 84 scales, 1000 x 10 s @ 44.1 kHz) and --config 5 (CQT 84 bins +
chroma, 30 s @ 44.1 kHz clips, 125 clips per GPU).

    python bench.py --gpus 1 --steps K --warmup W [--config 2|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch per GPU, inputs already resident in HBM,
features left in HBM; with N > 1 each rank owns its own batch (weak scaling: clips are independent,
no data-path collective) and the feature slab named by --gather goes to rank 0 over RCCL on a side
stream, overlapped with the next step (cfg 4: replicas only -- its output exceeds every link).
Timing: barrier + torch.cuda.synchronize() on both sides of exactly K steps, max over ranks.  Rank 0
prints ONE JSON line.

`roofline`: cfg 2 is ONE kernel launch per step (k_stft_mel_v2: STFT -> |S|^2 -> mel bank -> log10 ->
DCT-II); it is timed live with HIP events on the stream it is launched on (torch's current stream is
handed to the library).  `achieved` = SURVEY 8d's algorithmic bytes per unit x units per launch /
that duration; `sustained_ms` repeats the step back to back for >= 1 s (sustained clocks, where the
K-step region of a short run sees boost clocks); `traffic` = HBM bytes per launch from the round's
own rocprofv3 --pmc passes of this command (profiles/r06_bench_cfg<N>_pmc.json, written by
tools/prof_traffic.py), null when that file is absent.  After the timed region clip 0 of the
benchmarked outputs is checked against the oracle (1e-5 peak / L2).
`secondary` (--config 2 at one GPU, the driver's line): cfg 4 and cfg 5 measured in the same process after the
headline -- value, ms_per_step, frac, PMC traffic, oracle check each.  With N > 1 `gather` says whether the
exchange hides behind the compute: gather_ms (device time of the collective on rank 0's side stream),
exposed_ms (step time beyond the step's own compute) and overlap_hidden_ms.
`dense_bank` (same line): the dense filter-bank route north_star names as the MFMA path -- gammatone-128 at the headline shape as
frames/s with its own oracle check, and the route's product alone (k_gemm_bank_bf16x3, one chunk's shape) as bf16 matrix FLOP/s
against the 2.5 PF/s dense peak (`roofline.bound` "mfma"); mfma_busy is replayed from profiles/r06_rocprofv3_dense.txt.
`cpu_baseline` times the reference's own C path (oracle/_ref, built-in FFT + naive
double-accumulating matmul: no FFTW/MKL in this image) on the host cores for a bounded sample.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


# =================================================================================================
# host-side CPU baselines (compiled reference, oracle/_ref) -- rank 0, N = 1 only
def _effective_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_worker_cfg2(args):
    """one host process: reference mel+MFCC over `n` clips (objects pre-built, one warm-up)"""
    seed, n, threads = args
    if threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    else:
        os.environ.pop("OMP_NUM_THREADS", None)
    from oracle import ref
    x = (0.1 * np.random.default_rng(seed).standard_normal((n, 480000))).astype(np.float32)
    bft = ref.RefBFT(128, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1,
                     slide_length=512, scale_type=2, style_type=0, normal_type=0, data_type=0)
    bft.set_result_type(1)
    cc = ref.RefXXCC(128)
    re, _ = bft.bft(x[0])
    cc.xxcc(re, 13, 0)  # warm-up
    t0 = time.perf_counter()
    units = 0
    for i in range(n):
        re, _ = bft.bft(x[i])
        cc.xxcc(re, 13, 0)
        units += re.shape[0]
    return units, time.perf_counter() - t0


def cpu_worker_cfg4(args):
    """reference CWT on `n` chunks of 2^16 samples; returns (chunks, seconds)"""
    seed, n, threads = args
    if threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import ref
    r = ref.RefCWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                   wavelet_type=1, scale_type=5, is_padding=1)
    x = (0.1 * np.random.default_rng(seed).standard_normal((n + 1, 65536))).astype(np.float32)
    r.cwt(x[0])
    t0 = time.perf_counter()
    for i in range(n):
        r.cwt(x[1 + i])
    return n, time.perf_counter() - t0


def cpu_worker_cfg5(args):
    """reference CQT + chroma on `n` clips of 30 s @ 44.1 kHz; returns (top-octave frames, seconds)"""
    seed, n, threads = args
    if threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(threads)
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


def cpu_baseline(worker, unit, per_proc, what, budget_s=25.0):
    """reference CPU path on this box's host cores -- BASELINE.md section 3:
    (A) as shipped: one process, default OpenMP (frame loop on omp_get_max_threads()/2 threads,
        src/stft_algorithm.c:95-100; the matmul is serial);
    (B) all cores: P worker processes with one FFT thread each (OMP_NUM_THREADS=2), units sharded;
        P is swept because hosts differ (SMT, cgroup quotas) and the best aggregate is reported.
    Bounded sample, objects pre-built, one warm-up call per worker."""
    from oracle import ref
    if not ref.available():
        return None
    import multiprocessing as mp
    t_start = time.perf_counter()
    ncpu = _effective_cpus()
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:  # (A) in a fresh process so OMP defaults apply
        fa, ta = pool.map(worker, [(1000, per_proc[0], 0)])[0]
    best = {"value": fa / ta, "cores": ncpu, "how": "A: 1 process, default OpenMP", "units": fa}
    tried = [f"A=1proc:{fa / ta:.4g}"]
    for p in sorted({p for p in (8, 16, 32, 64, 128, 256, ncpu) if p <= ncpu}):
        if time.perf_counter() - t_start > budget_s:
            break
        with ctx.Pool(p) as pool:
            res = pool.map(worker, [(2000 + i, per_proc[1], 2) for i in range(p)])
        wall = max(r[1] for r in res)  # workers run concurrently; the slowest bounds the job
        rate = sum(r[0] for r in res) / wall
        tried.append(f"B={p}procs:{rate:.4g}")
        if rate > best["value"]:
            best = {"value": rate, "cores": p, "how": f"B: {p} processes x 1 FFT thread",
                    "units": sum(r[0] for r in res), "procs": p}
    # the sweep samples are short; re-time the best configuration on a sample sized for ~12 s
    if "procs" in best and per_proc[2] > per_proc[1] and time.perf_counter() - t_start < budget_s:
        p = best["procs"]
        with ctx.Pool(p) as pool:
            res = pool.map(worker, [(3000 + i, per_proc[2], 2) for i in range(p)])
        wall = max(r[1] for r in res)
        units = sum(r[0] for r in res)
        tried.append(f"final={p}procsx{per_proc[2]}:{units / wall:.4g}")
        best.update(value=units / wall, units=units)
    return {"value": best["value"], "unit": unit, "cores": best["cores"], "kind": "reference",
            "sample": f"{best['how']}, {best['units']} {unit.split('/')[0]} of {what}; built-in radix-2 FFT + "
                      f"naive matmul (no FFTW/MKL); visible cpus {ncpu}; sweep {unit}: " + " ".join(tried),
            "wall_s": round(time.perf_counter() - t_start, 1)}


# =================================================================================================
# workloads
def _parity(got, want):
    got = np.asarray(got)
    want = np.asarray(want)
    den = np.abs(want).max() or 1.0
    peak = float(np.abs(got - want).max() / den)
    l2 = float(np.linalg.norm((got - want).ravel()) / (np.linalg.norm(want.ravel()) or 1.0))
    return max(peak, l2)


class Cfg2:
    """batched STFT -> mel-128 -> MFCC-13, 1000 x 30 s @ 16 kHz per GPU, n_fft 2048, hop 512"""
    config = 2
    metric = "audio frames/sec (mel+MFCC, n_fft=2048 hop=512)"
    unit = "frames/s"
    default_clips = 1000
    bytes_per_unit = 4 * 512 + 4 * 128 + 4 * 13  # each sample once, each output once
    kernel = ("k_stft_mel_v2 (framed FFT -> |S|^2 -> banded mel bank -> log10 -> DCT-II, "
              "one launch per step)")
    gather_choices = ("mfcc", "mel")
    # what binds the kernel (roofline.compute): the vector unit + LDS of every SIMD, not the memory system
    bound = "VALU+LDS"
    bound_note = ("vector unit and LDS pipe each about 61 % busy and only partly overlapped; knock-out builds price LDS work like vector work "
                  "and instruction issue at nothing (profiles/r05_ab_headline.txt); frac prices the kernel against HBM, which is not what limits it")

    def __init__(self, torch, af, dev, rank, clips):
        self.torch, self.af, self.clips = torch, af, clips
        n = 16000 * 30
        gen = torch.Generator(device=dev).manual_seed(1 + rank)
        self.x = 0.1 * torch.randn((clips, n), generator=gen, device=dev, dtype=torch.float32)
        self.bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                          scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        self.bft.set_result_type(1)
        self.xx = af.XXCC(128)
        self.T = self.bft.cal_time_length(n)
        self.units = clips * self.T
        self.mel = torch.empty((clips, self.T, 128), device=dev, dtype=torch.float32)
        self.cc = [torch.empty((clips, self.T, 13), device=dev, dtype=torch.float32) for _ in range(2)]
        self.workload = (f"batched STFT->mel128->MFCC13, {clips} x 30 s @16 kHz per GPU, n_fft=2048 hop=512 "
                         f"(BASELINE cfg 2)")
        self.outputs = "mel[clips,T,128] + mfcc[clips,T,13] f32 in HBM"

    def step(self, i):
        self.af.mel_mfcc_device(self.bft, self.xx, self.x, 13, out_mel=self.mel, out_cc=self.cc[i & 1])

    def slab(self, i, which):
        return self.cc[i & 1] if which == "mfcc" else self.mel

    def check(self, i):
        from oracle import ref
        if not ref.available():
            return None
        rmel, rcc = ref.mel_mfcc(self.x[:1].cpu().numpy())
        return max(_parity(self.mel[0].cpu().numpy(), rmel[0]), _parity(self.cc[i & 1][0].cpu().numpy(), rcc[0]))

    def cpu(self):
        return cpu_baseline(cpu_worker_cfg2, "frames/s", (8, 4, 96), "30 s @16 kHz clips (mel-128 + MFCC-13, "
                            "n_fft 2048, hop 512)")


class Cfg4:
    """CWT morlet, 84 scales, 1000 x 10 s @ 44.1 kHz = 7000 chunks of 2^16 samples (padded, L = 2^17)"""
    config = 4
    metric = "CWT chunks/sec (morlet, 84 scales, 2^16-sample chunks @44.1 kHz, padded L=2^17)"
    unit = "chunks/s"
    default_clips = 1000
    bytes_per_unit = 65536 * (4 + 8 * 84)  # 676 B per input sample
    kernel = ("all launches of one step: k_cwt_td (36 short-kernel scales in the time domain, f16 matrix cores, own stream), "
              "k_cwt_fwd_*, k_cwt_inv_cols256_nb<R> (44 narrow-band scales) + k_cwt_inv_cols256_nb2<4> (4 scales of 17-20 rows)")
    dtype = "f32 (36 of 84 scales: f32 operands as (hi, lo) f16 words on the f16 matrix cores, f32 accumulation)"
    gather_choices = ()
    bound = "matrix+LDS"
    bound_note = ("the step is the sum of its launches' work: the f16 matrix kernels (time-domain scales, 44 % of the summed kernel time, matrix pipe "
                  "45 % busy) and the narrow-band inverse transforms (45 %; their R-term sums on the f32 matrix pipe since round 6: vector unit 47-53 %, "
                  "no unit saturated) run at the same time on two streams and slow each other (profiles/r06_cwt_phases.txt, r06_cfg4_occupancy.json)")
    GROUP = int(os.environ.get("AFX_CFG4_GROUP", "32"))  # chunks per device call: the [84, 2^16] complex outputs (44 MB per chunk) are ring-buffered

    def __init__(self, torch, af, dev, rank, clips):
        self.torch, self.af, self.clips = torch, af, clips
        self.o = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                        wavelet_type=af.WaveletContinueType.MORLET,
                        scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
        gen = torch.Generator(device=dev).manual_seed(3 + rank)
        # every clip = 7 chunks of 65536 samples, the last one zero padded (441 000 = 6 x 65536 + 47 784)
        x = torch.zeros((clips, 7 * 65536), device=dev, dtype=torch.float32)
        x[:, :441000] = 0.1 * torch.randn((clips, 441000), generator=gen, device=dev, dtype=torch.float32)
        self.x = x.view(clips * 7, 65536)
        self.units = clips * 7
        g = min(self.GROUP, self.units)
        self.ring = [(torch.empty((g, 84, 65536), device=dev), torch.empty((g, 84, 65536), device=dev))
                     for _ in range(2)]
        self.workload = (f"CWT morlet 84 scales, {clips} x 10 s @44.1 kHz per GPU = {self.units} chunks of 2^16 "
                         f"samples, padded (L=2^17) (BASELINE cfg 4)")
        self.outputs = f"[84, 65536] complex per chunk, ring-buffered in HBM ({g} chunks per call)"

    def step(self, i):
        g = self.ring[0][0].shape[0]
        k = 0
        for c0 in range(0, self.units, g):
            n = min(g, self.units - c0)
            re, im = self.ring[k & 1]
            self.o.cwt_device(self.x[c0:c0 + n], re[:n], im[:n])
            k += 1
        self.last = (k - 1) & 1, self.units - n, n

    def check(self, i):
        from oracle import ref
        if not ref.available():
            return None
        slot, c0, n = self.last
        r = ref.RefCWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                       wavelet_type=1, scale_type=5, is_padding=1)
        rre, rim = r.cwt(self.x[c0].cpu().numpy())
        re, im = self.ring[slot]
        return _parity(re[0].cpu().numpy() + 1j * im[0].cpu().numpy(), rre + 1j * rim)

    def cpu(self, budget_s=25.0):
        return cpu_baseline(cpu_worker_cfg4, "chunks/s", (2, 1, 2), "2^16-sample chunks (84 morlet scales, padded)", budget_s)


class Cfg5:
    """CQT 84 bins (12 / octave) + chroma-12, 30 s @ 44.1 kHz clips, 125 clips per GPU"""
    config = 5
    metric = "CQT top-octave frames/sec (84 bins, 12/octave, + chroma, 30 s @44.1 kHz clips)"
    unit = "frames/s"
    default_clips = 125
    bytes_per_unit = 4 * 128 + 8 * 84 + 4 * 12  # hop 128 samples in, 84 complex + 12 chroma out
    kernel = ("k_cqt_pyramid: ONE persistent launch per step -- eight role-specialised waves per workgroup walk runs of 32-frame "
              "tiles (octave products and the 2:1 resampler on the f16 matrix cores, level signals in L2-resident rings, chroma-12 as "
              "partial sums through the output rows)")
    dtype = "f32 (octave products: f32 operands as (hi, lo) f16 words on the f16 matrix cores, f32 accumulation)"
    gather_choices = ("chroma", "cqt")
    bound = "matrix+power"
    bound_note = ("f16 matrix pipe 44-47 % busy at 1.6-1.8 GHz: on real data the launch is held by its data-dependent power (the same code on zeros / "
                  "constants holds 2.3-2.4 GHz) and, at full clock, by 3.3 GB per step of poorly shaped traffic (48-byte row pieces, ring write-backs, "
                  "chroma read-modify-writes) -- both limits within 4 % of each other; the one class worth more than 8 % is the requested output "
                  "itself (knock-out and price builds, profiles/r06_ko_cqt.txt)")

    def __init__(self, torch, af, dev, rank, clips):
        self.torch, self.af, self.clips = torch, af, clips
        n = 1323000
        self.o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12,
                        normal_type=af.SpectralFilterBankNormalType.AREA)
        gen = torch.Generator(device=dev).manual_seed(4 + rank)
        self.x = 0.1 * torch.randn((clips, n), generator=gen, device=dev, dtype=torch.float32)
        self.T = self.o.cal_time_length(n)
        self.units = clips * self.T
        self.re = torch.empty((clips, self.T, 84), device=dev)
        self.im = torch.empty_like(self.re)
        self.ch = [torch.empty((clips, self.T, 12), device=dev) for _ in range(2)]
        self.workload = (f"CQT 84 bins (12/octave) + chroma-12, {clips} x 30 s @44.1 kHz per GPU "
                         f"(BASELINE cfg 5)")
        self.outputs = "cqt[clips,T,84] complex (split planes) + chroma[clips,T,12] f32 in HBM"

    def step(self, i):
        if os.environ.get("AFX_BENCH_CQT_SPLIT") == "1":  # the two separate calls (A/B)
            self.o.cqt_device(self.x, self.re, self.im)
            self.o.chroma_device(self.re, self.im, out=self.ch[i & 1])
        else:
            self.o.cqt_chroma_device(self.x, out_real=self.re, out_imag=self.im, out=self.ch[i & 1])

    def slab(self, i, which):
        return self.ch[i & 1] if which == "chroma" else self.re

    def check(self, i):
        from oracle import ref
        if not ref.available():
            return None
        r = ref.RefCQT(num=84, samplate=44100, min_fre=32.703, bin_per_octave=12, normal_type=1)
        rre, rim = r.cqt(self.x[0].cpu().numpy())
        rch = r.chroma(rre, rim)
        return max(_parity(self.re[0].cpu().numpy() + 1j * self.im[0].cpu().numpy(), rre + 1j * rim),
                   _parity(self.ch[i & 1][0].cpu().numpy(), rch))

    def cpu(self, budget_s=25.0):
        return cpu_baseline(cpu_worker_cfg5, "frames/s", (1, 1, 1), "30 s @44.1 kHz clips (CQT-84 + chroma-12)", budget_s)


class DryRun:
    """AFX_BENCH_DRYRUN=1: a CPU stand-in for the kernels (tests/test_dist_cpu.py runs this file under
    torch.distributed.run with the gloo backend): the distributed control flow of a bench run -- clip
    ownership, double-buffered slabs, the side-stream gather, fences, the max-over-ranks clock, the
    JSON line -- executes unchanged; every clip's "features" carry its global index.  `config` picks
    the slab names and widths of the real workload (cfg 2: mfcc[.,13] + mel[.,128]; cfg 5: chroma[.,12]
    + cqt[.,84]; cfg 4: none -- replicas only, nothing is gathered); `first_clip` is the rank's offset
    in the job's clip order (shard_range)."""
    metric, unit = "dry run (no device)", "frames/s"
    default_clips = 4
    bytes_per_unit = 2612
    kernel = "none (CPU stand-in)"
    SLABS = {2: (("mfcc", 13), ("mel", 128)), 4: (("", 1), ("", 1)), 5: (("chroma", 12), ("cqt", 84))}

    def __init__(self, torch, af, dev, rank, clips, config=2, first_clip=None):
        self.torch, self.clips, self.rank = torch, clips, rank
        self.first = rank * clips if first_clip is None else first_clip
        self.T, self.units = 7, clips * 7
        (self.small, ws), (self.big, wb) = self.SLABS[config]
        self.gather_choices = tuple(n for n in (self.small, self.big) if n)
        self.wide = torch.zeros((clips, self.T, wb))
        self.cc = [torch.zeros((clips, self.T, ws)) for _ in range(2)]
        self.workload, self.outputs = f"dry run of cfg {config}, {clips} clips on this rank", "CPU tensors"

    def step(self, i):
        ids = self.torch.arange(self.first, self.first + self.clips, dtype=self.torch.float32)
        self.cc[i & 1][:] = (ids * 1000 + i)[:, None, None]   # clip id and step, checkable after the gather
        self.wide[:] = ids[:, None, None]

    def slab(self, i, which):
        return self.cc[i & 1] if which == self.small else self.wide

    def check(self, i):
        return None

    def cpu(self):
        return None


WORKLOADS = {2: Cfg2, 4: Cfg4, 5: Cfg5}


def pmc_traffic(config, clips):
    """HBM bytes per step from the round's rocprofv3 --pmc passes of this command
    (tools/prof_traffic.py): 2 x FETCH_SIZE (gfx950 tallies wide coalesced reads at half,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KiB, scaled to this run's clip count"""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_bench_cfg{config}_pmc.json")
        try:
            rec = json.load(open(path))
            per_step = (2.0 * rec["fetch_kib_per_step"] + rec["write_kib_per_step"]) * 1024.0
            return per_step * clips / rec["clips"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


class CpuEvent:
    """stand-in for torch.cuda.Event in the dry run"""
    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def measure(w, torch, dist, dev, world, steps, warmup, clock_warmup, sustained_s, gathers=(), which=(), dry=False):
    """W untimed steps, then exactly `steps` steps between barrier + synchronize fences; the steps' average device time from
    ONE pair of events on the launch stream around them; optionally a back-to-back loop of >= sustained_s seconds.  With `gathers` the slabs
    of step i travel to rank 0 on a side stream while step i + 1 computes (double-buffered sources)."""
    Event = CpuEvent if dry else torch.cuda.Event
    sync = (lambda: None) if dry else torch.cuda.synchronize
    comm = torch.cuda.Stream(device=dev) if (gathers and not dry) else None
    ev, gev = [], []

    def step(i, timed):
        # (no events between the steps: an event record on the launch stream drains it -- 2.5 % of a 1.4 ms step; ONE pair
        #  brackets the K timed steps, the average launch duration is their difference / K)
        w.step(i)
        if gathers:
            # the slabs of this step go to rank 0 while the next step computes (double-buffered
            # sources; the previous gather has had a whole step to finish)
            for g in gathers:
                g.wait()
            if comm is not None:
                comm.wait_stream(torch.cuda.current_stream())
            with (torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()):
                g0, g1 = Event(enable_timing=True), Event(enable_timing=True)
                g0.record()
                for g, name in zip(gathers, which):
                    g.start(w.slab(i, name))
                g1.record()
                if timed:
                    gev.append((g0, g1))

    def fence():
        for g in gathers:
            g.wait()
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
        sync()
        if world > 1:
            dist.barrier()
        sync()

    if clock_warmup > 0 and not dry:
        # untimed, like the W steps below: brings the clocks to their loaded state.  Compute only --
        # the ranks run different numbers of steps in a fixed wall time, and every collective must be
        # entered the same number of times by all of them
        tw, i = time.perf_counter(), 0
        while time.perf_counter() - tw < clock_warmup:
            w.step(i)
            i += 1
            if i % 8 == 0:
                sync()
        sync()
    for i in range(warmup):
        step(i, False)
    fence()
    t0 = time.perf_counter()
    if steps > 0:
        ev.append((Event(enable_timing=True), Event(enable_timing=True)))
        ev[0][0].record()
    for i in range(steps):
        step(i, True)
    if steps > 0:
        ev[0][1].record()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    res = {"elapsed": elapsed, "last": max(steps - 1, 0),
           "kern_ms": float(ev[0][0].elapsed_time(ev[0][1]) / steps) if ev else None,
           "gather_ms": float(np.mean([g0.elapsed_time(g1) for g0, g1 in gev])) if gev else None,
           "sustained_ms": None, "clock": None}
    # ---- outside the timed region: the shader clock THIS device holds under THIS workload (a sleeping wave on a side stream
    # samples s_memtime against the constant reference clock while the steps below run on the other CUs: afx_clock_probe_start)
    probe = None
    if not dry:
        try:
            from audioflux_amd.batch import ClockProbe
            probe = ClockProbe(torch, dev)
        except Exception:
            probe = None
    if sustained_s > 0 and world == 1 and not dry:  # outside the timed region
        n, s0, s1 = 0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        if probe is not None:
            probe.start()
        s0.record()
        while True:
            w.step(n)
            n += 1
            if n % 8 == 0 or sustained_s < 1.0:
                # (the launch stream only: a device-wide synchronize would wait for the probe's sleeping wave on its side stream,
                #  which ends when the flag behind this loop is written -- i.e. at its 20 s limit)
                torch.cuda.current_stream(dev).synchronize()
                if time.perf_counter() - t1 >= sustained_s:
                    break
        s1.record()
        if probe is not None:
            res["clock"] = dict(probe.stop(), over="sustained loop", steps=n)
        torch.cuda.synchronize()
        res["sustained_ms"] = s0.elapsed_time(s1) / n
        res["last"] = n - 1
    elif probe is not None and steps > 0 and world > 1:  # N > 1 (no sustained loop): K more steps, untimed, under the probe
        # (N = 1 with --no-sustained is a profiling run: tools/prof_*.py count on warm-up + K launches exactly)
        probe.start()
        for i in range(steps):
            w.step(res["last"] + 1 + i)
        res["clock"] = dict(probe.stop(), over="untimed repeat of the K steps", steps=steps)
        sync()
        res["last"] += steps
    return res


PEAK_CLOCK_HZ = 2.4e9          # MI355X_MICROARCH.md: peak engine clock
VALU_PEAK_WAVE_INSTS = 1024 * PEAK_CLOCK_HZ / 4.0   # 256 CUs x 4 SIMDs, one 64-lane f32 instruction per 4 cycles
MFMA_F16_PEAK_FLOPS = 2.5e15   # dense f16 / bf16 matrix peak


def pmc_compute(config, units=None, kern_ms=None):
    """the compute side of the roofline from the round's PMC passes (tools/gpu_call6.sh evidence + tools/prof_compute.py): how busy
    the vector unit, the LDS, the SIMDs' issue ports and the matrix pipe were under the dominant kernel -- what bounds a
    kernel that is nowhere near the HBM roofline; replayed from profiles/, like `traffic`.  `frac_of_peak` relates THIS
    run's rate to the unit that binds: cfg 2 vector instructions per second / (1024 SIMDs x 2.4 GHz / 4); cfg 5 matrix
    flop/s (tiles x 807 MFMAs x 32768 flop) / the 2.5 PF dense f16 peak."""
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_bench_cfg{config}_compute.json")
        try:
            rec = json.load(open(path))
            out = {k: rec.get(k) for k in ("valu_busy", "lds_busy", "issue_busy", "mfma_busy", "lds_bank_conflict_share",
                                           "shader_clock_mhz", "valu_insts_per_unit_and_wave", "lds_insts_per_unit_and_wave")}
            busiest = max((v, k) for k, v in out.items() if k.endswith("_busy") and k != "issue_busy" and v is not None)
            out["busiest_unit"] = {"valu_busy": "vector unit", "lds_busy": "LDS", "mfma_busy": "matrix pipe"}[busiest[1]]
            if units and kern_ms:
                if config == 5:  # 32-frame tiles x 807 v_mfma_f32_32x32x16_f16 of 32768 flop each
                    flops = (units / 32.0) * 807 * 32768 / (kern_ms * 1e-3)
                    out["mfma_flops"], out["frac_of_peak"], out["peak_of"] = flops, flops / MFMA_F16_PEAK_FLOPS, "2.5 PF dense f16 MFMA"
                elif out.get("valu_insts_per_unit_and_wave"):
                    rate = units * out["valu_insts_per_unit_and_wave"] / (kern_ms * 1e-3)
                    out["valu_wave_insts_per_s"], out["frac_of_peak"] = rate, rate / VALU_PEAK_WAVE_INSTS
                    out["peak_of"] = "vector issue: 1024 SIMDs x 2.4 GHz / 4 cycles per 64-lane instruction"
            out["kernel"] = rec.get("kernel")
            for k in ("classes", "share_of_summed_kernel_time", "occupancy"):  # (cfg 4: per kernel class + the step's union / sum of busy intervals)
                if rec.get(k) is not None:
                    out[k] = rec[k]
            out["source"] = os.path.relpath(path, ROOT)
            return out
        except (OSError, KeyError, ValueError):
            continue
    return None


def roofline(W, w, clips, m):
    kern_ms, sus = m["kern_ms"], m["sustained_ms"]
    achieved = w.units * W.bytes_per_unit / (kern_ms * 1e-3) / 1e9 if kern_ms else None
    traffic, traffic_src = pmc_traffic(W.config, clips) if hasattr(W, "config") else (None, None)
    alg = w.units * W.bytes_per_unit
    return {"bound": getattr(W, "bound", "hbm"), "bound_note": getattr(W, "bound_note", None), "priced_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_algorithmic": (traffic / alg) if traffic else None,
            "kernel": W.kernel, "kernel_ms": kern_ms,
            "algorithmic_bytes_per_unit": W.bytes_per_unit, "units_per_launch": w.units,
            "sustained_ms": sus, "sustained_value": (w.units / (sus * 1e-3)) if sus else None,
            "sustained_frac": (alg / (sus * 1e-3) / 1e9 / HBM_PEAK_GBS) if sus else None,
            # the shader clock of THIS run on THIS device (box-to-box spread of the step time is clock spread: power-limited kernels)
            "clock_mhz_this_run": (m.get("clock") or {}).get("clock_mhz"), "clock_probe": m.get("clock"),
            "compute": pmc_compute(W.config, w.units, kern_ms) if hasattr(W, "config") else None}


def dense_bank_block(torch, af, dev, check=True):
    """north_star's MFMA evidence on the driver's line: the dense filter-bank route (gammatone-128 at the headline shape: spectrum rows
    from the headline transform, afxk_stft2k -> k_gemm_bank_bf16x3, three bf16 words per operand on the bf16 matrix cores) as frames/s,
    and its product alone (one chunk's shape through the library's C entry points) as matrix FLOP/s against the dense bf16 peak.
    Reference: __mdot1, src/vector/flux_vector.c:55-86, on the rows of src/stft_algorithm.c:717-803."""
    import ctypes as C
    clips, n = 1000, 16000 * 30
    gen = torch.Generator(device=dev).manual_seed(7)
    x = 0.1 * torch.randn((clips, n), generator=gen, device=dev, dtype=torch.float32)
    o = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
               scale_type=af.SpectralFilterBankScaleType.ERB, style_type=af.SpectralFilterBankStyleType.GAMMATONE,
               data_type=af.SpectralDataType.POWER)
    o.set_result_type(1)
    T = o.cal_time_length(n)
    out = torch.empty((clips, T, 128), device=dev, dtype=torch.float32)
    t_end = time.time() + 0.3
    while time.time() < t_end:  # clock warm-up
        o.bft_device(x, out_real=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k = 10
    e0.record()
    for _ in range(k):
        o.bft_device(x, out_real=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / k
    err = None
    if check:
        from oracle import ref
        if ref.available():
            r = ref.RefBFT(128, 11, samplate=16000, low_fre=0.0, high_fre=8000.0, window_type=1, slide_length=512, scale_type=4,
                           style_type=2, normal_type=0, data_type=0)
            r.set_result_type(1)
            want, _ = r.bft(x[0].cpu().numpy())
            err = _parity(out[0].cpu().numpy(), want)
            assert err <= 1e-5, f"dense bank: benchmarked output differs from the oracle: {err:.3e}"
    # the product alone: one chunk of the route (250 clips: 233 500 rows x 1025 bins, pitch 1028) x the 128 x 1025 bank
    lib = af.get_lib()
    vp, ll = C.c_void_p, C.c_longlong
    lib.afxk_gemm_bank_prepare.restype = C.c_int
    lib.afxk_gemm_bank_prepare.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), vp]
    lib.afxk_gemm_nt_bank.restype = C.c_int
    lib.afxk_gemm_nt_bank.argtypes = [vp, ll, vp, C.c_int, C.c_int, vp, ll, ll, C.c_int, C.c_float, vp]
    lib.afxdev_free.argtypes = [vp]
    M, N, K, P = 250 * T, 128, 1025, 1028
    del x
    A = torch.zeros((M, P), device=dev)
    A[:, :K] = torch.randn((M, K), device=dev, generator=gen) ** 2 * 10.0 ** (10 * torch.rand((M, K), device=dev, generator=gen) - 5)
    B = torch.zeros((N, P), device=dev)
    B[:, :K] = torch.rand((N, K), device=dev, generator=gen)
    Cm = out.view(-1)[:M * N].view(M, N)
    img, stream = vp(), torch.cuda.current_stream().cuda_stream
    assert lib.afxk_gemm_bank_prepare(B.data_ptr(), P, N, K, C.byref(img), stream) == 0
    run = lambda: lib.afxk_gemm_nt_bank(A.data_ptr(), P, img, N, K, Cm.data_ptr(), N, M, 0, 0.0, stream)
    for _ in range(40):
        assert run() == 0, af.last_error()
    torch.cuda.synchronize()
    kk = 40
    e0.record()
    for _ in range(kk):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / kk
    rows = torch.arange(0, M, M // 256, device=dev)
    want = A[rows, :K].double() @ B[:, :K].double().T
    gerr = ((Cm[rows].double() - want).abs() / want.abs().clamp_min(1e-300)).max().item()
    lib.afxdev_free(img)
    flop = 2.0 * M * N * K
    blk = {"metric": "audio frames/sec (gammatone-128 dense bank, n_fft=2048 hop=512)", "value": clips * T / ms * 1e3, "unit": "frames/s",
           "steps": k, "ms_per_step": ms, "dtype": "f32 (operands as three bf16 words each on the bf16 matrix cores, f32 accumulation)",
           "workload": f"batched STFT -> dense gammatone-128 bank, {clips} x 30 s @16 kHz, n_fft=2048 hop=512 (the MFMA route north_star names)",
           "kernels": "k_stft_mel_v2<STFT> (afxk_stft2k: spectrum rows) -> k_gemm_bank_bf16x3 (128 x 128 tiles, v_mfma_f32_32x32x16_bf16)",
           "oracle_check": {"clip0_max_rel_err": err, "bar": 1e-5},
           "roofline": {"bound": "mfma", "kernel": "k_gemm_bank_bf16x3", "kernel_us": us, "shape": [M, N, K],
                        "achieved": 6.0 * flop / us / 1e6, "peak": 2500.0, "unit": "TFLOP/s", "frac": 6.0 * flop / us / 1e6 / 2500.0,
                        "f32_equivalent_tflops": flop / us / 1e6, "products_per_f32_product": 6,
                        "elementwise_err_vs_f64": gerr,
                        "mfma_busy": 0.495, "mfma_busy_source": "profiles/r06_rocprofv3_dense.txt (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8))",
                        "note": "bound by the registers' operand fill rate at a 64 x 64 wave tile (profiles/r06_dense.txt (c)), not by the matrix pipe"}}
    return blk


def device_info(torch, dev):
    """what ran the numbers: name / arch / CU count / PCI bus id from the runtime, power cap and temperature from rocm-smi
    when it answers within a few seconds (never a reason to lose the line)"""
    p = torch.cuda.get_device_properties(dev)
    info = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
            "hbm_gib": round(p.total_memory / 2**30, 1),
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
            "hip": getattr(torch.version, "hip", None)}
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "-d", str(dev.index or 0), "--showmaxpower", "--showpower", "--showtemp", "--json"],
                           capture_output=True, text=True, timeout=8)
        card = next(iter(json.loads(r.stdout).values()))
        for k, v in card.items():
            kl = k.lower()
            if "max graphics package power" in kl:
                info["power_cap_w"] = float(v)
            elif "package power" in kl and "max" not in kl:
                info["power_w_idle_sample"] = float(v)
            elif "temperature" in kl and "junction" in kl:
                info["junction_c"] = float(v)
    except Exception:
        pass
    return info


def self_launch(n):
    """re-exec this command line as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` (one
    process per GPU); the ranks inherit stdout / stderr, so the launcher adds nothing to the JSON line"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU per step (0: the configuration's own)")
    ap.add_argument("--total-clips", type=int, default=0, help="clips of the whole job, sharded over the ranks in "
                    "contiguous ceil-sized blocks (dist.shard_range; the last ranks may get fewer) instead of --clips each")
    ap.add_argument("--gather", default="", help="comma list of feature slabs gathered to rank 0 when N > 1 "
                    "(cfg 2: mfcc[,mel]; cfg 5: chroma[,cqt]); default: the first one")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="--config 2 at one GPU also measures cfg 4 and cfg 5 "
                    "after the headline (outside its timed region) and reports them under `secondary`")
    ap.add_argument("--no-legacy", action="store_true", help="skip the reference's published benchmark through its own wrapper "
                    "(tools/legacy_bench.py; --config 2 at one GPU)")
    ap.add_argument("--n1-value", type=float, default=0.0, help="the N = 1 value of the same configuration (BENCH line): with N > 1 the "
                    "line then carries efficiency_vs_n1 = value / (N x n1-value), the weak-scaling efficiency north_star asks for")
    ap.add_argument("--clock-warmup", type=float, default=0.5, help="seconds of untimed steps before the W warm-up "
                    "steps: the device ramps its clocks over tens of ms after idling, and W steps of ~1.5 ms end "
                    "long before that (0 disables; reported as config.clock_warmup_s)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher environment, so this process becomes the launcher -- the same
        # command under torch.distributed.run, one rank per GPU on a free port of 127.0.0.1; rank 0 prints the one
        # JSON line on the inherited stdout, the exit status is the job's
        sys.exit(self_launch(a.gpus))

    import torch
    import torch.distributed as dist

    import audioflux_amd as af
    from audioflux_amd import dist as afd

    dry = os.environ.get("AFX_BENCH_DRYRUN") == "1"  # CPU stand-in for the kernels, gloo backend (tests)
    rank, local, world = afd.init_from_env(backend="gloo" if dry else None)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if dry:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        af._lib.check(af.get_lib().afx_set_device(local), "afx_set_device")

    W = DryRun if dry else WORKLOADS[a.config]
    if a.total_clips > 0:
        spans = [afd.shard_range(a.total_clips, r, world) for r in range(world)]
        counts = [hi - lo for lo, hi in spans]
        clips, first = counts[rank], spans[rank][0]
        assert min(counts) > 0, f"--total-clips {a.total_clips} leaves a rank of {world} without clips"
    else:
        clips = a.clips or W.default_clips
        counts, first = [clips] * world, rank * clips
    w = W(torch, af, dev, rank, clips, a.config, first) if dry else W(torch, af, dev, rank, clips)
    choices = w.gather_choices if dry else W.gather_choices
    which = [g for g in (a.gather.split(",") if a.gather else list(choices[:1])) if g]
    for g in which:
        assert g in choices, f"--gather {g}: config {a.config} offers {choices}"
    gathers = ([afd.FeatureGather(dst=0, counts=counts) for _ in which] if (world > 1 and not a.no_gather) else [])

    m = measure(w, torch, dist, dev, world, a.steps, a.warmup, a.clock_warmup,
                0.0 if a.no_sustained else 1.0, gathers, which, dry)
    elapsed = m["elapsed"]

    # ---- outside the timed region ----------------------------------------------------------------
    if dry and gathers and rank == 0:
        # the gathered slab of the last step: rank r's clips follow rank r-1's, every clip carries its id
        got = gathers[0].wait()
        total = sum(counts)
        want = torch.arange(total, dtype=torch.float32) * 1000 + (a.steps - 1)
        assert got.shape[0] == total and got.shape[1] == w.T and bool((got[:, 0, 0] == want).all()), "gathered slab order"
    err = None
    if rank == 0 and not a.no_check:
        err = w.check(m["last"])
        assert err is None or err <= 1e-5, f"benchmarked output differs from the oracle: {err:.3e}"

    units_all = w.units
    if world > 1:  # shards may differ (--total-clips): the job's units are the sum over ranks
        tu = torch.tensor([w.units], device=dev, dtype=torch.float64)
        dist.all_reduce(tu, op=dist.ReduceOp.SUM)
        units_all = float(tu.item())
    else:
        units_all = float(w.units)

    if rank == 0:
        value = units_all * a.steps / elapsed
        par = f"clips sharded x{world}"
        if gathers:
            par += f", RCCL gather of {'+'.join(which)} to rank 0 (side stream, overlapped)"
        elif world > 1:
            par += ", replicas only (no gather)"
        out = {
            "metric": W.metric, "value": value, "unit": W.unit, "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "strong" if a.total_clips > 0 else "weak", "vs_baseline": None,
            "dtype": getattr(w, "dtype", "f32"), "data": "synthetic",
            "config": {"workload": w.workload, "clips_per_gpu": clips, "units_per_step_per_gpu": w.units,
                       "outputs": w.outputs, "parallelism": par, "clock_warmup_s": a.clock_warmup,
                       "device": None if dry else device_info(torch, dev)},
            "roofline": roofline(W, w, clips, m),
            "oracle_check": {"clip0_max_rel_err": err, "bar": 1e-5},
        }
        if a.total_clips > 0:
            out["config"]["total_clips"], out["config"]["clips_per_rank"] = a.total_clips, counts
        if world > 1:
            # does the gather hide behind the next step?  gather_ms: device time of the collective(s) on the side
            # stream of rank 0; exposed_ms: what a step costs beyond its own compute; hidden = the rest
            ms_step = 1e3 * elapsed / a.steps
            exposed = max(0.0, ms_step - (m["kern_ms"] or 0.0))
            gm = m["gather_ms"]
            out["gather"] = {"slabs": which if gathers else [], "gather_ms": gm, "compute_ms": m["kern_ms"],
                             "exposed_ms": exposed if gathers else 0.0,
                             "overlap_hidden_ms": max(0.0, gm - exposed) if (gathers and gm is not None) else None}
            assert not gathers or out["gather"]["gather_ms"] is not None, "N > 1 with a gather must report gather_ms"
            out["config"]["clips_per_rank"] = counts
            if a.n1_value > 0:  # weak scaling: N ranks x the one-GPU rate is 1.0
                out["efficiency_vs_n1"] = value / (world * a.n1_value) if a.total_clips <= 0 else value / a.n1_value / world
                out["n1_value"] = a.n1_value
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = w.cpu()
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and a.config == 2 and not dry and not a.no_secondary:
            # BASELINE configs[3] / [4] on the same line: measured in this process after the headline, each with
            # its own fences, device-event step time, sustained loop and oracle check (a few seconds in all)
            del w
            torch.cuda.empty_cache()
            out["secondary"] = {}
            for cfg, (k, wu) in ((4, (2, 1)), (5, (10, 2))):
                try:
                    W2 = WORKLOADS[cfg]
                    w2 = W2(torch, af, dev, rank, W2.default_clips)
                    m2 = measure(w2, torch, dist, dev, 1, k, wu, 0.3, 0.0 if a.no_sustained else 0.7)
                    e2 = None if a.no_check else w2.check(m2["last"])
                    assert e2 is None or e2 <= 1e-5, f"cfg {cfg}: benchmarked output differs from the oracle: {e2:.3e}"
                    r2 = roofline(W2, w2, W2.default_clips, m2)
                    out["secondary"][f"cfg{cfg}"] = {
                        "metric": W2.metric, "value": w2.units * k / m2["elapsed"], "unit": W2.unit, "steps": k,
                        "warmup": wu, "ms_per_step": 1e3 * m2["elapsed"] / k, "workload": w2.workload,
                        "dtype": getattr(w2, "dtype", "f32"), "frac": r2["frac"], "sustained_frac": r2["sustained_frac"],
                        "traffic": r2["traffic"], "traffic_over_algorithmic": r2["traffic_over_algorithmic"],
                        "roofline": r2, "oracle_check": {"clip0_max_rel_err": e2, "bar": 1e-5}}
                    if not a.no_cpu_baseline:  # the reference on the host cores, a few seconds per configuration
                        try:
                            out["secondary"][f"cfg{cfg}"]["cpu_baseline"] = w2.cpu(budget_s=6.0)
                        except Exception as e:
                            out["secondary"][f"cfg{cfg}"]["cpu_baseline"] = {"error": repr(e)}
                    del w2
                    torch.cuda.empty_cache()
                except Exception as e:  # never lose the headline line to a secondary configuration
                    out["secondary"][f"cfg{cfg}"] = {"error": repr(e)}
        if world == 1 and a.config == 2 and not dry and not a.no_secondary:
            try:  # the dense filter-bank route and its MFMA product (a few seconds; never a reason to lose the line)
                out["dense_bank"] = dense_bank_block(torch, af, dev, check=not a.no_check)
            except Exception as e:
                out["dense_bank"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if world == 1 and a.config == 2 and not dry and not a.no_legacy:
            # the reference's own published benchmark through its unmodified wrapper + this library (host pointers, one
            # clip per call, PCIe inside the clock) -- a fresh interpreter, after everything else
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "legacy_bench.py")], capture_output=True, text=True,
                                   timeout=90,  # (1000 calls of ~0.3 ms + one interpreter start; never a reason to lose the line)
                                   env=dict(os.environ, AFX_HIP_RUNTIME="system"))
                out["legacy"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out["legacy"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
