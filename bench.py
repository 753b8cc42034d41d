#!/usr/bin/env python
"""bench.py -- audio frames/sec for batched STFT -> mel-128 -> MFCC-13
(n_fft = 2048, hop = 512) on synthetic 16 kHz mono clips, BASELINE.json's metric.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: `--clips` clips (default 1000,
BASELINE cfg 2) of 30 s per GPU, inputs already resident in HBM, features (mel and
MFCC) left in HBM; with N > 1 each rank owns its own batch (weak scaling) and the MFCC
slab is gathered to rank 0 over RCCL, overlapped with the next step.  Timing: barrier +
torch.cuda.synchronize() on both sides of exactly K steps, max over ranks.  Rank 0
prints ONE JSON line.

`roofline` is for the dominant kernel of the step, timed live with HIP events on the
stream it is launched on (torch's current stream is handed to the library);
`cpu_baseline` times the reference's own C path (oracle/_ref, built-in FFT + naive
double-accumulating matmul) on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, NFFT, HOP, NMEL, NCC = 16000, 2048, 512, 128, 13
CLIP_SECONDS = 30
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_FRAME = 4 * HOP + 4 * NMEL + 4 * NCC  # each sample read once, each output written once
BYTES_PER_FRAME_MEL_KERNEL = 4 * HOP + 4 * NMEL   # what the dominant kernel itself must move


def cpu_worker(args):
    """one host process: reference mel+MFCC over `n` clips (objects pre-built, one warm-up)"""
    seed, n, threads = args
    if threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    else:
        os.environ.pop("OMP_NUM_THREADS", None)
    from oracle import ref
    length = SR * CLIP_SECONDS
    x = (0.1 * np.random.default_rng(seed).standard_normal((n, length))).astype(np.float32)
    bft = ref.RefBFT(NMEL, 11, samplate=SR, low_fre=0.0, high_fre=SR / 2.0, window_type=1,
                     slide_length=HOP, scale_type=2, style_type=0, normal_type=0, data_type=0)
    bft.set_result_type(1)
    cc = ref.RefXXCC(NMEL)
    re, _ = bft.bft(x[0])
    cc.xxcc(re, NCC, 0)  # warm-up
    t0 = time.perf_counter()
    frames = 0
    for i in range(n):
        re, _ = bft.bft(x[i])
        cc.xxcc(re, NCC, 0)
        frames += re.shape[0]
    return frames, time.perf_counter() - t0


def _effective_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(budget_s=25.0):
    """reference CPU path (oracle/_ref: the reference's own C, built-in radix-2 FFT + naive
    double-accumulating matmul, no FFTW/MKL) on this box's host cores -- BASELINE.md section 3:
    (A) as shipped: one process, default OpenMP (frame loop on omp_get_max_threads()/2 threads,
        src/stft_algorithm.c:95-100; the matmul is serial);
    (B) all cores: P worker processes with one FFT thread each (OMP_NUM_THREADS=2), clips
        sharded; P is swept because hosts differ (SMT, cgroup quotas) and the best aggregate
        is reported.  Bounded sample, objects pre-built, one warm-up call per worker."""
    from oracle import ref
    if not ref.available():
        return None
    import multiprocessing as mp
    t_start = time.perf_counter()
    ncpu = _effective_cpus()
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:  # (A) in a fresh process so OMP defaults apply
        fa, ta = pool.map(cpu_worker, [(1000, 8, 0)])[0]
    best = {"value": fa / ta, "cores": ncpu, "how": "A: 1 process, default OpenMP", "frames": fa}
    tried = [f"A=1proc:{fa / ta:.0f}"]
    sizes = sorted({p for p in (8, 16, 32, 64, 128, 256, ncpu) if p <= ncpu})
    for p in sizes:
        if time.perf_counter() - t_start > budget_s:
            break
        with ctx.Pool(p) as pool:
            res = pool.map(cpu_worker, [(2000 + i, 4, 2) for i in range(p)])
        wall = max(r[1] for r in res)  # workers run concurrently; the slowest bounds the job
        frames = sum(r[0] for r in res)
        rate = frames / wall
        tried.append(f"B={p}procs:{rate:.0f}")
        if rate > best["value"]:
            best = {"value": rate, "cores": p, "how": f"B: {p} processes x 1 FFT thread", "frames": frames,
                    "procs": p}
    # the sweep samples are short; re-time the best configuration on a sample sized for ~12 s of
    # wall time (bounded: <= 96 clips per worker) and report that measurement
    if "procs" in best and time.perf_counter() - t_start < budget_s:
        p = best["procs"]
        per_proc = best["value"] / p
        n = int(max(8, min(96, 12.0 * per_proc / ((SR * CLIP_SECONDS - NFFT) // HOP + 1))))
        with ctx.Pool(p) as pool:
            res = pool.map(cpu_worker, [(3000 + i, n, 2) for i in range(p)])
        wall = max(r[1] for r in res)
        frames = sum(r[0] for r in res)
        tried.append(f"final={p}procsx{n}clips:{frames / wall:.0f}")
        best.update(value=frames / wall, frames=frames)
    return {"value": best["value"], "unit": "frames/s", "cores": best["cores"], "kind": "reference",
            "sample": f"{best['how']}, {best['frames']} frames of {CLIP_SECONDS} s @16 kHz clips "
                      f"(mel-128 + MFCC-13, n_fft 2048, hop 512); built-in radix-2 FFT + naive matmul "
                      f"(no FFTW/MKL); visible cpus {ncpu}; sweep frames/s: " + " ".join(tried),
            "wall_s": round(time.perf_counter() - t_start, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=1000, help="clips per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import audioflux_amd as af
    from audioflux_amd import dist as afd

    rank, local, world = afd.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    os.environ.setdefault("AFX_DEVICE", str(local))
    get = af.get_lib()
    get.afx_set_device(local)

    length = SR * CLIP_SECONDS
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    x = 0.1 * torch.randn((a.clips, length), generator=gen, device=dev, dtype=torch.float32)

    bft = af.BFT(NMEL, radix2_exp=11, samplate=SR, low_fre=0.0, high_fre=SR / 2.0, slide_length=HOP,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(NMEL)
    T = bft.cal_time_length(length)
    frames_per_step = a.clips * T
    mel = torch.empty((a.clips, T, NMEL), device=dev, dtype=torch.float32)
    ccs = [torch.empty((a.clips, T, NCC), device=dev, dtype=torch.float32) for _ in range(2)]
    gather = afd.FeatureGather(dst=0) if (world > 1 and not a.no_gather) else None
    comm = torch.cuda.Stream(device=dev) if gather else None

    ev_pairs = []

    def step(i, timed):
        cc = ccs[i & 1]
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e2 = torch.cuda.Event(enable_timing=True)
        # the two launches of afx_bftXxccBatchDevice, issued separately so that HIP events
        # on the launch stream bracket the dominant kernel alone
        e0.record()
        bft.bft_device(x, out_real=mel)          # k_stft_mel_banded: STFT -> |S|^2 -> mel bank
        e1.record()
        xx.xxcc_device(mel, NCC, out=cc)         # k_cepstrum_mfma: log10 -> DCT-II -> 13 coeffs
        e2.record()
        if timed:
            ev_pairs.append((e0, e1, e2))
        if gather:
            # MFCC slab of this step goes to rank 0 while the next step computes
            gather.wait()
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                gather.start(cc)

    def fence():
        if gather:
            gather.wait()
            torch.cuda.current_stream().wait_stream(comm)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i, False)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in ev_pairs])) if ev_pairs else None
    cep_ms = float(np.mean([e1.elapsed_time(e2) for _, e1, e2 in ev_pairs])) if ev_pairs else None

    if rank == 0:
        total_frames = frames_per_step * world * a.steps
        value = total_frames / elapsed
        achieved = frames_per_step * BYTES_PER_FRAME_MEL_KERNEL / (kern_ms * 1e-3) / 1e9 if kern_ms else None
        out = {
            "metric": "audio frames/sec (mel+MFCC, n_fft=2048 hop=512)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batched STFT->mel{NMEL}->MFCC{NCC}, {a.clips} x {CLIP_SECONDS} s "
                                   f"@16 kHz per GPU, n_fft={NFFT} hop={HOP} (BASELINE cfg 2)",
                       "clips_per_gpu": a.clips, "frames_per_step_per_gpu": frames_per_step,
                       "outputs": "mel[clips,T,128] + mfcc[clips,T,13] f32 in HBM",
                       "parallelism": f"clips sharded x{world}" + (", RCCL gather of MFCC to rank 0" if gather else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         # HBM bytes per launch from rocprofv3 PMC passes of this same command
                         # (profiles/r01_final_rocprofv3_bench_summary.txt): 2*FETCH_SIZE (gfx950 counts
                         # wide coalesced reads at half) + WRITE_SIZE, in KiB -> bytes; scales with clips
                         "traffic": (2 * 955752 + 467001) * 1024 * (a.clips / 1000.0),
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01_final_*",
                         "kernel": "k_stft_mel_banded (framed FFT -> |S|^2 -> banded mel bank), "
                                   "one launch per step",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_frame": BYTES_PER_FRAME_MEL_KERNEL,
                         "frames_per_launch": frames_per_step,
                         "second_kernel": {"name": "k_cepstrum_mfma (log10 + DCT-II)", "kernel_ms": cep_ms,
                                           "algorithmic_bytes_per_frame": 4 * NMEL + 4 * NCC},
                         "step_algorithmic_bytes_per_frame": BYTES_PER_FRAME},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
