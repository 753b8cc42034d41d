"""Do CWT objects on different streams disturb each other?  Round 3 found that afx_cwt_td.hip's kernel, run on a side
stream beside the FFT-path kernels of the SAME call, left wrong 64-byte pieces in rows it never writes
(profiles/r03_cwt_td_schedules.txt).  This probe separates the candidates with independent objects / co-runner kernels
on two streams, each result compared bitwise with the object's own solo run:
   fft  = the 48 lowest scales (FFT path only)      td = the 36 highest scales (time-domain kernel only)
   pairs: fft|td, fft|fft, td|td, full|full, fft|<co-runner of tools/micro/occupant.hip>
   python tools/gpu_concurrency.py [chunks]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import audioflux_amd as af
from audioflux_amd import _lib

CH = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N = 65536
LOW = 32.703


def make(num, low):
    return af.CWT(num=num, radix2_exp=16, samplate=44100, low_fre=low, bin_per_octave=12,
                  wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE,
                  is_padding=True)


class Job:
    def __init__(self, name, num, low, seed):
        self.name, self.o, self.num = name, make(num, low), num
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.x = 0.1 * torch.randn((CH, N), device="cuda", generator=g)
        self.stream = torch.cuda.Stream()
        self.gold = None

    def fresh(self):
        return (torch.full((CH, self.num, N), float("nan"), device="cuda"),
                torch.full((CH, self.num, N), float("nan"), device="cuda"))

    def launch(self, out):
        self.o.cwt_device(self.x, out[0], out[1], stream=self.stream)

    def solo(self):
        out = self.fresh()
        torch.cuda.synchronize()
        self.launch(out)
        torch.cuda.synchronize()
        return out

    def check(self, out, tag):
        d = (out[0] != self.gold[0]) | (out[1] != self.gold[1])
        n = int(d.sum())
        line = f"   {tag}: {self.name} wrong elements {n}"
        if n:
            rows = d.sum(dim=(0, 2)).nonzero().flatten().tolist()
            per = d.sum(dim=2)
            c, s = [int(v) for v in (per > 0).nonzero()[0]]
            w = d[c, s].nonzero().flatten()
            runs = int(((w[1:] - w[:-1]) != 1).sum()) + 1
            nan = int((torch.isnan(out[0]) & d).sum())
            line += f" rows {rows}; e.g. chunk {c} row {s}: {int(w.numel())} wrong in {runs} runs, first n {int(w[0])}; NaN among wrong {nan}"
        print(line, flush=True)
        if n and Job.explain > 0:
            Job.explain -= 1
            self.explain_runs(out, d)
        return n

    explain = 6
    others = []

    def explain_runs(self, out, d):
        """where does the data of the first wrong 16-sample runs come from?  exact search of the run in the golden
        planes of this object and of every other job"""
        idx = d.flatten().nonzero().flatten()
        starts = idx[torch.cat([torch.ones(1, dtype=torch.bool, device=idx.device), (idx[1:] - idx[:-1]) != 1])][:4].tolist()
        for st in starts:
            c, s, n = st // (self.num * N), (st // N) % self.num, st % N
            for pl, name in ((0, "re"), (1, "im")):
                v = out[pl].flatten()[st:st + 16]
                gv = self.gold[pl].flatten()[st:st + 16]
                same = bool((v == gv).all())
                msg = f"      run at chunk {c} row {s} n {n} plane {name}: " + ("equal to golden" if same else f"got {v[:3].tolist()} want {gv[:3].tolist()}")
                if not same:
                    hits = []
                    for j in [self] + [o for o in Job.others if o is not self]:
                        for pl2, name2 in ((0, "re"), (1, "im")):
                            g = j.gold[pl2].flatten()
                            cand = (g == v[0]).nonzero().flatten()[:64].tolist()
                            for q in cand:
                                if q + 16 <= g.numel() and bool((g[q:q + 16] == v).all()):
                                    hits.append((j.name, name2, q // (j.num * N), (q // N) % j.num, q % N))
                    msg += f"; found in golden data at {hits[:4]}" if hits else "; not found in any golden plane"
                print(msg, flush=True)


def pair(a, b, reps=3, calls=3):
    print(f"== {a.name} | {b.name}", flush=True)
    bad = 0
    for rep in range(reps):
        oa = [a.fresh() for _ in range(calls)]
        ob = [b.fresh() for _ in range(calls)]
        torch.cuda.synchronize()
        for i in range(calls):
            a.launch(oa[i])
            b.launch(ob[i])
        torch.cuda.synchronize()
        for i in range(calls):
            bad += a.check(oa[i], f"rep {rep} call {i}")
            bad += b.check(ob[i], f"rep {rep} call {i}")
        del oa, ob
    return bad


def with_corunner(a, name, launch_co, reps=3, calls=3):
    print(f"== {a.name} | {name}", flush=True)
    s2 = torch.cuda.Stream()
    bad = 0
    for rep in range(reps):
        oa = [a.fresh() for _ in range(calls)]
        torch.cuda.synchronize()
        for i in range(calls):
            for _ in range(6):
                launch_co(s2, oa[i])
            a.launch(oa[i])
            for _ in range(6):
                launch_co(s2, oa[i])
        torch.cuda.synchronize()
        for i in range(calls):
            bad += a.check(oa[i], f"rep {rep} call {i}")
        del oa
    return bad


def main():
    so = os.path.join(ROOT, "tools", "micro", "libocc.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "micro", "occupant.hip"), "-o", so])
    _lib.get_lib()
    occ = ctypes.CDLL(so)
    occ.occ_lds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    occ.occ_mem.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    occ.occ_oob.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

    fft = Job("fft48", 48, LOW, 1)
    fft2 = Job("fft48b", 48, LOW, 2)
    td = Job("td36", 36, LOW * 16, 3)
    td2 = Job("td36b", 36, LOW * 16, 4)
    full = Job("full84", 84, LOW, 5)
    full2 = Job("full84b", 84, LOW, 6)
    for j in (fft, fft2, td, td2, full, full2):
        j.gold = j.solo()
        again = j.solo()
        j.check(again, "solo repeat")
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o = j.fresh()
        torch.cuda.synchronize()
        with torch.cuda.stream(j.stream):
            t0.record()
            j.launch(o)
            t1.record()
        torch.cuda.synchronize()
        print(f"   {j.name}: {t0.elapsed_time(t1):.2f} ms per call of {CH} chunks", flush=True)
        del o, again

    Job.others = [fft, fft2, td, td2, full, full2]
    res = {}
    if len(sys.argv) > 2 and sys.argv[2] == "co":
        Job.explain = 0
    res["fft|td"] = pair(fft, td)
    res["full|full"] = pair(full, full2, reps=2)
    res["full|fft"] = pair(full, fft, reps=2)
    res["full|td"] = pair(full, td, reps=2)
    res["fft|td"] = pair(fft, td)
    res["fft|fft"] = pair(fft, fft2)
    res["td|td"] = pair(td, td2)
    res["full alone, 3 calls back to back"] = with_corunner(full, "nothing (3 calls back to back)", lambda s, o: 0, reps=2)

    sink = torch.zeros(1 << 20, device="cuda")
    src = torch.randn(1 << 26, device="cuda")
    dst = torch.empty_like(src)
    res["fft|lds150k+mfma"] = with_corunner(fft, "co-runner: 150,272 B of LDS per workgroup, LDS sweeps + MFMA, no global traffic",
                                            lambda s, o: occ.occ_lds(s.cuda_stream, 256, 150272, 24, sink.data_ptr()))
    res["fft|lds68k+mfma"] = with_corunner(fft, "co-runner: 68,096 B of LDS per workgroup (two per CU), LDS sweeps + MFMA",
                                           lambda s, o: occ.occ_lds(s.cuda_stream, 512, 68096, 48, sink.data_ptr()))
    res["fft|lds8k+mfma"] = with_corunner(fft, "co-runner: 8 KB of LDS per workgroup, LDS sweeps + MFMA",
                                          lambda s, o: occ.occ_lds(s.cuda_stream, 2048, 8192, 200, sink.data_ptr()))
    res["fft|mem"] = with_corunner(fft, "co-runner: 16-byte raw buffer loads + stores streaming 256 MB (private buffers)",
                                   lambda s, o: occ.occ_mem(s.cuda_stream, 2048, src.data_ptr(), dst.data_ptr(), (1 << 26) // 4, 2))
    res["fft|oob"] = with_corunner(fft, "co-runner: out-of-range (dropped) raw buffer stores / loads against the live output buffer",
                                   lambda s, o: occ.occ_oob(s.cuda_stream, 2048, o[0].data_ptr(), 400, sink.data_ptr()))
    print("summary (wrong elements over all repetitions):", res, flush=True)


if __name__ == "__main__":
    main()
