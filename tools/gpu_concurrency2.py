"""Follow-up of tools/gpu_concurrency.py: an FFT-only CWT object (the victim) beside different partners on another
stream; the victim's output is compared bitwise with its solo run.
   python tools/gpu_concurrency2.py <case> [chunks]        (AFX_LIB selects a probe build of the library)
cases: full (a full 84-scale object), seq (time-domain-only object, then an FFT-only object, on ONE stream),
       cqt (the CQT + chroma call), mel (the fused mel + MFCC call), fwdonly (partner = an FFT-only object)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
case = sys.argv[1] if len(sys.argv) > 1 else "full"
sys.argv = [sys.argv[0]] + sys.argv[2:]
import torch
import audioflux_amd as af
import tools.gpu_concurrency as gc
from tools.gpu_concurrency import Job, LOW, N, CH


def run(victim, partner_launch, partner_sync=None, reps=3, calls=3):
    bad = 0
    for rep in range(reps):
        ov = [victim.fresh() for _ in range(calls)]
        torch.cuda.synchronize()
        for i in range(calls):
            partner_launch(i)
            victim.launch(ov[i])
            partner_launch(i)
        torch.cuda.synchronize()
        for i in range(calls):
            bad += victim.check(ov[i], f"rep {rep} call {i}")
    return bad


def main():
    Job.explain = 0
    victim = Job("fft48", 48, LOW, 1)
    victim.gold = victim.solo()
    victim.check(victim.solo(), "solo repeat")
    s2 = torch.cuda.Stream()
    tag = os.environ.get("AFX_LIB", "shipped").split("/")[-1] + " queues=" + os.environ.get("GPU_MAX_HW_QUEUES", "default")
    if case == "full":
        p = Job("full84", 84, LOW, 5)
        outs = [p.fresh() for _ in range(2)]
        bad = run(victim, lambda i: p.launch(outs[i & 1]))
    elif case == "fwdonly":
        p = Job("fft48b", 48, LOW, 5)
        outs = [p.fresh() for _ in range(2)]
        bad = run(victim, lambda i: p.launch(outs[i & 1]))
    elif case == "seq":
        # the partner's stream carries what a full object's stream carries, as two objects: time-domain kernels, then FFT path
        t = Job("td36", 36, LOW * 16, 3)
        f = Job("fft48b", 48, LOW, 4)
        f.stream = t.stream
        ot, of = t.fresh(), f.fresh()

        def both(i):
            t.launch(ot)
            f.launch(of)
        bad = run(victim, both)
    elif case == "cqt":
        o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12)
        x = 0.1 * torch.randn((16, 44100 * 30), device="cuda")
        bufs = o.cqt_chroma_device(x, stream=s2)
        torch.cuda.synchronize()
        bad = run(victim, lambda i: o.cqt_chroma_device(x, out_real=bufs[0], out_imag=bufs[1], out=bufs[2], stream=s2))
    elif case == "mel":
        bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                     scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
        bft.set_result_type(1)
        xx = af.XXCC(128)
        x = 0.1 * torch.randn((500, 480000), device="cuda")

        def mel(i):
            with torch.cuda.stream(s2):
                af.mel_mfcc_device(bft, xx, x, 13)
        mel(0)
        torch.cuda.synchronize()
        bad = run(victim, mel)
    else:
        raise SystemExit("unknown case " + case)
    print(f"RESULT case {case} lib {tag}: victim fft48 wrong elements {bad}", flush=True)


if __name__ == "__main__":
    main()
