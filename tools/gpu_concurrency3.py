"""Every transform family as the victim: its batched device call on one stream while a matrix-core-heavy partner (a full
84-scale CWT object with its time-domain kernel, or the CQT + chroma call) runs on another; all outputs are compared
bitwise with the family's solo run.  Run with GPU_MAX_HW_QUEUES=8 so that the streams do not share a hardware queue.
   python tools/gpu_concurrency3.py [victim ...]      victims: mel stft ceps cqt cwt spec (default: all)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import audioflux_amd as af


def noise(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return 0.1 * torch.randn(shape, device="cuda", generator=g)


def victims():
    v = {}
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(128)
    xm = noise((300, 480000), 11)
    v["mel"] = lambda s: af.mel_mfcc_device(bft, xx, xm, 13, stream=s)
    st = af.STFT(radix2_exp=11, slide_length=512)
    xs = noise((40, 480000), 12)
    v["stft"] = lambda s: st.stft_device(xs, stream=s)
    ce = af.Cepstrogram(radix2_exp=11, samplate=16000, slide_length=512)
    xc = noise((60, 480000), 13)
    v["ceps"] = lambda s: ce.cepstrogram_device(xc, 4, stream=s)
    cq = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12)
    xq = noise((24, 44100 * 30), 14)
    v["cqt"] = lambda s: cq.cqt_chroma_device(xq, stream=s)
    cw = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
    xw = noise((48, 65536), 15)
    v["cwt"] = lambda s: cw.cwt_device(xw, stream=s)
    sp = af.BFT(128, radix2_exp=12, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=1024,
                scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    sp.set_result_type(1)
    xp = noise((100, 480000), 16)
    v["spec"] = lambda s: sp.bft_device(xp, stream=s)
    return v


def partners():
    p = {}
    cw = af.CWT(num=84, radix2_exp=16, samplate=44100, low_fre=32.703, bin_per_octave=12,
                wavelet_type=af.WaveletContinueType.MORLET, scale_type=af.SpectralFilterBankScaleType.OCTAVE, is_padding=True)
    xw = noise((64, 65536), 21)
    bw = cw.cwt_device(xw)
    p["full CWT"] = lambda s: cw.cwt_device(xw, bw[0], bw[1], stream=s)
    cq = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12)
    xq = noise((16, 44100 * 30), 22)
    bq = cq.cqt_chroma_device(xq)
    p["CQT + chroma"] = lambda s: cq.cqt_chroma_device(xq, out_real=bq[0], out_imag=bq[1], out=bq[2], stream=s)
    torch.cuda.synchronize()
    return p


def outs_of(r):
    if isinstance(r, torch.Tensor):
        return [r]
    return [t for t in r if t is not None]


def main():
    want = sys.argv[1:] or ["mel", "stft", "ceps", "cqt", "cwt", "spec"]
    vs, ps = victims(), partners()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    total = 0
    for name in want:
        run = vs[name]
        first = outs_of(run(sa))
        torch.cuda.synchronize()
        gold = [t.clone() for t in first]
        torch.cuda.synchronize()
        again = outs_of(run(sa))
        torch.cuda.synchronize()
        solo = sum(int((a != b).sum()) for a, b in zip(again, gold))
        for pname, prun in ps.items():
            bad = 0
            for rep in range(3):
                outs = []
                for _ in range(2):
                    prun(sb)
                    outs.append(outs_of(run(sa)))
                    prun(sb)
                torch.cuda.synchronize()
                for o in outs:
                    for k, (a, b) in enumerate(zip(o, gold)):
                        d = a != b
                        n = int(d.sum())
                        bad += n
                        if n and bad == n:
                            idx = d.flatten().nonzero().flatten()
                            runs = int(((idx[1:] - idx[:-1]) != 1).sum()) + 1
                            print(f"   output {k} shape {tuple(a.shape)}: {n} wrong in {runs} runs, first flat index {int(idx[0])}, "
                                  f"got {a.flatten()[idx[:3]].tolist()} want {b.flatten()[idx[:3]].tolist()}", flush=True)
                del outs
            total += bad
            print(f"RESULT victim {name} beside {pname}: wrong elements {bad} (solo repeat: {solo})", flush=True)
    print(f"TOTAL wrong elements {total}", flush=True)


if __name__ == "__main__":
    main()
