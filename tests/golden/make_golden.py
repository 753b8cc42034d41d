"""Generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (the library
compiled from /root/reference by oracle/Makefile) on the seeded inputs of
tests/cases.py.  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so these fixtures are what pins the oracle and the GPU path
where /root/reference is not mounted.

    python tests/golden/make_golden.py        # needs oracle/_ref/libaudioflux_ref.so
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests import cases  # noqa: E402


def run_bft_case(c):
    kw = cases.ctor_kwargs(c)
    o = ref.RefBFT(kw.pop("num"), kw.pop("radix2_exp"), **kw)
    assert o.status == 0, o.status
    o.set_result_type(c["result_type"])
    if "norm" in c:
        o.set_norm(c["norm"])
    x = cases.make_input(c["x"], c["samplate"])
    re, im = o.bft(x)
    out = {"re": re, "fre": o.fre_band(), "bin": o.bin_band()}
    if c["result_type"] == 0:
        out["im"] = im
    if c.get("is_temporal"):
        e, r, z = o.temporal(re.shape[0])
        out.update(energy=e, rms=r, zcr=z)
    return out


def make_stft():
    """stft.npz: STFT object (padding modes, streaming, inverse) from the compiled reference"""
    out = {}
    for name, c in cases.STFT_CASES.items():
        o = ref.RefSTFT(c["radix2_exp"], c["window_type"], c["slide_length"])
        assert o.status == 0
        pad = cases.stft_pad_args(c)
        if pad:
            o.enable_padding(1)
            o.set_padding(*pad)
        re, im = o.stft(cases.make_input(c["x"], 16000))
        out[f"{name}/re"], out[f"{name}/im"] = re, im
    for name, c in cases.STFT_STREAMS.items():
        o = ref.RefSTFT(c["radix2_exp"], c["window_type"], c["slide_length"], 1)
        x = cases.noise(c["seed"], sum(c["chunks"]))
        off, res, ims, tl = 0, [], [], []
        for n in c["chunks"]:
            re, im = o.stft(x[off:off + n])
            off += n
            tl.append(re.shape[0])
            res.append(re)
            ims.append(im)
        out[f"{name}/re"], out[f"{name}/im"] = np.concatenate(res), np.concatenate(ims)
        out[f"{name}/tl"] = np.array(tl, np.int32)
    for name, (src, method, acc) in cases.ISTFT_CASES.items():
        c = cases.STFT_CASES[src]
        o = ref.RefSTFT(c["radix2_exp"], c["window_type"], c["slide_length"])
        re, im = out[f"{src}/re"], out[f"{src}/im"]
        init = cases.noise(80, o.data_length(re.shape[0])) if acc else None
        out[f"{name}/y"] = o.istft(re, im, method, init)
    np.savez_compressed(os.path.join(HERE, "stft.npz"), **out)
    print("stft.npz", os.path.getsize(os.path.join(HERE, "stft.npz")) // 1024, "KiB")


def run_spec_case(c):
    kw = cases.spec_ctor(c)
    o = ref.RefSpectrogram(kw.pop("num"), **kw)
    assert o.status == 0, o.status
    if "norm" in c:
        o.set_norm(c["norm"])
    if "chroma_norm" in c:
        o.set_chroma_norm(c["chroma_norm"])
    x = cases.make_input(c["x"], c["samplate"])
    out = {"num": np.array([o.num], np.int32)}
    if c.get("phase"):
        out["spec"], out["phase"] = o.spectrogram(x, phase=True)
    else:
        out["spec"] = o.spectrogram(x)
    if c["scale_type"] not in (8, 9):
        out["fre"], out["bin"] = o.fre_band(), o.bin_band()
    if "cc" in c:
        kind, ccn = c["cc"][0], c["cc"][1]
        out["cc"] = o.cc(kind, np.abs(out["spec"]), ccn, c["cc"][2] if len(c["cc"]) > 2 else None)
    if c.get("deconv"):
        out["timbre"], out["pitch"] = o.deconv(out["spec"])
    return out


def make_spectrogram():
    """spectrogram.npz: the spectrogram object (every supported scale, switches, streaming)"""
    out = {}
    for name, c in cases.SPEC_CASES.items():
        for k, v in run_spec_case(c).items():
            out[f"{name}/{k}"] = v
    c = cases.SPEC_STREAM
    kw = cases.spec_ctor(c)
    o = ref.RefSpectrogram(kw.pop("num"), **kw)
    x = cases.noise(c["seed"], sum(c["chunks"]))
    off, rows, tl = 0, [], []
    for n in c["chunks"]:
        tl.append(o.time_length(n))
        rows.append(o.spectrogram(x[off:off + n]))
        off += n
    out["stream/spec"], out["stream/tl"] = np.concatenate(rows), np.array(tl, np.int32)
    np.savez_compressed(os.path.join(HERE, "spectrogram.npz"), **out)
    print("spectrogram.npz", os.path.getsize(os.path.join(HERE, "spectrogram.npz")) // 1024, "KiB")


def make_pwt():
    """pwt.npz: pseudo wavelet transform (time axis strided like the CWT fixtures)"""
    out = {}
    for name, c in cases.PWT_CASES.items():
        kw = {k: v for k, v in c.items() if k != "x"}
        o = ref.RefPWT(kw.pop("num"), kw.pop("radix2_exp"), **kw)
        assert o.status == 0, (name, o.status)
        x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
        re, im = o.pwt(x)
        st = cases.cwt_stride(c)
        out[f"{name}/re"], out[f"{name}/im"] = re[:, ::st], im[:, ::st]
        out[f"{name}/fre"], out[f"{name}/bin"] = o.fre_band(), o.bin_band()
        if name in ("octave84_pad", "mel40_area_nopad"):
            dre, dim = o.pwt(x, det=True)
            out[f"{name}/det_re"], out[f"{name}/det_im"] = dre[:, ::st], dim[:, ::st]
    np.savez_compressed(os.path.join(HERE, "pwt.npz"), **out)
    print("pwt.npz", os.path.getsize(os.path.join(HERE, "pwt.npz")) // 1024, "KiB")


def make_wsst():
    """wsst.npz: squeezed coefficients of the reference; the squeezing moves coefficients along the
    band axis only, so fixtures keep whole columns at the time stride of the CWT fixtures"""
    out = {}
    for name, c in cases.WSST_CASES.items():
        kw = {k: v for k, v in c.items() if k != "x"}
        o = ref.RefWSST(kw.pop("num"), kw.pop("radix2_exp"), **kw)
        assert o.status == 0, (name, o.status)
        x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
        s, w = o.wsst(x)
        out[f"{name}/s"] = s[:, ::cases.cwt_stride(c)].astype(np.complex64)
        out[f"{name}/fre"] = o.fre_band()
    np.savez_compressed(os.path.join(HERE, "wsst.npz"), **out)
    print("wsst.npz", os.path.getsize(os.path.join(HERE, "wsst.npz")) // 1024, "KiB")


def make_reassign():
    """reassign.npz: reassignment object and bftObj_new(isReassign = 1)"""
    out = {}
    for name, c in cases.REASSIGN_CASES.items():
        o = ref.RefReassign(c["radix2_exp"], **cases.reassign_ctor(c))
        assert o.status == 0
        if "result_type" in c:
            o.set_result_type(c["result_type"])
        if "order" in c:
            o.set_order(c["order"])
        a = o.reassign(cases.make_input(c["x"], c["samplate"]))
        out[f"{name}/re"], out[f"{name}/im"] = a[0], a[1]
    for name, c in cases.BFT_REASSIGN_CASES.items():
        kw = cases.ctor_kwargs(c)
        o = ref.RefBFT(kw.pop("num"), kw.pop("radix2_exp"), is_reassign=1, **kw)
        assert o.status == 0
        o.set_result_type(c["result_type"])
        re, im = o.bft(cases.make_input(c["x"], c["samplate"]))   # first call: the scratch is still zero
        out[f"bft_{name}/re"], out[f"bft_{name}/im"] = re, im
    np.savez_compressed(os.path.join(HERE, "reassign.npz"), **out)
    print("reassign.npz", os.path.getsize(os.path.join(HERE, "reassign.npz")) // 1024, "KiB")


def make_synsq():
    """synsq.npz: synchrosqueezing of synthetic matrices (whole columns at a time stride)"""
    out = {}
    for name, c in cases.SYNSQ_CASES.items():
        fre, W = cases.synsq_input(c)
        o = ref.RefSynsq(c["num"], c["radix2_exp"], samplate=c["samplate"])
        assert o.status == 0
        out[f"{name}/s"] = o.synsq(fre, c["scale_type"], W)[:, ::cases.cwt_stride(c)].astype(np.complex64)
    np.savez_compressed(os.path.join(HERE, "synsq.npz"), **out)
    print("synsq.npz", os.path.getsize(os.path.join(HERE, "synsq.npz")) // 1024, "KiB")


def main():
    assert ref.available(), "build the reference oracle first: make -C oracle"
    bft_out = {}
    for name, c in cases.BFT_CASES.items():
        for k, v in run_bft_case(c).items():
            bft_out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "bft.npz"), **bft_out)

    xx_out = {}
    rng = np.random.default_rng(99)
    for name, c in cases.XXCC_CASES.items():
        m = np.abs(bft_out[c["src"] + "/re"])
        o = ref.RefXXCC(c["num"])
        if "standard" in c:
            dlen, etype = c["standard"]
            energy = np.abs(rng.standard_normal(m.shape[0])).astype(np.float32) + 1e-3
            coe, d1, d2 = o.standard(m, energy, c["cc_num"], dlen, etype, c["rectify"])
            xx_out[f"{name}/energy"] = energy
            xx_out[f"{name}/coe"], xx_out[f"{name}/d1"], xx_out[f"{name}/d2"] = coe, d1, d2
        else:
            xx_out[f"{name}/cc"] = o.xxcc(m, c["cc_num"], c["rectify"])
    np.savez_compressed(os.path.join(HERE, "xxcc.npz"), **xx_out)
    cp_out = {}
    for name, c in cases.CEPS_CASES.items():
        o = ref.RefCepstrogram(c["radix2_exp"], c["window_type"], c["slide_length"])
        outs = o.cepstrogram(cases.make_input(c["x"], 16000), c["cep_num"])
        for k, v in zip(("cep", "env", "det"), outs):
            cp_out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "cepstrogram.npz"), **cp_out)
    cq_out = {}
    for name, c in cases.CQT_CASES.items():
        kw = {k: v for k, v in c.items() if k != "x"}
        o = ref.RefCQT(kw.pop("num"), **kw)
        assert o.status == 0
        x = cases.make_input(c["x"], c["samplate"])
        re, im = o.cqt(x)
        cq_out[f"{name}/re"], cq_out[f"{name}/im"] = re, im
        cq_out[f"{name}/fre"] = o.fre_band()
        cq_out[f"{name}/fft"] = np.array([o.fft_length()])
        if c["bin_per_octave"] == 12:
            for cname, (cn, dt, nt) in cases.CQT_CHROMA.items():
                cq_out[f"{name}/chroma_{cname}"] = o.chroma(re, im, cn, dt, nt)
        mag = np.abs(re + 1j * im).astype(np.float32)
        cq_out[f"{name}/cqcc"] = o.cqcc(mag, 13, 0)
        cq_out[f"{name}/cqhc"] = o.cqhc(mag, 20)
        cq_out[f"{name}/timbre"], cq_out[f"{name}/pitch"] = o.deconv(mag)
    np.savez_compressed(os.path.join(HERE, "cqt.npz"), **cq_out)
    cw_out = {}
    for name, c in cases.CWT_CASES.items():
        kw = {k: v for k, v in c.items() if k != "x"}
        o = ref.RefCWT(kw.pop("num"), kw.pop("radix2_exp"), **kw)
        assert o.status == 0, (name, o.status)
        x = cases.make_input((c["x"][0], c["x"][1], 1 << c["radix2_exp"]), c["samplate"])
        re, im = o.cwt(x)
        st = cases.cwt_stride(c)  # fixtures keep <= 512 time samples per scale
        cw_out[f"{name}/re"], cw_out[f"{name}/im"] = re[:, ::st], im[:, ::st]
        cw_out[f"{name}/fre"], cw_out[f"{name}/bin"] = o.fre_band(), o.bin_band()
        if name in ("morlet_84_pad", "morse_nopad"):
            dre, dim = o.cwt(x, det=True)
            cw_out[f"{name}/det_re"], cw_out[f"{name}/det_im"] = dre[:, ::st], dim[:, ::st]
    np.savez_compressed(os.path.join(HERE, "cwt.npz"), **cw_out)
    for f in ("bft.npz", "xxcc.npz", "cepstrogram.npz", "cqt.npz", "cwt.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    # `make_golden.py stft ...` regenerates only the named fixture files
    if len(sys.argv) > 1:
        for which in sys.argv[1:]:
            globals()["make_" + which]()
    else:
        main()
        make_stft()
        make_spectrogram()
        make_pwt()
        make_wsst()
        make_reassign()
        make_synsq()
