"""Drop-in proof, run in a FRESH interpreter (no torch, no audioflux_amd in the process):
the reference's own, unmodified `python/audioflux` ctypes wrapper is unpacked into a scratch
directory, the stock library (the compiled reference) is installed as its `lib/libaudioflux.so`
and the product as `lib/libaudioflux_mi355x.so`, and every wrapper flow on the hot path is run
twice -- stock, then after `audioflux.fftlib.set_fft_lib(lib_ext='mi355x')` (the selection
mechanism of python/audioflux/fftlib.py:88-129; `Base.__init__` snapshots the handle,
base.py:4-8).  Results of both runs go to one .npz which tests/dropin/test_dropin.py compares.

TEST INFRASTRUCTURE: the wrapper archive (oracle/_ref/audioflux_pywrapper.zip) and the stock
library (oracle/_ref/libaudioflux_ref.so) are build outputs of oracle/Makefile.

usage: python flows.py WORKDIR OUT.npz [gpu|cpu]
  gpu: run every flow on both libraries
  cpu: no device -- import the wrapper, select the product library, resolve every symbol the
       wrapper looks up for the path, construct every object (status -2, NULL handle) and call
       the cheap entry points: nothing may crash.  Writes OUT.npz with the symbol table.
"""
import json
import os
import re
import sys
import types
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WRAPPER_ZIP = os.path.join(ROOT, "oracle", "_ref", "audioflux_pywrapper.zip")
STOCK = os.path.join(ROOT, "oracle", "_ref", "libaudioflux_ref.so")
PRODUCT = os.environ.get("AFX_LIB") or os.path.join(ROOT, "audioflux_amd", "lib", "libaudioflux_mi355x.so")

# wrapper modules that make up the path; every `self._lib['name']` in them must resolve
PATH_MODULES = ["bft.py", "feature/xxcc.py", "cwt.py", "cqt.py", "cepstrogram.py", "stft.py", "pwt.py",
                "wsst.py", "synsq.py", "reassign.py"]
# spectrogram.py also binds the spectral-descriptor family (out of scope); the core it
# needs for MelSpectrogram / BarkSpectrogram / ErbSpectrogram / Spectrogram.spectrogram / cepstra:
SPECTROGRAM_CORE = [
    "spectrogramObj_new", "spectrogramObj_newLinear", "spectrogramObj_newMel", "spectrogramObj_newBark",
    "spectrogramObj_newErb", "spectrogramObj_newChroma", "spectrogramObj_newDeep", "spectrogramObj_newDeepChroma",
    "spectrogramObj_calTimeLength", "spectrogramObj_getFreBandArr", "spectrogramObj_getBinBandArr",
    "spectrogramObj_getBandNum", "spectrogramObj_getBinBandLength", "spectrogramObj_setDataNormValue",
    "spectrogramObj_setChromaDataNormalType", "spectrogramObj_setDeepOrder", "spectrogramObj_spectrogram",
    "spectrogramObj_deconv", "spectrogramObj_mfcc", "spectrogramObj_bfcc", "spectrogramObj_gtcc",
    "spectrogramObj_xxcc", "spectrogramObj_free",
]


def stage(workdir):
    """unpack the wrapper, install both libraries under its lib/ (fftlib.py:118-126 looks there)"""
    if not os.path.exists(WRAPPER_ZIP):
        # this container: build it from /root/reference where it lies (oracle/stage_wrapper.py)
        sys.path.insert(0, ROOT)
        from oracle import stage_wrapper
        stage_wrapper.stage()
    with zipfile.ZipFile(WRAPPER_ZIP) as z:
        z.extractall(workdir)
    lib = os.path.join(workdir, "audioflux", "lib")
    os.makedirs(lib, exist_ok=True)
    for src, name in ((STOCK, "libaudioflux.so"), (PRODUCT, "libaudioflux_mi355x.so")):
        dst = os.path.join(lib, name)
        if os.path.lexists(dst):
            os.remove(dst)
        os.symlink(src, dst)
    return lib


def import_wrapper(workdir):
    # audio.py:6 imports soundfile (file I/O, out of scope, not installed here)
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    sys.path.insert(0, workdir)
    import audioflux as af
    assert os.path.realpath(af.__file__).startswith(os.path.realpath(workdir)), af.__file__
    return af


def wrapper_symbols(workdir):
    """every symbol name the path's wrapper modules look up by `_lib['...']`"""
    names = set()
    for m in PATH_MODULES:
        with open(os.path.join(workdir, "audioflux", m)) as f:
            names.update(re.findall(r"_lib\['([A-Za-z0-9_]+)'\]", f.read()))
    names.update(SPECTROGRAM_CORE)
    return sorted(names)


# ------------------------------------------------------------------------------------------------
# inputs
def signals():
    rng = np.random.default_rng(20)
    n = 40000
    t = np.arange(n) / 16000.0
    noise = (0.1 * rng.standard_normal(n)).astype(np.float32)
    tones = (0.5 * np.sin(2 * np.pi * 220 * t) + 0.3 * np.sin(2 * np.pi * 880 * t)
             + 0.2 * np.sin(2 * np.pi * 3520 * t) + 1e-3 * rng.standard_normal(n)).astype(np.float32)
    return noise, tones


# ------------------------------------------------------------------------------------------------
# flows: each takes the wrapper module and returns {name: array}
def flow_bft_mel_real(af, x, y):
    o = af.BFT(num=128, radix2_exp=11, samplate=16000, low_fre=0., high_fre=8000., slide_length=512,
               scale_type=af.type.SpectralFilterBankScaleType.MEL, data_type=af.type.SpectralDataType.POWER)
    return {"spec": o.bft(x, result_type=1), "fre": o.get_fre_band_arr(), "bin": o.get_bin_band_arr(),
            "T": np.array(o.cal_time_length(len(x)))}


def flow_bft_default_complex(af, x, y):
    # the wrapper's defaults: radix2_exp 12, 32 kHz, linear scale would need num; use mel, MAG, complex
    o = af.BFT(num=128, radix2_exp=12, samplate=32000, scale_type=af.type.SpectralFilterBankScaleType.MEL)
    return {"spec": o.bft(y)}


def flow_bft_temporal_channels(af, x, y):
    o = af.BFT(num=64, radix2_exp=10, samplate=16000, slide_length=256,
               scale_type=af.type.SpectralFilterBankScaleType.BARK, is_temporal=True)
    spec = o.bft(np.stack([x, y]), result_type=1)
    e, r, z = o.get_temporal_data()
    return {"spec": spec, "energy": e, "rms": r, "zcr": z}


def flow_bft_linear(af, x, y):
    o = af.BFT(num=257, radix2_exp=9, samplate=16000, data_type=af.type.SpectralDataType.POWER)
    return {"spec": o.bft(x, result_type=1), "fre": o.get_fre_band_arr()}


def flow_xxcc(af, x, y):
    o = af.BFT(num=128, radix2_exp=11, samplate=16000, slide_length=512,
               scale_type=af.type.SpectralFilterBankScaleType.MEL, data_type=af.type.SpectralDataType.POWER)
    spec = o.bft(x, result_type=1)
    xx = af.XXCC(num=128)
    xx.set_time_length(time_length=spec.shape[-1])
    cc = xx.xxcc(spec, cc_num=13)
    cc3 = xx.xxcc(spec, cc_num=20, rectify_type=af.type.CepstralRectifyType.CUBIC_ROOT)
    energy = np.sum(spec, axis=0).astype(np.float32)
    std = xx.xxcc_standard(spec, energy, cc_num=13)
    out = {"cc": cc, "cc_cubic": cc3}
    for i, a in enumerate(std if isinstance(std, (tuple, list)) else [std]):
        out[f"std{i}"] = np.asarray(a)
    return out


def flow_core_mfcc(af, x, y):
    cc, fre = af.mfcc(x, cc_num=13, mel_num=128, radix2_exp=11, samplate=16000, slide_length=512)
    return {"cc": cc, "fre": fre}


def flow_core_mel_spectrogram(af, x, y):
    spec, fre = af.mel_spectrogram(y, num=128, radix2_exp=11, samplate=16000, slide_length=512)
    return {"spec": spec, "fre": fre}


def flow_mel_spectrogram_obj(af, x, y):
    o = af.MelSpectrogram(num=128, samplate=16000, radix2_exp=11, slide_length=512)
    spec = o.spectrogram(x)
    return {"spec": spec, "mfcc": o.mfcc(spec, cc_num=13), "fre": o.get_fre_band_arr(),
            "T": np.array(o.cal_time_length(len(x)))}


def flow_bark_erb_spectrogram(af, x, y):
    b = af.BarkSpectrogram(num=64, samplate=16000, radix2_exp=10)
    e = af.ErbSpectrogram(num=64, samplate=16000, radix2_exp=10)
    return {"bark": b.spectrogram(x), "erb": e.spectrogram(x)}


def flow_cwt(af, x, y):
    o = af.CWT(num=84, radix2_exp=12, samplate=16000, wavelet_type=af.type.WaveletContinueType.MORLET)
    m = af.CWT(num=48, radix2_exp=12, samplate=16000)  # wrapper default: morse
    return {"morlet": o.cwt(y[:4096]), "fre": o.get_fre_band_arr(), "bin": o.get_bin_band_arr(),
            "morse": m.cwt(x[:4096])}


def flow_cqt(af, x, y):
    o = af.CQT(num=84, samplate=32000)
    q = o.cqt(y)
    out = {"cqt": q, "fre": o.get_fre_band_arr(), "fft_length": np.array(o.get_fft_length()),
           "T": np.array(o.cal_time_length(len(y)))}
    out["chroma"] = o.chroma(q)
    # the log-domain consumers chain from the CQT of the NOISE clip: on the tonal clip most bins sit at
    # the 1e-3 noise floor and log10 turns the 1e-6-of-peak float32 differences there into 5e-5 of the
    # cepstral peak (conditioning of the chain, not of cqcc; cqcc on equal inputs is in tests/test_cqt_gpu.py)
    qa = np.abs(o.cqt(x)).astype(np.float32)
    out["cqcc"] = o.cqcc(qa)
    out["cqhc"] = o.cqhc(qa)
    dec = o.deconv(qa)
    for i, a in enumerate(dec):
        out[f"deconv{i}"] = np.asarray(a)
    return out


def flow_core_cqt_chroma(af, x, y):
    q, fre = af.cqt(x, samplate=32000)
    return {"cqt_abs": q, "fre": fre, "chroma": af.chroma_cqt(x, samplate=32000)}


def flow_cepstrogram(af, x, y):
    o = af.Cepstrogram(radix2_exp=11, samplate=16000, slide_length=512)
    c, e, d = o.cepstrogram(x, cep_num=4)
    return {"cep": c, "env": e, "det": d, "T": np.array(o.cal_time_length(len(x)))}


def flow_stft(af, x, y):
    o = af.STFT(radix2_exp=10, window_type=af.type.WindowType.HANN, slide_length=256)
    s = o.stft(y)
    out = {"stft": s, "window": o.get_window_data_arr(), "T": np.array(o.cal_time_length(len(y)))}
    out["istft"] = o.istft(s)
    return out


def flow_pwt(af, x, y):
    o = af.PWT(num=84, radix2_exp=12, samplate=16000)
    return {"pwt": o.pwt(y[:4096]), "fre": o.get_fre_band_arr()}


def flow_wsst(af, x, y):
    o = af.WSST(num=84, radix2_exp=12, samplate=16000)
    a, b = o.wsst(y[:4096])
    return {"wsst": a, "cwt": b}


def flow_synsq(af, x, y):
    c = af.CWT(num=84, radix2_exp=12, samplate=16000, wavelet_type=af.type.WaveletContinueType.MORLET)
    w = c.cwt(y[:4096])
    o = af.Synsq(num=c.num, radix2_exp=c.radix2_exp, samplate=c.samplate)
    return {"synsq": o.synsq(w, filter_bank_type=c.scale_type, fre_arr=c.get_fre_band_arr())}


def flow_reassign(af, x, y):
    o = af.Reassign(radix2_exp=10, samplate=16000)
    a, b = o.reassign(y[:16000])
    return {"reassign": a, "stft": b}


FLOWS = [flow_bft_mel_real, flow_bft_default_complex, flow_bft_temporal_channels, flow_bft_linear,
         flow_xxcc, flow_core_mfcc, flow_core_mel_spectrogram, flow_mel_spectrogram_obj,
         flow_bark_erb_spectrogram, flow_cwt, flow_cqt, flow_core_cqt_chroma, flow_cepstrogram,
         flow_stft, flow_pwt, flow_wsst, flow_synsq, flow_reassign]


def run_gpu(workdir, out):
    stage(workdir)
    af = import_wrapper(workdir)
    x, y = signals()
    res = {}
    meta = {"flows": [f.__name__[5:] for f in FLOWS], "errors": {}}
    for tag, ext in (("stock", None), ("mi355x", "mi355x")):
        af.fftlib.set_fft_lib(lib_ext=ext)
        meta[tag + "_lib"] = os.path.realpath(af.fftlib.get_fft_lib_fp())
        for f in FLOWS:
            name = f.__name__[5:]
            try:
                for k, v in f(af, x, y).items():
                    res[f"{tag}/{name}/{k}"] = np.asarray(v)
            except Exception as e:  # recorded, the test fails on it
                meta["errors"][f"{tag}/{name}"] = f"{type(e).__name__}: {e}"
    # the handle every object snapshots really is the product library
    probe = af.BFT(num=8, radix2_exp=8)
    meta["object_lib"] = os.path.realpath(probe._lib._name)
    res["meta"] = np.array(json.dumps(meta))
    np.savez(out, **res)


def run_cpu(workdir, out):
    """no device needed: symbols resolve, the -2 (no gfx950 device) path never crashes"""
    stage(workdir)
    af = import_wrapper(workdir)
    af.fftlib.set_fft_lib(lib_ext="mi355x")
    lib = af.fftlib.get_fft_lib()
    meta = {"lib": os.path.realpath(af.fftlib.get_fft_lib_fp()), "missing": [], "calls": []}
    names = wrapper_symbols(workdir)
    for n in names:
        try:
            lib[n]
        except AttributeError:
            meta["missing"].append(n)
    meta["symbols"] = len(names)
    T = af.type
    x, y = signals()

    def attempt(what, fn):
        try:
            r = fn()
            meta["calls"].append([what, "ok", repr(type(r).__name__)])
        except Exception as e:  # a Python exception is fine (e.g. from_address(None)); a crash is not
            meta["calls"].append([what, "raised", f"{type(e).__name__}: {e}"[:120]])

    # constructors first (no device here => status -2 and a NULL handle), then the cheap entry points
    objs = {}
    attempt("BFT()", lambda: objs.setdefault("bft", af.BFT(num=128, radix2_exp=11, samplate=16000,
                                                           scale_type=T.SpectralFilterBankScaleType.MEL)))
    attempt("XXCC()", lambda: objs.setdefault("xxcc", af.XXCC(num=128)))
    attempt("CWT()", lambda: objs.setdefault("cwt", af.CWT(num=84, radix2_exp=12)))
    attempt("CQT()", lambda: objs.setdefault("cqt", af.CQT(num=84)))
    attempt("Cepstrogram()", lambda: objs.setdefault("cep", af.Cepstrogram(radix2_exp=11)))
    attempt("STFT()", lambda: objs.setdefault("stft", af.STFT(radix2_exp=10)))
    attempt("MelSpectrogram()", lambda: objs.setdefault("mel", af.MelSpectrogram(num=128, radix2_exp=11)))
    attempt("PWT()", lambda: objs.setdefault("pwt", af.PWT(num=84, radix2_exp=12)))
    attempt("WSST()", lambda: objs.setdefault("wsst", af.WSST(num=84, radix2_exp=12)))
    attempt("Synsq()", lambda: objs.setdefault("synsq", af.Synsq(num=84, radix2_exp=12)))
    attempt("Reassign()", lambda: objs.setdefault("reassign", af.Reassign(radix2_exp=10)))
    for key in ("bft", "cqt", "cep", "stft", "mel", "reassign"):
        if key in objs:
            attempt(f"{key}.cal_time_length", lambda k=key: objs[k].cal_time_length(40000))
    for key in ("bft", "cwt", "cqt", "mel", "pwt", "wsst"):
        if key in objs:
            attempt(f"{key}.get_fre_band_arr", lambda k=key: objs[k].get_fre_band_arr())
    if "bft" in objs:
        attempt("bft.bft", lambda: objs["bft"].bft(x, result_type=1))
        attempt("bft.set_data_norm_value", lambda: objs["bft"].set_data_norm_value(2.0))
    if "xxcc" in objs:
        attempt("xxcc.xxcc", lambda: objs["xxcc"].xxcc(np.ones((128, 10), np.float32)))
    if "cwt" in objs:
        attempt("cwt.cwt", lambda: objs["cwt"].cwt(x[:4096]))
    if "cqt" in objs:
        attempt("cqt.cqt", lambda: objs["cqt"].cqt(x))
    if "cep" in objs:
        attempt("cep.cepstrogram", lambda: objs["cep"].cepstrogram(x))
    if "stft" in objs:
        attempt("stft.stft", lambda: objs["stft"].stft(x))
    if "mel" in objs:
        attempt("mel.spectrogram", lambda: objs["mel"].spectrogram(x))
    objs.clear()  # __del__ -> *_free on NULL handles
    import gc
    gc.collect()
    meta["survived"] = True
    np.savez(out, meta=np.array(json.dumps(meta)))


def run_cqt_functional(workdir, out):
    """CPU: AFX_LIB = the host objects linked with tests/hoststub/cqt_functional.c (CQT launchers that compute);
    the CQT / chroma flows through the wrapper, stock library vs that build (tests/test_hoststub.py)"""
    stage(workdir)
    af = import_wrapper(workdir)
    x, y = signals()
    res = {}
    for tag, ext in (("stock", None), ("mi355x", "mi355x")):
        af.fftlib.set_fft_lib(lib_ext=ext)
        o = af.CQT(num=84, samplate=32000)
        q = o.cqt(y)
        res[f"{tag}/cqt"], res[f"{tag}/chroma"] = q, o.chroma(q)
        res[f"{tag}/fre"], res[f"{tag}/T"] = o.get_fre_band_arr(), np.array(o.cal_time_length(len(y)))
        q2, _ = af.cqt(x, samplate=32000)
        res[f"{tag}/core_cqt_abs"], res[f"{tag}/core_chroma"] = q2, af.chroma_cqt(x, samplate=32000)
    np.savez(out, **res)


if __name__ == "__main__":
    workdir, out = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "gpu"
    {"gpu": run_gpu, "cpu": run_cpu, "cqt_functional": run_cqt_functional}[mode](workdir, out)
    print("flows done:", mode)
