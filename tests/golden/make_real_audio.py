"""Generates tests/golden/real_audio.npz: excerpts of the reference's own sample clips
(/root/reference/python/audioflux/utils/sample_data, listed in python/audioflux/utils/sample.py:9-31) as int16
INPUTS -- the WAV files are not present on the GPU box -- and the outputs of the compiled reference
(oracle/_ref, built by oracle/Makefile) on them: mel-128 + MFCC-13 (n_fft 2048, hop 512), CQT-84 + chroma-12,
CWT morlet-84 of the first 2^16 samples, cepstrogram.  Large outputs are kept at a stride (the GPU tests compare
against the compiled reference itself when it is present, and against these rows otherwise).

    python tests/golden/make_real_audio.py      # needs /root/reference and oracle/_ref/libaudioflux_ref.so
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402
from tests import cases  # noqa: E402

SRC = "/root/reference/python/audioflux/utils/sample_data"
EXCERPT = {"voice": ("voice.wav", 40000), "guitar": ("guitar_chord1.wav", 4000), "metronome": ("chord_metronome1.wav", 20000)}
CQT_STRIDE, CWT_STRIDE = 4, 256


def reference_outputs(x):
    """what the GPU tests compare: every output of the compiled reference for one 32 kHz clip"""
    sr = cases.REAL_AUDIO_SR
    out = {}
    b = ref.RefBFT(128, 11, samplate=sr, low_fre=0.0, high_fre=sr / 2.0, window_type=1, slide_length=512, scale_type=2,
                   style_type=0, normal_type=0, data_type=0)
    b.set_result_type(1)
    mel, _ = b.bft(x)
    out["mel"], out["mfcc"] = mel, ref.RefXXCC(128).xxcc(mel, 13, 0)
    q = ref.RefCQT(num=84, samplate=sr, min_fre=32.703, bin_per_octave=12, normal_type=1)
    re, im = q.cqt(x)
    out["cqt"], out["chroma"] = (re + 1j * im).astype(np.complex64), q.chroma(re, im)
    w = ref.RefCWT(num=84, radix2_exp=16, samplate=sr, low_fre=32.703, bin_per_octave=12, wavelet_type=1, scale_type=5,
                   is_padding=1)
    wre, wim = w.cwt(x[:65536])
    out["cwt"] = (wre + 1j * wim).astype(np.complex64)
    c = ref.RefCepstrogram(11, 1, 512)
    out["cep"], out["env"], out["det"] = c.cepstrogram(x, 4)
    return out


def main():
    import scipy.io.wavfile as wavfile
    assert ref.available(), "build the reference oracle first: make -C oracle"
    z = {}
    for name in cases.REAL_AUDIO:
        fn, start = EXCERPT[name]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sr, raw = wavfile.read(os.path.join(SRC, fn))
        assert sr == cases.REAL_AUDIO_SR and raw.ndim == 1
        seg = raw[start:start + cases.REAL_AUDIO_LEN]
        assert len(seg) == cases.REAL_AUDIO_LEN
        if seg.dtype == np.int32:   # 32-bit files: keep the upper 16 bits (the fixture DEFINES the input)
            seg = (seg >> 16).astype(np.int16)
        z[f"{name}/x"] = seg.astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "real_audio.npz"), **z)   # cases.real_audio reads the inputs back
    for name in cases.REAL_AUDIO:
        o = reference_outputs(cases.real_audio(name, HERE))
        z[f"{name}/mel"], z[f"{name}/mfcc"], z[f"{name}/chroma"] = o["mel"], o["mfcc"], o["chroma"]
        z[f"{name}/cqt"] = o["cqt"][::CQT_STRIDE]
        z[f"{name}/cwt"] = o["cwt"][:, ::CWT_STRIDE]
        z[f"{name}/cep"] = o["cep"][::4]
    np.savez_compressed(os.path.join(HERE, "real_audio.npz"), **z)
    print("real_audio.npz", os.path.getsize(os.path.join(HERE, "real_audio.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
