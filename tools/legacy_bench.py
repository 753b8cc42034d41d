#!/usr/bin/env python3
"""The reference's own published benchmark (benchmark/run_audioflux.py:13-32: MelSpectrogram(num=128, samplate=32000,
radix2_exp=11, slide_length=512).spectrogram on random-normal input of 1000 frames, 1000 runs, data generation and one
warm-up call outside the clock) through the reference's UNMODIFIED ctypes wrapper with this library selected by
audioflux.fftlib.set_fft_lib(lib_ext='mi355x') -- i.e. the legacy host-pointer entry points, one clip per call, PCIe
both ways inside the clock.  A fresh interpreter: the wrapper, ctypes and the library only.  Prints one JSON line.
usage: tools/legacy_bench.py [runtimes] [time_step]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np

import flows  # staging of the wrapper archive + the library under its lib/ (tests/dropin/flows.py)


def main():
    runtimes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    time_step = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    if not (os.path.exists(flows.WRAPPER_ZIP) and os.path.exists(flows.PRODUCT)):
        print(json.dumps({"error": "wrapper archive or library missing"}))
        return
    work = tempfile.mkdtemp(prefix="afx_legacy_")
    try:
        run(work, runtimes, time_step)
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)


def run(work, runtimes, time_step):
    flows.stage(work)
    af = flows.import_wrapper(work)
    from audioflux import fftlib
    fftlib.set_fft_lib(lib_ext="mi355x")
    import audioflux.type as aft
    radix2_exp, slide_length = 11, 512
    mel_obj = af.MelSpectrogram(num=128, samplate=32000, low_fre=0, high_fre=16000, radix2_exp=radix2_exp,
                                window_type=aft.WindowType.HANN, slide_length=slide_length, data_type=aft.SpectralDataType.POWER,
                                style_type=aft.SpectralFilterBankStyleType.SLANEY, normal_type=aft.SpectralFilterBankNormalType.NONE)

    def gen_data():  # benchmark/utils.py:4-6
        return np.random.randn((1 << radix2_exp) + (time_step - 1) * slide_length).astype(np.float32)

    r = mel_obj.spectrogram(gen_data())
    assert r.shape == (128, time_step) and np.isfinite(r).all() and r.max() > 0, r.shape
    total = 0.0
    for _ in range(runtimes):
        x = gen_data()
        s = time.time()
        r = mel_obj.spectrogram(x)
        total += time.time() - s
    print(json.dumps({
        "protocol": "benchmark/run_audioflux.py: MelSpectrogram(128, 32 kHz, n_fft 2048, hop 512).spectrogram, "
                    f"{runtimes} runs x {time_step} frames, warm-up and data generation excluded",
        "through": "the reference's unmodified python wrapper, set_fft_lib(lib_ext='mi355x'): host pointers, one clip per call, PCIe inside the clock",
        "seconds": total, "frames_per_s": runtimes * time_step / total, "ms_per_call": 1e3 * total / runtimes,
        "reference_published": {"seconds": 1.43854, "frames_per_s": 695e3, "where": "BASELINE.md: Threadripper 3970X 32C, MKL + OpenMP (benchmark/README.md:68-84)"},
    }))


if __name__ == "__main__":
    main()
