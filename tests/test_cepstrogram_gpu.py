"""GPU parity of the cepstrogram object against the reference's golden vectors."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.conftest import assert_parity

pytestmark = pytest.mark.gpu
# cepstrum / envelope: 1e-5.  details: the reference's own float32 result is 0.7-1.1e-5
# (peak-relative) away from a float64 evaluation of the same formulas on these inputs
# (tests/test_oracle.py), so two float32 implementations can differ by ~sqrt(2) of that.
TOL = {"cep": 1e-5, "env": 1e-5, "det": 3e-5}


@pytest.mark.parametrize("name", list(cases.CEPS_CASES))
def test_cepstrogram_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "cepstrogram.npz"))
    c = cases.CEPS_CASES[name]
    o = af.Cepstrogram(radix2_exp=c["radix2_exp"], window_type=af.WindowType(c["window_type"]),
                       slide_length=c["slide_length"])
    x = cases.make_input(c["x"], 16000)
    outs = o.cepstrogram(x, cep_num=c["cep_num"])
    for k, got in zip(("cep", "env", "det"), outs):
        assert_parity(got.T, gold[f"{name}/{k}"], TOL[k], f"{name}/{k}")


def test_envelope_plus_details_reconstruct_log_spectrum():
    """size-independent property: the two lifters partition the cepstrum (index N-cepNum is in
    both), so envelope + details = ln|S|^2 + the doubled term's cosine"""
    n, hop, q = 1024, 256, 12
    x = cases.noise(77, 8000)
    o = af.Cepstrogram(radix2_exp=10, window_type=af.WindowType.HANN, slide_length=hop)
    cep, env, det = [a.T for a in o.cepstrogram(x, cep_num=q)]
    from oracle import restate
    fr = restate.frames_of(x, n, hop) * restate.fft_window(1, n)[None, :]
    logS = np.log(np.maximum(np.abs(np.fft.rfft(fr, axis=1)) ** 2, 1e-16))
    k = np.arange(n // 2 + 1)
    # c is even, so c[N-q] = c[q] = cep[:, q]
    extra = cep[:, q:q + 1] * np.cos(2 * np.pi * k * (n - q) / n)[None, :]
    assert_parity(env + det, logS + extra, 1e-4, "lifter partition")


def test_wave_kernel_2048_matches_golden_through_device_call(golden_dir):
    """n_fft 2048 through the batched device entry point runs k_cepstrogram_w2048 (four wave-level
    real transforms per frame); same golden vectors as the one-clip entry point above"""
    import torch
    gold = np.load(os.path.join(golden_dir, "cepstrogram.npz"))
    c = cases.CEPS_CASES["hann_2048"]
    o = af.Cepstrogram(radix2_exp=11, window_type=af.WindowType(c["window_type"]), slide_length=c["slide_length"])
    x = cases.make_input(c["x"], 16000)
    outs = o.cepstrogram_device(torch.from_numpy(x[None]).cuda(), cep_num=c["cep_num"])
    torch.cuda.synchronize()
    for k, got in zip(("cep", "env", "det"), outs):
        assert_parity(got[0].cpu().numpy(), gold[f"hann_2048/{k}"], TOL[k], f"wave hann_2048/{k}")


@pytest.mark.parametrize("hop,length,stride_pad,cep_num,window", [
    (512, 30000, 0, 20, "hann"),     # aligned frames: float2 loads
    (512, 30000, 1, 0, "rect"),      # odd row pitch: dword loads; cep_num 0: empty mirror
    (300, 20000, 0, 1, "hamm"),      # hop not a multiple of 64
    (333, 9000, 0, 1022, "hann"),    # odd hop; the largest cep_num the wave kernel takes
    (2048, 16384, 0, 4, "hann"),     # no overlap
])
def test_wave_kernel_2048_matches_compiled_reference(hop, length, stride_pad, cep_num, window):
    import torch
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    wt = cases.WIN[window]
    rng = np.random.default_rng(1000 + hop + cep_num)
    clips = 3
    x = (0.1 * rng.standard_normal((clips, length + stride_pad))).astype(np.float32)
    x[1] += np.sin(np.arange(length + stride_pad) * 0.05).astype(np.float32)  # a tonal clip: peaky cepstrum
    xd = torch.from_numpy(x).cuda()[:, :length]  # row pitch length + stride_pad
    o = af.Cepstrogram(radix2_exp=11, window_type=af.WindowType(wt), slide_length=hop)
    outs = o.cepstrogram_device(xd, cep_num=cep_num)
    torch.cuda.synchronize()
    r = ref.RefCepstrogram(11, wt, hop)
    from oracle import restate
    for i in range(clips):
        want = r.cepstrogram(x[i, :length], cep_num)
        # conditioning: ln|S|^2 amplifies the float32 error of the spectrum where |S| is far below
        # the frame's peak (the tonal clip at Nyquist: the reference itself is 5e-5 of the peak away
        # from a float64 evaluation there; independent float32 evaluations scatter by that much
        # around the exact value), so the bar is the larger of TOL and 3x the reference's own
        # distance from float64 -- for the comparison with the reference AND with float64
        f64 = restate.cepstrogram(x[i, :length].astype(np.float64), 2048, hop, cep_num, window_type=wt)
        for k, name in enumerate(("cep", "env", "det")):
            ref_err = np.abs(want[k] - f64[k]).max() / np.abs(f64[k]).max()
            tol = max(TOL[name], 3.0 * ref_err)
            got = outs[k][i].cpu().numpy()
            assert_parity(got, want[k], tol, f"clip {i} {name} hop {hop} q {cep_num}")
            assert_parity(got, f64[k], tol, f"clip {i} {name} hop {hop} q {cep_num} vs float64")
    # and against the size-generic kernel behind the one-clip entry point
    loop = o.cepstrogram(x[0, :length], cep_num=cep_num)  # clip 0 is plain noise: well conditioned
    for k, name in enumerate(("cep", "env", "det")):
        assert_parity(outs[k][0].cpu().numpy().T, loop[k], TOL[name], f"vs generic {name}")


def test_wave_kernel_2048_cep_num_beyond_its_range_takes_the_generic_kernel():
    import torch
    x = cases.noise(78, 6000)
    o = af.Cepstrogram(radix2_exp=11, window_type=af.WindowType.HANN, slide_length=512)
    outs = o.cepstrogram_device(torch.from_numpy(x[None]).cuda(), cep_num=1023)
    torch.cuda.synchronize()
    loop = o.cepstrogram(x, cep_num=1023)
    for k in range(3):
        assert np.array_equal(outs[k][0].cpu().numpy().T, loop[k])
