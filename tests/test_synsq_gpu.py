"""GPU parity of synsqObj_synsq against the reference's golden vectors, under the boundary-aware
criterion pinned in tests/test_synsq_host.py; plus the accumulate semantics of the C entry."""
import os

import numpy as np
import pytest

import audioflux_amd as af
from tests import cases
from tests.test_synsq_host import explained

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(cases.SYNSQ_CASES))
def test_synsq_matches_golden(name, golden_dir):
    gold = np.load(os.path.join(golden_dir, "synsq.npz"))
    c = cases.SYNSQ_CASES[name]
    fre, W = cases.synsq_input(c)
    o = af.Synsq(c["num"], radix2_exp=c["radix2_exp"], samplate=c["samplate"])
    got = o.synsq(W, af.SpectralFilterBankScaleType(c["scale_type"]), fre)
    n_diff = explained(got, gold[f"{name}/s"], c, fre, W, name)
    assert n_diff < 0.01 * gold[f"{name}/s"].size
    # the float32 unwrap recurrence is replicated operation for operation: apart from atan2f's last
    # bit the two implementations take the same decisions, so the squeezed mass matches closely
    assert abs(np.abs(got[:, ::cases.cwt_stride(c)]).sum() - np.abs(gold[f"{name}/s"]).sum()) <= 2e-3 * np.abs(gold[f"{name}/s"]).sum()
    assert np.array_equal(o.synsq(W, af.SpectralFilterBankScaleType(c["scale_type"]), fre), got)  # deterministic
