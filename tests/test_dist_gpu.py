"""RCCL on the GPU box.  Only one GPU is visible there, so these run the REAL backend at world
size 1 through the same code paths bench.py --gpus N uses: communicator creation, side-stream
ordering, the double-buffered hand-off (FeatureGather with always_collective) and the library's own
afx_gather export.  World size > 1 is covered on the CPU by tests/test_dist_cpu.py (gloo)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_feature_gather_runs_the_rccl_path_at_world_size_1(nccl_world1):
    import torch
    import audioflux_amd as af
    from audioflux_amd import dist as afd
    x = 0.1 * torch.randn((6, 40000), device="cuda")
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    xx = af.XXCC(128)
    ccs = [torch.empty((6, bft.cal_time_length(40000), 13), device="cuda") for _ in range(2)]
    g = afd.FeatureGather(dst=0, counts=[6], always_collective=True)
    comm = torch.cuda.Stream()
    outs = []
    for i in range(4):  # bench.py's step loop: compute, then the gather of this step on the side stream
        af.mel_mfcc_device(bft, xx, x * (i + 1), 13, out_cc=ccs[i & 1])
        g.wait()
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            g.start(ccs[i & 1])
        if i >= 2:
            outs.append(i)
    got = g.wait()
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()
    assert g.work is None and got.shape == ccs[0].shape
    assert torch.equal(got, ccs[1])  # step 3 wrote buffer 1


def test_native_afx_gather_world_size_1(nccl_world1):
    import torch
    from audioflux_amd import dist as afd
    g = afd.NativeGather(dst=0)
    assert g.world == 1 and g.rank == 0
    slab = torch.randn((5, 7, 13), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g.start(slab, stream=side)
    side.synchronize()
    assert torch.equal(g.wait(), slab)
    slab2 = torch.randn((5, 7, 13), device="cuda")
    g.start(slab2)  # current stream, buffer re-used
    torch.cuda.synchronize()
    assert torch.equal(g.wait(), slab2)
    g.close()


def test_afx_comm_argument_checks():
    import ctypes
    import audioflux_amd as af
    lib = af.get_lib()
    assert lib.afx_comm_get_unique_id(None) == -6
    c = ctypes.c_void_p(None)
    ident = ctypes.create_string_buffer(128)
    assert lib.afx_comm_create(ctypes.byref(c), 2, 5, ident) == -6 and not c.value
    lib.afx_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int,
                               ctypes.c_void_p]
    assert lib.afx_gather(None, None, 0, None, 0, None) == -6
