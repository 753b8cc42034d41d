"""CPU-only, world_size 2 over gloo: clip sharding and the feature gather that
bench.py --gpus N uses over RCCL (same code path, different backend)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n_clips=6):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from audioflux_amd import dist as afd
    import torch.distributed as dist
    r, _, w = afd.init_from_env(backend="gloo")
    t, c = 5, 3
    start, stop = afd.shard_range(n_clips, r, w)
    # "features" of clip i are filled with i so the gathered layout is checkable
    slab = torch.stack([torch.full((t, c), float(i)) for i in range(start, stop)])
    g = afd.FeatureGather(dst=0)
    for _ in range(2):  # re-use across steps
        g.start(slab)
        out = g.wait()
    if r == 0:
        ok = out.shape == (n_clips, t, c) and all(bool((out[i] == i).all()) for i in range(n_clips))
        q.put(("gather", ok))
    else:
        q.put(("none", out is None))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from audioflux_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


@pytest.mark.timeout(120)
@pytest.mark.parametrize("n_clips", [6, 7])  # 7: ceil-sized blocks 4 + 3, the short shard is padded / trimmed
def test_gather_world2_gloo(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, n_clips)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == {"gather": True, "none": True}
