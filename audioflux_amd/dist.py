"""Multi-GPU layer: one process per GPU, clips sharded contiguously, features
gathered to rank 0 over RCCL/xGMI (torch.distributed backend "nccl" on ROCm;
"gloo" in the CPU tests).  The transforms themselves need no collective: every
clip is independent; the gather is the only exchange."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run.
    Returns (rank, local_rank, world_size); single-process when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n_items, rank, world):
    """contiguous clip-major block of rank: [start, stop) with ceil-sized blocks, so
    each rank's output is one contiguous slab of the gathered tensor"""
    per = -(-n_items // world)
    start = min(rank * per, n_items)
    return start, min(start + per, n_items)


class FeatureGather:
    """Gathers equally-shaped per-rank feature slabs [clips_local, T, C] to `dst`.
    Asynchronous: start() enqueues the collective behind the producing stream,
    wait() returns the [world*clips_local, T, C] tensor on dst (None elsewhere)."""

    def __init__(self, dst=0, group=None):
        self.dst, self.group = dst, group
        self.work, self.out = None, None

    def start(self, slab):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            self.out, self.work = slab, None
            return
        rank = dist.get_rank(self.group)
        if rank == self.dst:
            if self.out is None or self.out.shape[0] != world * slab.shape[0] or self.out.shape[1:] != slab.shape[1:]:
                self.out = torch.empty((world * slab.shape[0],) + tuple(slab.shape[1:]),
                                       dtype=slab.dtype, device=slab.device)
            chunks = list(self.out.chunk(world, dim=0))
        else:
            chunks = None
        self.work = dist.gather(slab, chunks, dst=self.dst, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1 or dist.get_rank(self.group) == self.dst:
            return self.out
        return None
