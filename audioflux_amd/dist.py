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
    if "LOCAL_RANK" in os.environ and torch.cuda.is_available():
        # one process per GPU: torch's current device AND the library's default device (new objects, their
        # streams and scratch; the device afx_comm_create binds a communicator to) are this rank's GPU
        from . import _lib
        torch.cuda.set_device(local)
        _lib.check(_lib.get_lib().afx_set_device(local), "afx_set_device")
    return rank, local, world


def shard_range(n_items, rank, world):
    """contiguous clip-major block of rank: [start, stop) with ceil-sized blocks, so
    each rank's output is one contiguous slab of the gathered tensor"""
    per = -(-n_items // world)
    start = min(rank * per, n_items)
    return start, min(start + per, n_items)


class FeatureGather:
    """Gathers per-rank feature slabs [clips_local, ...] to `dst`, clip-major (rank r's clips follow
    rank r-1's).  Asynchronous: start() enqueues the collective on the CURRENT stream (callers put it
    on a side stream behind the producing one), wait() returns the gathered tensor on dst (None
    elsewhere).  Ranks may hold different clip counts (shard_range gives ceil-sized blocks, the last
    ranks get fewer): every slab is padded to `clips_max` rows for the equal-chunk collective and the
    result is trimmed -- `clips_max` / `counts` are fixed at construction or learnt once by an
    all_gather of the counts.  `always_collective` runs the same RCCL code path at world size 1
    (communicator creation, stream ordering), which is what the single-GPU box can exercise."""

    def __init__(self, dst=0, group=None, counts=None, always_collective=False):
        self.dst, self.group = dst, group
        self.counts = list(counts) if counts is not None else None
        self.always = always_collective
        self.work, self.out, self.pad = None, None, None

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _learn_counts(self, slab):
        world = self._world()
        mine = torch.tensor([slab.shape[0]], dtype=torch.int64, device=slab.device)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine, group=self.group)
        self.counts = [int(g.item()) for g in got]

    def start(self, slab):
        world = self._world()
        if world == 1 and not (self.always and dist.is_initialized()):
            self.out, self.work = slab, None
            return
        if self.counts is None:
            self._learn_counts(slab)
        assert len(self.counts) == world and slab.shape[0] == self.counts[dist.get_rank(self.group)], \
            "FeatureGather: slab rows differ from the announced per-rank counts"
        cmax = max(self.counts)
        send = slab
        if slab.shape[0] != cmax:  # short last shard: pad for the equal-chunk collective
            if self.pad is None or self.pad.shape[1:] != slab.shape[1:] or self.pad.dtype != slab.dtype:
                self.pad = torch.zeros((cmax,) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
            self.pad[: slab.shape[0]].copy_(slab)
            send = self.pad
        rank = dist.get_rank(self.group)
        if rank == self.dst:
            shape = (world * cmax,) + tuple(slab.shape[1:])
            if self.out is None or tuple(self.out.shape) != shape or self.out.dtype != slab.dtype:
                self.out = torch.empty(shape, dtype=slab.dtype, device=slab.device)
            chunks = list(self.out.chunk(world, dim=0))
        else:
            chunks = None
        self.work = dist.gather(send.contiguous(), chunks, dst=self.dst, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        world = self._world()
        if world == 1 and not (self.always and dist.is_initialized()):
            return self.out
        if self.out is None or dist.get_rank(self.group) != self.dst:
            return None
        if self.counts is None or all(c == self.counts[0] for c in self.counts):
            return self.out
        cmax = max(self.counts)  # trim the padding rows of the short shards
        return torch.cat([self.out[r * cmax: r * cmax + c] for r, c in enumerate(self.counts)], dim=0)


def save_npy(path, tensor):
    """Write a (gathered) float32 feature tensor as .npy through the library's own writer
    (include/afx_batch.h: afx_write_npy_f32 -- the on-wire format a C / C++ host uses too)."""
    import ctypes
    from . import _lib
    t = tensor.detach().to("cpu", torch.float32).contiguous()
    lib = _lib.get_lib()
    lib.afx_write_npy_f32.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
    dims = (ctypes.c_longlong * t.dim())(*t.shape)
    _lib.check(lib.afx_write_npy_f32(str(path).encode(), t.data_ptr(), t.dim(), dims), "afx_write_npy_f32")


class NativeGather:
    """The same exchange through the library's own C-ABI export (include/afx_batch.h: afx_comm_*,
    afx_gather = ncclGather over RCCL / xGMI, bound at run time) -- what a C / C++ host of the
    library uses; torch.distributed only carries the 128-byte bootstrap id here (any launcher
    side channel does).  Slabs must have the same shape on every rank."""

    def __init__(self, dst=0, group=None):
        import ctypes
        from . import _lib
        self._lib, self._check = _lib.get_lib(), _lib.check
        self.dst = dst
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self._lib.afx_comm_get_unique_id(ident), "afx_comm_get_unique_id")
        if world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = ctypes.create_string_buffer(box[0], 128)
        self._comm = ctypes.c_void_p(None)
        self._lib.afx_comm_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p]
        self._check(self._lib.afx_comm_create(ctypes.byref(self._comm), world, rank, ident), "afx_comm_create")
        self._lib.afx_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p]
        self._lib.afx_comm_free.argtypes = [ctypes.c_void_p]
        self._lib.afx_comm_free.restype = None
        self.world, self.rank, self.out = world, rank, None

    def start(self, slab, stream=None):
        """asynchronous on `stream` (default: torch's current stream)"""
        assert slab.is_cuda and slab.dtype == torch.float32 and slab.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream(slab.device)
        recv = None
        if self.rank == self.dst:
            shape = (self.world * slab.shape[0],) + tuple(slab.shape[1:])
            if self.out is None or tuple(self.out.shape) != shape:
                self.out = torch.empty(shape, dtype=slab.dtype, device=slab.device)
            recv = self.out.data_ptr()
        self._check(self._lib.afx_gather(self._comm, slab.data_ptr(), slab.numel(), recv, self.dst, s.cuda_stream),
                    "afx_gather")

    def wait(self):
        """stream-ordered like every device call of the library: the result is valid for work
        enqueued behind the gather on its stream (synchronise that stream to read it on the host)"""
        return self.out if self.rank == self.dst else None

    def close(self):
        if self._comm:
            self._lib.afx_comm_free(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
