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
    # BASELINE cfg 5 plus one clip over the eight GPUs of a node: ceil-sized blocks, the last rank gets the remainder
    spans = [shard_range(1001, r, 8) for r in range(8)]
    assert [hi - lo for lo, hi in spans] == [126] * 7 + [119] and spans[-1][1] == 1001


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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gather,extra", [("mfcc", []), ("mfcc,mel", []), ("chroma", ["--config", "5", "--total-clips", "7"]),
                                          ("chroma,cqt", ["--config", "5"])])
def test_bench_control_flow_world2_gloo(gather, extra):
    """bench.py itself under `torch.distributed.run --nproc-per-node 2` with the kernels replaced by a CPU
    stand-in (AFX_BENCH_DRYRUN=1, gloo): the launch line the driver uses, clip ownership, the
    double-buffered side-stream gather(s), fences, the max-over-ranks clock and the JSON contract; rank 0
    checks the order of the gathered slab."""
    import json
    import subprocess
    env = dict(os.environ, AFX_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--gather", gather, *extra]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    sharded = "--total-clips" in extra  # 7 clips over 2 ranks: 4 + 3 through dist.shard_range, the short slab padded
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == ("strong" if sharded else "weak")
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    units = 7 * 7 if sharded else d["config"]["units_per_step_per_gpu"] * 2
    assert units * 3 / (d["ms_per_step"] * 3e-3) == pytest.approx(d["value"], rel=1e-6)
    if sharded:
        assert d["config"]["clips_per_rank"] == [4, 3]
    # N > 1 says whether the exchange hides behind the compute
    g = d["gather"]
    assert g["slabs"] == gather.split(",") and g["gather_ms"] is not None and g["gather_ms"] >= 0
    assert g["overlap_hidden_ms"] is not None and g["exposed_ms"] >= 0
    assert "RCCL gather of " + "+".join(gather.split(",")) in d["config"]["parallelism"]
    assert "cpu_baseline" not in d  # N > 1: no CPU baseline leg


@pytest.mark.timeout(300)
def test_bench_replicas_only_world2_gloo():
    """`bench.py --config 4 --gpus 2` dry: the CWT configuration has no exchange step (replicas
    only), so N > 1 is the launch line, the barriers, the max-over-ranks clock and the summed units --
    no gather object, no `gather` block on the line."""
    import json
    import subprocess
    env = dict(os.environ, AFX_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--config", "4", "--gpus", "2", "--steps", "3", "--warmup", "1"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    units = d["config"]["units_per_step_per_gpu"] * 2
    assert units * 3 / (d["ms_per_step"] * 3e-3) == pytest.approx(d["value"], rel=1e-6)
    assert "gather" not in d or d["gather"] is None or not d["gather"].get("slabs")
    assert "RCCL gather" not in d["config"]["parallelism"]
    assert "cpu_baseline" not in d


@pytest.mark.timeout(300)
@pytest.mark.parametrize("config", [2, 4, 5])
def test_bench_launches_its_own_ranks(config):
    """plain `python bench.py --gpus 2` (no launcher environment -- how a driver that runs `python bench.py --gpus 1`
    would ask for more GPUs): the command re-executes itself under torch.distributed.run, one rank per GPU, and
    rank 0 still prints the ONE JSON line; `rccl_ranks` is the size of the process group that ran."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AFX_BENCH_DRYRUN"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", str(config)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    if config == 4:
        assert not (d.get("gather") or {}).get("slabs")
    else:
        assert d["gather"]["slabs"] and d["gather"]["gather_ms"] is not None


def test_bench_launcher_passes_failures_on():
    """a rank that dies must fail the plain command too (the driver reads the exit status)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AFX_BENCH_DRYRUN"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--gather", "nonsense"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config", [2, 4, 5])
def test_bench_world8_line_is_complete(config):
    """the launch the driver's SCALE run makes at N = 8 -- `python bench.py --gpus 8 --config C` -- dry
    (AFX_BENCH_DRYRUN=1, gloo, 8 processes on this box's CPUs): ONE JSON line with n_gpus / rccl_ranks 8, every rank's clip
    span, the gather block (gather_ms, exposed_ms; none for cfg 4: replicas only) and, with --n1-value, the weak-scaling
    efficiency north_star asks for (value / (8 x the one-GPU value))."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AFX_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--config", str(config),
           "--n1-value", "1000.0"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=560)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "weak"
    assert d["config"]["clips_per_rank"] == [d["config"]["clips_per_gpu"]] * 8
    assert d["value"] > 0 and abs(d["efficiency_vs_n1"] - d["value"] / 8000.0) < 1e-9 and d["n1_value"] == 1000.0
    g = d["gather"]
    if config == 4:
        assert not g["slabs"] and g["exposed_ms"] == 0.0 and "replicas only" in d["config"]["parallelism"]
    else:
        assert g["slabs"] and g["gather_ms"] is not None and g["exposed_ms"] is not None and g["overlap_hidden_ms"] is not None
        assert "RCCL gather" in d["config"]["parallelism"]
    # the rank's units are summed over the ranks: 8 x one rank's
    assert d["config"]["units_per_step_per_gpu"] * 8 * d["steps"] == pytest.approx(d["value"] * d["ms_per_step"] * d["steps"] / 1e3, rel=1e-6)
