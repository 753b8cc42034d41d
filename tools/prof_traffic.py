#!/usr/bin/env python
"""HBM traffic per bench.py step from rocprofv3 PMC passes (run on the GPU box):

    python tools/prof_traffic.py <config> [--clips N]   -> gpurun_out/<round>_bench_cfg<config>_pmc.json (AFX_ROUND, default r05)

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one pass on gfx950, and
counters are never combined with trace domains other than kernel-trace); the library's kernels
(everything that is not a torch / rocclr kernel) are summed over the timed + warm-up steps and
divided by their number.  bench.py reads the committed copy under profiles/ for `roofline.traffic`
(2 x FETCH_SIZE -- gfx950 tallies wide coalesced reads at half, MI355X_MICROARCH.md -- + WRITE_SIZE)."""
import argparse
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, out_dir, bench_args, script="bench.py", extra_env=None):
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", **(extra_env or {}))
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, script)] + bench_args
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
    assert dbs, res.stdout[-2000:]
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    ours = [(n, c, v) for n, c, v in rows if "at::native" not in n and "rocclr" not in n and "hipcub" not in n]
    for d in dbs:
        os.remove(d)
    return ours


def kname(n):
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int)
    ap.add_argument("--clips", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    warm = 1
    args = ["--config", str(a.config), "--steps", str(a.steps), "--warmup", str(warm), "--no-cpu-baseline",
            "--no-sustained", "--no-check", "--no-secondary", "--no-legacy", "--clock-warmup", "0"]  # (traffic does not depend on the clock state)
    if a.clips:
        args += ["--clips", str(a.clips)]
    tmp = os.path.join(ROOT, "gpurun_out", f"prof_traffic_cfg{a.config}")
    fetch = run_pass("FETCH_SIZE", os.path.join(tmp, "fetch"), args)
    write = run_pass("WRITE_SIZE", os.path.join(tmp, "write"), args)
    steps = a.steps + warm
    sys.path.insert(0, ROOT)
    import bench
    rec = {
        "config": a.config,
        "clips": a.clips or bench.WORKLOADS[a.config].default_clips,
        "steps_profiled": steps,
        "fetch_kib_per_step": sum(v for _, _, v in fetch) / steps,
        "write_kib_per_step": sum(v for _, _, v in write) / steps,
        "kernels": {kname(n): {"dispatches": c, "fetch_kib": v, "write_kib": dict((kname(m), w) for m, _, w in write).get(kname(n))}
                    for n, c, v in fetch},
        "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py " + " ".join(args),
        "note": "KiB as reported by rocprofv3; FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 (x2 in bench.py)",
    }
    out = os.path.join(ROOT, "gpurun_out", f"{os.environ.get('AFX_ROUND', 'r05')}_bench_cfg{a.config}_pmc.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
