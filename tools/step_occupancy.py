#!/usr/bin/env python3
"""Step-level occupancy of a multi-kernel, multi-stream bench configuration (cfg 4) from ONE rocprofv3 --kernel-trace database:
the kernels of the library run on two streams and their traced durations overlap, so the sum of the per-kernel averages is not
a decomposition of the step.  Reports, over the traced span: the UNION of the kernels' busy intervals (time during which at
least one of them executes), the SUM of their durations, and the same per kernel family.
usage: step_occupancy.py <results.db> [steps]      (PROF_DB_HOOK of tools/prof_cmd.sh)  -> JSON on stdout"""
import json
import re
import sqlite3
import sys


def fam(name):
    m = re.search(r"(k_cwt_\w+(?:<[^>]*>)?|k_\w+)", name)
    return m.group(1) if m and "at::native" not in name and "rocclr" not in name else None


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    by = {}
    for n, s, e in rows:
        f = fam(n)
        if f:
            by.setdefault(f, []).append((s, e))
    allv = [x for v in by.values() for x in v]
    t0, t1 = min(s for s, _ in allv), max(e for _, e in allv)
    u, sm = union(allv), sum(e - s for s, e in allv)
    # concurrency histogram: time with exactly k kernels executing
    ev = sorted([(s, 1) for s, _ in allv] + [(e, -1) for _, e in allv])
    hist, n, last = {}, 0, ev[0][0]
    for t, d in ev:
        hist[n] = hist.get(n, 0) + (t - last)
        n += d
        last = t
    out = {"span_ms": (t1 - t0) / 1e6, "union_busy_ms": u / 1e6, "sum_of_durations_ms": sm / 1e6, "sum_over_union": sm / u,
           "idle_share_of_span": 1.0 - u / (t1 - t0),
           "time_with_k_kernels_executing_ms": {str(k): v / 1e6 for k, v in sorted(hist.items()) if v > 0},
           "families": {f: {"dispatches": len(v), "sum_ms": sum(e - s for s, e in v) / 1e6, "avg_us": sum(e - s for s, e in v) / len(v) / 1e3,
                            "union_ms": union(v) / 1e6} for f, v in sorted(by.items())}}
    if steps:
        out["steps"] = steps
        out["union_busy_ms_per_step"] = u / 1e6 / steps
        out["sum_of_durations_ms_per_step"] = sm / 1e6 / steps
    print(json.dumps(out, indent=1))


main()
