#!/usr/bin/env python
"""Do the time-domain CWT launches (k_cwt_td) run beside the FFT-path launches of the same step?  Reads a rocprofv3
kernel-trace database (rocpd) and reports, per kernel family: dispatches, summed duration, and how much of that time
another family was executing too; plus the hardware queue / stream every family was dispatched on.
usage: td_overlap.py <results.db>"""
import re
import sqlite3
import sys


def fam(name):
    if "k_cwt_td" in name:
        return "td<1024>" if "1024" in name else "td<384>"
    if "nb2" in name:
        return "inv_nb2"
    if "_nb" in name:
        return "inv_nb"
    if "k_cwt_fwd" in name or "k_cwt_small" in name:
        return "fwd"
    if "k_cwt_inv" in name:
        return "inv_wide"
    return None


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("columns of `kernels`:", ", ".join(cols))
    extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
    rows = cur.execute(f"select name, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
    iv = {}
    where = {}
    for r in rows:
        f = fam(r[0])
        if not f:
            continue
        iv.setdefault(f, []).append((r[1], r[2]))
        where.setdefault(f, set()).add(tuple(r[3:]))
    t0 = min(s for v in iv.values() for s, _ in v)
    t1 = max(e for v in iv.values() for _, e in v)
    print(f"span of the CWT launches: {(t1 - t0) / 1e6:.2f} ms")
    for f, v in sorted(iv.items()):
        print(f"{f:10s} {len(v):6d} dispatches, sum {sum(e - s for s, e in v) / 1e6:9.2f} ms, avg {sum(e - s for s, e in v) / len(v) / 1e3:8.1f} us, "
              f"{extra} = {sorted(where[f])[:6]}")
    # time during which td<1024> is executing and a given other family is executing too (sweep)
    for a in ("td<1024>", "td<384>"):
        if a not in iv:
            continue
        A = iv[a]
        for b, B in sorted(iv.items()):
            if b == a:
                continue
            ev = [(s, 0, 1) for s, _ in A] + [(e, 0, -1) for _, e in A] + [(s, 1, 1) for s, _ in B] + [(e, 1, -1) for _, e in B]
            ev.sort()
            n = [0, 0]
            last = ev[0][0]
            both = 0
            for t, k, d in ev:
                if n[0] > 0 and n[1] > 0:
                    both += t - last
                n[k] += d
                last = t
            print(f"  {a} and {b} both executing: {both / 1e6:8.2f} ms")


main()
