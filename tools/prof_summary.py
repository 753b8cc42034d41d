#!/usr/bin/env python
"""Condense rocprofv3 rocpd databases into the small text summaries kept under
profiles/: per-kernel time statistics (kernel-trace runs) and per-kernel counter
sums (--pmc runs).   usage: prof_summary.py <results.db> [...]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name[:70]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        rows = cur.execute(
            "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
            "max(grid_x), max(workgroup_x) from kernels group by name order by sum(end-start) desc").fetchall()
        total = sum(r[5] for r in rows) or 1
        print(f"{'kernel':70s} {'calls':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}  vgpr agpr sgpr lds scratch grid wg")
        for r in rows:
            print(f"{short(r[0]):70s} {r[1]:5d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {100*r[5]/total:6.2f}  "
                  f"{r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]} {r[12]}")
        try:
            pm = cur.execute(
                "select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                "group by kernel_name, counter_name").fetchall()
        except sqlite3.Error as e:
            pm = []
            print("(no counter data:", e, ")")
        if pm:
            print("-- counters (sum over dispatches; dispatches listed)")
            for name, ctr, n, val in pm:
                if "at::native" in name or "rocclr" in name:
                    continue
                print(f"{short(name):50s} {ctr:28s} dispatches={n:3d} sum={val:.6g} per_dispatch={val/n:.6g}")


if __name__ == "__main__":
    main()
