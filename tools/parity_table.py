#!/usr/bin/env python
"""Condense an AFX_PARITY_LOG file (tests/conftest.py::parity_log, one JSON line per parity decision of
`pytest -m gpu`) into the table of DESIGN.md section 2: every decision taken at a bar OTHER than plain
1e-5 peak + L2, worst measured value per test and label group, plus the headroom of the 1e-5 decisions.
usage: python tools/parity_table.py gpurun_out/parity_r02.jsonl"""
import collections
import json
import re
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
plain = [r for r in rows if r["kind"] == "peak/l2" and abs(r["bar"] - 1e-5) < 1e-12]
other = [r for r in rows if r not in plain]
print(f"{len(rows)} parity decisions, {len(plain)} at plain 1e-5 (worst measured {max(r['measured'] for r in plain):.2e}, "
      f"median {sorted(r['measured'] for r in plain)[len(plain) // 2]:.2e})\n")
groups = collections.OrderedDict()
for r in other:
    label = re.sub(r"\d+", "#", r["what"])[:60] if r["kind"] != "peak/l2" else ""
    key = (r["test"].split("::")[0].replace("tests/", ""), r["test"].split("::")[-1].split("[")[0], r["kind"], label.split(":")[0])
    g = groups.setdefault(key, {"n": 0, "worst": 0.0, "ratio": 0.0, "bars": set(), "extra": {}})
    g["n"] += 1
    g["bars"].add(r["bar"])
    g["worst"] = max(g["worst"], r["measured"])
    if r["bar"] > 0 and r["measured"] / r["bar"] >= g["ratio"]:
        g["ratio"] = r["measured"] / r["bar"]
        g["extra"] = {k: v for k, v in r.items() if k not in ("what", "measured", "bar", "kind", "test")}
print("| test | bar | decisions | worst measured | worst measured / bar | notes |")
print("|---|---|---|---|---|---|")
for (f, t, kind, label), g in groups.items():
    extra = ", ".join(f"{k} {v:.2e}" if isinstance(v, float) else f"{k} {v}" for k, v in g["extra"].items())
    bars = sorted(g["bars"])
    bar = f"{bars[0]:.1e}" if len(bars) == 1 else f"{bars[0]:.1e} .. {bars[-1]:.1e}"
    print(f"| `{f}::{t}` {label} | {bar} ({kind}) | {g['n']} | {g['worst']:.2e} | {g['ratio']:.2f} | {extra} |")
