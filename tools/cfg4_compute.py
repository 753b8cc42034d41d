#!/usr/bin/env python3
"""cfg 4's compute record: tools/prof_compute.py per kernel CLASS of the step (the time-domain matrix kernels, the narrow-band
inverse kernels, the forward transform) from one prof_cmd.sh summary, plus the step-level occupancy (tools/step_occupancy.py).
The top-level busy figures are the classes' figures weighted by their share of the summed kernel time.
usage: cfg4_compute.py <summary.txt> <occupancy.json>  -> profiles/<round>_bench_cfg4_compute.json on stdout"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
summary, occ_path = sys.argv[1], sys.argv[2]
names = []
for line in open(summary, errors="replace"):
    m = re.match(r"(k_cwt_\w+(?:<[^>]*>)?)\s", line)
    if m and m.group(1) not in names:
        names.append(m.group(1))
classes = {}
for n in names:
    r = subprocess.run([sys.executable, os.path.join(HERE, "prof_compute.py"), summary, n, "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode == 0:
        d = json.loads(r.stdout)
        d.pop("counters_per_dispatch", None)
        d.pop("source", None)
        classes[n] = d
try:
    occ = json.load(open(occ_path))
except (OSError, ValueError):
    occ = None
w = {n: (occ or {}).get("families", {}).get(n, {}).get("sum_ms", 0.0) or c.get("kernel_us_in_pmc_passes", 0.0) for n, c in classes.items()}
tot = sum(w.values()) or 1.0
out = {"config": 4, "kernel": "all kernels of the step, weighted by their share of the summed kernel time", "source": summary}
for k in ("valu_busy", "lds_busy", "issue_busy", "mfma_busy", "lds_bank_conflict_share", "shader_clock_mhz"):
    vals = [(classes[n].get(k), w[n]) for n in classes if classes[n].get(k) is not None]
    out[k] = sum(v * ww for v, ww in vals) / sum(ww for _, ww in vals) if vals else None
out["classes"] = classes
out["share_of_summed_kernel_time"] = {n: w[n] / tot for n in classes}
out["occupancy"] = occ
print(json.dumps(out, indent=1))
