#!/usr/bin/env python3
"""The compute side of a bench configuration's roofline, from the PMC passes of tools/prof.sh (its summary.txt): how busy
the vector unit, the LDS and the matrix pipe were while the dominant kernel ran -- for kernels that are bound by
instruction issue, not by HBM (`roofline.compute` of the bench line).

    python tools/prof_compute.py <summary.txt> <kernel substring> <config> [units per launch]  -> profiles-style JSON on stdout

Counters (MI355X_MICROARCH.md, PMC section): SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (x 4 = cycles of a
SIMD's issue port taken), SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles per CU, SQ_VALU_MFMA_BUSY_CYCLES
cycles of a SIMD's matrix pipe, GRBM_GUI_ACTIVE the shader clock summed over the 8 XCDs.  Every figure is per dispatch
(the passes see different numbers of dispatches)."""
import json
import re
import sys

SIMDS, CUS, XCDS = 1024, 256, 8


def main():
    path, kern, config = sys.argv[1], sys.argv[2], int(sys.argv[3])
    units = float(sys.argv[4]) if len(sys.argv) > 4 else None
    per, avg_us = {}, []
    for line in open(path, errors="replace"):
        if kern not in line:
            continue
        m = re.search(r"\s(\S+)\s+dispatches=\s*(\d+) sum=(\S+) per_dispatch=(\S+)", line)
        if m:
            per[m.group(1)] = float(m.group(4))
            continue
        m = re.match(r"\S.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s", line[len(kern):]) if line.startswith(kern) else None
        f = line.split()
        if len(f) > 6 and f[0].startswith(kern.split("<")[0]):
            try:
                avg_us.append(float(f[-11]))
            except (ValueError, IndexError):
                pass
    assert "GRBM_GUI_ACTIVE" in per and "SQ_ACTIVE_INST_VALU" in per, sorted(per)
    cyc = per["GRBM_GUI_ACTIVE"] / XCDS  # shader cycles of one dispatch
    cycles_from = "GRBM_GUI_ACTIVE / 8 XCDs"
    if "SQ_BUSY_CYCLES" in per:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; the two agree within a few percent for short kernels, but the
        # GRBM_GUI_ACTIVE-only pass of a long persistent launch has read several times the launch's duration -- the SQ figure
        # (same pass as the busy counters) is taken when they disagree
        sq = per["SQ_BUSY_CYCLES"] / 32.0
        if abs(cyc - sq) > 0.25 * sq:
            cyc, cycles_from = sq, "SQ_BUSY_CYCLES / 32 shader engines (GRBM_GUI_ACTIVE / 8 read %.3g)" % (per["GRBM_GUI_ACTIVE"] / XCDS)
    out = {"config": config, "kernel": kern, "source": path, "shader_cycles_per_dispatch": cyc, "shader_cycles_from": cycles_from,
           "valu_busy": 4.0 * per["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc),
           "lds_busy": per.get("SQ_LDS_IDX_ACTIVE", 0.0) / (CUS * cyc),
           "lds_bank_conflict_share": per.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(per.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
           # (the counter and the cycle count come from different passes, i.e. different runs of the kernel: a saturated port
           # can read a percent over 1 -- capped)
           "issue_busy": min(1.0, 4.0 * per.get("SQ_ACTIVE_INST_ANY", 0.0) / (SIMDS * cyc)),
           "mfma_busy": (per["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc)) if "SQ_VALU_MFMA_BUSY_CYCLES" in per else None,
           "counters_per_dispatch": per}
    if avg_us:
        out["kernel_us_in_pmc_passes"] = sum(avg_us) / len(avg_us)
        out["shader_clock_mhz"] = cyc / (sum(avg_us) / len(avg_us))
    if units:
        out["units_per_dispatch"] = units
        for k, name in (("SQ_INSTS_VALU", "valu_insts_per_unit_and_wave"), ("SQ_INSTS_LDS", "lds_insts_per_unit_and_wave"),
                        ("SQ_INSTS_SALU", "salu_insts_per_unit_and_wave")):
            if k in per:
                out[name] = per[k] / units
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
