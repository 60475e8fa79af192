#!/usr/bin/env python
"""MFMA-pipe utilisation of the sweep launches from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py ...
Per dispatch: duration from the kernel trace, counters from the collection file (matched by dispatch id).
  effective shader clock      = GRBM_GUI_ACTIVE / 8 / duration            (the counter is summed over the 8 XCDs; the guide's
                                                                          DVFS note: reads high on dispatches < 0.3 ms)
  MFMA pipe utilisation       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
  algorithmic MFMA cycles     = 32 cycles x matrix instructions the arithmetic needs (stated per workload below)
    python tools/pmc_mfma_summary.py DIR OUT.json WORKLOAD"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prmf_amd.synthetic import CONFIGS  # noqa: E402


def short(name):
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    return re.sub(r"^void ", "", name).replace("prmf::", "")


def main():
    d, out, wl = sys.argv[1:4]
    dur = {}
    for fp in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(fp)):
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    per = {}
    for fp in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fp)):
            k = short(row["Kernel_Name"])
            if not k.startswith("sweep_kernel"):
                continue
            e = per.setdefault(row["Dispatch_Id"], {"kernel": k})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    m, n, k, P, xd = CONFIGS[wl]
    KP = (k + 31) // 32 * 32
    per_frag = 2 if xd == "bf16" else 5                      # hi/lo factor pieces | three-piece split of f32 X
    pad = lambda a, b: (a + b - 1) // b * b
    n_mfma = (pad(m, 256) / 32) * (pad(n, 128) / 16) * (KP / 32) * per_frag
    for did, e in per.items():
        if did in dur:
            e["duration_s"] = dur[did]
    out_rows = []
    for kname in sorted({e["kernel"] for e in per.values()}):
        rows = [e for e in per.values() if e["kernel"] == kname and "duration_s" in e and "GRBM_GUI_ACTIVE" in e]
        rows = rows[len(rows) // 4:]                          # drop the warm-up launches
        if not rows:
            continue
        avg = lambda key: sum(e[key] for e in rows) / len(rows)
        gui, mf, t = avg("GRBM_GUI_ACTIVE"), avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("duration_s")
        # a workload whose two sweeps run different kernels (C5: 4-wave X V, 8-wave X^T U) has half the launches each
        res = {"workload": wl, "kernel": kname, "sweep_launches_averaged": len(rows),
               "avg_duration_ms": t * 1e3, "GRBM_GUI_ACTIVE_per_launch": gui, "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": mf,
               "SQ_BUSY_CYCLES_per_launch": avg("SQ_BUSY_CYCLES") if "SQ_BUSY_CYCLES" in rows[0] else None,
               "effective_clock_ghz": gui / 8.0 / t / 1e9,
               "mfma_pipe_utilisation": mf / (1024.0 * gui / 8.0),
               "algorithmic_mfma_instructions_per_launch": n_mfma, "algorithmic_mfma_cycles_per_launch": 32.0 * n_mfma,
               "counter_over_algorithmic": mf / (32.0 * n_mfma),
               "note": "under rocprofv3 --pmc the launches run slower than un-profiled ones (the guide's DVFS note (2)); utilisation = "
                       "busy cycles / (1024 SIMDs x shader cycles of the launch)"}
        out_rows.append(res)
        print("%-8s %s: %.3f ms/launch, clock %.2f GHz, MFMA pipe busy %.1f%% (counter / algorithmic = %.3f)" % (
            wl, kname, t * 1e3, res["effective_clock_ghz"], 100 * res["mfma_pipe_utilisation"], res["counter_over_algorithmic"]))
    if not out_rows:
        sys.exit("no sweep dispatches with counters found in %s" % d)
    from prmf_amd._build import sweep_sources_sha16
    nl = sum(r["sweep_launches_averaged"] for r in out_rows)
    top = dict(out_rows[0]) if len(out_rows) == 1 else {"workload": wl, "kernels": out_rows}
    # launch-weighted over the sweep kernels of the workload (its two sweeps may be different instantiations): what bench.py reads
    top["effective_clock_ghz"] = sum(r["effective_clock_ghz"] * r["sweep_launches_averaged"] for r in out_rows) / nl
    top["mfma_pipe_utilisation"] = sum(r["mfma_pipe_utilisation"] * r["sweep_launches_averaged"] for r in out_rows) / nl
    top["sweep_sources_sha16"] = sweep_sources_sha16()       # the kernel these counters belong to (prmf_amd/_build.py)
    json.dump(top, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
