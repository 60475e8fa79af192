#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as the guide
prescribes; csv output).  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B-per-lane coalesced streams at
64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Units: KB in the csv.
    python tools/pmc_summary.py FETCH_DIR WRITE_DIR OUT.json [workload]"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name).replace("prmf::", "")


def collect(d, counter):
    acc = {}
    for fp in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fp)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    wl = sys.argv[4] if len(sys.argv) > 4 else ""
    f, w = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    res = {"workload": wl, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE exact; KB -> bytes x1024; memory-side counters "
                                         "(Infinity-Cache hits included)", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if k.startswith("at::") or "rocclr" in k or k.startswith("Cijk"):
            continue
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fb = 2.0 * 1024.0 * sf / nf if nf else 0.0
        wb = 1024.0 * sw / nw if nw else 0.0
        res["kernels"][k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_bytes_per_launch": fb,
                             "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    sweeps = [v for k, v in res["kernels"].items() if k.startswith("sweep_kernel")]
    if sweeps:
        res["hbm_bytes_per_sweep_launch"] = sum(v["hbm_bytes_per_launch"] for v in sweeps) / len(sweeps)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from prmf_amd._build import sweep_sources_sha16
    res["sweep_sources_sha16"] = sweep_sources_sha16()       # the kernel these counters belong to (bench.py reports them only while it matches)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-60s fetch %10.2f MB  write %9.2f MB per launch (%d launches)" % (k[:60], v["fetch_bytes_per_launch"] / 1e6,
                                                                                v["write_bytes_per_launch"] / 1e6, v["launches_fetch_pass"]))


if __name__ == "__main__":
    main()
