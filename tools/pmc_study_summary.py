#!/usr/bin/env python
"""Per-kernel means of every counter collected by tools/pmc_study.sh (one directory per pass), as a table
kernel x counter (mean per launch), plus a few ratios (average latency = LATENCY / REQ, in-flight = LEVEL / cycles ...).
    python3 tools/pmc_study_summary.py gpurun_out/r05pmc"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    name = re.sub(r"^void ", "", name).replace("prmf::", "")
    m = re.match(r"sweep_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>", name)
    return name


root = sys.argv[1]
data = {}      # kernel -> counter -> [sum, n]
order = {}     # kernel -> dispatch order list (for telling X V from X^T U)
for fp in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(fp)))
    # the two sweeps share a kernel name when both take the same form: tell them apart by launch order (X V first)
    seen = {}
    for row in rows:
        k = short(row["Kernel_Name"])
        if k.startswith("at::") or "rocclr" in k or k.startswith("Cijk") or "elementwise" in k or "vectorized" in k:
            continue
        did = row.get("Dispatch_Id") or row.get("Dispatch_ID")
        if k.startswith("sweep_kernel"):
            key = (k, did)
            if key not in seen:
                ids = sorted({r.get("Dispatch_Id") or r.get("Dispatch_ID") for r in rows if short(r["Kernel_Name"]) == k}, key=int)
                seen.update({(k, d): ("xv" if i % 2 == 0 else "xtu") for i, d in enumerate(ids)})
            nkern = len({short(r["Kernel_Name"]) for r in rows if short(r["Kernel_Name"]).startswith("sweep_kernel")})
            k = k + (" [" + seen[key] + "]" if nkern == 1 else "")
        a = data.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
        a[0] += float(row["Counter_Value"])
        a[1] += 1
for k in sorted(data):
    print("== " + k)
    c = {name: v[0] / v[1] for name, v in data[k].items()}
    for name in sorted(c):
        print("   %-42s %18.1f   (%d launches)" % (name, c[name], data[k][name][1]))

    def ratio(label, a, b, scale=1.0):
        if a in c and b in c and c[b]:
            print("   -> %-39s %18.3f" % (label, scale * c[a] / c[b]))
    ratio("avg TCP->TCC read latency (cycles)", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum")
    ratio("avg TCP->TCC write latency (cycles)", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum")
    ratio("wave cycles waiting (s_waitcnt/barrier)", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    ratio("wave cycles issue-stalled", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
    ratio("wave cycles issuing", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
    ratio("VMEM instructions in flight per CU-cycle", "SQ_INST_LEVEL_VMEM", "SQ_BUSY_CYCLES")
    ratio("L2 hit rate", "TCC_HIT_sum", "TCC_REQ_sum")
    ratio("UTCL1 miss rate", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum")
    ratio("LDS bank-conflict cycles / LDS active", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
    ratio("EA read requests in flight (LEVEL / TCC busy)", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_BUSY_sum")
