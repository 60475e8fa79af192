#!/usr/bin/env python
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 --kernel-trace run kept in the
rocpd sqlite format (`*_results.db`), plus the gaps between consecutive dispatches; writes the CSV kept under profiles/.
    python tools/rocpd_stats.py gpurun_out/x/c3_results.db [out.csv] [--skip-first N]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.replace("prmf::", "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0
    for i, a in enumerate(sys.argv):
        if a == "--skip-first":
            skip = int(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    con = sqlite3.connect(args[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(con.execute("select %s, start, end from kernels order by start" % name_col))
    rows = rows[skip:]
    stats = {}
    for name, st, en in rows:
        d = (en - st) / 1000.0
        s = stats.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        s[0] += 1
        s[1] += d
        s[2] = min(s[2], d)
        s[3] = max(s[3], d)
    tot = sum(s[1] for s in stats.values())
    out = sorted(stats.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(args[1], "w", newline="")) if len(args) > 1 else None
    hdr = ["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"]
    if w:
        w.writerow(hdr)
    print("%-70s %7s %12s %10s %10s %10s %6s" % tuple(hdr))
    for k, (n, t, mn, mx) in out:
        row = [k, n, "%.1f" % t, "%.2f" % (t / n), "%.2f" % mn, "%.2f" % mx, "%.1f" % (100 * t / tot)]
        if w:
            w.writerow(row)
        print("%-70s %7d %12s %10s %10s %10s %6s" % tuple([k[:70]] + row[1:]))
    # gaps between consecutive kernels (end -> next start) in the steady part
    gaps = [(rows[i + 1][1] - rows[i][2]) / 1000.0 for i in range(len(rows) - 1)]
    gaps = [g for g in gaps if 0 <= g < 50]
    if gaps:
        gaps.sort()
        print("gaps between consecutive dispatches (<50us): n=%d median %.2f us mean %.2f us p90 %.2f us" %
              (len(gaps), gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[int(0.9 * len(gaps))]))


if __name__ == "__main__":
    main()
