#!/usr/bin/env python
"""A/B of library builds in ONE GPU session (boxes differ by up to 8%, so variants are only comparable on the same box):
runs tools/sweep_knobs.py once per (variant, workload), interleaved ROUNDS times, each in its own process with
PRMF_HIP_LIB pointing at the variant, and prints the per-variant medians.
    python tools/ab_sweep.py c3,c2 base=prmf_amd/lib/libprmf_hip.so glds0=prmf_amd/lib/variants/libprmf_hip_glds0.so"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workloads = sys.argv[1].split(",")
variants = [a.split("=", 1) for a in sys.argv[2:]]
rounds = int(os.environ.get("ROUNDS", 3))
res = {}
for rnd in range(rounds):
    for wl in workloads:
        for name, path in variants:
            # a variant may carry environment settings behind its library:  name=lib.so:PRMF_SWEEP_LDX=64:OTHER=1
            path, *extra = path.split(":")
            env = dict(os.environ, PRMF_HIP_LIB=os.path.join(ROOT, path), WORKLOAD=wl)
            env.update(kv.split("=", 1) for kv in extra)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_knobs.py")], env=env, capture_output=True, text=True)
            m = re.search(r"xv ([0-9.]+) ms .* xtu ([0-9.]+) ms .* iteration ([0-9.]+) ms", r.stdout)
            if not m:
                print("FAILED", name, wl, r.stdout[-300:], r.stderr[-300:], flush=True)
                sys.exit(1)
            res.setdefault((wl, name), []).append(tuple(float(x) for x in m.groups()))
            print("round %d %-8s %-10s xv %.4f xtu %.4f iteration %.4f" % ((rnd, wl, name) + res[(wl, name)][-1]), flush=True)
print()
for (wl, name), v in res.items():
    med = [sorted(x[i] for x in v)[len(v) // 2] for i in range(3)]
    print("MEDIAN %-8s %-10s xv %.4f ms  xtu %.4f ms  iteration %.4f ms" % (wl, name, med[0], med[1], med[2]))
