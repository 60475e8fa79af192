#!/usr/bin/env python
"""What does a hand-off inside the sweep's tail cost?  (epilogue in the sweep's tail behind per-panel
arrival counters.)  Needs the -DPRMF_TAIL_PROBE=1 build:

    python -m prmf_amd._build --variant tailprobe PRMF_TAIL_PROBE=1
    PRMF_HIP_LIB=prmf_amd/lib/variants/libprmf_hip_tailprobe.so WORKLOAD=c3 python tools/tail_probe.py

Times both sweeps (HIP events inside the library, 20 iterations) without the probe and with it: every workgroup then
publishes its slabs (release fence, one agent-scope atomic per panel), claims groups of 4 output tiles from a ticket
counter, waits for the group's panel (acquire) and reads the group's tiles of one slab.  The difference is what a fused
epilogue pays for the hand-off before it has done any arithmetic; the epilogue launch it would save is printed next to it
(kernel trace numbers of the same workloads: profiles/r03_*_kernel_stats.csv)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine

wl = os.environ.get("WORKLOAD", "c3")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=xd, timing=True)
eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
buf = torch.zeros(64 + 2 * 4096, dtype=torch.int32, device=dev)


def run(tag, probe_which):
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    eng.invalidate_factors()
    eng.step(U, V, 5, 1.0, 1.0)
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        eng.step(U, V, 20, 1.0, 1.0)
        torch.cuda.synchronize()
        res.append(eng.timing()[:2])
    a = sorted(r[0] for r in res)[1]
    b = sorted(r[1] for r in res)[1]
    flag = int(buf[3].item())
    print("%-8s %-22s xv %.4f ms  xtu %.4f ms%s" % (wl, tag, a, b, "  (A WAIT GAVE UP)" if flag else ""), flush=True)
    return a, b


base = run("no probe", None)
out = {}
for which, name in ((0, "probe in X V"), (1, "probe in X^T U")):
    rc = eng.lib.prmf_debug_set_tail_probe(eng.ctx, which, C.c_void_p(buf.data_ptr()), 4)
    if rc != 0:
        print("%-8s %-22s not available: %s" % (wl, name, eng.lib.prmf_last_error(eng.ctx).decode()))
        continue
    buf[0] = 0; buf[3] = 0; buf[64:64 + 4096] = 0      # the probe is keyed to this plan's launches only
    out[which] = run(name, which)
    eng.lib.prmf_debug_set_tail_probe(None, 0, None, 0)
for which, r in out.items():
    d = (r[which] - base[which]) * 1e3
    print("%-8s hand-off cost in %s: %+.1f us per launch" % (wl, "X V" if which == 0 else "X^T U", d))
