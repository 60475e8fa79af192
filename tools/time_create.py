#!/usr/bin/env python
"""What prmf_create / prmf_destroy cost (wall clock) at several shapes: fixed part against the part that grows with the buffers."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

torch.cuda.init()
torch.zeros(1, device="cuda:0")
for (m, n, k, xd) in [(100, 1000, 6, "f32"), (5000, 10000, 16, "f32"), (50000, 20000, 64, "bf16"), (25000, 25000, 128, "bf16"), (50000, 20000, 64, "bf16"), (100, 1000, 6, "f32")]:
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = PrmfEngine(m, n, k, x_dtype=xd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e.close()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    print("%6d x %6d k=%3d %s: create %s ms, destroy %s ms" % (m, n, k, xd, " ".join("%.1f" % (1e3 * a) for a, _ in ts), " ".join("%.1f" % (1e3 * b) for _, b in ts)), flush=True)
