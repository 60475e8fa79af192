#!/usr/bin/env python
"""Where the time of a prmf_step(K) call goes beyond its K iterations (small workloads feel it: 8 % at C2 with K = 20):
begin, the iterations, and prmf_end (last prologue, the two record launches, three small copies, the stream wait)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c2")
K = int(os.environ.get("K", 20))
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=xd)
eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k)
V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.step(U, V, K, 1.0, 1.0)
torch.cuda.synchronize()
T = {"step": [], "begin": [], "iters_host": [], "iters_dev": [], "end": []}
for rep in range(15):
    t0 = time.perf_counter()
    eng.step(U, V, K, 1.0, 1.0)
    T["step"].append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    eng.begin(U, V, 1.0, 1.0)
    t1 = time.perf_counter()
    for _ in range(K):
        eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    eng.end(K)
    t4 = time.perf_counter()
    T["begin"].append(t1 - t0); T["iters_host"].append(t2 - t1); T["iters_dev"].append(t3 - t1); T["end"].append(t4 - t3)
med = {key: 1e6 * sorted(v)[len(v) // 2] for key, v in T.items()}
print("%s K=%d: prmf_step %.0f us = %.1f us per iteration | begin %.0f us | %d iterations: host enqueue %.0f us, device done after %.0f us (%.1f us each) | prmf_end %.0f us"
      % (wl, K, med["step"], med["step"] / K, med["begin"], K, med["iters_host"], med["iters_dev"], med["iters_dev"] / K, med["end"]))
