#!/usr/bin/env python
"""The setup calls of one solve, timed one by one (wall clock, device idle before and after each): prmf_create, the upload,
prmf_set_x, prmf_set_pathway_library, prmf_set_active, the first and second batch of 10 iterations, prmf_destroy -- twice in a
row, as two nmf_pathway calls of one process would issue them.   WORKLOAD=c3 python tools/time_setup.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c3")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
X = np.empty((m, n), dtype=np.float32)
for r0 in range(0, m, 16384):
    X[r0:r0 + 16384] = Xd[r0:r0 + 16384].float().cpu().numpy()
del Xd
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
lib.arrays()


def timed(what, f, out):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    out.append((what, 1e3 * (time.perf_counter() - t0)))
    return r


for rep in range(3):
    t = []
    eng = timed("create", lambda: PrmfEngine(m, n, k, x_dtype=xd), t)
    Xdev = timed("upload", lambda: eng.device_tensor(X, eng.torch_x_dtype), t)
    nx = timed("norm", lambda: eng.local_norm_sq(Xdev), t)
    timed("set_x", lambda: eng.set_x(Xdev, nx), t)
    del Xdev
    timed("set_library", lambda: eng.set_library(lib), t)
    timed("set_active", lambda: eng.set_active([kk % P for kk in range(k)]), t)
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    timed("step(10) #1", lambda: eng.step(U, V, 10, 1.0, 1.0), t)
    timed("step(10) #2", lambda: eng.step(U, V, 10, 1.0, 1.0), t)
    timed("destroy", eng.close, t)
    print("%s pass %d: " % (wl, rep) + " | ".join("%s %.1f" % x for x in t) + "  (ms)", flush=True)
