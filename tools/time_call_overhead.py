#!/usr/bin/env python
"""Where does a prmf_step call spend its time outside the iterations?  Times step(K) for several K at a benchmark
shape (fence, perf_counter, step, fence -- exactly bench.py's bracket), REPS repetitions each, and fits
t(K) = call_overhead + K * per_iteration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c3")
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
eng.step(U, V, 10, 1.0, 1.0)
Ks = [1, 2, 5, 10, 20, 50, 100]
reps = int(os.environ.get("REPS", 7))
med = []
for K in Ks:
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(U, V, K, 1.0, 1.0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med.append(ts[len(ts) // 2])
    print("%s K=%3d  median %.3f ms  (%.1f us per iteration; min %.3f max %.3f ms)" % (wl, K, 1e3 * med[-1], 1e6 * med[-1] / K,
                                                                                      1e3 * ts[0], 1e3 * ts[-1]))
A = np.stack([np.ones(len(Ks)), np.array(Ks, dtype=float)], axis=1)
a, b = np.linalg.lstsq(A, np.array(med), rcond=None)[0]
print("%s fit: call overhead %.1f us + %.1f us per iteration" % (wl, 1e6 * a, 1e6 * b))
