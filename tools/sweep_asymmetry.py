#!/usr/bin/env python
"""Is the slower of the two C3 sweeps slower because of its SHAPE (98 panels x 314 chunks against 40 x 782) or because of
which buffer it streams?  Times both sweeps (HIP events in the library) for 50000 x 20000 and for the transposed problem
20000 x 50000 (where X V has the 40-panel shape and X^T U the 98-panel one), k = 64 bf16, alternating engines."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

dev = "cuda:0"
k, P = 64, 300
res = {}
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for (m, n) in ((50000, 20000), (20000, 50000)):
        Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
        eng = PrmfEngine(m, n, k, x_dtype="bf16", timing=True)
        eng.set_x(Xd)
        del Xd
        eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
        eng.set_active([kk % P for kk in range(k)])
        U = syn.init_factor_torch(torch, dev, 1, m, k)
        V = syn.init_factor_torch(torch, dev, 2, n, k)
        eng.step(U, V, 5, 1.0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(U, V, 20, 1.0, 1.0)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20
        a, b, _ = eng.timing()
        info = eng.sweep_info()
        print("round %d  %6d x %6d: X V %.4f ms (grid %d, slabs %d)  X^T U %.4f ms (grid %d, slabs %d)  iteration %.4f ms" % (
            rnd, m, n, a, info.grid_xv, info.splits_xv, b, info.grid_xtu, info.splits_xtu, el * 1e3), flush=True)
        res.setdefault((m, n), []).append((a, b))
        eng.close()
        torch.cuda.empty_cache()
for key, v in res.items():
    med = [sorted(x[i] for x in v)[len(v) // 2] for i in range(2)]
    print("MEDIAN %6d x %6d: X V %.4f ms  X^T U %.4f ms" % (key[0], key[1], med[0], med[1]))
