#!/usr/bin/env python
"""The exact short-reduction product against the streaming sweep on the same shapes, in one session (PRMF_SWEEP_NO_EXACT=1 forces
the streaming form): device time per iteration and of the two products.
    python tools/ab_exact_product.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

SHAPES = [(500, 20000, 6, "f32"), (100, 1000, 6, "f32"), (500, 20000, 64, "bf16"), (512, 20000, 128, "bf16"), (300, 9000, 20, "bf16"),
          (500, 500, 16, "f32"), (20000, 400, 64, "bf16"), (13, 1000, 6, "f32"), (1000, 20000, 64, "bf16"), (1000, 20000, 6, "f32"),
          (1000, 20000, 32, "bf16"), (250, 20000, 64, "bf16")]
for m, n, k, xd in SHAPES:
    row = []
    for no_exact in (True, False, True, False):
        if no_exact:
            os.environ["PRMF_SWEEP_NO_EXACT"] = "1"
        else:
            os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
        eng = PrmfEngine(m, n, k, x_dtype=xd)
        os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
        Xd = syn.x_rows_torch(torch, "cuda:0", 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
        eng.set_x(Xd)
        P = min(50, max(2, n // 400))
        eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, min(300, n // 4), 0)))
        eng.set_active([kk % P for kk in range(k)])
        U = syn.init_factor_torch(torch, "cuda:0", 1, m, k)
        V = syn.init_factor_torch(torch, "cuda:0", 2, n, k)
        eng.step(U, V, 20, 1.0, 1.0)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.step(U, V, 100, 1.0, 1.0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 100)
        eng.set_timing(True)
        eng.step(U, V, 20, 1.0, 1.0)
        a, b, _ = eng.timing()
        row.append((sorted(ts)[1] * 1e6, a * 1e3, b * 1e3, int(eng.sweep_info().reserved)))
        eng.close()
    f = lambda r: "%6.1f us/it (X V %5.1f, X^T U %5.1f) [exact bits %d]" % r
    print("%6d x %6d k=%3d %s | stream %s | exact %s | again: stream %.1f exact %.1f" % (m, n, k, xd, f(row[0]), f(row[1]), row[2][0], row[3][0]), flush=True)
