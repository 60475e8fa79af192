#!/usr/bin/env python
"""Why is X*V slower inside the solve than launched alone?  Times the X*V sweep (events around the launch)
(a) in the full iteration, (b) with only phase_u repeated (sweep + U epilogue + Gram/pathway kernel; the packed V
never changes), (c) with phase_u + phase_q repeated (adds the X^T*U sweep in between)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

m, n, k, P, xd = syn.CONFIGS["c3"]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
eng = PrmfEngine(m, n, k, x_dtype="bf16")
eng.set_x(Xd)
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
eng.set_timing(True)
eng.comm_buffers()


def fresh():
    return syn.init_factor_torch(torch, dev, 1, m, k, 0, m), syn.init_factor_torch(torch, dev, 2, n, k)


for rep in range(2):
    U, V = fresh()
    eng.step(U, V, 30, 1.0, 1.0)
    torch.cuda.synchronize()
    print("(a) full iteration      : xv %.4f  xtu %.4f" % eng.timing()[:2])
    U, V = fresh()
    eng.invalidate_factors()
    eng.begin(U, V, 1.0, 1.0)
    for i in range(30):
        eng.phase_u()
        eng.lib.prmf_debug_bump_cursor(eng.ctx)
    torch.cuda.synchronize()
    try:
        eng.end(30)
    except FloatingPointError:
        pass
    print("(b) phase_u only        : xv %.4f" % eng.timing()[0])
    U, V = fresh()
    eng.invalidate_factors()
    eng.begin(U, V, 1.0, 1.0)
    for i in range(30):
        eng.phase_u()
        eng.phase_q()
        eng.lib.prmf_debug_bump_cursor(eng.ctx)
    torch.cuda.synchronize()
    try:
        eng.end(30)
    except FloatingPointError:
        pass
    print("(c) phase_u + phase_q   : xv %.4f  xtu %.4f" % eng.timing()[:2])
