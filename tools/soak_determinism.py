#!/usr/bin/env python
"""Long run-to-run determinism soak at the C3 shape: REPS times (reset U, V; 3 iterations), every result compared
bitwise (float32 sums of U and V as checksums, full compare on mismatch) with the first one."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

REPS = int(os.environ.get("REPS", 1000))
wl = os.environ.get("WORKLOAD", "c3")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
tdt = torch.bfloat16 if xd == "bf16" else torch.float32
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, tdt)
eng = PrmfEngine(m, n, k, x_dtype=xd)
nsq = eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
normx = math.sqrt(nsq)
U0 = syn.init_factor_torch(torch, dev, 1, m, k, 0, m)
V0 = syn.init_factor_torch(torch, dev, 2, n, k)
U, V = U0.clone(), V0.clone()
first = None
bad = 0
for rep in range(REPS):
    U.copy_(U0)
    V.copy_(V0)
    eng.invalidate_factors()
    objs, _ = eng.step(U, V, 3, normx / k, 10 / normx)
    sig = (U.view(torch.int32).sum(dtype=torch.int64).item(), V.view(torch.int32).sum(dtype=torch.int64).item(), objs[-1]["obj"])
    if first is None:
        first, Uf, Vf = sig, U.clone(), V.clone()
    elif sig != first:
        bad += 1
        du = (U != Uf).sum().item()
        dv = (V != Vf).sum().item()
        print("rep %d differs: %d elements of U, %d of V" % (rep, du, dv), flush=True)
    if rep % 200 == 199:
        print("rep %d ok so far, mismatches %d" % (rep + 1, bad), flush=True)
print("%s: %d repetitions x 3 iterations (%d sweep launches), mismatching repetitions: %d" % (wl, REPS, REPS * 6, bad))
sys.exit(1 if bad else 0)
