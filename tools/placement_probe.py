#!/usr/bin/env python
"""Does the sweep time depend on where the buffers landed?  Several engines with identical inputs are
created in one process (all kept alive, so every one gets different memory) and timed in turn; then the
slowest and the fastest engine exchange one buffer at a time (debug entry prmf_debug_swap) to see which
buffer carries the difference."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

m, n, k, P, xd = syn.CONFIGS["c3"]
dev = "cuda:0"
NE = int(os.environ.get("NE", 6))
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
engs, pads = [], []
for e in range(NE):
    eng = PrmfEngine(m, n, k, x_dtype="bf16")
    eng.set_x(Xd)
    eng.set_library(lib)
    eng.set_active([kk % P for kk in range(k)])
    eng.set_timing(True)
    engs.append(eng)
    pads.append(torch.empty((e + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device=dev))   # perturb the allocator


def timed(eng):
    U = syn.init_factor_torch(torch, dev, 1, m, k, 0, m)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    eng.invalidate_factors()
    eng.step(U, V, 30, 1.0, 1.0)
    torch.cuda.synchronize()
    a, b, _ = eng.timing()
    return a, b


res = []
for rep in range(2):
    for i, eng in enumerate(engs):
        a, b = timed(eng)
        print("rep %d engine %d: xv %.4f ms  xtu %.4f ms" % (rep, i, a, b), flush=True)
        if rep == 1:
            res.append((a, b))
dbg = engs[0].lib.prmf_debug_swap
dbg.restype = C.c_int
names = ["X", "XT", "slabs", "Vp", "Up"]
for side, col in (("xv", 0), ("xtu", 1)):
    order = sorted(range(NE), key=lambda i: res[i][col])
    fast, slow = engs[order[0]], engs[order[-1]]
    print("%s: fastest engine %d (%.4f), slowest engine %d (%.4f)" % (side, order[0], res[order[0]][col], order[-1], res[order[-1]][col]))
    for which, nm in enumerate(names):
        assert dbg(fast.ctx, slow.ctx, which) == 0
        tf, ts = timed(fast)[col], timed(slow)[col]
        print("   swapped %-5s: fast-engine %.4f  slow-engine %.4f" % (nm, tf, ts), flush=True)
        assert dbg(fast.ctx, slow.ctx, which) == 0
