#!/usr/bin/env python
"""Does the sweep time depend on where the buffers landed?  Several engines with identical inputs are
created in one process (all kept alive, so every one gets different memory) and timed in turn, twice."""
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
engs = []
for e in range(NE):
    eng = PrmfEngine(m, n, k, x_dtype="bf16")
    eng.set_x(Xd)
    eng.set_library(lib)
    eng.set_active([kk % P for kk in range(k)])
    eng.set_timing(True)
    engs.append(eng)
    pad = torch.empty((e + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device=dev)   # perturb the allocator between engines
    engs.append(pad)
for rep in range(2):
    for i, eng in enumerate(engs[::2]):
        U = syn.init_factor_torch(torch, dev, 1, m, k, 0, m)
        V = syn.init_factor_torch(torch, dev, 2, n, k)
        eng.step(U, V, 30, 1.0, 1.0)
        torch.cuda.synchronize()
        a, b, _ = eng.timing()
        print("rep %d engine %d: xv %.4f ms  xtu %.4f ms" % (rep, i, a, b), flush=True)
