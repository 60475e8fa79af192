#!/usr/bin/env python
"""Sweep time vs byte offset of the X / X^T copies inside their (over-sized) allocations, one engine,
nothing re-allocated between trials (debug entry prmf_debug_set_offsets + prmf_set_x again)."""
import ctypes as C
import os
import sys

os.environ.setdefault("PRMF_PLACEMENT_SLACK_MB", "64")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

m, n, k, P, xd = syn.CONFIGS["c3"]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
eng = PrmfEngine(m, n, k, x_dtype="bf16")
eng.set_x(Xd)
eng.set_library(lib)
eng.set_active([kk % P for kk in range(k)])
eng.set_timing(True)
setoff = eng.lib.prmf_debug_set_offsets
setoff.restype = C.c_int
setoff.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
KB, MB = 1024, 1024 * 1024
offs = [0, 256, 4 * KB, 64 * KB, 512 * KB, MB, 2 * MB, 3 * MB, 4 * MB, 6 * MB, 8 * MB, 16 * MB, 24 * MB, 32 * MB, 48 * MB, 64 * MB, 0, 2 * MB]
for off in offs:
    assert setoff(eng.ctx, off, off) == 0
    eng.set_x(Xd)
    U = syn.init_factor_torch(torch, dev, 1, m, k, 0, m)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    eng.invalidate_factors()
    objs, _ = eng.step(U, V, 30, 1.0, 1.0)
    torch.cuda.synchronize()
    a, b, _ = eng.timing()
    print("offset %9d: xv %.4f ms  xtu %.4f ms   obj %.6e" % (off, a, b, objs[-1]["obj"]), flush=True)
