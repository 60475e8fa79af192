#!/usr/bin/env python
"""Sweep-shape experiments: times the two sweep launches (HIP events inside the library) of a benchmark workload
under several settings of the PRMF_SWEEP_* knobs, one engine per setting in one process.
    WORKLOAD=c5shard python tools/sweep_knobs.py "MT=1" "MT=1,WGS_PER_CU=2" "NW=8" ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c5shard")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
settings = [""] + sys.argv[1:]
for st in settings:
    for key in list(os.environ):
        if key.startswith("PRMF_SWEEP_"):
            del os.environ[key]
    for kv in filter(None, st.split(",")):
        a, b = kv.split("=")
        os.environ["PRMF_SWEEP_" + a] = b
    eng = PrmfEngine(m, n, k, x_dtype=xd, timing=True)
    eng.set_x(Xd)
    eng.set_library(lib)
    eng.set_active([kk % P for kk in range(k)])
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    import time
    nwarm, ntimed = (2, 4) if os.environ.get("SHORT") else (5, 20)   # SHORT: for ablation builds whose arithmetic is wrong
    eng.step(U, V, nwarm, 1.0, 1.0)
    if os.environ.get("SHORT"):
        U = syn.init_factor_torch(torch, dev, 1, m, k)
        V = syn.init_factor_torch(torch, dev, 2, n, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(U, V, ntimed, 1.0, 1.0)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / ntimed
    a, b, _ = eng.timing()
    info = eng.sweep_info()
    eb = 2 if xd == "bf16" else 4
    gbs = lambda ms: (m * n * eb + 4 * k * (m + n)) / ms / 1e6
    print("%-8s %-28s xv %.4f ms (%.0f GB/s) xtu %.4f ms (%.0f GB/s) grid %d/%d slabs %d/%d  iteration %.4f ms" % (
        wl, st or "default", a, gbs(a), b, gbs(b), info.grid_xv, info.grid_xtu, info.splits_xv, info.splits_xtu, el * 1e3), flush=True)
    eng.close()
