"""Iteration / sweep times across shapes (m samples x n genes) for (k, storage) pairs: looks for planner cliffs.
    [PAIRS=64:bf16,6:f32] python tools/shape_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
dev, P = "cuda:0", 50
PAIRS = [(int(a.split(":")[0]), a.split(":")[1]) for a in os.environ.get("PAIRS", "64:bf16,6:f32").split(",")]
for k, xd in PAIRS:
    for n in (2000, 20000, 60000):
        lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
        for m in (100, 500, 2000, 8000, 30000, 100000):
            if m * n > 3.2e9:
                continue
            Xd = syn.x_rows_torch(torch, dev, 0, n, 16, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
            eng = PrmfEngine(m, n, k, x_dtype=xd, timing=True)
            eng.set_x(Xd); eng.set_library(lib); eng.set_active([kk % P for kk in range(k)])
            del Xd
            U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
            eng.step(U, V, 5, 1.0, 1.0); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.step(U, V, 30, 1.0, 1.0); torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 30
            a, b, _ = eng.timing(); info = eng.sweep_info()
            eb = 2 if xd == "bf16" else 4
            print("%-4s k=%2d  %6d x %5d (%7.1f MB)  iteration %.4f ms  sweeps %.4f + %.4f ms (%5.0f / %5.0f GB/s)  grid %d/%d slabs %d/%d" % (
                xd, k, m, n, m * n * eb / 1e6, el * 1e3, a, b, m * n * eb / a / 1e6, m * n * eb / b / 1e6,
                info.grid_xv, info.grid_xtu, info.splits_xv, info.splits_xtu), flush=True)
            eng.close()
