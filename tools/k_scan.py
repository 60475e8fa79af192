"""Iteration time across k (and both storages of X) at one shape: looks for cliffs between the kernel variants
(column tiles NT = ceil(k/32), aligned / element-wise factor loads, 32- vs 64-row wave tiles).
    SHAPE=20000,10000 python tools/k_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n = [int(x) for x in os.environ.get("SHAPE", "20000,10000").split(",")]
P, dev = 50, "cuda:0"
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
for xd in ("bf16", "f32"):
    Xd = syn.x_rows_torch(torch, dev, 0, n, 16, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
    for k in (4, 6, 13, 16, 32, 33, 50, 64, 65, 96, 100, 128):
        eng = PrmfEngine(m, n, k, x_dtype=xd, timing=True)
        eng.set_x(Xd); eng.set_library(lib); eng.set_active([kk % P for kk in range(k)])
        U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
        eng.step(U, V, 5, 1.0, 1.0); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.step(U, V, 30, 1.0, 1.0); torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 30
        a, b, _ = eng.timing()
        print("%-4s k=%3d  iteration %.4f ms  sweeps %.4f + %.4f ms  (%.0f GB/s of X per sweep)" % (
            xd, k, el * 1e3, a, b, m * n * (2 if xd == "bf16" else 4) / (0.5 * (a + b)) / 1e6), flush=True)
        eng.close()
    del Xd
