import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P, dev = 20000, 20000, 64, 64, "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, 16, 0, m, torch.bfloat16)
for order in (300, 1000, 3000, 10000):
    lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, order, 0))
    eng = PrmfEngine(m, n, k, x_dtype="bf16")
    eng.set_x(Xd); eng.set_library(lib); eng.set_active([kk % P for kk in range(k)])
    U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
    eng.step(U, V, 5, 1.0, 1.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.step(U, V, 30, 1.0, 1.0); torch.cuda.synchronize()
    print("pathway order %5d: iteration %.4f ms" % (order, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    eng.close()
