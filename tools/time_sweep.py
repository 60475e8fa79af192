import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = 50000, 20000, 64, 300
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
eng = PrmfEngine(m, n, k, x_dtype="bf16", timing=True)
nsq = eng.set_x(Xd); del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.comm_buffers()
for rep in range(2):
    eng.begin(U, V, 1.0, 1.0)
    for i in range(15):
        eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
    try:
        eng.end(15)
    except FloatingPointError:
        pass
a, b, _ = eng.timing()
info = eng.sweep_info()
print("ms_xv %.4f ms_xtu %.4f grid_xv %d grid_xtu %d" % (a, b, info.grid_xv, info.grid_xtu))
