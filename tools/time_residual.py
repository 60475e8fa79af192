"""Device time of the direct-residual sweep (prmf_residual_sq, prmf_amd/csrc/residual.hip) next to one iteration of
the same workload:  python tools/time_residual.py c3 [c2 c3f32 c5shard ...]"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine

for wl in sys.argv[1:] or ["c3"]:
    m, n, k, P, xd = syn.CONFIGS[wl]
    dev = "cuda:0"
    Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
    eng = PrmfEngine(m, n, k, x_dtype=xd)
    nsq = eng.set_x(Xd)
    eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
    eng.set_active([kk % P for kk in range(k)])
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    nx = math.sqrt(nsq)
    eng.step(U, V, 10, nx / k, 10.0 / nx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(U, V, 50, nx / k, 10.0 / nx)
    torch.cuda.synchronize()
    it_ms = (time.perf_counter() - t0) / 50 * 1e3
    eng.residual_sq(U, V)
    t0 = time.perf_counter()
    for _ in range(20):
        r = eng.residual_sq(U, V)
    res_ms = (time.perf_counter() - t0) / 20 * 1e3
    acc = 0.0
    for r0 in range(0, m, 4096):
        R = Xd[r0:r0 + 4096].double() - U[r0:r0 + 4096].double() @ V.double().T
        acc += float((R * R).sum())
    print("%-10s iteration %.3f ms | direct residual %.3f ms per call (incl. sync) = %.2f iterations | recon/||X|| %.4f  rel err of recon %.2e"
          % (wl, it_ms, res_ms, res_ms / it_ms, math.sqrt(acc) / nx, abs(math.sqrt(r) - math.sqrt(acc)) / math.sqrt(acc)))
    eng.close()
    del Xd
