#!/usr/bin/env python
"""Time of prmf_score_candidates (k x P match scores of `restrict`, prmf_runner.py:115-127, 129-194)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

for (n, k, P) in ((20000, 64, 300), (25000, 128, 1000)):
    eng = PrmfEngine(256, n, k, x_dtype="bf16")
    lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
    eng.set_library(lib)
    V = syn.init_factor_torch(torch, "cuda:0", 2, n, k)
    eng.score_candidates(V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sc = eng.score_candidates(V)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    nnz = int(lib.w.shape[0])
    sup = int(lib.sup_idx.shape[0])
    alg = nnz * 12 + sup * 28 + k * sup * 4 + k * P * 8          # CSR + support arrays once, one V value per (support node, latent), scores
    print("n=%d k=%d P=%d: %.3f ms per call (incl. the k*P float64 download); CSR nnz %d, support %d; algorithmic %.1f MB -> %.1f GB/s"
          % (n, k, P, dt * 1e3, nnz, sup, alg / 1e6, alg / dt / 1e9))
    eng.close()
