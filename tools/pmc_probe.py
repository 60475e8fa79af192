#!/usr/bin/env python
"""The kernels of one workload and nothing else, for rocprofv3 counter passes (tools/pmc_study.sh): ITER iterations through
the phase calls plus the pure-stream calibration of X and X^T (stream_read_kernel) -- so that the sweep and a plain stream of
the same bytes can be compared counter by counter in one run.
    WORKLOAD=c3 ITERS=3 python3 tools/pmc_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c3")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=xd)
eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k)
V = syn.init_factor_torch(torch, dev, 2, n, k)
iters = int(os.environ.get("ITERS", 3))
eng.step(U, V, iters, 1.0, 1.0)
torch.cuda.synchronize()
eng.stream_read("x", iters)
eng.stream_read("xt", iters)
torch.cuda.synchronize()
eng.close()
print("probe done", wl, iters)
