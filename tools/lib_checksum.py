#!/usr/bin/env python
"""Bitwise fingerprint of a library build: sha256 of U, V and the objective records after STEPS iterations of a benchmark
workload from the seeded initial factors.  Two builds whose change must not touch the arithmetic (a different load
schedule, a seamless segment boundary) print the same line:
    WORKLOAD=c3 PRMF_HIP_LIB=prmf_amd/lib/variants/libprmf_hip_nochain.so python tools/lib_checksum.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

for wl in os.environ.get("WORKLOAD", "c3").split(","):
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
    objs, _ = eng.step(U, V, int(os.environ.get("STEPS", 3)), 1.0, 1.0)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    h.update(U.cpu().numpy().tobytes())
    h.update(V.cpu().numpy().tobytes())
    h.update(np.array([[o[key] for key in ("recon", "manifold", "ignore", "fro", "obj")] for o in objs]).tobytes())
    info = eng.sweep_info()
    print("%-10s %s  obj %.9g  grid %d/%d slabs %d/%d  lib %s" % (wl, h.hexdigest()[:32], objs[-1]["obj"], info.grid_xv, info.grid_xtu,
                                                                 info.splits_xv, info.splits_xtu, os.path.basename(os.environ.get("PRMF_HIP_LIB", "libprmf_hip.so"))), flush=True)
    eng.close()
