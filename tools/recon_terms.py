"""Where does the error of the Gram-identity recon^2 come from?  The three terms of record `it` (prmf_debug_recon_terms)
against float64: <X^T U, V>, <U^T U, V^T V> (and U^T U, V^T V entry-wise), each relative to ||X||^2.
    python tools/recon_terms.py c3 c2"""
import ctypes as C
import math
import sys

import numpy as np
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
    steps = 10
    objs, _ = eng.step(U, V, steps, nx / k, 10.0 / nx)
    cross = C.c_double()
    gu, gv = np.zeros((k, k)), np.zeros((k, k))
    rc = eng.lib.prmf_debug_recon_terms(eng.ctx, steps - 1, C.byref(cross), C.c_void_p(gu.ctypes.data), C.c_void_p(gv.ctypes.data))
    assert rc == 0, rc
    U64, V64 = U.double(), V.double()
    Q = torch.zeros((n, k), dtype=torch.float64, device=dev)
    acc = 0.0
    for r0 in range(0, m, 4096):
        Xc = Xd[r0:r0 + 4096].double()
        Q += Xc.T @ U64[r0:r0 + 4096]
        R = Xc - U64[r0:r0 + 4096] @ V64.T
        acc += float((R * R).sum())
    # note: the library's <Q, V> pairs the Q of this iteration (from U_new) with V_new: the same U, V as here
    cross64 = float((Q * V64).sum())
    gu64, gv64 = (U64.T @ U64).cpu().numpy(), (V64.T @ V64).cpu().numpy()
    gg, gg64 = float((gu * gv).sum()), float((gu64 * gv64).sum())
    ident = nsq - 2 * cross.value + gg
    print("%-8s recon/||X|| %.4f | recon^2: identity err %.2e ||X||^2 (record err on recon %.2e) | cross err %.2e | <Gu,Gv> err %.2e | max entry err Gu %.2e Gv %.2e"
          % (wl, math.sqrt(acc) / nx, (ident - acc) / nsq, (math.sqrt(max(ident, 0)) - math.sqrt(acc)) / math.sqrt(acc),
             2 * (cross.value - cross64) / nsq, (gg - gg64) / nsq, np.abs(gu - gu64).max() / np.abs(gu64).max(), np.abs(gv - gv64).max() / np.abs(gv64).max()))
    eng.close()
    del Xd
