#!/usr/bin/env python
"""What does the systematic finish-time spread of the sweep's equal shares go with?  

Per workgroup of several launches of both sweeps: duration (debug stamps), WHERE it ran (XCC_ID, SE / SH / CU / SIMD from
HW_ID), which SLOT of its CU it took (first or second workgroup), and WHICH BYTES it streamed (first row / first chunk of
its share -> address bits of the first line, a proxy for the HBM stack / channel interleave).  Prints, for each candidate
factor, the spread of the per-group mean durations and the share of the per-workgroup variance the factor explains
(between-group variance / total variance of the launch-averaged durations).

    SHAPE=5000,10000,16,50 XDTYPE=f32 python tools/wg_where.py        (C2)
    SHAPE=6250,20000,64,300 python tools/wg_where.py                  (one rank's share of C3 on 8 GPUs)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine

m, n, k, P = [int(x) for x in os.environ.get("SHAPE", "6250,20000,64,300").split(",")]
XD = os.environ.get("XDTYPE", "bf16")
REPS = int(os.environ.get("REPS", 6))
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if XD == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=XD, eager_objective=True)
eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k)
V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.comm_buffers()
eng.begin(U, V, 1.0, 1.0)
for i in range(3):
    eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
torch.cuda.synchronize()
_w, _ph, _mx = C.c_int(), C.c_int(), C.c_int()
st = torch.zeros(eng.lib.prmf_debug_stamp_layout(C.byref(_w), C.byref(_ph), C.byref(_mx)), dtype=torch.int64, device=dev)
WHERE = _w.value
info = eng.sweep_info()


def collect(which):
    durs, hw = [], []
    for rep in range(REPS):
        if which == "xv":
            st.zero_(); torch.cuda.synchronize()
            eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel())
            eng.phase_u(); torch.cuda.synchronize()
            eng.lib.prmf_debug_set_stamps(None, 0)
            a = st.cpu().numpy()
            eng.phase_q(); eng.phase_v(); eng.phase_post(); torch.cuda.synchronize()
        else:
            eng.phase_u(); torch.cuda.synchronize()
            st.zero_(); torch.cuda.synchronize()
            eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel())
            eng.phase_q(); torch.cuda.synchronize()
            eng.lib.prmf_debug_set_stamps(None, 0)
            a = st.cpu().numpy()
            eng.phase_v(); eng.phase_post(); torch.cuda.synchronize()
        g = info.grid_xv if which == "xv" else info.grid_xtu
        se = a[:2 * g].reshape(-1, 2)
        durs.append((se[:, 1] - se[:, 0]) / 100.0)
        hw.append(a[WHERE:WHERE + g].copy())
    return np.array(durs), np.array(hw)


def explain(name, groups, mean_d):
    tot = mean_d.var()
    keys = np.unique(groups)
    gm = np.array([mean_d[groups == q].mean() for q in keys])
    cnt = np.array([(groups == q).sum() for q in keys])
    between = float(((gm - mean_d.mean()) ** 2 * cnt).sum() / cnt.sum())
    print("   %-34s %3d groups, group means %.1f .. %.1f us, explains %.0f %% of the per-workgroup variance"
          % (name, len(keys), gm.min(), gm.max(), 100.0 * between / max(tot, 1e-12)))


for which in ("xv", "xtu"):
    d, hw = collect(which)
    g = d.shape[1]
    mean_d = d.mean(axis=0)
    print("== %s: %d workgroups, launch %.1f us (slowest workgroup), mean duration %.1f, std over workgroups %.2f us; "
          "launch-to-launch correlation %.2f" % (which, g, d.max(axis=1).mean(), mean_d.mean(), mean_d.std(),
                                                  np.corrcoef(d[0], d[-1])[0, 1]))
    h = hw[0]
    xcc = (h >> 32) & 0xF
    simd = (h >> 4) & 3
    cu = (h >> 8) & 0xF
    sh = (h >> 12) & 1
    se = (h >> 13) & 7
    same_place = float((hw == hw[0]).all(axis=0).mean())
    print("   placement identical in all %d launches for %.0f %% of the workgroups; blockIdx %% 8 == XCC for %.0f %%"
          % (REPS, 100 * same_place, 100 * float((xcc == (np.arange(g) % 8)).mean())))
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    order = np.argsort(np.arange(g), kind="stable")
    first = np.ones(g, dtype=bool)
    seen = set()
    for b in order:                                         # the older workgroup of a CU = the lower block index
        first[b] = cuid[b] not in seen
        seen.add(cuid[b])
    explain("XCC", xcc, mean_d)
    explain("XCC x SE", xcc * 8 + se, mean_d)
    explain("CU (XCC, SE, SH, CU)", cuid, mean_d)
    explain("first / second workgroup of its CU", first.astype(int), mean_d)
    explain("block index // 64 (position in the grid)", np.arange(g) // 64, mean_d)
    # the bytes: a workgroup's share starts at unit b * T / G of the (panel, unit) sequence (unsliced plans)
    rows_pp = 32 * (2 if not (XD == "f32" and (k <= 32 or k > 64)) else 1) * (8 if False else 4)
    print("   ten slowest workgroups (block, us, xcc, se, sh, cu, simd):",
          [(int(b), round(float(mean_d[b]), 1), int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b]), int(simd[b])) for b in np.argsort(-mean_d)[:10]])
    print("   ten fastest:", [(int(b), round(float(mean_d[b]), 1), int(xcc[b]), int(se[b]), int(cu[b])) for b in np.argsort(mean_d)[:10]])
    np.save(os.path.join(os.environ.get("OUT", "/tmp"), "wg_where_%s.npy" % which), np.vstack([mean_d, h.astype(np.float64)]))
