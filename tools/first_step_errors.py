#!/usr/bin/env python
"""Which stage of ONE iteration carries how much error, on a small problem where nothing averages out (301 x 1500 f32, k = 8:
the case of tools/tradeoff_digits.py): P = X V, U_new, Q = X^T U_new, V_new and the manifold term of the objective against the
float64 oracle, next to the same stages of an all-float32 NumPy restatement (BLAS sgemm) -- the yardstick of "what any float32
implementation shows".  rms = relative Frobenius error; sup = the same over the entries on the active supports only (the
manifold term reads nothing else).
    python tools/first_step_errors.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import torch  # noqa: E402
from oracle import prmf_oracle as orc  # noqa: E402
from prmf_amd import PathwayLibrary  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

rel = lambda a, b: np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b)
f = np.float32
EPS = f(np.finfo(np.float32).eps)
for m, n, k in ((301, 1500, 8), (3000, 1500, 8), (301, 1500, 64)):
    case = synth.small_case(seed=13, m=m, n=n, k=k, P=6, order=80)
    X = case["X"].astype(f).astype(np.float64)            # the values the GPU holds
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    act = [pws[i % 6] for i in range(k)]
    U0, V0, gamma, delta = case["U0"].astype(f).astype(np.float64), case["V0"].astype(f).astype(np.float64), case["gamma"], case["delta"]
    mask = np.zeros((n, k), dtype=bool)
    for kk, pw in enumerate(act):
        mask[pw.support, kk] = True
    # float64
    P = X.dot(V0)
    U1, V1 = orc.update_step(X, U0, V0, act, gamma, delta)
    Q = X.T.dot(U1)
    man = orc.nmf_manifold_vec_obj(X, U1, V1, act, gamma, delta)["manifold"]
    # all-float32 NumPy
    X32, U32, V32 = X.astype(f), U0.astype(f), V0.astype(f)
    P32 = X32.dot(V32)
    den = U32.dot(V32.T.dot(V32)) + U32
    U32n = U32 * (P32 / den)
    Q32 = X32.T.dot(U32n)
    ng, dg = np.zeros((n, k), f), np.zeros((n, k), f)
    for kk, pw in enumerate(act):
        v = V32[:, kk]
        ng[pw.support, kk] = f(delta) * (v[pw.support] + f(1)) ** f(-2)
        ng[:, kk] += f(gamma) * pw.W.astype(f).dot(v)
        dg[:, kk] = f(gamma) * (pw.deg.astype(f) * v)
    d32 = V32.dot(U32n.T.dot(U32n)) + dg
    d32[d32 < EPS] = EPS
    V32n = V32 * ((Q32 + ng) / d32)
    V32n[V32n < EPS] = EPS
    man32 = orc.nmf_manifold_vec_obj(X, U32n.astype(np.float64), V32n.astype(np.float64), act, gamma, delta)["manifold"]
    # the library, phase by phase
    eng = PrmfEngine(m, n, k, x_dtype="f32")
    eng.set_x(X)
    eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
    eng.set_active(np.arange(k) % 6)
    U = eng.device_tensor(U0, torch.float32)
    V = eng.device_tensor(V0, torch.float32)
    eng.comm_buffers()
    eng.begin(U, V, gamma, delta)
    eng.phase_u()
    Pg = eng.product("xv").cpu().numpy()
    Ug = U.cpu().numpy()
    eng.phase_q()
    Qg = eng.product("xtu").cpu().numpy()
    eng.phase_v()
    eng.phase_post()
    objs, _ = eng.end(1)
    Vg = V.cpu().numpy()
    print("%d x %d, k = %d   (library | all-float32 NumPy)" % (m, n, k))
    for name, g, s, r in (("P = X V", Pg, P32, P), ("U_new", Ug, U32n, U1), ("Q = X^T U_new", Qg, Q32, Q), ("V_new", Vg, V32n, V1)):
        extra = ""
        if r.shape == (n, k):
            extra = "   on the supports: %.2e | %.2e" % (rel(np.asarray(g, np.float64)[mask], r[mask]), rel(np.asarray(s, np.float64)[mask], r[mask]))
        print("   %-14s rms %.2e | %.2e%s" % (name, rel(g, r), rel(s, r), extra))
    # is the error of a product random or a BIAS?  mean signed relative error (a bias of b in Q moves ||v||^2, and with it
    # every normalised quantity of the objective, by 2 b -- random errors of the same rms average out over the column)
    Qx = X.T.dot(Ug.astype(np.float64))                     # exact product of the library's own U_new
    Q32x = X.T.dot(U32n.astype(np.float64))
    nz = Qx > 0
    print("   signed mean relative error:  P %.2e | %.2e    Q (against X^T of the library's own U_new) %.2e, rms %.2e | NumPy %.2e, rms %.2e" % (
        np.mean((Pg - P) / P), np.mean((P32 - P) / P), np.mean((Qg[nz] - Qx[nz]) / Qx[nz]), rel(Qg, Qx),
        np.mean((Q32[nz] - Q32x[nz]) / Q32x[nz]), rel(Q32, Q32x)))
    Vx = V1.copy()
    print("   V_new: signed mean relative error OFF the supports %.2e | %.2e, ON them %.2e | %.2e;  ||v||^2 per column (library): %s" % (
        np.mean(((Vg - Vx) / Vx)[~mask]), np.mean(((V32n - Vx) / Vx)[~mask]), np.mean(((Vg - Vx) / Vx)[mask]), np.mean(((V32n - Vx) / Vx)[mask]),
        " ".join("%.1e" % ((np.sum(Vg[:, kk].astype(np.float64) ** 2) - np.sum(Vx[:, kk] ** 2)) / np.sum(Vx[:, kk] ** 2)) for kk in range(min(k, 8)))))
    print("   manifold       rel %.2e | %.2e" % (abs(objs[0]["manifold"] - man) / man, abs(man32 - man) / man))
    # the float64 evaluation of the manifold term on the LIBRARY's V_new: separates the state from the evaluation
    og = orc.nmf_manifold_vec_obj(X, Ug.astype(np.float64), Vg.astype(np.float64), act, gamma, delta)
    print("   manifold of the library's V_new evaluated by the oracle: rel to exact %.2e; the library's record against THAT: %.2e  (ignore %.2e, fro %.2e, recon %.2e)" % (
        abs(og["manifold"] - man) / man, abs(objs[0]["manifold"] - og["manifold"]) / og["manifold"], abs(objs[0]["ignore"] - og["ignore"]) / og["ignore"],
        abs(objs[0]["fro"] - og["fro"]) / og["fro"], abs(objs[0]["recon"] - og["recon"]) / og["recon"]))
    # per column: which columns carry the difference
    per = []
    for kk, pw in enumerate(act):
        vh = Vg[:, kk].astype(np.float64) / np.linalg.norm(Vg[:, kk].astype(np.float64))
        ve = V1[:, kk] / np.linalg.norm(V1[:, kk])
        Ln = pw.L_normal()
        per.append((Ln.dot(vh).dot(vh), Ln.dot(ve).dot(ve)))
    print("   per-column manifold, library state vs exact state (float64 evaluation): " + " ".join("%.1e" % (abs(a - b) / b) for a, b in per[:8]))
    # what the manifold of the float64 V_new rounded to float32 shows (state rounding alone)
    man_r = orc.nmf_manifold_vec_obj(X, U1, V1.astype(f).astype(np.float64), act, gamma, delta)["manifold"]
    print("   manifold of the float64 V_new rounded to float32: rel %.2e" % (abs(man_r - man) / man))
    eng.close()
