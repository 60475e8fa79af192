#!/usr/bin/env python
"""Where the tradeoff feedback (gamma = delta = (1 - t) recon / (t manifold), prmf_runner.py:542-548) loses digits: the case of
tests/test_gpu_eightway.py::test_tradeoff_feedback_on_eight_ranks (301 x 1500 f32, k = 8, t = 0.4, six steps; gamma runs from
12 to 2e33) through the single-rank step, every record component of every step against the float64 oracle -- and against the
oracle with nothing but its STATE rounded to float32 (the yardstick: 6e-7 on the final gamma).
    python tools/tradeoff_digits.py"""
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

for m, steps, round_x in ((301, 6, False), (301, 6, True), (5, 3, False)):
    case = synth.small_case(seed=13, m=m, n=1500, k=8, P=6, order=80)
    X = case["X"].astype(np.float32).astype(np.float64) if round_x else case["X"]
    pws = [orc.pathway_from_edges(1500, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    act = [pws[i % 6] for i in range(8)]
    tr = []
    Ur, Vr, od, g_ref, d_ref = orc.nmf_manifold_vec_update_tradeoff(X, case["U0"], case["V0"], act, n_steps=steps, tradeoff=0.4,
                                                                   gamma=case["gamma"], delta=case["delta"], trace=tr)
    eng = PrmfEngine(m, 1500, 8, x_dtype="f32")
    eng.set_x(X)
    eng.set_library(PathwayLibrary.from_edge_lists(1500, case["pathways"]))
    eng.set_active(np.arange(8) % 6)
    U = eng.device_tensor(case["U0"], torch.float32)
    V = eng.device_tensor(case["V0"], torch.float32)
    objs, gd = eng.step(U, V, steps, case["gamma"], case["delta"], 0.4)
    print("m = %d, X %s: final gamma %.6e (oracle %.6e) rel %.2e | U %.2e V %.2e" % (
        m, "rounded to float32 for the oracle too" if round_x else "float64 for the oracle, float32 on the GPU", gd[0], g_ref,
        abs(gd[0] - g_ref) / abs(g_ref), np.linalg.norm(U.cpu().numpy() - Ur) / np.linalg.norm(Ur), np.linalg.norm(V.cpu().numpy() - Vr) / np.linalg.norm(Vr)))
    for s, (a, b) in enumerate(zip(objs, tr)):
        print("   step %d  gamma %.4e | rel: recon %.1e manifold %.1e ignore %.1e fro %.1e gamma(in) %.1e obj %.1e  flags %d" % (
            s, b["gamma"], *[abs(a[key] - b[key]) / abs(b[key]) for key in ("recon", "manifold", "ignore", "fro", "gamma", "obj")], eng.last_flags[s]))
    eng.close()
