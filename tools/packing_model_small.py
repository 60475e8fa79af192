#!/usr/bin/env python
"""How much of the library's error on the reference's own small problem (100 samples x 1000 genes, k = 6: test/script/PRMF) is the
16-17-bit hi/lo packing of the factors the streaming sweeps multiply with?  Float64 model of the whole solve (the oracle's
nmf_pathway) with (a) nothing rounded, (b) the state rounded to float32 after every update, (c) additionally V and / or U
rounded to hi + lo bf16 inside the two big products -- everything else exact.  The last row is the library as it was through
round 5 (measured on the GPU: trace 1.5e-6, U 1.2e-6, V 1.7e-6): the packing IS its error there; the float32 state alone costs
1e-8.  Round 6 multiplies short reductions exactly (csrc/gemm.hip: exact_product_kernel).  CPU only.
    python tools/packing_model_small.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import test_oracle_golden as tg  # noqa: E402
from oracle import prmf_oracle as orc  # noqa: E402

f = np.float32


def bf16(a):
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + (((u >> 16) & 1) + 0x7FFF)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def pack(F):
    hi = bf16(F)
    return hi + bf16(F - hi)


def make(pack_u, pack_v):
    def step(X, U, V, active, gamma, delta):
        n, k = V.shape
        num_u = X.dot(pack(V) if pack_v else V)
        den_u = U.dot(V.T.dot(V)) + U
        ratio = np.ones_like(num_u)
        np.divide(num_u, den_u, out=ratio, where=den_u != 0)
        U = (U * ratio).astype(f).astype(np.float64)
        num_recon = X.T.dot(pack(U) if pack_u else U)
        ng, dg = np.zeros((n, k)), np.zeros((n, k))
        for kk, pw in enumerate(active):
            v = V[:, kk]
            ng[pw.support, kk] = delta * (v[pw.support] + 1.0) ** -2
            ng[:, kk] += gamma * pw.W.dot(v)
            dg[:, kk] = gamma * (pw.deg * v)
        den = np.maximum(V.dot(U.T.dot(U)) + dg, orc.EPSILON)
        V = np.maximum(V * ((num_recon + ng) / den), orc.EPSILON).astype(f).astype(np.float64)
        return U, V
    return step


c = synth.cli_case(tempfile.mkdtemp(), variant=1)
pws = tg._graphs_to_pathways(c)
m, n = c["X"].shape
exact = orc.update_step
runs = {}
for name, st in (("exact (float64)", exact), ("float32 state only", make(False, False)), ("+ V packed (X V, 1000 genes)", make(False, True)),
                 ("+ U packed (X^T U, 100 samples)", make(True, False)), ("+ both packed = the streaming sweeps", make(True, True))):
    orc.update_step = st
    np.random.seed(1)
    U0 = 3 * (1 - np.random.rand(m, 6))
    V0 = 3 * (1 - np.random.rand(n, 6))
    tr = []
    U, V, od = orc.nmf_pathway(c["X"], pws, gamma=1.0, k_latent=6, U_init=U0, V_init=V0, trace=tr)
    runs[name] = (np.array([t["obj"] for t in tr]), U, V)
orc.update_step = exact
t0, Ue, Ve = runs["exact (float64)"]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("%d x %d, k = 6, the reference's test problem, whole solve (%d iterations): drift against the exact run" % (m, n, len(t0)))
for name, (t1, U1, V1) in runs.items():
    L = min(len(t0), len(t1))
    print("  %-40s objective trace %.1e   U %.1e   V %.1e" % (name, np.max(np.abs(t1[:L] - t0[:L]) / np.abs(t0[:L])), rel(U1, Ue), rel(V1, Ve)))
