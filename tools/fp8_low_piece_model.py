#!/usr/bin/env python
"""VERDICT r5 item 4(b), priced before any kernel is written: what would the factor's LOW piece on the fp8 matrix pipe cost
in accuracy?  The sweeps multiply X (bf16) with the factor split as hi = bf16(v), lo = bf16(v - hi) -- 16-17 mantissa bits,
two bf16 MFMAs per product.  The proposal: X . lo on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4: half the
cycles of bf16 per product; e4m3 operands, one e8m0 scale per 32 elements of the reduction), i.e. lo -> e4m3 with a shared
power-of-two scale per 32-element block, and X -> e4m3 the same way FOR THAT PRODUCT ONLY.

Float64 model of the iteration (the oracle's update_step) with exactly these operand roundings in the two products
P = X V and Q = X^T U, everything else exact, against the exact run from the same start: relative error of U, V after
10 .. 200 iterations (SURVEY Appendix D's experiment; its rows for "bf16 factor operands" and "hi + lo bf16" are reproduced
as a check of the model).  CPU only.
    python tools/fp8_low_piece_model.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
from oracle import prmf_oracle as orc  # noqa: E402


def bf16(a):
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + (((u >> 16) & 1) + 0x7FFF)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def e4m3_blocks(a, axis):
    """e4m3 (4 significant bits, exponents -6 .. 8, subnormals) after a shared power-of-two scale per 32 consecutive elements
    along `axis` (the reduction index), chosen so that the block's largest magnitude lands in the top binade (max 448)."""
    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, -1)
    shp = a.shape
    L = shp[-1]
    pad = (-L) % 32
    b = np.concatenate([a, np.zeros(shp[:-1] + (pad,))], axis=-1).reshape(shp[:-1] + (-1, 32))
    mx = np.abs(b).max(axis=-1, keepdims=True)
    e = np.where(mx > 0, np.floor(np.log2(np.where(mx > 0, mx, 1.0))), 0.0)
    scale = 2.0 ** (8 - e)                                   # top binade of e4m3 is 2^8 (256 .. 448)
    x = b * scale
    ex = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -6)))  # binade; below 2^-6: subnormal spacing 2^-9
    q = 2.0 ** (ex - 3)
    y = np.clip(np.round(x / q) * q, -448, 448) / scale
    y = y.reshape(shp[:-1] + (-1,))[..., :L]
    return np.moveaxis(y, -1, axis)


def product(X, F, mode, Xe):
    """X (rows x L) times F (L x k) with the operand roundings of `mode`; X itself is bf16-exact"""
    if mode == "exact":
        return X.dot(F)
    hi = bf16(F)
    if mode == "bf16 factor":
        return X.dot(hi)
    lo = F - hi
    if mode == "hi+lo bf16":
        return X.dot(hi) + X.dot(bf16(lo))
    if mode == "hi bf16 + lo e4m3 (X bf16)":                   # (not a hardware form: isolates the factor's share)
        return X.dot(hi) + X.dot(e4m3_blocks(lo, 0))
    if mode == "hi bf16 + lo e4m3 x X e4m3":
        return X.dot(hi) + Xe.dot(e4m3_blocks(lo, 0))
    raise ValueError(mode)


def run(case, act, mode, iters, marks):
    X = case["X"]
    Xe_rows, Xe_cols = e4m3_blocks(X, 1), e4m3_blocks(X.T, 1)
    U, V = case["U0"].copy(), case["V0"].copy()
    out = {}
    real_dot = np.ndarray.dot
    for it in range(1, iters + 1):
        # the oracle's update with the two big products replaced (prmf_runner.py:420, :424)
        P = product(X, V, mode, Xe_rows)
        den_u = U.dot(V.T.dot(V)) + U
        U = U * np.where(den_u != 0, P / np.where(den_u != 0, den_u, 1), 1.0)
        Q = product(X.T, U, mode, Xe_cols)
        n, k = V.shape
        ng, dg = np.zeros((n, k)), np.zeros((n, k))
        for kk, pw in enumerate(act):
            v = V[:, kk]
            ng[pw.support, kk] = case["delta"] * (v[pw.support] + 1.0) ** -2
            ng[:, kk] += case["gamma"] * pw.W.dot(v)
            dg[:, kk] = case["gamma"] * (pw.deg * v)
        den = np.maximum(V.dot(U.T.dot(U)) + dg, orc.EPSILON)
        V = np.maximum(V * ((Q + ng) / den), orc.EPSILON)
        if it in marks:
            out[it] = (U.copy(), V.copy())
    return out


def main():
    m, n, k = 1000, 4000, 16
    case = synth.small_case(seed=2, m=m, n=n, k=k, P=8, order=300, bf16_x=True)
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    act = [pws[i % 8] for i in range(k)]
    marks = (10, 50, 100, 200)
    ref = run(case, act, "exact", 200, marks)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    print("m = %d, n = %d, k = %d, bf16-exact X; relative error of U / V against the exact iteration" % (m, n, k))
    for mode in ("bf16 factor", "hi+lo bf16", "hi bf16 + lo e4m3 (X bf16)", "hi bf16 + lo e4m3 x X e4m3"):
        got = run(case, act, mode, 200, marks)
        print("  %-32s " % mode + "  ".join("it %3d: %.1e / %.1e" % (it, rel(got[it][0], ref[it][0]), rel(got[it][1], ref[it][1])) for it in marks))


if __name__ == "__main__":
    main()
