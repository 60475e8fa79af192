"""Short reductions take the EXACT product (round 6; prmf_amd/csrc/gemm.hip: exact_product_kernel).

The streaming sweep multiplies with the factor packed to hi + lo bf16 (16-17 mantissa bits).  Over a long reduction that
rounding averages out; over a short one -- X^T U of the reference's own test problems sums 100 samples (script/prmf_runner.py:424
on test/script/PRMF's data) -- it does not: X is positive, every gene of a latent column sums the SAME rounded factor entries and
gets the same relative shift.  A short reduction (padded length x 32-column tiles <= 512: 512 samples at k <= 32, 128 at k = 128) therefore runs on the float32 MFMA (exact products, a fused
multiply-add chain) against the unpacked factor.  These tests hold: which products take it, that its results carry no such
shift (the streaming form, forced by PRMF_SWEEP_NO_EXACT=1, shows it), and what it does to parity on the reference's 100 x 1000
problem -- the golden traces of gitter-lab/prmf itself."""
import os

import numpy as np
import pytest

import synth
from measure import rel, relmax, report

pytestmark = pytest.mark.gpu


def _products(case, x_dtype, no_exact):
    """(P = X V, Q = X^T U_new, flags) of one iteration through the phase calls"""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    old = os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
    if no_exact:
        os.environ["PRMF_SWEEP_NO_EXACT"] = "1"
    try:
        m, n = case["X"].shape
        eng = PrmfEngine(m, n, case["k"], x_dtype=x_dtype)
    finally:
        os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
        if old is not None:
            os.environ["PRMF_SWEEP_NO_EXACT"] = old
    try:
        eng.set_x(case["X"])
        eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
        eng.set_active(np.arange(case["k"]) % len(case["pathways"]))
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        eng.comm_buffers()
        eng.begin(U, V, case["gamma"], case["delta"])
        eng.phase_u()
        P = eng.product("xv").cpu().numpy().astype(np.float64)
        Un = U.cpu().numpy().astype(np.float64)
        eng.phase_q()
        Q = eng.product("xtu").cpu().numpy().astype(np.float64)
        eng.phase_v()
        eng.phase_post()
        objs, _ = eng.end(1)
        return P, Un, Q, int(eng.sweep_info().reserved), objs[0], U.cpu().numpy(), V.cpu().numpy()
    finally:
        eng.close()


@pytest.mark.parametrize("m,n,k,x_dtype", [(100, 1000, 6, "f32"),      # the reference's test problem: X^T U sums 100 samples
                                           (301, 1500, 8, "f32"),      # X^T U short (301 samples), X V streams (1500 genes)
                                           (128, 150, 70, "bf16"),     # three column tiles, bf16 storage, both short (128 x 3 <= 512)
                                           (500, 20000, 6, "f32"),     # the reference's typical use: 500 samples at the default k
                                           (128, 100, 128, "f32")])    # at the limit for k = 128: 128 samples, both short
def test_short_reductions_are_exact_and_unbiased(m, n, k, x_dtype):
    case = synth.small_case(seed=21, m=m, n=n, k=k, P=6, order=60, bf16_x=(x_dtype == "bf16"))
    if x_dtype == "f32":
        case["X"] = case["X"].astype(np.float32).astype(np.float64)            # the values the GPU holds
    X, V0 = case["X"], case["V0"].astype(np.float32).astype(np.float64)
    got = {}
    for no_exact in (False, True):
        P, Un, Q, flags, od, U, V = _products(case, x_dtype, no_exact)
        Pr, Qr = X.dot(V0), X.T.dot(Un)                              # float64 products of what the library multiplied
        nz = Qr > 0
        got[no_exact] = dict(flags=flags, rms_P=rel(P, Pr), rms_Q=rel(Q, Qr), bias_Q=float(np.mean((Q[nz] - Qr[nz]) / Qr[nz])),
                             col_bias_Q=float(np.max(np.abs([np.mean(((Q - Qr) / np.where(nz, Qr, 1.0))[:, c][nz[:, c]]) for c in range(k)]))),
                             U=U, V=V, obj=od)
    nt = (k + 31) // 32
    pad = lambda d: (d + 127) // 128 * 128
    short = lambda d, rows: pad(d) * nt <= 512 or (nt == 1 and pad(d) <= 1024 and rows <= 4096)
    want = (1 if short(n, m) else 0) | (2 if short(m, n) else 0)
    assert want & 2
    assert got[False]["flags"] == want and got[True]["flags"] == 0
    e, s = got[False], got[True]
    report("exact_short_reduction[%dx%d-k%d-%s]" % (m, n, k, x_dtype), rms_Q_exact=e["rms_Q"], rms_Q_stream=s["rms_Q"],
           column_shift_exact=e["col_bias_Q"], column_shift_stream=s["col_bias_Q"], rms_P_exact=e["rms_P"])
    # float32 fused multiply-add chains of 32 products, added chunk by chunk: random rounding only (6-9e-8 measured) -- and no
    # column carries a common shift (2-7e-9 measured; the streaming form on the same problems: 3e-7 .. 9e-7)
    assert e["rms_Q"] < 2e-7 and (e["rms_P"] < 2e-7 or not want & 1)
    assert e["col_bias_Q"] < 2e-8, e["col_bias_Q"]
    assert s["col_bias_Q"] > 10 * e["col_bias_Q"]
    # and both are the same iteration to the bar
    assert rel(e["U"], s["U"]) < 1e-5 and rel(e["V"], s["V"]) < 1e-5
    for key in ("recon", "manifold", "ignore", "fro", "obj"):
        assert abs(e["obj"][key] - s["obj"][key]) <= 1e-5 * abs(s["obj"][key])


@pytest.mark.parametrize("name,kw,x_dtype", [("g2_inner25.npz", {}, "f32"), ("g9_bf16x.npz", {"bf16_x": True}, "bf16")])
def test_reference_goldens_with_exact_products(golden_dir, name, kw, x_dtype):
    """The reference's own inner loop on its small problem (60 samples x 400 genes, 25 steps, generated by the unmodified
    reference): with both products exact the float32 state follows the float64 reference to a few 1e-7 -- the streaming form
    (measured in the same test) to ~2e-6.  The bar itself (1e-4) is asserted in test_gpu_parity.py."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    g = np.load(os.path.join(golden_dir, name))
    case = synth.small_case(**({"seed": 1} | kw))
    m, n = case["X"].shape
    steps = int(g["snap_steps"][-1])
    out = {}
    for no_exact in (False, True):
        if no_exact:
            os.environ["PRMF_SWEEP_NO_EXACT"] = "1"
        try:
            eng = PrmfEngine(m, n, case["k"], x_dtype=x_dtype)
        finally:
            os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
        try:
            eng.set_x(case["X"])
            eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
            eng.set_active(np.arange(case["k"]))
            U = eng.device_tensor(case["U0"], torch.float32)
            V = eng.device_tensor(case["V0"], torch.float32)
            objs, _ = eng.step(U, V, steps, case["gamma"], case["delta"])
            eo = relmax([objs[-1][key] for key in ("recon", "manifold", "ignore", "fro", "obj")], g["obj"][steps - 1][[0, 1, 2, 3, 6]])
            out[no_exact] = (rel(U.cpu().numpy(), g["U"][-1]), rel(V.cpu().numpy(), g["V"][-1]), eo)
        finally:
            eng.close()
    report("reference_golden_exact_vs_stream[%s-%s]" % (name, x_dtype), steps=steps, rel_U=out[False][0], rel_V=out[False][1], rel_obj=out[False][2],
           rel_U_stream=out[True][0], rel_V_stream=out[True][1], rel_obj_stream=out[True][2])
    # measured: U 2.9-3.6e-7, V 3.2-4.0e-7, objective 1.3-6.7e-7 (streaming form: 1.0-1.8e-6, 1.8-2.1e-6, 3.5-5.1e-6)
    assert out[False][0] < 1e-6 and out[False][1] < 1e-6 and out[False][2] < 2e-6
    assert out[False][1] < out[True][1] and out[False][2] < out[True][2]
