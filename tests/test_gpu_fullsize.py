"""Parity at the sizes BASELINE.json names.

* C2 (5000 x 10000, k = 16, P = 50, f32) runs the oracle itself at full size: U, V and all five
  objective components of every step against oracle.nmf_manifold_vec_update at the 1e-4 bar.
* C3 (50000 x 20000 k = 64 bf16), the same shape with f32 X, and one rank's share of C5
  (25000 x 25000 k = 128 bf16) are too large for the NumPy oracle to finish in seconds, so ONE
  iteration is checked array by array -- X V, U_new, X^T U_new, V_new -- against a float64
  restatement of oracle.update_step (script/prmf_runner.py:420-444) written
  with torch on the GPU in this file (float64 matmuls in row chunks; not the HIP path).  Besides the
  Frobenius error of each array, the worst element of EVERY 32-row tile is bounded, so that a single
  wrong MFMA tile (the failure class the sweep once had at this size only) cannot hide in a norm.
"""
import math

import numpy as np
import pytest

from measure import rel, relmax, report

pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)
OBJ_KEYS = ("recon", "manifold", "ignore", "fro", "obj")


def _engine(torch, workload, x_dtype, seed):
    from prmf_amd import PathwayLibrary, synthetic as syn
    from prmf_amd.solver import PrmfEngine
    m, n, k, P = syn.CONFIGS[workload][:4]
    dev = "cuda:0"
    Xd = syn.x_rows_torch(torch, dev, seed, n, k, 0, m, torch.bfloat16 if x_dtype == "bf16" else torch.float32)
    eng = PrmfEngine(m, n, k, x_dtype=x_dtype)
    nsq = eng.set_x(Xd)
    pathways = syn.tree_pathways(n, P, 300, seed)
    eng.set_library(PathwayLibrary.from_edge_lists(n, pathways))
    active = [kk % P for kk in range(k)]
    eng.set_active(active)
    normx = math.sqrt(nsq)
    U = syn.init_factor_torch(torch, dev, seed + 1, m, k)
    V = syn.init_factor_torch(torch, dev, seed + 2, n, k)
    return eng, Xd, U, V, pathways, active, normx / k, 10.0 / normx


def test_c2_full_size_vs_oracle():
    """BASELINE.json configs[1] at full size against the oracle, 6 steps."""
    import torch
    from oracle import prmf_oracle as orc
    eng, Xd, U, V, pathways, active, gamma, delta = _engine(torch, "c2", "f32", 3)
    n, k = V.shape
    X = Xd.cpu().numpy().astype(np.float64)
    U0, V0 = U.cpu().numpy().astype(np.float64), V.cpu().numpy().astype(np.float64)
    steps = 6
    objs, _ = eng.step(U, V, steps, gamma, delta)
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in pathways]
    tr = []
    Ur, Vr, _ = orc.nmf_manifold_vec_update(X, U0, V0, [pws[p] for p in active], n_steps=steps, gamma=gamma,
                                            delta=delta, trace=tr)
    eu, ev = rel(U.cpu().numpy(), Ur), rel(V.cpu().numpy(), Vr)
    eo = {key: relmax([o[key] for o in objs], [t[key] for t in tr]) for key in OBJ_KEYS}
    report("c2_full_size_vs_oracle", steps=steps, rel_U=eu, rel_V=ev, **{"rel_" + k_: v for k_, v in eo.items()})
    assert eu < 1e-4 and ev < 1e-4
    for key in OBJ_KEYS:
        assert eo[key] < 1e-4, key
    assert float(V.min()) >= EPS
    eng.close()


def _fp64_iteration(torch, Xd, U, V, pathways, active, gamma, delta):
    """One iteration of oracle.update_step in float64 on the GPU (tests/fp64_model.py), from float32 U, V and the
    stored X.  Returns P = X V, U_new, Q = X^T U_new, V_new (float64 tensors)."""
    from fp64_model import Fp64Model
    return Fp64Model(torch, Xd, pathways, active, gamma, delta).step(U.double(), V.double())


def test_fp64_model_equals_oracle():
    """The torch float64 restatement used at the sizes the NumPy oracle cannot reach, against the oracle itself at a
    size both run (weighted edges, isolated support nodes, all-zero genes): 4 steps, factors and every record."""
    import torch
    import synth
    from fp64_model import Fp64Model
    from oracle import prmf_oracle as orc
    case = synth.small_case(seed=31, m=300, n=900, k=7, P=5, order=60, zero_genes=4, isolated=2)
    n, k = case["V0"].shape
    active = [kk % 5 for kk in range(k)]
    Xd = torch.as_tensor(case["X"], dtype=torch.float32, device="cuda:0")
    X = Xd.cpu().numpy().astype(np.float64)
    mdl = Fp64Model(torch, Xd, case["pathways"], active, case["gamma"], case["delta"], chunk=128)
    U = torch.as_tensor(case["U0"], device="cuda:0")
    V = torch.as_tensor(case["V0"], device="cuda:0")
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, _ = orc.nmf_manifold_vec_update(X, case["U0"], case["V0"], [pws[p] for p in active], n_steps=4,
                                            gamma=case["gamma"], delta=case["delta"], trace=tr)
    for it in range(4):
        _, U, _, V = mdl.step(U, V)
        od = mdl.objective(U, V)
        for key in OBJ_KEYS:
            assert abs(od[key] - tr[it][key]) <= 1e-11 * abs(tr[it][key]), (it, key)
    assert rel(U.cpu().numpy(), Ur) < 1e-12 and rel(V.cpu().numpy(), Vr) < 1e-12


def _tile_check(torch, name, got, want, tol_tile):
    """Frobenius relative error and, per 32-row tile, max |got - want| relative to the tile's largest |want|."""
    got = got.double()
    r = float(torch.linalg.norm(got - want) / torch.linalg.norm(want))
    rows, k = want.shape
    pad = (-rows) % 32
    d = (got - want).abs()
    w = want.abs()
    if pad:
        z = torch.zeros((pad, k), dtype=torch.float64, device=want.device)
        d, w = torch.cat([d, z]), torch.cat([w, z])
    dt = d.reshape(-1, 32 * k).max(dim=1).values
    wt = w.reshape(-1, 32 * k).max(dim=1).values.clamp_min(1e-300)
    worst = float((dt / wt).max())
    bad = int(((dt / wt) > tol_tile).sum())
    return {"rel_" + name: r, "tile_" + name: worst, "bad_tiles_" + name: bad}


@pytest.mark.parametrize("workload,x_dtype", [("c3", "bf16"), ("c3", "f32"), ("c5shard", "bf16"), ("c5shard", "f32"),
                                              ("c5", "bf16"),
                                              # 8-wave sweeps (factor >= 8 MB) over matrices of one or two 512-row panels
                                              ("tallthin", "bf16"), ("tallthin", "f32"), ("tallthin128", "bf16"),
                                              ("wideshort", "bf16"), ("wideshort", "f32")])
def test_one_iteration_vs_fp64_at_full_size(workload, x_dtype):
    import torch
    eng, Xd, U, V, pathways, active, gamma, delta = _engine(torch, workload, x_dtype, 11)
    P64, U64, Q64, V64 = _fp64_iteration(torch, Xd, U, V, pathways, active, gamma, delta)
    eng.begin(U, V, gamma, delta)
    eng.phase_u()
    P = eng.product("xv")
    eng.phase_q()
    Q = eng.product("xtu")
    eng.phase_v()
    eng.phase_post()
    objs, _ = eng.end(1)
    # factor operands carry 16 mantissa bits (bf16 hi + lo) or are f32, X is exact, accumulation is float32 over up
    # to 50000 terms: measured 1e-7..1e-6 per array; the bar below leaves one order of magnitude
    tol_arr, tol_tile = 1e-5, 5e-5
    res = {}
    for name, got, want in (("XV", P, P64), ("U", U, U64), ("XtU", Q, Q64), ("V", V, V64)):
        res.update(_tile_check(torch, name, got, want, tol_tile))
    # objective of the same iteration: direct float64 residual against the Gram-identity record
    acc = 0.0
    Vn = V.double()
    for r0 in range(0, Xd.shape[0], 4096):
        R = Xd[r0:r0 + 4096].double() - U[r0:r0 + 4096].double() @ Vn.T
        acc += float((R * R).sum())
    res["rel_recon"] = abs(objs[-1]["recon"] - math.sqrt(acc)) / math.sqrt(acc)
    report("one_iteration_vs_fp64[%s-%s]" % (workload, x_dtype), **res)
    for name in ("XV", "U", "XtU", "V"):
        assert res["rel_" + name] < tol_arr, (name, res)
        assert res["bad_tiles_" + name] == 0, (name, res)
    assert res["rel_recon"] < 1e-4
    assert float(V.min()) >= EPS and bool(torch.isfinite(U).all())
    eng.close()


@pytest.mark.parametrize("workload,x_dtype,steps", [("c3", "bf16", 10),
                                                    ("c5shard", "bf16", 10),      # k = 128: a whole round at one rank's share of C5
                                                    ("c5", "bf16", 3)])           # BASELINE configs[4] whole: the records of 3 iterations
def test_whole_round_vs_fp64_at_full_size(workload, x_dtype, steps):
    """One whole round (the reference's `modulus` = 10 iterations, prmf_runner.py:740) at BASELINE.json's headline size
    against the float64 restatement run next to it: U, V after the round and all five objective components of EVERY
    iteration -- recon from the materialised residual as :337 computes it, manifold and ignore from the normalised
    Laplacians -- at the 1e-4 bar.  Measures the drift of the float32 state over a round at 50000 x 20000, and at
    k = 128 (25000 x 25000 for a round; 200000 x 25000 for three iterations of records), where the identity-based recon
    carries the float32 accumulation error of X^T U (4.5e-5 after one iteration at the full C5 shape):
    the records of these ordinary fits (recon / ||X|| ~ 10%) must stay identity-based (no flag) AND within the bar -- except at
    the whole C5 shape since round 6: with 200000 samples per accumulation chain the threshold below which the batch's LAST
    record takes the direct residual scales with the chain (capi.hip direct_below), so that record is exact to 1e-5 and flagged
    DIRECT, the earlier ones of the batch keep the identity value (within the bar) and are flagged IMPRECISE."""
    import torch
    from fp64_model import Fp64Model
    eng, Xd, U, V, pathways, active, gamma, delta = _engine(torch, workload, x_dtype, 17)
    mdl = Fp64Model(torch, Xd, pathways, active, gamma, delta)
    U64, V64 = U.double(), V.double()
    objs, _ = eng.step(U, V, steps, gamma, delta)
    worst = {key: 0.0 for key in OBJ_KEYS}
    for it in range(steps):
        _, U64, _, V64 = mdl.step(U64, V64)
        od = mdl.objective(U64, V64)
        for key in OBJ_KEYS:
            worst[key] = max(worst[key], abs(objs[it][key] - od[key]) / abs(od[key]))
    res = {"rel_" + key: v for key, v in worst.items()}
    res.update(_tile_check(torch, "U", U, U64, 1e-3))
    res.update(_tile_check(torch, "V", V, V64, 1e-3))
    res["flags"] = int(max(eng.last_flags))
    res["recon_over_normx"] = objs[-1]["recon"] / math.sqrt(eng.norm_x_sq)
    res["rel_recon_last_record"] = abs(objs[-1]["recon"] - od["recon"]) / od["recon"]
    report("whole_round_vs_fp64[%s-%s]" % (workload, x_dtype), steps=steps, **res)
    if workload == "c5":
        assert eng.last_flags[-1] == 1 and all(f == 2 for f in eng.last_flags[:-1]), eng.last_flags      # DIRECT last, IMPRECISE before
        assert res["rel_recon_last_record"] < 1e-5, res
    else:
        assert res["flags"] == 0, eng.last_flags          # an ordinary fit: the identity value stands, unflagged
    assert res["rel_U"] < 1e-4 and res["rel_V"] < 1e-4, res
    assert res["bad_tiles_U"] == 0 and res["bad_tiles_V"] == 0, res
    for key in OBJ_KEYS:
        assert worst[key] < 1e-4, (key, worst)
    assert float(V.min()) >= EPS and bool(torch.isfinite(U).all())
    eng.close()


@pytest.mark.parametrize("workload,x_dtype,rounds", [("c3", "bf16", 10), ("c3", "f32", 10), ("c5shard", "bf16", 10)])
def test_hundred_iterations_vs_fp64_at_full_size(workload, x_dtype, rounds):
    """A whole solve's worth of inner iterations at BASELINE.json's headline size: 10 rounds of 10 (prmf_runner.py:715-775)
    with a NEW active pathway per latent column every round, as the outer loop re-samples them (:717-729), the float32
    state carried across the rounds -- against the float64 restatement run next to it.  After rounds 1, 3 and 10 (10, 30,
    100 iterations): U, V per array and per 32-row tile, and all five objective components of the round's last iteration,
    at the 1e-4 bar north_star states.  SURVEY Appendix D measured this drift at 1000 x 4000 only (1e-6 after 200
    iterations, saturating); this is the same measurement at 50000 x 20000, k = 64 (bf16 and f32 storage of X) and at one
    rank's share of BASELINE configs[4] (25000 x 25000, k = 128)."""
    import torch
    from fp64_model import Fp64Model
    eng, Xd, U, V, pathways, active, gamma, delta = _engine(torch, workload, x_dtype, 23)
    P, k = len(pathways), V.shape[1]
    U64, V64 = U.double(), V.double()
    out = {}
    for rnd in range(rounds):
        active = [(kk + 7 * rnd) % P for kk in range(k)]
        eng.set_active(active)
        mdl = Fp64Model(torch, Xd, pathways, active, gamma, delta)
        objs, _ = eng.step(U, V, 10, gamma, delta)
        for _it in range(10):
            _, U64, _, V64 = mdl.step(U64, V64)
        if rnd + 1 in (1, 3, rounds):
            od = mdl.objective(U64, V64)
            res = {"rel_" + key: abs(objs[-1][key] - od[key]) / abs(od[key]) for key in OBJ_KEYS}
            res.update(_tile_check(torch, "U", U, U64, 1e-3))
            res.update(_tile_check(torch, "V", V, V64, 1e-3))
            res["flags"] = int(max(eng.last_flags))
            out[10 * (rnd + 1)] = res
            report("hundred_iterations_vs_fp64[%s-%s]" % (workload, x_dtype), iterations=10 * (rnd + 1), **res)
            assert res["rel_U"] < 1e-4 and res["rel_V"] < 1e-4, (rnd, res)
            assert res["bad_tiles_U"] == 0 and res["bad_tiles_V"] == 0, (rnd, res)
            for key in OBJ_KEYS:
                assert res["rel_" + key] < 1e-4, (rnd, key, res)
    assert float(V.min()) >= EPS and bool(torch.isfinite(U).all())
    eng.close()
