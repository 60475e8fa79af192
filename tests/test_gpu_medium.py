"""Medium-size checks of the HIP path against the oracle: many reduction chunks,
several splits, ragged shapes -- and run-to-run bitwise determinism."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def run_gpu(case, x_dtype, steps, repeats=1):
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    m, n = case["X"].shape
    eng = PrmfEngine(m, n, case["k"], x_dtype=x_dtype)
    eng.set_x(case["X"])
    eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
    eng.set_active(np.arange(case["k"]) % len(case["pathways"]))
    outs = []
    for _ in range(repeats):
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        eng.invalidate_factors()
        objs, _ = eng.step(U, V, steps, case["gamma"], case["delta"])
        outs.append((U.cpu().numpy(), V.cpu().numpy(), objs))
    eng.close()
    return outs


@pytest.mark.parametrize("m,n,k,x_dtype", [
    (2000, 3000, 64, "bf16"),
    (1537, 2111, 16, "f32"),
    (1000, 5000, 33, "bf16"),
    (700, 900, 128, "bf16"),
    (515, 1300, 100, "f32"),
])
def test_medium_vs_oracle_and_deterministic(m, n, k, x_dtype):
    from oracle import prmf_oracle as orc
    case = synth.small_case(seed=7, m=m, n=n, k=k, P=8, order=120, bf16_x=(x_dtype == "bf16"))
    steps = 6
    outs = run_gpu(case, x_dtype, steps, repeats=3)
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    active = [pws[i % len(pws)] for i in range(k)]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], active, n_steps=steps,
                                             gamma=case["gamma"], delta=case["delta"], trace=tr)
    for U, V, objs in outs:
        assert np.isfinite(U).all() and np.isfinite(V).all()
        assert rel(U, Ur) < 1e-4 and rel(V, Vr) < 1e-4
        for a, b in zip(objs, tr):
            for key in ("recon", "manifold", "ignore", "fro", "obj"):
                assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key
    for U, V, objs in outs[1:]:
        assert np.array_equal(U, outs[0][0]) and np.array_equal(V, outs[0][1]), "run-to-run difference"
        assert objs == outs[0][2]
