"""Edge shapes of the HIP path against the oracle: one sample, one gene block, one latent factor, a pathway of
two nodes, fewer rows than a wave tile; and the deferred objective against the per-iteration form."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check(case, x_dtype, steps=5):
    import torch
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    m, n = case["X"].shape
    k = case["k"]
    eng = PrmfEngine(m, n, k, x_dtype=x_dtype)
    eng.set_x(case["X"])
    eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
    P = len(case["pathways"])
    eng.set_active(np.arange(k) % P)
    U = eng.device_tensor(case["U0"], torch.float32)
    V = eng.device_tensor(case["V0"], torch.float32)
    objs, _ = eng.step(U, V, steps, case["gamma"], case["delta"])
    eng.close()
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % P] for i in range(k)],
                                             n_steps=steps, gamma=case["gamma"], delta=case["delta"], trace=tr)
    assert rel(U.cpu().numpy(), Ur) < 1e-4 and rel(V.cpu().numpy(), Vr) < 1e-4
    for a, b in zip(objs, tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key


@pytest.mark.parametrize("m,n,k,P,order,x_dtype", [
    (1, 40, 1, 1, 2, "f32"),          # one sample, one latent factor, a two-node pathway
    (3, 33, 2, 1, 5, "bf16"),
    (31, 64, 4, 2, 10, "f32"),        # fewer rows than one MFMA tile, fused epilogues (k % 4 == 0)
    (33, 31, 3, 3, 8, "bf16"),        # fewer genes than one tile
    (257, 129, 8, 2, 20, "bf16"),     # one row past a 256-row panel, one chunk past 128 genes
    (64, 2049, 32, 4, 30, "f32"),
    (5, 70, 40, 2, 10, "f32"),        # the 16x16x32 MFMA forms on tiny ragged shapes: f32 X at 32 < k <= 64 ...
    (40, 33, 64, 3, 8, "f32"),
    (33, 257, 100, 3, 20, "bf16"),    # ... and bf16 X at k > 64 (column-split epilogues)
    (2, 65, 128, 2, 12, "f32"),
])
def test_tiny_and_ragged_shapes(m, n, k, P, order, x_dtype):
    case = synth.small_case(seed=21, m=m, n=n, k=k, P=P, order=order, zero_genes=min(3, n // 8), isolated=1,
                            bf16_x=(x_dtype == "bf16"))
    check(case, x_dtype)


def test_deferred_objective_equals_per_iteration_form():
    """The default single-rank iteration is 4 launches: the Gram / column reductions and the pathway terms ride in
    the sweeps' prologues and the objective records of a batch are evaluated in prmf_end.  PRMF_FLAG_EAGER_OBJECTIVE
    keeps the per-iteration launches (mid / post / finalize).  Same arithmetic in the same order: U, V and every
    record must be bit-identical -- also across a call boundary and with k = 13 (unfused path: no difference at all)."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    for (m, n, k, xd) in [(777, 1900, 16, "f32"), (1500, 2600, 64, "bf16"), (300, 1000, 13, "f32"), (640, 900, 128, "bf16"), (500, 800, 48, "f32")]:
        case = synth.small_case(seed=5, m=m, n=n, k=k, P=6, order=90, bf16_x=(xd == "bf16"))
        outs = []
        for lazy in (True, False):
            eng = PrmfEngine(m, n, k, x_dtype=xd, eager_objective=not lazy)
            eng.set_x(case["X"])
            eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
            eng.set_active(np.arange(k) % 6)
            U = eng.device_tensor(case["U0"], torch.float32)
            V = eng.device_tensor(case["V0"], torch.float32)
            o1, _ = eng.step(U, V, 4, case["gamma"], case["delta"])
            eng.set_active((np.arange(k) + 1) % 6)                  # a new round: other pathways
            o2, _ = eng.step(U, V, 3, case["gamma"], case["delta"])
            outs.append((U.cpu().numpy(), V.cpu().numpy(), o1 + o2))
            eng.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (m, n, k)
        assert outs[0][2] == outs[1][2], (m, n, k)


def test_long_call_is_batched_like_short_calls():
    """prmf_step keeps at most 256 iterations of deferred-objective state on the device: a 300-iteration call runs as two
    batches inside the library and must equal three 100-iteration calls bit for bit (factors and all 300 records)."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=8, m=300, n=700, k=8, P=4, order=50)
    outs = []
    for calls in ((300,), (100, 100, 100)):
        eng = PrmfEngine(300, 700, 8)
        eng.set_x(case["X"])
        eng.set_library(PathwayLibrary.from_edge_lists(700, case["pathways"]))
        eng.set_active(np.arange(8) % 4)
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        objs = []
        for nst in calls:
            o, _ = eng.step(U, V, nst, case["gamma"], case["delta"])
            objs += o
        outs.append((U.cpu().numpy(), V.cpu().numpy(), objs))
        eng.close()
    assert len(outs[0][2]) == 300 and outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception):      # the active pathways belong to a whole batch: changing them mid-batch is an error
        eng2 = PrmfEngine(300, 700, 8)
        eng2.set_x(case["X"])
        eng2.set_library(PathwayLibrary.from_edge_lists(700, case["pathways"]))
        eng2.set_active(np.arange(8) % 4)
        U = eng2.device_tensor(case["U0"], torch.float32)
        V = eng2.device_tensor(case["V0"], torch.float32)
        eng2.begin(U, V, case["gamma"], case["delta"])
        eng2.phase_u(); eng2.phase_q(); eng2.phase_v(); eng2.phase_post()
        try:
            eng2.set_active((np.arange(8) + 1) % 4)
        finally:
            eng2.end(1)
            eng2.close()


@pytest.mark.parametrize("x_dtype", ["f32", "bf16"])
def test_all_zero_matrix_matches_oracle(x_dtype):
    """X = 0 with the caller's finite gamma / delta (the CLI would stop earlier: delta = 10 / ||X||, prmf_runner.py:645): P = X V = 0
    drives U to exact zeros in one step (zeros are absorbing, :422), V follows the pathway terms alone -- finite, equal to
    the oracle, recon exactly 0 (flagged as clamped or direct, never a NaN from sqrt of a negative identity value)."""
    case = synth.small_case(seed=3, m=45, n=200, k=5, P=3, order=20, bf16_x=(x_dtype == "bf16"))
    case["X"] = np.zeros_like(case["X"])
    case["gamma"], case["delta"] = 0.7, 0.3
    check(case, x_dtype, steps=4)


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_input_is_an_error_not_a_silent_nan(bad):
    """A NaN or an infinity in X: the reference would carry NaNs through every array without a word (np.seterr only raises
    on divide); the library must never hand back non-finite factors silently (risk 4): PRMF_ERR_NUMERIC ->
    FloatingPointError on the Python side."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=4, m=50, n=300, k=6, P=3, order=20)
    X = case["X"].copy()
    X[7, 11] = bad
    eng = PrmfEngine(50, 300, 6, x_dtype="f32")
    try:
        eng.set_x(X, norm_x_sq=float(np.sum(case["X"] ** 2)))
        eng.set_library(PathwayLibrary.from_edge_lists(300, case["pathways"]))
        eng.set_active(np.arange(6) % 3)
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        with pytest.raises(FloatingPointError):
            eng.step(U, V, 3, case["gamma"], case["delta"])
    finally:
        eng.close()


@pytest.mark.parametrize("gamma,delta", [(0.0, 0.0), (0.0, 0.5), (2.5, 0.0), (1e5, 1e-6), (1e-8, 1e4)])
@pytest.mark.parametrize("x_dtype,k", [("f32", 6), ("bf16", 70)])
def test_extreme_hyper_parameters(gamma, delta, x_dtype, k):
    """gamma = 0 and / or delta = 0 (plain NMF: the pathway terms vanish from the update, not from the record), a gamma
    that lets the Laplacian term dominate the V update, a delta that lets the ignore term dominate: the association
    num = Q + (gamma W v + delta (v + 1)^-2) and the two clamps of prmf_runner.py:440-444 at both ends of the scale."""
    case = synth.small_case(seed=8, m=90, n=260, k=k, P=4, order=30, bf16_x=(x_dtype == "bf16"))
    case["gamma"], case["delta"] = gamma, delta
    check(case, x_dtype, steps=4)


@pytest.mark.parametrize("n,order,k,x_dtype", [(4000, 3000, 8, "f32"),       # support beyond the 1024 floats of LDS the k <= 32 epilogue can stage
                                               (4000, 3000, 64, "bf16"),     # the same pathway within the 4096 floats of the k = 64 one
                                               (7000, 6000, 64, "f32"),      # beyond them
                                               (7000, 6900, 128, "bf16")])   # nearly every gene on one pathway, k = 128
def test_pathways_larger_than_the_lds_staging(n, order, k, x_dtype):
    """Pathways of thousands of genes (the reference's KEGG-sized graphs have ~300): the pathway terms stage a support's V
    values in LDS when they fit and walk global memory when they do not (update.hip: graph_terms_body<true / false>); the
    deferred objective and the candidate scorer have the same two paths."""
    import torch
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=6, m=70, n=n, k=k, P=2, order=order, zero_genes=3, isolated=2, bf16_x=(x_dtype == "bf16"))
    check(case, x_dtype, steps=3)
    # the candidate scorer on the same library (restrict, prmf_runner.py:115-127)
    eng = PrmfEngine(70, n, k, x_dtype=x_dtype)
    try:
        eng.set_x(case["X"])
        eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
        V = eng.device_tensor(case["V0"], torch.float32)
        S = eng.score_candidates(V)
    finally:
        eng.close()
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    want = orc.score_matrix(case["V0"].astype(np.float32).astype(np.float64), pws)
    assert np.max(np.abs(S - want) / np.maximum(np.abs(want), 1e-12)) < 1e-5


def test_self_loops_duplicate_edges_and_components():
    """Graph oddities the library format has to carry: self-loops (on the diagonal of W once, so L_ii = deg - W_ii differs
    from deg: normalize_laplacian reads L_ii, prmf_runner.py:60), duplicate edges (summed), two components, a hub, isolated
    support nodes -- inner loop, objective and candidate scores against the oracle."""
    import torch
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=12, m=40, n=120, k=4, P=1, order=10, zero_genes=2, isolated=0)
    nodes0 = np.arange(5, 25)
    e0 = [(5, 6), (6, 7), (7, 7), (5, 6), (8, 9), (9, 10), (10, 8), (12, 12), (5, 13), (5, 14), (5, 15), (5, 16), (20, 21)]      # 11, 17-19, 22-24: isolated
    w0 = [1.0, 0.5, 2.0, 0.25, 1.0, 1.0, 3.0, 0.75, 1, 1, 1, 1, 0.125]
    nodes1 = np.array([100, 3, 50, 51, 52])
    e1 = [(100, 3), (3, 50), (50, 50), (51, 52)]
    case["pathways"] = [(np.array(e0), np.array(w0), nodes0), (np.array(e1), np.ones(len(e1)), nodes1)]
    check(case, "f32", steps=5)
    check(dict(case, X=synth.round_to_bf16(case["X"])), "bf16", steps=3)      # (the oracle gets the X the GPU stores)
    eng = PrmfEngine(40, 120, 4, x_dtype="f32")
    try:
        eng.set_x(case["X"])
        eng.set_library(PathwayLibrary.from_edge_lists(120, case["pathways"]))
        S = eng.score_candidates(eng.device_tensor(case["V0"], torch.float32))
    finally:
        eng.close()
    pws = [orc.pathway_from_edges(120, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    want = orc.score_matrix(case["V0"].astype(np.float32).astype(np.float64), pws)
    assert np.max(np.abs(S - want) / np.maximum(np.abs(want), 1e-12)) < 1e-5


def test_contexts_release_their_device_memory():
    """Thirty contexts in a row (create, X, library, a few iterations in three forms, scores, destroy), then several alive at
    once: the free device memory comes back every time -- a CLI run is one context, a cross-validation or a server process many."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=2, m=300, n=2000, k=33, P=4, order=60)
    lib = PathwayLibrary.from_edge_lists(2000, case["pathways"])

    def one(flags):
        eng = PrmfEngine(300, 2000, 33, x_dtype="f32", **flags)
        eng.set_x(case["X"])
        eng.set_library(lib)
        eng.set_active(np.arange(33) % 4)
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        eng.step(U, V, 2, case["gamma"], case["delta"])
        eng.score_candidates(V)
        return eng

    one({}).close()                                        # first use: module load, runtime pools
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(30):
        one([{}, {"eager_objective": True}, {"direct_residual": True}, {"timing": True}][i % 4]).close()
    alive = [one({}) for _ in range(5)]
    for e in alive:
        e.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, "device memory not returned: %.1f MB" % ((free0 - free1) / 2**20)


def test_independent_contexts_on_concurrent_threads():
    """Four unrelated problems (different shapes, storages, k) solved at the same time by four host threads, each with a
    context and a stream of its own: no state is shared between contexts (include/prmf_hip.h), so every thread gets,
    bit for bit, what the same problem gives when it runs alone."""
    import threading
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    specs = [dict(seed=21, m=400, n=1500, k=16, P=4, order=50, x="f32"), dict(seed=22, m=90, n=5000, k=64, P=6, order=80, x="bf16"),
             dict(seed=23, m=700, n=600, k=100, P=3, order=40, x="bf16"), dict(seed=24, m=33, n=257, k=6, P=2, order=12, x="f32")]
    cases = [synth.small_case(seed=s["seed"], m=s["m"], n=s["n"], k=s["k"], P=s["P"], order=s["order"], bf16_x=(s["x"] == "bf16")) for s in specs]

    def solve(i, out, stream=None):
        s, case = specs[i], cases[i]
        ctx = torch.cuda.stream(stream) if stream is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            eng = PrmfEngine(s["m"], s["n"], s["k"], x_dtype=s["x"])
            eng.set_x(case["X"])
            eng.set_library(PathwayLibrary.from_edge_lists(s["n"], case["pathways"]))
            eng.set_active(np.arange(s["k"]) % s["P"])
            U = eng.device_tensor(case["U0"], torch.float32)
            V = eng.device_tensor(case["V0"], torch.float32)
            objs = []
            for _ in range(6):                      # several calls, so that the threads interleave
                o, _gd = eng.step(U, V, 3, case["gamma"], case["delta"])
                objs += o
            torch.cuda.current_stream().synchronize()
            out[i] = (U.cpu().numpy(), V.cpu().numpy(), objs)
            eng.close()
        except BaseException as e:      # noqa: B902
            out[i] = e
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    alone, together = {}, {}
    for i in range(4):
        solve(i, alone)
    threads = [threading.Thread(target=solve, args=(i, together, torch.cuda.Stream())) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    for i in range(4):
        assert not isinstance(alone[i], BaseException), alone[i]
        assert not isinstance(together.get(i), BaseException) and together.get(i) is not None, together.get(i)
        assert np.array_equal(alone[i][0], together[i][0]) and np.array_equal(alone[i][1], together[i][1]) and alone[i][2] == together[i][2], i


def test_sum_squares_kernel_vs_float64():
    """prmf_sum_squares (||X||_F^2 of the stored values, prmf_runner.py:640) against numpy in float64: both storages, a leading
    dimension wider than the row, one row, no rows, 3 x 10^7 values; the same bits on a second call."""
    import ctypes as C
    import torch
    from prmf_amd import _capi
    lib = _capi.load()
    rs = np.random.RandomState(6)

    def ss(t, m, n, ld, xd):
        out = C.c_double(-1.0)
        rc = lib.prmf_sum_squares(0, C.c_void_p(t.data_ptr()), xd, m, n, ld, C.byref(out), None)
        assert rc == _capi.PRMF_OK, lib.prmf_last_error(None)
        return out.value

    for m, n, ld in ((1, 1, 1), (1, 5000, 5000), (257, 33, 40), (3000, 10000, 10000), (0, 7, 7)):
        a = (rs.gamma(2.0, size=(m, ld)) * 3).astype(np.float32)
        for xd, tdt in ((_capi.PRMF_X_F32, torch.float32), (_capi.PRMF_X_BF16, torch.bfloat16)):
            t = torch.from_numpy(a).to("cuda:0").to(tdt).contiguous()
            want = float((t[:, :n].double().cpu().numpy() ** 2).sum())
            got = ss(t, m, n, ld, xd)
            assert got == ss(t, m, n, ld, xd)
            assert abs(got - want) <= 1e-13 * max(want, 1.0), (m, n, ld, xd, got, want)
    out = C.c_double()
    assert lib.prmf_sum_squares(0, None, _capi.PRMF_X_F32, 3, 3, 3, C.byref(out), None) == _capi.PRMF_ERR_ARG
    assert lib.prmf_sum_squares(0, None, 7, 0, 0, 0, C.byref(out), None) == _capi.PRMF_ERR_ARG
    assert lib.prmf_sum_squares(0, None, _capi.PRMF_X_F32, 2, 5, 4, C.byref(out), None) == _capi.PRMF_ERR_ARG
