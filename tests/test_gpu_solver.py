"""GPU tests of everything around the inner loop: tradeoff feedback, candidate
scoring, the outer solver and the CLI -- against reference-generated goldens."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def engine_for(case, x_dtype="f32", **kw):
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import PrmfEngine
    m, n = case["X"].shape
    eng = PrmfEngine(m, n, case["k"], x_dtype=x_dtype, **kw)
    eng.set_x(case["X"])
    eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
    return eng


def test_tradeoff_variant_vs_reference(golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, "g4_tradeoff15.npz"))
    case = synth.small_case(seed=1)
    eng = engine_for(case)
    eng.set_active(np.arange(case["k"]))
    U = eng.device_tensor(case["U0"], torch.float32)
    V = eng.device_tensor(case["V0"], torch.float32)
    # The reference's gamma feedback runs away on this input (gamma 1e5 at step 7, 1e11 at
    # step 9, 1e295 at step 14 in float64): only the first 7 steps fit float32 state.
    nst = 7
    objs, gd = eng.step(U, V, nst, case["gamma"], case["delta"], tradeoff=float(g["tradeoff"]))
    got = np.array([[o[k] for k in ["recon", "manifold", "ignore", "fro", "gamma", "delta", "obj"]] for o in objs])
    assert np.allclose(got, g["obj"][:nst], rtol=5e-4)
    assert np.allclose(gd, g["obj"][nst][4:6], rtol=5e-4)        # gamma/delta handed to step nst+1
    objs5, _ = eng.step(U, V, 0, 1.0, 1.0)                        # n_steps = 0 is a no-op
    assert objs5 == []
    eng.close()


def test_drop_in_inner_loop_function(golden_dir, capsys):
    """nmf_manifold_vec_update with the reference's own argument types."""
    import scipy.sparse as sp
    from prmf_amd.solver import nmf_manifold_vec_update
    g = np.load(os.path.join(golden_dir, "g2_inner25.npz"))
    case = synth.small_case(seed=1)
    n = case["X"].shape[1]
    kW, kD, kL, kF = {}, {}, {}, {}
    for k in range(case["k"]):
        e, w, nodes = case["pathways"][k]
        W = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]]))),
                          shape=(n, n)).tocsr()
        D = sp.dok_matrix(sp.dia_matrix((W.sum(axis=0), np.array([0])), shape=(n, n)))
        kW[k], kD[k], kL[k], kF[k] = W, D, D - W, list(nodes)
    U, V, od = nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], kW, kD, kL, kF, n_steps=5,
                                       gamma=case["gamma"], delta=case["delta"], i=10)
    assert rel(U, g["U"][1]) < 1e-4 and rel(V, g["V"][1]) < 1e-4
    assert abs(od["obj"] - g["obj"][4][6]) < 1e-4 * g["obj"][4][6]
    lines = capsys.readouterr().out.strip().splitlines()
    assert [l.split()[0] for l in lines] == ["11", "12", "13", "14", "15"]     # "<iter> <obj>" trace


def test_score_candidates_vs_reference(golden_dir):
    import torch
    g = np.load(os.path.join(golden_dir, "g5_restrict.npz"))
    case = synth.small_case(seed=3, P=12)
    eng = engine_for(case)
    V = eng.device_tensor(g["V"], torch.float32)
    S = eng.score_candidates(V)
    assert np.allclose(S, g["scores"], rtol=1e-5, atol=1e-6)
    eng.close()


def _graphs(c):
    import networkx as nx
    from prmf_amd import runner
    return [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]


@pytest.mark.parametrize("name,tradeoff", [("g6_api.npz", None)])
def test_outer_solver_vs_reference_trace(golden_dir, tmp_path, name, tradeoff):
    """nmf_pathway end to end: same RNG stream -> same per-round pathway assignment,
    same final latent->pathway map, objective trace within fp32 tolerance."""
    from prmf_amd.solver import nmf_pathway
    g = np.load(os.path.join(golden_dir, name))
    c = synth.cli_case(str(tmp_path), variant=1)
    rounds, trace = [], []
    np.random.seed(1)
    U, V, od = nmf_pathway(c["X"], _graphs(c), nodelist=c["nodelist_ids"], gamma=1.0, tradeoff=tradeoff,
                           k_latent=6, round_log=rounds, trace=trace)
    assert np.array_equal(np.array(rounds), g["rounds"])
    assert np.allclose([t["obj"] for t in trace], g["trace"], rtol=1e-4)
    l2p = od["latent_to_pathway_data"]
    assert [int(l2p[k][0][0]) for k in sorted(l2p)] == list(g["final_map"])
    assert rel(U, g["U"]) < 1e-3 and rel(V, g["V"]) < 1e-3
    with pytest.raises(ValueError):
        nmf_pathway(c["X"], _graphs(c), nodelist=c["nodelist_ids"], k_latent=6, U_init=np.ones((3, 6)))


def _cli(args, cwd):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py")] + args, cwd=cwd,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _read_obj(fp):
    vals, mapping = {}, []
    for line in open(fp):
        if " = " in line:
            a, b = line.rstrip().split(" = ")
            vals[a] = float(b)
        elif "->" in line:
            a, b = line.rstrip().split("->")
            mapping.append([int(a), os.path.basename(b.strip())])
    return vals, mapping


@pytest.mark.parametrize("variant,tag", [(1, "g6"), (2, "g7")])
def test_cli_outputs_vs_reference(golden_dir, tmp_path, variant, tag):
    import pandas as pd
    g = np.load(os.path.join(golden_dir, tag + "_nonorm.npz"))
    c = synth.cli_case(str(tmp_path), variant=variant)
    out = str(tmp_path / "out")
    os.makedirs(out)
    stdout = _cli(["--data", c["data"], "--manifolds"] + c["manifolds"] +
                  ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t",
                   "--seed", "1", "--no-normalize"], str(tmp_path))
    assert "Printing manifold node representation in nodelist:" in stdout and "norm(X) = " in stdout
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)                               # same keys, same order
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 2e-4 * max(abs(ref[key]), 1e-3), key
    assert mapping == json.loads(str(g["mapping"]))
    U = pd.read_csv(os.path.join(out, "U.csv"))
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
    assert list(U.columns) == ["LV%d" % i for i in range(6)] and list(V.index) == c["nodelist_ids"]
    assert rel(U.to_numpy(), g["U"]) < 1e-3 and rel(V.to_numpy(), g["V"]) < 1e-3
    head = open(os.path.join(out, "V.csv")).readline().strip()
    assert head == '"","LV0","LV1","LV2","LV3","LV4","LV5"'       # QUOTE_NONNUMERIC contract
    if variant == 2:
        # unmeasured genes (all-zero columns of X) pin their V rows at float32 eps
        assert np.allclose(V.to_numpy()[1000:], np.finfo(np.float32).eps)


@pytest.mark.parametrize("tag", ["g10_init", "g11_fewer_pathways", "g12_weighted"])
def test_cli_more_variants_vs_reference(golden_dir, tmp_path, tag):
    """--manifolds-init (NNLS-based U/V start, init_pathways.txt), fewer pathways than latent factors,
    weighted graphml edges: obj.txt, the latent->pathway mapping, U.csv and V.csv against reference runs."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    man = c["manifolds"]
    extra = []
    if tag == "g10_init":
        extra = ["--manifolds-init"] + [man[i] for i in (0, 1, 3, 2, 5, 4)]
    elif tag == "g11_fewer_pathways":
        man = man[:3]
    else:
        wdir = str(tmp_path / "weighted")
        os.makedirs(wdir)
        man = synth.weighted_copies(man, wdir)
    out = str(tmp_path / "out")
    os.makedirs(out)
    stdout = _cli(["--data", c["data"], "--manifolds"] + man + extra +
                  ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t",
                   "--seed", "1", "--no-normalize"], str(tmp_path))
    trace = np.array([float(l.split()[1]) for l in stdout.splitlines()
                      if len(l.split()) == 2 and l.split()[0].isdigit()])
    ref_trace = g["trace"]
    assert len(trace) == len(ref_trace)
    assert np.allclose(trace, ref_trace, rtol=2e-4)
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 2e-4 * max(abs(ref[key]), 1e-3), key
    assert mapping == json.loads(str(g["mapping"]))
    U = pd.read_csv(os.path.join(out, "U.csv")).to_numpy()
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0).to_numpy()
    assert rel(U, g["U"]) < 2e-3 and rel(V, g["V"]) < 2e-3
    if tag == "g10_init":
        got = [os.path.basename(x) for x in open(os.path.join(out, "init_pathways.txt")).read().split("\n")]
        assert got == json.loads(str(g["init_pathways"]))
        assert "Using the following manifolds for initialization:" in stdout


def test_cli_npy_input(golden_dir, tmp_path):
    """--data file.npy (extension for large matrices) gives the run of the delimited file."""
    g = np.load(os.path.join(golden_dir, "g6_nonorm.npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    npy = str(tmp_path / "data.npy")
    np.save(npy, c["X"])
    out = str(tmp_path / "out")
    os.makedirs(out)
    _cli(["--data", npy, "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--nodelist", c["nodelist"],
                                                           "--outdir", out, "--seed", "1", "--no-normalize"], str(tmp_path))
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 2e-4 * max(abs(ref[key]), 1e-3), key
    assert mapping == json.loads(str(g["mapping"]))


def test_cli_config1_plumbing(golden_dir, tmp_path):
    """BASELINE.json configs[0]: header-inferred nodelist, -k 4, 2 manifolds, --cross-validation 0.2
    (quantile-normalised)."""
    g = np.load(os.path.join(golden_dir, "g8_config1.npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    out = str(tmp_path / "out")
    os.makedirs(out)
    _cli(["--data", c["data_header"], "--manifolds"] + c["manifolds"][:2] +
         ["--node-attribute", "name", "--outdir", out, "--delimiter", "\t", "--seed", "1", "-k", "4",
          "--cross-validation", "0.2"], str(tmp_path))
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 5e-4 * max(abs(ref[key]), 1e-3), key
    assert mapping == json.loads(str(g["mapping"]))
    err = np.loadtxt(os.path.join(out, "test_error.csv"), delimiter=",")
    assert err.shape == (20,) and np.allclose(err, g["test_error"], rtol=5e-3)


@pytest.mark.parametrize("m,n,k,x_dtype", [(37, 500, 8, "f32"), (64, 1000, 13, "f32"), (50, 2000, 64, "f32"),
                                            (20, 900, 128, "f32"), (40, 700, 16, "bf16")])
def test_cv_error_vs_scipy_nnls(m, n, k, x_dtype):
    """prmf_cv_error (float64 X V and V^T V + batched active-set NNLS on the normal equations) against
    scipy.optimize.nnls per sample (the oracle's measure_cv_performance = prmf/__init__.py:768-798).
    V has strongly correlated columns and the samples use few of them, so the active set both grows
    and shrinks.  Tolerance 1e-7 relative: the residual comes from ||x||^2 - 2c.h + c^T G c in float64,
    cancellation factor ||x||^2 / err^2 ~ 1e3 here."""
    from oracle import prmf_oracle as orc
    from prmf_amd.solver import measure_cv_performance
    rs = np.random.RandomState(m + k)
    base = rs.rand(n, max(2, k // 3))
    V = np.abs(base[:, rs.randint(0, base.shape[1], k)] + 0.15 * rs.rand(n, k)).astype(np.float32).astype(np.float64)
    C = rs.rand(m, k) * (rs.rand(m, k) < 0.3)
    Xt = C.dot(V.T) + 0.2 * rs.rand(m, n)
    Xt[3] = 0.0                                           # all-zero sample: 0/0 -> nan in the reference
    Xt[5] = 2.0 * V[:, 1]                                 # exactly representable sample: error ~ 0
    if x_dtype == "bf16":
        Xt = synth.round_to_bf16(Xt)
    else:
        Xt = Xt.astype(np.float32).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = orc.measure_cv_performance(V, Xt)
    got = measure_cv_performance(V, Xt, x_dtype=x_dtype)
    assert got.shape == want.shape
    assert np.isnan(got[3]) and np.isnan(want[3])
    mask = np.ones(m, dtype=bool)
    mask[[3, 5]] = False
    assert np.allclose(got[mask], want[mask], rtol=1e-7, atol=0)
    if x_dtype == "bf16":                                  # 2 V[:,1] rounded to bf16 is no longer in the cone
        assert abs(got[5] - want[5]) < 1e-6 * want[5]
    else:                                                  # near-zero residual: sqrt of the float64 cancellation noise
        assert got[5] < 1e-6 and want[5] < 1e-6
