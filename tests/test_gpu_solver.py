"""GPU tests of everything around the inner loop: tradeoff feedback, candidate
scoring, the outer solver and the CLI -- against reference-generated goldens."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from measure import rel, relmax, report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Tolerances.  State is float32 against the float64 reference: ~1e-6 drift after hundreds of iterations on
# U and V, the bar is 1e-4.  Scalars parsed from obj.txt are printed by the reference with
# 5 decimals ("%.5f"): a value like manifold = 0.00215 carries 3 significant digits, so a scalar v from obj.txt is
# compared with |got - ref| <= TOL_OBJ * |ref| + OBJ_TXT_QUANTUM (half a unit of the last printed decimal, twice:
# both sides were rounded).  U.csv / V.csv hold full-precision reprs and get the plain 1e-4 bar.
TOL_FACTOR = 1e-4
TOL_OBJ = 1e-4
# Round 6: on the reference's own problems (100 samples) both big products are exact float32 (csrc/gemm.hip: exact_product_kernel)
# and whole solves of 70-1000 iterations follow the reference to 1-2e-6 (3-4e-5 before: profiles/r06_parity_measured.jsonl against
# r05_).  PIN is a regression pin a decade above that; the bar north_star states stays the TOL_* above.
PIN = 1e-5
OBJ_TXT_QUANTUM = 1.0e-5


def obj_txt_errors(vals, ref):
    """per key: |got - ref| minus the print quantum, relative to |ref| (<= TOL_OBJ passes)"""
    return {key: max(0.0, abs(vals[key] - ref[key]) - OBJ_TXT_QUANTUM) / max(abs(ref[key]), 1e-12) for key in ref}


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
    # the feedback gamma = (1 - t) recon / (t manifold) divides by a manifold term of 1e-3..1e-5 that is itself a
    # difference of nearly equal sums: float32 V carries ~1e-7 into it, amplified step by step (gamma x2..x100 per
    # step here).  Measured drift is reported; the bound is per step, loosening with the amplification.
    err = [relmax(got[i], g["obj"][i]) for i in range(nst)]
    report("tradeoff_variant_vs_reference", per_step=err, gd=relmax(gd, g["obj"][nst][4:6]))
    assert max(err[:4]) < 1e-4 and max(err) < 5e-4
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


@pytest.mark.parametrize("name,kw", [("g5_restrict.npz", dict(seed=3, P=12)),
                                     ("g19_restrict_k40.npz", dict(seed=5, m=60, n=1500, k=40, P=60, order=50))])      # round 5: 40 x 60 pairs
def test_score_candidates_vs_reference(golden_dir, name, kw):
    import torch
    g = np.load(os.path.join(golden_dir, name))
    case = synth.small_case(**kw)
    eng = engine_for(case)
    V = eng.device_tensor(g["V"], torch.float32)
    S = eng.score_candidates(V)
    assert np.allclose(S, g["scores"], rtol=1e-5, atol=1e-6)
    eng.close()


@pytest.mark.slow
def test_score_candidates_at_c5_shape_vs_oracle():
    """`restrict`'s score matrix (prmf_runner.py:115-127) at the shape of BASELINE.json configs[4] -- 25000 genes,
    k = 128, P = 1000 pathways of 300 genes -- against oracle.score_latent_pathway_match on sampled (latent, pathway)
    pairs: every latent column with 24 pathways drawn per column, plus every pathway with one column."""
    import torch
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary, synthetic as syn
    from prmf_amd.solver import PrmfEngine
    from measure import report
    n, k, P = 25000, 128, 1000
    pathways = syn.tree_pathways(n, P, 300, 23)
    eng = PrmfEngine(64, n, k, x_dtype="bf16")       # scoring needs the library and V only
    eng.set_library(PathwayLibrary.from_edge_lists(n, pathways))
    V = syn.init_factor_torch(torch, "cuda:0", 29, n, k)
    V[::7, 3] = float(np.finfo(np.float32).eps)      # clamped entries as the V update leaves them
    S = eng.score_candidates(V)
    eng.close()
    assert S.shape == (k, P)
    Vh = V.cpu().numpy().astype(np.float64)
    rs = np.random.RandomState(5)
    pairs = [(kk, int(p)) for kk in range(k) for p in rs.choice(P, size=24, replace=False)]
    pairs += [(int(rs.randint(k)), p) for p in range(P)]
    cache = {}
    worst = 0.0
    for kk, p in pairs:
        if p not in cache:
            e, w, nodes = pathways[p]
            cache[p] = orc.pathway_from_edges(n, e, nodes, w)
        want = orc.score_latent_pathway_match(Vh[:, kk], cache[p])
        worst = max(worst, abs(S[kk, p] - want) / max(abs(want), 1e-12))
    report("score_candidates_c5_shape", pairs=len(pairs), rel_worst=worst)
    assert worst < 1e-5


def _graphs(c):
    import networkx as nx
    from prmf_amd import runner
    return [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]


@pytest.mark.parametrize("name,tradeoff", [("g6_api.npz", None), ("g6_api_tradeoff.npz", 0.5)])
def test_outer_solver_vs_reference_trace(golden_dir, tmp_path, name, tradeoff):
    """nmf_pathway end to end (with and without --tradeoff, prmf_runner.py:740 / :742): same RNG stream -> same
    per-round pathway assignment, same final latent->pathway map, objective trace within fp32 tolerance."""
    from prmf_amd.solver import nmf_pathway
    g = np.load(os.path.join(golden_dir, name))
    c = synth.cli_case(str(tmp_path), variant=1)
    rounds, trace = [], []
    np.random.seed(1)
    # (with --tradeoff the reference never meets |d obj| / obj < tol on this input: it stops at max_iter = 1000 with the
    # per-round change at 1.0e-3..2e-3, i.e. AT the threshold -- the default stopping rule is kept here all the same)
    U, V, od = nmf_pathway(c["X"], _graphs(c), nodelist=c["nodelist_ids"], gamma=1.0, tradeoff=tradeoff,
                           k_latent=6, round_log=rounds, trace=trace)
    tr = np.array([t["obj"] for t in trace])
    ncmp = min(len(tr), len(g["trace"]))
    e_tr = relmax(tr[:ncmp], g["trace"][:ncmp])
    eu, ev = rel(U, g["U"]), rel(V, g["V"])
    drift = [relmax(tr[a:a + 100], g["trace"][a:a + 100]) for a in range(0, ncmp, 100)]
    report("outer_solver_vs_reference_trace[%s]" % name, iterations=len(tr), ref_iterations=len(g["trace"]),
           rel_trace=e_tr, rel_trace_per_100=drift, rel_U=eu, rel_V=ev, rel_obj=relmax([od[k] for k in ("recon", "manifold", "ignore", "fro", "obj")],
                                                              g["obj"][[0, 1, 2, 3, 6]]))
    assert np.array_equal(np.array(rounds), g["rounds"])
    assert len(tr) == len(g["trace"])
    l2p = od["latent_to_pathway_data"]
    assert [int(l2p[k][0][0]) for k in sorted(l2p)] == list(g["final_map"])
    # with the tradeoff feedback on every record's recon comes from the direct residual (it feeds gamma = recon / manifold
    # back every step); the trace then stays at the float32-state bar for all 1000 iterations, as without the feedback
    assert e_tr < TOL_OBJ and eu < TOL_FACTOR and ev < TOL_FACTOR
    assert e_tr < PIN and eu < PIN and ev < PIN, (e_tr, eu, ev)
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


def _trace_of(stdout):
    return np.array([float(l.split()[1]) for l in stdout.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit()])


@pytest.mark.parametrize("variant,tag,norm", [(1, "g6", "nonorm"), (2, "g7", "nonorm"), (1, "g6", "norm"), (2, "g7", "norm")])
def test_cli_outputs_vs_reference(golden_dir, tmp_path, variant, tag, norm):
    """The reference's two CLI test inputs, with --no-normalize (70 iterations) and with the default
    quantile normalisation (370 / 320 iterations): stdout trace, obj.txt, mapping, U.csv, V.csv."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "%s_%s.npz" % (tag, norm)))
    c = synth.cli_case(str(tmp_path), variant=variant)
    out = str(tmp_path / "out")
    os.makedirs(out)
    stdout = _cli(["--data", c["data"], "--manifolds"] + c["manifolds"] +
                  ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t",
                   "--seed", "1"] + (["--no-normalize"] if norm == "nonorm" else []), str(tmp_path))
    assert "Printing manifold node representation in nodelist:" in stdout and "norm(X) = " in stdout
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)                               # same keys, same order
    trace = _trace_of(stdout)
    U = pd.read_csv(os.path.join(out, "U.csv"))
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
    eo = obj_txt_errors(vals, ref)
    ncmp = min(len(trace), len(g["trace"]))
    eu, ev, et = rel(U.to_numpy(), g["U"]), rel(V.to_numpy(), g["V"]), relmax(trace[:ncmp], g["trace"][:ncmp])
    report("cli_outputs_vs_reference[%s_%s]" % (tag, norm), iterations=len(trace), ref_iterations=len(g["trace"]),
           rel_trace=et, rel_U=eu, rel_V=ev, obj_txt=eo)
    assert len(trace) == len(g["trace"])
    assert mapping == json.loads(str(g["mapping"]))
    assert list(U.columns) == ["LV%d" % i for i in range(6)] and list(V.index) == c["nodelist_ids"]
    assert max(eo.values()) < TOL_OBJ, eo
    assert et < TOL_OBJ and eu < TOL_FACTOR and ev < TOL_FACTOR
    assert et < PIN and eu < PIN and ev < PIN, (et, eu, ev)
    head = open(os.path.join(out, "V.csv")).readline().strip()
    assert head == '"","LV0","LV1","LV2","LV3","LV4","LV5"'       # QUOTE_NONNUMERIC contract
    if variant == 2 and norm == "nonorm":
        # unmeasured genes (all-zero columns of X) pin their V rows at float32 eps
        assert np.allclose(V.to_numpy()[1000:], np.finfo(np.float32).eps)


@pytest.mark.parametrize("tag", ["g10_init", "g11_fewer_pathways", "g12_weighted", "g13_quirk_flags", "g15_seed7", "g16_seed2024_k4"])
def test_cli_more_variants_vs_reference(golden_dir, tmp_path, tag):
    """--manifolds-init (NNLS-based U/V start, init_pathways.txt), fewer pathways than latent factors,
    weighted graphml edges, the flags the reference parses but does not forward (--delta, --tolerence, --m-samples next
    to a --gamma it does): obj.txt, the latent->pathway mapping, U.csv and V.csv against reference runs."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    man = c["manifolds"]
    extra = []
    if tag == "g10_init":
        extra = ["--manifolds-init"] + [man[i] for i in (0, 1, 3, 2, 5, 4)]
    elif tag == "g11_fewer_pathways":
        man = man[:3]
    elif tag == "g13_quirk_flags":
        # parsed by the reference, but only --gamma reaches its solver: --delta and --tolerence are not passed on, --m-samples
        # sets n_rows and then reads every row -- the drop-in CLI has to ignore them in the same way
        extra = ["--gamma", "2.0", "--delta", "7.0", "--tolerence", "0.5", "--m-samples", "10"]
    elif tag == "g15_seed7":
        pass                                    # another seed: other initial factors, pathway draws, restrict / matching decisions
    elif tag == "g16_seed2024_k4":
        extra = ["-k", "4"]                     # fewer latent factors than pathways
    else:
        wdir = str(tmp_path / "weighted")
        os.makedirs(wdir)
        man = synth.weighted_copies(man, wdir)
    out = str(tmp_path / "out")
    os.makedirs(out)
    stdout = _cli(["--data", c["data"], "--manifolds"] + man + extra +
                  ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t",
                   "--seed", {"g15_seed7": "7", "g16_seed2024_k4": "2024"}.get(tag, "1"), "--no-normalize"], str(tmp_path))
    trace = _trace_of(stdout)
    ref_trace = g["trace"]
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)
    U = pd.read_csv(os.path.join(out, "U.csv")).to_numpy()
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0).to_numpy()
    eo = obj_txt_errors(vals, ref)
    ncmp = min(len(trace), len(ref_trace))
    eu, ev, et = rel(U, g["U"]), rel(V, g["V"]), relmax(trace[:ncmp], ref_trace[:ncmp])
    report("cli_more_variants_vs_reference[%s]" % tag, iterations=len(trace), ref_iterations=len(ref_trace),
           rel_trace=et, rel_U=eu, rel_V=ev, obj_txt=eo)
    assert len(trace) == len(ref_trace)
    assert mapping == json.loads(str(g["mapping"]))
    assert max(eo.values()) < TOL_OBJ, eo
    assert et < TOL_OBJ and eu < TOL_FACTOR and ev < TOL_FACTOR
    assert et < PIN and eu < PIN and ev < PIN, (et, eu, ev)
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
    eo = obj_txt_errors(vals, ref)
    report("cli_npy_input", obj_txt=eo)
    assert max(eo.values()) < TOL_OBJ, eo
    assert mapping == json.loads(str(g["mapping"]))


@pytest.mark.parametrize("variant,tag", [(1, "g6"), (2, "g7")])
def test_cli_header_inferred_nodelist(golden_dir, tmp_path, variant, tag):
    """No --nodelist: the gene ids come from the data file's header (prmf_runner.py:955-975); the reference's
    own test (test_inferred_nodelist_1.py:55-56) expects the obj.txt of the explicit-nodelist run."""
    g = np.load(os.path.join(golden_dir, tag + "_header.npz"))
    g_explicit = np.load(os.path.join(golden_dir, tag + "_nonorm.npz"))
    c = synth.cli_case(str(tmp_path), variant=variant)
    out = str(tmp_path / "out")
    os.makedirs(out)
    _cli(["--data", c["data_header"], "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--outdir", out,
                                                                         "--delimiter", "\t", "--seed", "1", "--no-normalize"],
         str(tmp_path))
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)
    eo = obj_txt_errors(vals, ref)
    report("cli_header_inferred_nodelist[%s]" % tag, obj_txt=eo)
    assert max(eo.values()) < TOL_OBJ, eo
    assert mapping == json.loads(str(g["mapping"]))
    assert json.loads(str(g["obj_txt"])) == json.loads(str(g_explicit["obj_txt"]))     # what the reference's test asserts


def test_cli_embeds_graph_only_genes(golden_dir, tmp_path):
    """Header lists only the measured genes; the pathways name six more.  The CLI must append them to the
    nodelist and zero-embed X (prmf_runner.py:962-972, prmf.embed_arr): V.csv then has 1006 rows in the
    reference's order and the graph-only genes sit at float32 eps."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "g7_measured.npz"))
    c = synth.cli_case(str(tmp_path), variant=2)
    out = str(tmp_path / "out")
    os.makedirs(out)
    stdout = _cli(["--data", c["data_header_measured"], "--manifolds"] + c["manifolds"] +
                  ["--node-attribute", "name", "--outdir", out, "--delimiter", "\t", "--seed", "1", "--no-normalize"],
                  str(tmp_path))
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
    trace = _trace_of(stdout)
    ref_index = json.loads(str(g["V_index"]))
    assert sorted(V.index) == sorted(ref_index)
    Vr = pd.DataFrame(g["V"], index=ref_index).loc[list(V.index)].to_numpy()    # graph-only genes are appended from a set
    eo = obj_txt_errors(vals, ref)
    ncmp = min(len(trace), len(g["trace"]))
    ev, et = rel(V.to_numpy(), Vr), relmax(trace[:ncmp], g["trace"][:ncmp])
    report("cli_embeds_graph_only_genes", iterations=len(trace), rel_trace=et, rel_V=ev, obj_txt=eo)
    assert len(trace) == len(g["trace"])
    assert mapping == json.loads(str(g["mapping"]))
    assert max(eo.values()) < TOL_OBJ and et < TOL_OBJ and ev < TOL_FACTOR
    extra = [x for x in V.index if int(x[4:]) >= 1000]
    assert len(extra) == 6 and np.allclose(V.loc[extra].to_numpy(), np.finfo(np.float32).eps)


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
    eo = obj_txt_errors(vals, ref)
    err = np.loadtxt(os.path.join(out, "test_error.csv"), delimiter=",")
    report("cli_config1_plumbing", obj_txt=eo, rel_test_error=relmax(err, g["test_error"]))
    assert max(eo.values()) < TOL_OBJ, eo
    assert mapping == json.loads(str(g["mapping"]))
    assert err.shape == (20,) and np.allclose(err, g["test_error"], rtol=1e-4)


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


def test_cli_library_cache(golden_dir, tmp_path):
    """--library-cache: the second run loads the pathway library from the .npz instead of rebuilding it from the graphs;
    outputs are byte-identical to the first run's and match the reference golden."""
    g = np.load(os.path.join(golden_dir, "g6_nonorm.npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    cache = str(tmp_path / "lib.npz")
    outs = []
    for run in range(2):
        out = str(tmp_path / ("out%d" % run))
        os.makedirs(out)
        _cli(["--data", c["data"], "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--nodelist", c["nodelist"],
                                                                      "--outdir", out, "--delimiter", "\t", "--seed", "1", "--no-normalize",
                                                                      "--library-cache", cache], str(tmp_path))
        assert os.path.exists(cache)
        outs.append([open(os.path.join(out, f)).read() for f in ("obj.txt", "U.csv", "V.csv")])
    assert outs[0] == outs[1]
    vals, mapping = _read_obj(str(tmp_path / "out1" / "obj.txt"))
    assert mapping == json.loads(str(g["mapping"]))
    assert max(obj_txt_errors(vals, json.loads(str(g["obj_txt"]))).values()) < TOL_OBJ


@pytest.mark.slow
@pytest.mark.parametrize("m,n,k,P,x_dtype", [(700, 3000, 12, 40, "f32"), (300, 9000, 20, 64, "bf16")])
def test_outer_solver_at_medium_size_vs_oracle(m, n, k, P, x_dtype):
    """The whole outer loop (prmf_runner.py:556-792: per-round pathway sampling, candidate scoring, `restrict`, the
    matching, best iterate, stopping rule) at a size the reference's own goldens do not reach (they are 100 x 1000, k = 6,
    6 pathways: one sweep workgroup): 40 / 64 candidate pathways per latent column, several `restrict` rounds -- against
    the oracle's nmf_pathway (pinned to the reference by the goldens) consuming the same legacy RNG stream: the same
    assignment in every round, the same number of iterations, the same final map; trace, U, V at the 1e-4 bar."""
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary
    from prmf_amd.solver import nmf_pathway
    case = synth.small_case(seed=31, m=m, n=n, k=k, P=P, order=60, bf16_x=(x_dtype == "bf16"))
    X = case["X"]
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    r_ref, t_ref = [], []
    np.random.seed(5)
    Ur, Vr, odr = orc.nmf_pathway(X, pws, gamma=1.0, k_latent=k, max_iter=300, round_log=r_ref, trace=t_ref)
    rounds, trace = [], []
    np.random.seed(5)
    U, V, od = nmf_pathway(X, None, library=lib, gamma=1.0, k_latent=k, max_iter=300, x_dtype=x_dtype, round_log=rounds, trace=trace)
    tr, tref = np.array([t["obj"] for t in trace]), np.array([t["obj"] for t in t_ref])
    assert rounds == r_ref, "per-round assignment differs in round %d" % next(i for i, (a, b) in enumerate(zip(rounds, r_ref)) if a != b)
    assert len(tr) == len(tref)
    mp, mr = od["latent_to_pathway_data"], odr["latent_to_pathway_data"]
    assert [[int(p) for p, _s in mp[kk]] for kk in sorted(mp)] == [[int(p) for p, _s in mr[kk]] for kk in sorted(mr)]
    e_tr, eu, ev = relmax(tr, tref), rel(U, Ur), rel(V, Vr)
    report("outer_solver_medium_vs_oracle[%dx%d-k%d-P%d-%s]" % (m, n, k, P, x_dtype), iterations=len(tr), rounds=len(rounds), rel_trace=e_tr, rel_U=eu, rel_V=ev)
    assert e_tr < TOL_OBJ and eu < TOL_FACTOR and ev < TOL_FACTOR


def test_cli_genes_by_samples_input_fails_where_the_reference_fails(golden_dir, tmp_path):
    """--data stored genes x samples (1000 x 100): --high-dimensional (default True) turns the matrix so that the shorter
    side is the samples (prmf_runner.py:945-952) and the solve runs -- its trace is the g6 run's -- but the sample labels
    were taken BEFORE the turn (:943), so writing U.csv raises pandas' shape error.  The reference does exactly that
    (tried with oracle/ref_harness.py while looking for a fixture: ValueError "Shape of passed values is (100, 6), indices
    imply (1000, 6)"); a drop-in keeps the error behaviour, it does not guess a repair."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "g6_nonorm.npz"))
    c = synth.cli_case(str(tmp_path), variant=1)
    data_t = str(tmp_path / "data_T.tsv")
    pd.DataFrame(c["X"].T).to_csv(data_t, sep="\t", header=False, index=False)
    out = str(tmp_path / "out")
    os.makedirs(out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py"), "--data", data_t, "--manifolds"] + c["manifolds"] +
                       ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t", "--seed", "1", "--no-normalize"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0
    assert "Shape of passed values is (100, 6), indices imply (1000, 6)" in r.stderr
    trace = _trace_of(r.stdout)
    assert len(trace) == len(g["trace"]) and relmax(trace, g["trace"]) < TOL_OBJ      # the solve itself ran on the turned matrix
    assert not os.path.exists(os.path.join(out, "U.csv"))
