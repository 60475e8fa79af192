#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE
(/root/reference, imported unmodified through oracle/ref_harness.py) on seeded
inputs from tests/synth.py.

Run in the build container only (the reference does not travel):
    python oracle/gen_golden.py

Fixtures are data only -- inputs are regenerated from seeds by tests/synth.py
(a checksum of each regenerated input is stored), expected outputs are stored
as float64 arrays.  Library versions are recorded in every fixture.

  G1/G2  inner loop, 1 and 25 steps      (prmf_runner.py:374-451)
  G3     same with an all-zero sample row (U -> 0 path, den!=0 guard)
  G4     tradeoff variant, 15 steps       (:497-554)
  G5     restrict: k x P score matrix + kept candidate sets (:129-194, :115-127)
  G6     full nmf_pathway trace on the reference's test-1 generator
         (--no-normalize and default quantile-normalised), plus CLI outputs
  G7     same for the test-2 generator (unmeasured nodes -> zero columns)
  G8     config-1 plumbing: header-inferred nodelist, -k 4, 2 manifolds,
         --cross-validation 0.2
  G9     G2 with X pre-rounded to bf16
"""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import scipy
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from oracle import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def versions():
    import networkx
    import sklearn
    import pandas
    return json.dumps({"numpy": np.__version__, "scipy": scipy.__version__,
                       "networkx": networkx.__version__, "sklearn": sklearn.__version__,
                       "pandas": pandas.__version__, "python": sys.version.split()[0]})


def ref_graph_inputs(n, pathways, active_idx):
    """Build the dicts the reference's inner loop expects, the way
    nmf_pathway does at prmf_runner.py:679-690 (W CSR, D dok diagonal, L=D-W)."""
    k_to_W, k_to_D, k_to_L, k_to_feat = {}, {}, {}, {}
    for k, p in enumerate(active_idx):
        edges, w, nodes = pathways[p]
        u, v = edges[:, 0], edges[:, 1]
        W = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([u, v]), np.concatenate([v, u]))),
                          shape=(n, n)).tocsr()
        data = W.sum(axis=0)
        D = sp.dok_matrix(sp.dia_matrix((data, np.array([0])), shape=(n, n)))
        k_to_W[k] = W
        k_to_D[k] = D
        k_to_L[k] = D - W
        k_to_feat[k] = [int(x) for x in nodes]
    return k_to_W, k_to_D, k_to_L, k_to_feat


OBJ_KEYS = ["recon", "manifold", "ignore", "fro", "gamma", "delta", "obj"]


def obj_row(od):
    return [float(od[key]) for key in OBJ_KEYS]


def run_inner(ref, case, n_steps, snap_steps, tradeoff=None):
    X, U, V = case["X"], case["U0"].copy(), case["V0"].copy()
    n = X.shape[1]
    active = list(range(case["k"]))
    kW, kD, kL, kF = ref_graph_inputs(n, case["pathways"], active)
    gamma, delta = case["gamma"], case["delta"]
    objs, snaps_U, snaps_V = [], [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for step in range(1, n_steps + 1):
            if tradeoff is None:
                U, V, od = ref.nmf_manifold_vec_update(X, U, V, kW, kD, kL, kF, n_steps=1,
                                                       gamma=gamma, delta=delta, i=step - 1)
            else:
                U, V, od, g2, d2 = ref.nmf_manifold_vec_update_tradeoff(
                    X, U, V, kW, kD, kL, kF, n_steps=1, i=step - 1, tradeoff=tradeoff,
                    gamma=gamma, delta=delta)
            objs.append(obj_row(od))
            if tradeoff is not None:
                gamma, delta = g2, d2
            if step in snap_steps:
                snaps_U.append(U.copy())
                snaps_V.append(V.copy())
    return np.array(objs), np.array(snaps_U), np.array(snaps_V), (gamma, delta)


def save(name, **arrs):
    arrs["versions"] = np.array(versions())
    np.savez_compressed(os.path.join(GOLD, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


WIDE_INNER = {      # G17 / G18 (round 5): the reference's inner loop at LARGER k -- the shapes that take the other GPU code paths (k = 100: three
    # ... four 32-column tiles, 16x16x32 MFMA, column-split epilogues; k = 48 with f32 X: the three-piece split on 16x16x32)
    "g17_k100_bf16": dict(seed=3, m=80, n=900, k=100, P=100, order=30, bf16_x=True),
    "g18_k48": dict(seed=4, m=120, n=1200, k=48, P=48, order=40),
}


def gen_inner_wide(ref, only=None):
    for name, kw in WIDE_INNER.items():
        if only and name not in only:
            continue
        case = synth.small_case(**kw)
        objs, sU, sV, _ = run_inner(ref, case, 6, [6])
        save(name + ".npz", x_sha=np.array(synth.sha16(case["X"])), snap_steps=np.array([6]), obj=objs, U=sU, V=sV,
             case_kwargs=np.array(json.dumps(kw)))


def gen_inner(ref):
    snap = [1, 5, 10, 25]
    for name, kw in [("g2_inner25", {}), ("g3_zero_row", {"zero_row": True}),
                     ("g9_bf16x", {"bf16_x": True})]:
        case = synth.small_case(seed=1, **kw)
        objs, sU, sV, _ = run_inner(ref, case, 25, snap)
        save(name + ".npz", x_sha=np.array(synth.sha16(case["X"])), snap_steps=np.array(snap),
             obj=objs, U=sU, V=sV, case_kwargs=np.array(json.dumps(kw)))
    # G1 == first row/snapshot of G2, kept as an explicit tiny fixture with a second seed
    case = synth.small_case(seed=2)
    objs, sU, sV, _ = run_inner(ref, case, 1, [1])
    save("g1_inner1.npz", x_sha=np.array(synth.sha16(case["X"])), snap_steps=np.array([1]),
         obj=objs, U=sU, V=sV, case_kwargs=np.array(json.dumps({"seed": 2})))
    # G4 tradeoff
    case = synth.small_case(seed=1)
    objs, sU, sV, gd = run_inner(ref, case, 15, [1, 5, 15], tradeoff=0.7)
    save("g4_tradeoff15.npz", x_sha=np.array(synth.sha16(case["X"])), snap_steps=np.array([1, 5, 15]),
         obj=objs, U=sU, V=sV, final_gamma_delta=np.array(gd), tradeoff=np.array(0.7))


RESTRICT_CASES = {"g5_restrict": dict(seed=3, P=12),
                  # G19 (round 5): the same at k = 40 latent columns x 60 candidate pathways of 50 genes on 1500 genes
                  "g19_restrict_k40": dict(seed=5, m=60, n=1500, k=40, P=60, order=50)}


def gen_restrict(ref, name="g5_restrict"):
    """G5 / G19: scores for every (latent, pathway) pair and the kept sets of one
    `restrict` call, on V after 10 inner steps of the small case (G5: P = 12)."""
    case = synth.small_case(**RESTRICT_CASES[name])
    X, n, k = case["X"], case["X"].shape[1], case["k"]
    objs, sU, sV, _ = run_inner(ref, case, 10, [10])
    V = sV[0]
    P = len(case["pathways"])
    kW, kD, kL, kF = ref_graph_inputs(n, case["pathways"], list(range(P)))
    ref.PATHWAY_TO_SUPPORT = kF
    ref.LAPLACIANS = [kL[p] for p in range(P)]
    ref.NORMALIZED_LAPLACIANS = [ref.normalize_laplacian(kL[p], kF[p]) for p in range(P)]
    scores = np.zeros((k, P))
    for kk in range(k):
        for p in range(P):
            scores[kk, p] = ref.score_latent_pathway_match_global(V[:, kk], p)
    l2p = ref.init_latent_to_pathway_data(k, ref.LAPLACIANS)
    kept = ref.restrict(V, ref.LAPLACIANS, l2p, kF)
    kept_mask = np.zeros((k, P), dtype=np.int8)
    for kk, lst in kept.items():
        for p, _s in lst:
            kept_mask[kk, p] = 1
    ref_harness.reset_reference_globals(ref)
    save(name + ".npz", x_sha=np.array(synth.sha16(X)), V=V, scores=scores, kept=kept_mask)


def parse_trace(text):
    """'<iter> <obj>' lines of the reference's stdout (prmf_runner.py:447)."""
    out = []
    for line in text.splitlines():
        w = line.split()
        if len(w) == 2 and w[0].isdigit():
            try:
                out.append(float(w[1]))
            except ValueError:
                pass
    return np.array(out)


def run_ref_cli(ref, argv, cwd):
    old_argv, old_cwd = sys.argv, os.getcwd()
    buf = io.StringIO()
    err = io.StringIO()
    ref_harness.reset_reference_globals(ref)
    try:
        os.chdir(cwd)
        sys.argv = ["prmf_runner.py"] + argv
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):
            ref.main()
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    return buf.getvalue()


def read_obj_txt(fp):
    vals, mapping = {}, []
    for line in open(fp):
        line = line.rstrip()
        if " = " in line:
            a, b = line.split(" = ")
            vals[a] = float(b)
        elif "->" in line:
            a, b = line.split("->")
            mapping.append((int(a), os.path.basename(b.strip())))
    return vals, mapping


def gen_cli(ref):
    import pandas as pd
    tmp = tempfile.mkdtemp(prefix="prmf_gold_")
    try:
        # ----- G6: test-1 generator -----
        for variant, tag in [(1, "g6"), (2, "g7")]:
            d = os.path.join(tmp, tag)
            c = synth.cli_case(d, variant=variant)
            for norm_tag, extra in [("nonorm", ["--no-normalize"]), ("norm", [])]:
                out = os.path.join(d, "out_" + norm_tag)
                os.makedirs(out)
                argv = ["--data", c["data"], "--manifolds"] + c["manifolds"] + \
                       ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out,
                        "--delimiter", "\t", "--seed", "1"] + extra
                stdout = run_ref_cli(ref, argv, d)
                trace = parse_trace(stdout)
                vals, mapping = read_obj_txt(os.path.join(out, "obj.txt"))
                U = pd.read_csv(os.path.join(out, "U.csv")).to_numpy()
                V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0).to_numpy()
                save("{}_{}.npz".format(tag, norm_tag), x_sha=np.array(synth.sha16(c["X"])),
                     trace=trace, obj_txt=np.array(json.dumps(vals)),
                     mapping=np.array(json.dumps(mapping)), U=U, V=V,
                     obj_txt_raw=np.array(open(os.path.join(out, "obj.txt")).read().replace(d + os.sep, "")))
            # header-inferred nodelist run must give the same obj.txt (reference test :55-56)
            out = os.path.join(d, "out_header")
            os.makedirs(out)
            argv = ["--data", c["data_header"], "--manifolds"] + c["manifolds"] + \
                   ["--node-attribute", "name", "--outdir", out, "--delimiter", "\t", "--seed", "1",
                    "--no-normalize"]
            run_ref_cli(ref, argv, d)
            vals_h, mapping_h = read_obj_txt(os.path.join(out, "obj.txt"))
            save("{}_header.npz".format(tag), obj_txt=np.array(json.dumps(vals_h)),
                 mapping=np.array(json.dumps(mapping_h)))
            if variant == 2:
                # graph-only nodes absent from the header -> CLI embeds X (embed_arr)
                out = os.path.join(d, "out_measured")
                os.makedirs(out)
                argv = ["--data", c["data_header_measured"], "--manifolds"] + c["manifolds"] + \
                       ["--node-attribute", "name", "--outdir", out, "--delimiter", "\t", "--seed", "1",
                        "--no-normalize"]
                stdout = run_ref_cli(ref, argv, d)
                vals_m, mapping_m = read_obj_txt(os.path.join(out, "obj.txt"))
                V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
                save("g7_measured.npz", obj_txt=np.array(json.dumps(vals_m)),
                     mapping=np.array(json.dumps(mapping_m)), trace=parse_trace(stdout),
                     V=V.to_numpy(), V_index=np.array(json.dumps(list(V.index))))
        # ----- G8: config-1 plumbing -----
        d = os.path.join(tmp, "g8")
        c = synth.cli_case(d, variant=1)
        out = os.path.join(d, "out")
        os.makedirs(out)
        argv = ["--data", c["data_header"], "--manifolds"] + c["manifolds"][:2] + \
               ["--node-attribute", "name", "--outdir", out, "--delimiter", "\t", "--seed", "1",
                "-k", "4", "--cross-validation", "0.2"]
        stdout = run_ref_cli(ref, argv, d)
        vals, mapping = read_obj_txt(os.path.join(out, "obj.txt"))
        U = pd.read_csv(os.path.join(out, "U.csv")).to_numpy()
        V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0).to_numpy()
        test_err = np.loadtxt(os.path.join(out, "test_error.csv"), delimiter=",")
        save("g8_config1.npz", trace=parse_trace(stdout), obj_txt=np.array(json.dumps(vals)),
             mapping=np.array(json.dumps(mapping)), U=U, V=V, test_error=test_err,
             stdout_head=np.array("\n".join(stdout.splitlines()[:4]).replace(d + os.sep, "")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gen_solver_api(ref):
    """G6b: nmf_pathway called through its Python API (no CLI), with the per-round
    active-pathway assignment captured -- what tests compare the oracle's and the
    product's outer loops against."""
    import networkx as nx
    tmp = tempfile.mkdtemp(prefix="prmf_gold_")
    try:
        c = synth.cli_case(tmp, variant=1)
        for tag, tradeoff in [("g6_api", None), ("g6_api_tradeoff", 0.5)]:
            Gs = []
            for fp in c["manifolds"]:
                G = nx.read_graphml(fp).to_undirected()
                G = nx.relabel_nodes(G, {x: G.nodes[x]["name"] for x in G.nodes()})
                Gs.append(G)
            ref_harness.reset_reference_globals(ref)
            rounds = []
            orig = ref.map_k_to_lapls

            def spy(k_to_lapl_ind, *a, **kw):
                rounds.append([int(k_to_lapl_ind[k]) for k in sorted(k_to_lapl_ind)])
                return orig(k_to_lapl_ind, *a, **kw)

            ref.map_k_to_lapls = spy
            buf = io.StringIO()
            try:
                np.random.seed(1)
                with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                    U, V, od = ref.nmf_pathway(c["X"], Gs, nodelist=c["nodelist_ids"], gamma=1.0,
                                               tradeoff=tradeoff, k_latent=6)
            finally:
                ref.map_k_to_lapls = orig
            l2p = od.pop("latent_to_pathway_data")
            final_map = [int(l2p[k][0][0]) for k in sorted(l2p)]
            save(tag + ".npz", x_sha=np.array(synth.sha16(c["X"])), trace=parse_trace(buf.getvalue()),
                 rounds=np.array(rounds), final_map=np.array(final_map), obj=np.array(obj_row(od)),
                 U=U, V=V)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gen_more_cli(ref):
    """G10: --manifolds-init (all 6 given in permuted order -> NNLS-based U/V initialisation, :1040-1059); G11: fewer pathways
    than latent factors (3 manifolds, k = 6: only 3 latents get a final pathway, prmf_runner.py
    Appendix-B4 behaviour); G12: weighted edges read from graphml."""
    import networkx as nx
    import pandas as pd
    tmp = tempfile.mkdtemp(prefix="prmf_gold_")
    try:
        d = os.path.join(tmp, "g10")
        c = synth.cli_case(d, variant=1)
        runs = {
            # all k given, permuted: the < k branch draws with random.sample from a set() of file names, whose
            # order depends on PYTHONHASHSEED, so only the == k branch is reproducible across processes
            "g10_init": ["--manifolds"] + c["manifolds"] + ["--manifolds-init"] + [c["manifolds"][i] for i in (0, 1, 3, 2, 5, 4)],
            "g11_fewer_pathways": ["--manifolds"] + c["manifolds"][:3],
        }
        # weighted copies of the graphs (weights in (0,1], 3 decimals)
        wdir = os.path.join(d, "weighted")
        os.makedirs(wdir)
        wfps = synth.weighted_copies(c["manifolds"], wdir)
        runs["g12_weighted"] = ["--manifolds"] + wfps
        # G13: the flags the reference parses but treats specially: --gamma reaches the solver, --delta and
        # --tolerence do not (main passes neither to nmf_pathway, prmf_runner.py:1067), --m-samples sets n_rows and then reads
        # with nrows=None (:933-942): the run is the --gamma 2 run of all samples with delta = 1 and tol = 1e-3
        runs["g13_quirk_flags"] = ["--manifolds"] + c["manifolds"] + ["--gamma", "2.0", "--delta", "7.0", "--tolerence", "0.5", "--m-samples", "10"]
        # G15 / G16: other seeds -- other initial factors, other pathway draws per round, other `restrict` / matching decisions
        runs["g15_seed7"] = ["--manifolds"] + c["manifolds"]
        runs["g16_seed2024_k4"] = ["--manifolds"] + c["manifolds"] + ["-k", "4"]      # fewer latent factors than pathways
        seeds = {"g15_seed7": "7", "g16_seed2024_k4": "2024"}
        if ONLY:
            runs = {k: v for k, v in runs.items() if k in ONLY}
        for tag, margs in runs.items():
            out = os.path.join(d, "out_" + tag)
            os.makedirs(out)
            argv = ["--data", c["data"]] + margs + ["--node-attribute", "name", "--nodelist", c["nodelist"],
                                                    "--outdir", out, "--delimiter", "\t", "--seed", seeds.get(tag, "1"), "--no-normalize"]
            stdout = run_ref_cli(ref, argv, d)
            vals, mapping = read_obj_txt(os.path.join(out, "obj.txt"))
            U = pd.read_csv(os.path.join(out, "U.csv")).to_numpy()
            V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0).to_numpy()
            extra = {}
            ip = os.path.join(out, "init_pathways.txt")
            if os.path.exists(ip):
                extra["init_pathways"] = np.array(json.dumps([os.path.basename(x) for x in open(ip).read().split("\n")]))
            save(tag + ".npz", x_sha=np.array(synth.sha16(c["X"])), trace=parse_trace(stdout),
                 obj_txt=np.array(json.dumps(vals)), mapping=np.array(json.dumps(mapping)), U=U, V=V, **extra)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]      # e.g. `python oracle/gen_golden.py g13_quirk_flags`: only that CLI fixture


def main():
    if not ref_harness.reference_available():
        sys.exit("reference not mounted; golden vectors can only be generated in the build container")
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_harness.load_reference_runner()
    if ONLY:
        if any(t in WIDE_INNER for t in ONLY):
            gen_inner_wide(ref, ONLY)
        for t in ONLY:
            if t in RESTRICT_CASES:
                gen_restrict(ref, t)
        if any(t not in WIDE_INNER and t not in RESTRICT_CASES for t in ONLY):
            gen_more_cli(ref)
        return
    gen_inner(ref)
    gen_inner_wide(ref)
    gen_restrict(ref, "g19_restrict_k40")
    gen_restrict(ref)
    gen_solver_api(ref)
    gen_cli(ref)
    gen_more_cli(ref)


if __name__ == "__main__":
    main()
