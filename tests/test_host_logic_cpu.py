"""Host-side logic of the product (no GPU): pathway library construction, the
restrict / matching control code and the CLI front end, against the oracle and the
reference-generated goldens."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from oracle import prmf_oracle as orc
from prmf_amd import PathwayLibrary
from prmf_amd import runner
from prmf_amd import solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_from_edge_lists_matches_oracle_pathways():
    case = synth.small_case(seed=4, P=7)
    n = case["X"].shape[1]
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    assert lib.P == 7 and lib.sup_ptr[-1] == len(lib.sup_idx)
    for p, (e, w, nodes) in enumerate(case["pathways"]):
        pw = orc.pathway_from_edges(n, e, nodes, w)
        s0, s1 = lib.sup_ptr[p], lib.sup_ptr[p + 1]
        assert np.array_equal(lib.sup_idx[s0:s1], pw.support)
        assert np.allclose(lib.sup_deg[s0:s1], pw.deg[pw.support])
        assert np.allclose(lib.sup_ldiag[s0:s1], pw.L.diagonal()[pw.support])
        # rebuild W from the support-local CSR
        W = np.zeros((n, n))
        for s in range(s0, s1):
            for e_ in range(lib.row_ptr[s], lib.row_ptr[s + 1]):
                W[lib.sup_idx[s], lib.sup_idx[s0 + lib.col_local[e_]]] = lib.w[e_]
        assert np.allclose(W, pw.W.toarray())
        v = np.random.RandomState(p).rand(n)
        assert abs(lib.laplacian_quadratic(p, v) - pw.L.dot(v).dot(v)) < 1e-9
        # the batched form force_distinct_lapls uses (one sparse product per pathway): the same numbers, bit for bit
        Vm = np.random.RandomState(100 + p).rand(n, 9)
        cols = [7, 0, 3, 3, 8]
        assert lib.laplacian_quadratics(p, Vm, cols) == [lib.laplacian_quadratic(p, Vm[:, c]) for c in cols]


def test_library_from_graphs_and_from_sparse(tmp_path):
    import networkx as nx
    c = synth.cli_case(str(tmp_path), variant=2)
    Gs = [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]
    lib = PathwayLibrary.from_graphs(Gs, c["nodelist_ids"])
    pws = [orc.pathway_from_graph(G, c["nodelist_ids"]) for G in Gs]
    for p, pw in enumerate(pws):
        s0, s1 = lib.sup_ptr[p], lib.sup_ptr[p + 1]
        assert np.array_equal(lib.sup_idx[s0:s1], pw.support)
        assert np.allclose(lib.sup_deg[s0:s1], pw.deg[pw.support])
    # the isolated unmeasured node is in the support with degree 0
    assert (lib.sup_deg == 0).sum() == len(Gs)
    lib2 = PathwayLibrary.from_sparse([pw.W for pw in pws], [pw.support for pw in pws])
    for a, b in zip(lib.arrays(), lib2.arrays()):
        assert np.array_equal(a, b)
    bad = pws[0].W.tolil()
    outside = [i for i in range(lib.n) if i not in set(pws[0].support)][:2]
    bad[outside[0], outside[1]] = 1.0
    with pytest.raises(ValueError):
        PathwayLibrary.from_sparse([bad.tocsr()], [pws[0].support])


@pytest.mark.parametrize("name", ["g5_restrict.npz", "g19_restrict_k40.npz"])
def test_restrict_matches_reference_kept_sets(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    k, P = g["scores"].shape
    l2p = {kk: [(p, 1) for p in range(P)] for kk in range(k)}
    kept = solver.restrict(g["scores"], l2p)
    mask = np.zeros_like(g["kept"])
    for kk, lst in kept.items():
        for p, s in lst:
            mask[kk, p] = 1
            assert s == g["scores"][kk, p]
    assert np.array_equal(mask, g["kept"])
    # a latent with a single candidate is left alone; ties raise like the reference
    assert solver.restrict(g["scores"], {0: [(3, 2)]}) == {0: [(3, 2)]}
    # (sample size ceil(n*(1-19.9)/100) is negative from n = 6 candidates on, prmf_runner.py:179)
    with pytest.raises(ValueError):
        solver.restrict(np.ones((1, 8)), {0: [(p, 1) for p in range(8)]})


def test_force_distinct_matches_oracle():
    case = synth.small_case(seed=6, P=5)
    n, k = case["X"].shape[1], case["k"]
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    V = case["V0"]
    l2p_a = {kk: [(p, 1) for p in range(5)] for kk in range(k)}
    l2p_b = {kk: [(p, 1) for p in range(5)] for kk in range(k)}
    k_to_lapl = {kk: kk % 5 for kk in range(k)}
    a = solver.force_distinct_lapls(V, lib, l2p_a, k_to_lapl, case["gamma"], case["delta"])
    b = orc.force_distinct_lapls(V, pws, l2p_b, {kk: pws[k_to_lapl[kk]].support for kk in range(k)},
                                 case["gamma"], case["delta"])
    assert {kk: v[0][0] for kk, v in a.items()} == {kk: v[0][0] for kk, v in b.items()}
    assert len({v[0][0] for v in a.values()}) == k


def test_cli_front_end_helpers(tmp_path):
    c = synth.cli_case(str(tmp_path), variant=1)
    assert runner.check_header(c["data_header"], "\t") and not runner.check_header(c["data"], "\t")
    assert not runner.check_row_names(c["data"], "\t", False)
    with open(c["nodelist"]) as fh:
        assert runner.parse_nodelist(fh) == c["nodelist_ids"]
    arr = np.arange(6.0).reshape(2, 3)
    assert np.array_equal(runner.embed_arr(list("abcde"), list("eca"), arr), orc.embed_arr(list("abcde"), list("eca"), arr))


def _run_cli(args, cwd):
    return subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py")] + args,
                          cwd=cwd, capture_output=True, text=True)


def test_cli_exit_codes(tmp_path):
    c = synth.cli_case(str(tmp_path), variant=1)
    out = str(tmp_path)
    r = _run_cli(["--data", c["data"], "--outdir", out], out)
    assert r.returncode == 22 and "Exactly one of --manifolds" in r.stderr
    mf = os.path.join(out, "mf.txt")
    open(mf, "w").write("\n".join(c["manifolds"]))
    r = _run_cli(["--data", c["data"], "--outdir", out, "--manifolds", c["manifolds"][0], "--manifolds-file", mf], out)
    assert r.returncode == 23
    r = _run_cli(["--data", c["data"], "--outdir", out, "--manifolds", c["manifolds"][0], "--delimiter", "\t"], out)
    assert r.returncode == 25 and "--nodelist is not provided" in r.stderr
    bad_nodes = os.path.join(out, "bad_nodelist.txt")
    open(bad_nodes, "w").write("\n".join("X%d" % i for i in range(1000)))
    r = _run_cli(["--data", c["data"], "--outdir", out, "--manifolds", c["manifolds"][0], "--delimiter", "\t",
                  "--nodelist", bad_nodes], out)
    assert r.returncode == 24 and "Invalid manifolds" in r.stderr
    # the one limit this build has and the reference has not (prmf/prmf_args.py:14 takes any -k): more than 128 latent factors.
    # Declared before any input is read, with an exit code of its own; 128 itself is accepted by the parser stage.
    r = _run_cli(["--data", c["data"], "--outdir", out, "--manifolds", c["manifolds"][0], "--delimiter", "\t", "--nodelist", c["nodelist"],
                  "-k", "160"], out)
    assert r.returncode == 26 and "--k-latent 160 is not supported" in r.stderr and "at most 128" in r.stderr
    assert not os.path.exists(os.path.join(out, "U.csv"))


def test_library_refuses_more_than_128_latent_columns():
    """prmf_create returns PRMF_ERR_UNSUPPORTED for k > 128 (include/prmf_hip.h) -- checked before any device is touched"""
    import ctypes as C
    from prmf_amd import _capi
    lib = _capi.load()
    ctx = C.c_void_p()
    assert lib.prmf_create(C.byref(ctx), 0, 10, 10, 129, _capi.PRMF_X_F32, 1, 0, 0) == _capi.PRMF_ERR_UNSUPPORTED
    assert b"129 > 128" in lib.prmf_last_error(None) and not ctx.value


def test_verbose_candidate_table_format(capsys):
    """--verbose prints, after every restrict / matching round, the remaining candidates of each latent factor
    best score first (prmf_runner.py:67-88)."""
    solver.print_latent_to_pathway_data({0: [(3, 0.25), (1, 1.5)], 1: [(2, 2)]})
    out = capsys.readouterr().out.splitlines()
    assert out == ["-----------------------------", "Latent to pathway match data:", "-----------------------------",
                   "Latent vector: 0", "  1\t1.5", "  3\t0.25", "Latent vector: 1", "  2\t2"]


def test_pathway_library_cache_roundtrip(tmp_path):
    """the built library as one .npz -- every array the GPU gets and the host-side helpers survive a save / load;
    a cache built from other inputs (key mismatch) is refused."""
    import synth
    from prmf_amd import PathwayLibrary
    case = synth.small_case(seed=4, n=300, P=5, order=30)
    lib = PathwayLibrary.from_edge_lists(300, case["pathways"], names=["p%d" % i for i in range(5)])
    fp = str(tmp_path / "lib.npz")
    lib.save(fp, key="abc")
    assert PathwayLibrary.load(fp, key="other") is None and PathwayLibrary.load(str(tmp_path / "missing.npz"), key="abc") is None
    got = PathwayLibrary.load(fp, key="abc")
    assert got.n == lib.n and got.P == lib.P and got.names == lib.names and got.max_support == lib.max_support
    for a, b in zip(got.arrays(), lib.arrays()):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    v = np.random.RandomState(0).rand(300)
    assert all(got.laplacian_quadratic(p, v) == lib.laplacian_quadratic(p, v) for p in range(5))
    # a name without the suffix is the same cache (np.savez would have appended '.npz' behind load's back), a truncated or
    # corrupt archive is "no cache" (rebuilt), not a crash, and no temporary file is left behind
    bare = str(tmp_path / "cache_without_suffix")
    lib.save(bare, key="abc")
    assert PathwayLibrary.load(bare, key="abc").names == lib.names and (tmp_path / "cache_without_suffix.npz").exists()
    raw = open(fp, "rb").read()
    open(fp, "wb").write(raw[:len(raw) // 2])
    assert PathwayLibrary.load(fp, key="abc") is None
    open(fp, "wb").write(b"not a zip archive")
    assert PathwayLibrary.load(fp, key="abc") is None
    assert sorted(x.name for x in tmp_path.iterdir()) == ["cache_without_suffix.npz", "lib.npz"]


def test_fast_matching_equals_networkx():
    """force_distinct_lapls solves the latent <-> pathway matching as an assignment problem and proves the optimum unique
    before trusting it (solver._max_weight_matching); on random instances -- more latents than pathways, fewer, sparse
    candidate lists, exact ties, zero weights -- the result must be networkx.max_weight_matching's."""
    import networkx as nx
    rs = np.random.RandomState(3)
    for trial in range(60):
        K, P = rs.randint(1, 10), rs.randint(1, 30)
        per_latent = []
        for k in range(K):
            cand = sorted(rs.choice(P, size=rs.randint(1, P + 1), replace=False).tolist())
            if trial % 6 == 4:
                ws = [(p, float(rs.randint(1, 4))) for p in cand]              # ties
            elif trial % 6 == 5:
                ws = [(p, float(rs.rand() > 0.5) * rs.rand()) for p in cand]   # zero weights
            else:
                ws = [(p, float(rs.rand()) + 1e-3) for p in cand]
            per_latent.append((k, ws))
        G = nx.Graph()
        for k, ws in per_latent:
            for p, w in ws:
                G.add_edge("k%d" % k, "l%d" % p, weight=w)
        want = sorted((int(a[1:]), int(b[1:])) if a[0] == "k" else (int(b[1:]), int(a[1:])) for a, b in nx.max_weight_matching(G))
        assert sorted(solver._max_weight_matching(per_latent)) == want, trial


def _same_graph(a, b):
    assert type(a) is type(b)
    assert list(a.nodes()) == list(b.nodes())
    assert dict(a.nodes(data=True)) == dict(b.nodes(data=True))
    assert a.graph == b.graph
    for u in a.nodes():
        assert list(a.adj[u]) == list(b.adj[u]), u            # adjacency ORDER too
        for v in a.adj[u]:
            assert a.adj[u][v] == b.adj[u][v] and {k: type(x) for k, x in a.adj[u][v].items()} == {k: type(x) for k, x in b.adj[u][v].items()}
    assert list(a.edges(data=True)) == list(b.edges(data=True))


def test_fast_graphml_equals_networkx(tmp_path):
    """prmf_amd.graphml.read_pathway against the reference's three steps (prmf_runner.py:863-866): class, node order, node
    and edge attributes with their types, adjacency order, graph attributes -- for plain files built directly and for
    every kind of file that is handed to networkx instead."""
    import networkx as nx
    from prmf_amd import graphml
    from prmf_amd.runner import relabel_nodes
    rs = np.random.RandomState(4)

    def slow(fp, attr="name"):
        return relabel_nodes(nx.read_graphml(fp).to_undirected(), attr)

    fast_taken = 0
    files = []
    for case in range(12):
        n = int(rs.randint(1, 60))
        G = nx.Graph()
        order = rs.permutation(n)
        for i in order:
            attrs = {}
            if case % 4 != 3 or rs.rand() < 0.7:
                attrs["name"] = "ENSP%d" % (1000 + i)            # case 3, 7, 11: some nodes keep their id
            if case % 3 == 0:
                attrs["score"] = float(rs.rand())
                attrs["rank"] = int(rs.randint(100))
                attrs["flag"] = bool(rs.rand() < 0.5)
            G.add_node(int(i), **attrs)
        for _ in range(int(rs.randint(0, 3 * n + 1))):
            u, v = int(rs.randint(n)), int(rs.randint(n))
            if case % 5 == 4 or u != v:                           # case 4, 9: self loops too
                w = {}
                if case % 2:
                    w["weight"] = float(rs.rand()) if case % 4 == 1 else int(rs.randint(1, 9))
                G.add_edge(u, v, **w)
        fp = str(tmp_path / ("g%d.graphml" % case))
        nx.write_graphml(G, fp, named_key_ids=bool(case % 2))
        files.append(fp)
    # hand-written, like the reference's docstring example (:838-848), edges with ids, an empty <data/>, another attribute name
    fp = str(tmp_path / "hand.graphml")
    with open(fp, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<graphml xmlns="http://graphml.graphdrawing.org/xmlns">\n'
                 '<key attr.name="name" attr.type="string" for="node" id="d1"/><key attr.name="weight" attr.type="double" for="edge" id="d2"/>'
                 '<key attr.name="note" attr.type="string" for="edge" id="d3"/>\n<graph edgedefault="undirected">\n'
                 '<node id="11485"><data key="d1">ENSG11486</data></node><node id="7"><data key="d1">ENSG7</data></node><node id="9"/>\n'
                 '<edge id="e0" source="7" target="11485"><data key="d2">0.25</data></edge><edge source="9" target="7"><data key="d3"/></edge>'
                 '<edge id="e2" source="9" target="9"/>\n</graph></graphml>\n')
    files.append(fp)
    real_fast = graphml._fast
    for fp in files:
        for attr in ("name", None, "rank"):
            _same_graph(graphml.read_pathway(fp, attr), slow(fp, attr))
            try:
                real_fast(fp, attr)
                fast_taken += 1
            except graphml._NotPlain:
                pass
    assert fast_taken >= 30                                   # the plain files did take the direct path
    # not plain: each must come out as networkx makes it (or raise what networkx raises)
    head = '<?xml version="1.0"?>\n<graphml xmlns="http://graphml.graphdrawing.org/xmlns">\n<key attr.name="name" attr.type="string" for="node" id="d1"/>\n'
    odd = {
        "directed": head + '<graph edgedefault="directed"><node id="a"/><node id="b"/><edge source="a" target="b"/><edge source="b" target="a"/></graph></graphml>',
        "parallel edges": head + '<graph edgedefault="undirected"><node id="a"/><node id="b"/><edge source="a" target="b"/><edge source="b" target="a"/></graph></graphml>',
        "undeclared endpoint": head + '<graph edgedefault="undirected"><node id="a"/><edge source="a" target="zz"/></graph></graphml>',
        "two nodes, one name": head + '<graph edgedefault="undirected"><node id="a"><data key="d1">X</data></node><node id="b"><data key="d1">X</data></node>'
                                      '<node id="c"/><edge source="a" target="c"/><edge source="b" target="c"/></graph></graphml>',
        "name equal to another id": head + '<graph edgedefault="undirected"><node id="a"><data key="d1">b</data></node><node id="b"/><edge source="a" target="b"/></graph></graphml>',
        "default value": head.replace('id="d1"/>', 'id="d1"><default>none</default></key>') + '<graph edgedefault="undirected"><node id="a"/></graph></graphml>',
        "duplicate node id": head + '<graph edgedefault="undirected"><node id="a"><data key="d1">X</data></node><node id="a"/></graph></graphml>',
        "graph data": head + '<key attr.name="title" attr.type="string" for="graph" id="d9"/><graph edgedefault="undirected"><data key="d9">t</data><node id="a"/></graph></graphml>',
        "two graphs": head + '<graph edgedefault="undirected"><node id="a"/></graph><graph edgedefault="undirected"><node id="b"/></graph></graphml>',
        "untyped key": head.replace(' attr.type="string"', '') + '<graph edgedefault="undirected"><node id="a"><data key="d1">A</data></node></graph></graphml>',
    }
    for what, text in odd.items():
        fp = str(tmp_path / "odd.graphml")
        with open(fp, "w") as fh:
            fh.write(text)
        with pytest.raises(graphml._NotPlain):
            graphml._fast(fp, "name")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _same_graph(graphml.read_pathway(fp, "name"), slow(fp, "name"))
    # what networkx refuses is refused in its words
    for text in ("not xml at all", head + '<graph edgedefault="undirected"><node id="a"><data key="nokey">1</data></node></graph></graphml>'):
        fp = str(tmp_path / "bad.graphml")
        with open(fp, "w") as fh:
            fh.write(text)
        with pytest.raises(Exception) as mine:
            graphml.read_pathway(fp, "name")
        with pytest.raises(Exception) as theirs:
            slow(fp)
        assert type(mine.value) is type(theirs.value) and str(mine.value) == str(theirs.value)


def test_multinomial_draw_equals_scipy():
    """solver.multinomial_one against scipy.stats.multinomial.rvs(1, p) -- the reference's draw (prmf_runner.py:723) -- on the
    same legacy stream: the same one-hot vectors, draw by draw, and the same stream position afterwards; probabilities as the
    outer loop makes them (scores / sum), with zeros, a single candidate, sums that are off by rounding."""
    from scipy.stats import multinomial
    from prmf_amd.solver import multinomial_one
    rs = np.random.RandomState(3)
    cases = []
    for _ in range(400):
        n = int(rs.randint(1, 40))
        scores = rs.rand(n) * 10.0 ** rs.randint(-3, 4)
        if rs.rand() < 0.3:
            scores[rs.randint(n)] = 0.0
        if scores.sum() == 0:
            scores[0] = 1.0
        cases.append(scores / np.sum(scores))
    cases += [np.array([1.0]), np.array([0.5, 0.5]), np.array([0.1] * 10), np.array([1 / 3.0] * 3), np.array([1.0, 0.0, 0.0]),
              np.array([0.0, 0.0, 1.0]), np.array([0.2, 0.3, 0.5 + 1e-16]), np.array([0.25, 0.25, 0.25, 0.25 - 1e-12])]
    np.random.seed(17)
    want = [multinomial.rvs(1, p) for p in cases]
    after_want = np.random.rand()
    np.random.seed(17)
    got = [multinomial_one(p) for p in cases]
    after_got = np.random.rand()
    assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got, want))
    assert after_got == after_want


def test_hold_out_equals_scikit_learn_kfold():
    """runner.hold_out (the --cross-validation split, prmf_runner.py:1002-1013) against KFold(n_splits=round(1/f)) itself."""
    from sklearn.model_selection import KFold
    from prmf_amd.runner import hold_out
    for n in (2, 3, 7, 10, 99, 100, 101, 1000):
        X = np.arange(n * 3, dtype=float).reshape(n, 3)
        samples = ["s%d" % i for i in range(n)]
        for f in (0.5, 0.34, 0.2, 0.1, 0.05, 0.013):
            k = round(1 / f)
            if k > n:
                with pytest.raises(ValueError):
                    hold_out(X, samples, f)
                continue
            train, test = next(KFold(n_splits=k).split(X))
            a, b, c = hold_out(X, samples, f)
            assert np.array_equal(a, X[train]) and np.array_equal(b, X[test]) and c == [samples[i] for i in train]
    with pytest.raises(ValueError):
        hold_out(np.zeros((10, 2)), list(range(10)), 0.9)          # one fold: scikit-learn refuses


@pytest.mark.parametrize("what,cases", [("write", 400), ("graphml", 250)])
def test_bounded_run_of_the_host_io_fuzzer(what, cases):
    """tools/fuzz_host_io.py: the CSV writer against DataFrame.to_csv (the same bytes or a refusal) and the GraphML reader against
    networkx's three steps (the same graph or the same exception) on random and mutated inputs; long runs: profiles/r05_fuzz_text.txt."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_host_io.py"), what], env=dict(os.environ, SEED="5", CASES=str(cases)),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("%d " % cases) in r.stdout.splitlines()[-1], (r.stdout + r.stderr)[-1500:]
