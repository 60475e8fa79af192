"""Deterministic synthetic inputs shared by the golden generator and the tests.

Nothing here depends on the reference or on the oracle: plain numpy / networkx.
All randomness is `np.random.RandomState(seed)` or the legacy global stream
(`np.random.seed`) where the reference's own test generator uses it, both of
which are stable across numpy versions.
"""
import hashlib
import os

import numpy as np


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ---------------------------------------------------------------------------
# small inner-loop case (G1-G4, G9)
# ---------------------------------------------------------------------------
def small_case(seed=1, m=60, n=400, k=5, P=5, order=40, zero_genes=7, isolated=2,
               zero_row=False, bf16_x=False):
    """X (m x n) positive with `zero_genes` all-zero columns; P tree pathways of
    `order` nodes with mixed int/float weights and `isolated` extra support nodes
    that have no edge; U0, V0 as the reference initialises them (values in (0,3]).

    Returns dict with X, U0, V0, pathways=[(edges, weights, nodes)], gamma, delta."""
    rs = np.random.RandomState(seed)
    r = max(2, k // 2)
    A = rs.gamma(2.0, size=(m, r))
    B = rs.gamma(2.0, size=(r, n))
    X = A @ B
    X /= X.max()
    X += 0.05 * rs.rand(m, n)
    zg = rs.choice(n, size=zero_genes, replace=False)
    X[:, zg] = 0.0
    if zero_row:
        X[3, :] = 0.0
    if bf16_x:
        X = round_to_bf16(X)
    pathways = []
    for p in range(P):
        nodes = rs.choice(n, size=order + isolated, replace=False)
        tree = nodes[:order]
        parent = np.array([rs.randint(0, i) for i in range(1, order)], dtype=np.int64)
        edges = np.stack([tree[1:], tree[parent]], axis=1)
        if p % 2 == 0:
            w = np.ones(len(edges))
        else:
            w = np.round(0.25 + 0.75 * rs.rand(len(edges)), 3)
        pathways.append((edges, w, nodes))
    U0 = 3 * (1 - rs.rand(m, k))
    V0 = 3 * (1 - rs.rand(n, k))
    norm_X = np.linalg.norm(X)
    return {"X": X, "U0": U0, "V0": V0, "pathways": pathways,
            "gamma": norm_X / k, "delta": 10.0 / norm_X, "k": k}


def round_to_bf16(a):
    """Round float64/float32 values to the nearest bfloat16 (ties to even) and
    return them as float64 -- the same rounding torch/HIP `float -> bf16` does."""
    f = np.asarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


# ---------------------------------------------------------------------------
# the reference's CLI test generators, restated
# (test/script/PRMF/test_inferred_nodelist_1/test_inferred_nodelist_1.py:9-45,
#  test_inferred_nodelist_2.py:12-56)
# ---------------------------------------------------------------------------
def cli_case(outdir, variant=1, m_samples=100, n_genes=1000, k_latent=6, shape_param=5):
    """Write data.tsv / data_header.tsv / nodelist.txt / graph{k}.graphml into
    `outdir` the way the reference's test scripts do; returns dict of paths + X.

    variant 2 adds one isolated, unmeasured node per pathway and zero-embeds X
    (the reference's test 2: all-zero columns of X, V rows pinned at EPS)."""
    import networkx as nx
    import pandas as pd
    from scipy.stats import gamma

    np.random.seed(seed=1)
    U = gamma.rvs(shape_param, size=m_samples * k_latent).reshape(m_samples, k_latent)
    V = gamma.rvs(shape_param, size=n_genes * k_latent).reshape(n_genes, k_latent)
    X = U.dot(V.transpose())
    measured = ["ENSP" + str(x) for x in range(n_genes)]
    nodelist = list(measured)
    if variant == 2:
        nodelist += ["ENSP" + str(n_genes + k) for k in range(k_latent)]
    X_full = np.zeros((m_samples, len(nodelist)))
    X_full[:, :n_genes] = X
    index = [str(x) for x in range(m_samples)]
    X_df = pd.DataFrame(data=X_full, index=index, columns=nodelist)
    os.makedirs(outdir, exist_ok=True)
    data_fp = os.path.join(outdir, "data.tsv")
    header_fp = os.path.join(outdir, "data_header.tsv")
    measured_fp = os.path.join(outdir, "data_header_measured.tsv")
    X_df.to_csv(data_fp, sep="\t", header=False, index=False)
    X_df.to_csv(header_fp, sep="\t", header=nodelist, index=False)
    # only the measured genes in the header: the CLI must add the graph-only
    # nodes to the nodelist and zero-embed X (prmf_runner.py:962-972)
    pd.DataFrame(data=X, index=index, columns=measured).to_csv(
        measured_fp, sep="\t", header=measured, index=False)
    manifold_fps = []
    for k in range(k_latent):
        inds = np.where(V[:, k] > np.percentile(V[:, k], 95))[0]
        G = nx.path_graph(len(inds))
        for i in range(len(inds)):
            G.nodes[i]["name"] = "ENSP" + str(inds[i])
        if variant == 2:
            G.add_node(n_genes + k, name="ENSP" + str(n_genes + k))   # isolated, unmeasured
        fp = os.path.join(outdir, "graph{}.graphml".format(k))
        nx.write_graphml(G, fp)
        manifold_fps.append(fp)
    nodelist_fp = os.path.join(outdir, "nodelist.txt")
    with open(nodelist_fp, "w") as fh:
        fh.write("\n".join(nodelist))
    return {"data": data_fp, "data_header": header_fp, "data_header_measured": measured_fp,
            "nodelist": nodelist_fp, "manifolds": manifold_fps, "X": X_full,
            "nodelist_ids": nodelist}


def weighted_copies(manifold_fps, outdir):
    """Copies of the graphml files with a deterministic 'weight' attribute in (0,1] on every edge."""
    import networkx as nx
    out = []
    for gi, fp in enumerate(manifold_fps):
        G = nx.read_graphml(fp)
        rs = np.random.RandomState(100 + gi)
        for a, b in G.edges():
            G[a][b]["weight"] = float(np.round(0.2 + 0.8 * rs.rand(), 3))
        ofp = os.path.join(outdir, os.path.basename(fp))
        nx.write_graphml(G, ofp)
        out.append(ofp)
    return out


# ---------------------------------------------------------------------------
# benchmark-shaped inputs, generated in row chunks
# ---------------------------------------------------------------------------
def bench_x_rows(seed, m, n, k, row0, row1):
    """Rows [row0,row1) of the synthetic X of SURVEY.md 8(d):
    X = A@B/max + 0.05*rand, A~Gamma(2) (m x r), B~Gamma(2) (r x n), r=k/2.
    Chunk-wise reproducible: B comes from stream `seed`, each 1024-row block of
    A and of the noise from its own stream; the global max is replaced by a
    fixed analytic scale so chunks do not need a first pass."""
    r = max(2, k // 2)
    B = np.random.RandomState(seed).gamma(2.0, size=(r, n))
    scale = 1.0 / (4.0 * r * 2.5)      # E[A]E[B]r = 4r ; ~2.5x headroom for the max
    out = np.empty((row1 - row0, n), dtype=np.float32)
    blk = 1024
    b0 = (row0 // blk) * blk
    while b0 < row1:
        rs = np.random.RandomState([seed, 7919, b0 // blk])
        A = rs.gamma(2.0, size=(blk, r))
        noise = rs.rand(blk, n)
        lo, hi = max(b0, row0), min(b0 + blk, row1)
        out[lo - row0:hi - row0] = ((A[lo - b0:hi - b0] @ B) * scale
                                    + 0.05 * noise[lo - b0:hi - b0]).astype(np.float32)
        b0 += blk
    return out


def tree_pathways(n, P, order, seed):
    from prmf_amd.synthetic import tree_pathways as _tp
    return _tp(n, P, order, seed)
