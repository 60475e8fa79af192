"""CPU oracle: a float64 NumPy/SciPy restatement of the PRMF hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module, and only as the checker /
reported CPU baseline -- never as the product path.  The product (prmf_amd/)
does not import it and fails loudly when the HIP library is missing.

What it restates (reference = gitter-lab/prmf, paths relative to the reference
root; the arithmetic itself lives in numpy/scipy, which the reference also calls):

  * nmf_manifold_vec_update        script/prmf_runner.py:374-451  (loop :419-449)
  * nmf_manifold_vec_update_tradeoff script/prmf_runner.py:497-554
  * nmf_manifold_vec_obj           script/prmf_runner.py:336-372
  * normalize_laplacian            script/prmf_runner.py:56-65
  * score_latent_pathway_match_global / restrict   :115-127 / :129-194
  * force_distinct_lapls           :209-258
  * nmf_pathway (setup + outer control)            :556-792
  * embed_arr                      prmf/__init__.py:331-352
  * measure_cv_performance         prmf/__init__.py:768-798

Parity pin: every function here is checked against outputs of the reference
itself (imported unmodified in the build container by oracle/gen_golden.py) on
the fixtures committed under tests/golden/ -- see tests/test_oracle_golden.py.
The reference's own single golden file (test_inferred_nodelist_1_expected_obj.txt)
is stale; only its gamma/delta lines reproduce, and those
are asserted too.

The representation differs from the reference on purpose (this is a restatement,
not a copy): pathways are held as a `Pathway` record (CSR W restricted to the
nodelist, support list, degree vector, cached normalised Laplacian) instead of
four parallel dicts of n x n sparse matrices, and the per-column Python loops are
vectorised over the support.
"""
import math

import numpy as np
import scipy.optimize
import scipy.sparse as sp
from scipy.stats import multinomial

# prmf_runner.py:23 -- float32 machine epsilon used as the float64 clamp value
EPSILON = np.finfo(np.float32).eps


# --------------------------------------------------------------------------
# pathway representation
# --------------------------------------------------------------------------
class Pathway(object):
    """One pathway graph embedded in the nodelist index space.

    W       : (n, n) CSR weighted adjacency, rows/cols indexed by nodelist
              (prmf_runner.py:679; nodes outside the graph give empty rows)
    support : int array, nodelist indices of the graph's nodes in G.nodes()
              order (prmf_runner.py:689-690, `feat_inds`)
    deg     : (n,) column sums of W (prmf_runner.py:680-682, diagonal of D)
    """

    def __init__(self, W, support):
        self.W = sp.csr_matrix(W)
        self.n = self.W.shape[0]
        self.support = np.asarray(support, dtype=np.int64)
        self.deg = np.asarray(self.W.sum(axis=0)).ravel().astype(np.float64)
        self._L = None
        self._Ln = None

    @property
    def L(self):
        """L = D - W (prmf_runner.py:683)."""
        if self._L is None:
            self._L = (sp.diags(self.deg) - self.W).tocsr()
        return self._L

    def dinv_sqrt(self):
        """Diagonal of D^-1/2 as built by normalize_laplacian
        (prmf_runner.py:57-63): L[i,i]**-0.5 on support indices whose
        Laplacian diagonal is non-zero, 0 elsewhere."""
        ldiag = self.L.diagonal()
        vals = np.zeros(self.n)
        s = self.support
        nz = ldiag[s] != 0
        vals[s[nz]] = ldiag[s[nz]] ** (-0.5)
        return vals

    def L_normal(self, cached=True):
        """D^-1/2 L D^-1/2 (prmf_runner.py:64)."""
        if cached and self._Ln is not None:
            return self._Ln
        d = sp.diags(self.dinv_sqrt())
        Ln = (d @ (self.L @ d)).tocsr()
        if cached:
            self._Ln = Ln
        return Ln


    def L_normal_rebuilt(self):
        """normalize_laplacian as the reference RUNS it, every iteration, for every latent factor
        (prmf_runner.py:56-65, called at :349): one sparse element access L[ind, ind] per support index in a
        Python loop, a dia_matrix with setdiag, two sparse products.  Same matrix as L_normal(); kept separate
        because its cost is part of the reference's iteration (bench.py's `rebuild_laplacians` baseline)."""
        L = self.L
        D_to_minus_half = sp.dia_matrix(L.shape)
        values = np.zeros(L.shape[0])
        for ind in self.support:
            v = L[ind, ind]
            if v != 0:
                values[ind] = v ** (-1 / 2)
        D_to_minus_half.setdiag(values)
        return D_to_minus_half.dot(L.dot(D_to_minus_half))


def pathway_from_edges(n, edges, nodes, weights=None):
    """Build a Pathway from an undirected edge list over nodelist indices.

    Mirrors networkx-1.11 `adjacency_matrix(G, nodelist)` as the reference uses
    it (prmf_runner.py:679): symmetric, self-loops counted once, default
    weight 1.  `nodes` is the support in G.nodes() order."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if weights is None:
        weights = np.ones(len(edges))
    weights = np.asarray(weights, dtype=np.float64)
    u, v = edges[:, 0], edges[:, 1]
    off = u != v
    rows = np.concatenate([u, v[off]])
    cols = np.concatenate([v, u[off]])
    data = np.concatenate([weights, weights[off]])
    W = sp.coo_matrix((data, (rows, cols)), shape=(n, n)).tocsr()
    return Pathway(W, nodes)


def pathway_from_graph(G, nodelist, node_to_index=None):
    """networkx graph -> Pathway, following prmf_runner.py:670-690:
    G is first restricted to the nodelist, W is indexed by the nodelist and the
    support is the restricted graph's node order."""
    if node_to_index is None:
        node_to_index = {x: i for i, x in enumerate(nodelist)}
    Gs = G.subgraph([x for x in G.nodes() if x in node_to_index])
    edges, w = [], []
    for a, b, d in Gs.edges(data=True):
        edges.append((node_to_index[a], node_to_index[b]))
        w.append(d.get("weight", 1))
    nodes = [node_to_index[x] for x in Gs.nodes()]
    return pathway_from_edges(len(nodelist), edges, nodes, w)


# --------------------------------------------------------------------------
# objective  (prmf_runner.py:336-372)
# --------------------------------------------------------------------------
def objective_graph_terms(V, active, rebuild_laplacians=False):
    """manifold = sum_k vhat_k^T Lhat_k vhat_k ; ignore = sum_k sum_{i in S_k} 1/(vhat_ik + 1)
    with vhat_k = V[:,k]/||V[:,k]||  (prmf_runner.py:344-352).  rebuild_laplacians: normalise every Laplacian
    anew as the reference does at :349 (same numbers, the reference's cost)."""
    manifold = 0.0
    ignore = 0.0
    for k, pw in enumerate(active):
        v = V[:, k]
        vhat = v / np.linalg.norm(v)
        Ln = pw.L_normal_rebuilt() if rebuild_laplacians else pw.L_normal()
        manifold += Ln.dot(vhat).dot(vhat)
        ignore += np.sum(1.0 / (vhat[pw.support] + 1.0))
    return manifold, ignore


def nmf_manifold_vec_obj(X, U, V, active, gamma=1.0, delta=1.0, lean=False, norm_X_sq=None, rebuild_laplacians=False):
    """Objective record of prmf_runner.py:336-372.

    faithful (lean=False): recon = ||X - U V^T||_F on the materialised residual
    (:337).  lean=True uses the Gram identity
        ||X-UV^T||^2 = ||X||^2 - 2<X^T U, V> + <U^T U, V^T V>
    which is what the GPU path evaluates; both agree to ~1e-12 relative."""
    if lean:
        if norm_X_sq is None:
            norm_X_sq = float(np.sum(np.multiply(X, X)))
        cross = float(np.sum(np.multiply(X.T.dot(U), V)))
        gg = float(np.sum(np.multiply(U.T.dot(U), V.T.dot(V))))
        recon = math.sqrt(max(norm_X_sq - 2.0 * cross + gg, 0.0))
    else:
        recon = np.linalg.norm(X - U.dot(V.T))
    manifold, ignore = objective_graph_terms(V, active, rebuild_laplacians)
    fro = np.sum(np.multiply(U, U))
    obj = recon + gamma * manifold + delta * ignore + fro
    return {"recon": recon, "manifold": manifold, "ignore": ignore, "fro": fro,
            "gamma": gamma, "delta": delta, "obj": obj}


# --------------------------------------------------------------------------
# one multiplicative update  (prmf_runner.py:420-444)
# --------------------------------------------------------------------------
def update_step(X, U, V, active, gamma, delta):
    """Returns (U_new, V_new).  U-step :420-422, V-step :424-444.
    V-step uses the NEW U and the OLD V everywhere on the right-hand side."""
    n, k = V.shape
    # U <- U * (XV) / (U V^T V + U), 0/0 := 1   (:420-422)
    num_u = X.dot(V)
    den_u = U.dot(V.T.dot(V)) + U
    ratio = np.ones_like(num_u)
    np.divide(num_u, den_u, out=ratio, where=den_u != 0)
    U = U * ratio

    num_recon = X.T.dot(U)                 # :424
    den_recon = V.dot(U.T.dot(U))          # :425
    num_graph = np.zeros((n, k))           # gamma*W_k v_k + delta*(v+1)^-2 on S_k  (:434,:438)
    den_graph = np.zeros((n, k))           # gamma*deg_k*v_k                     (:435)
    for kk, pw in enumerate(active):
        v = V[:, kk]
        ign = np.zeros(n)
        s = pw.support
        ign[s] = delta * (v[s] + 1.0) ** -2
        num_graph[:, kk] = gamma * pw.W.dot(v) + ign
        den_graph[:, kk] = gamma * (pw.deg * v)
    num = num_recon + num_graph            # association as in :440
    den = den_recon + den_graph            # :441
    den[den < EPSILON] = EPSILON           # :442
    V = V * (num / den)                    # :443 (den is never 0 after the clamp)
    V[V < EPSILON] = EPSILON               # :444
    return U, V


def nmf_manifold_vec_update(X, U, V, active, n_steps=10, gamma=1.0, delta=1.0,
                            trace=None, lean=False, norm_X_sq=None, rebuild_laplacians=False):
    """prmf_runner.py:374-451.  `active[k]` is the Pathway of latent column k
    (the reference passes k_to_W / k_to_D / k_to_L / k_to_feat_inds dicts).
    If `trace` is a list, the per-step objective records are appended to it
    (the reference prints "<iter> <obj>" at :447)."""
    obj_data = None
    for _ in range(n_steps):
        U, V = update_step(X, U, V, active, gamma, delta)
        obj_data = nmf_manifold_vec_obj(X, U, V, active, gamma, delta, lean=lean, norm_X_sq=norm_X_sq,
                                        rebuild_laplacians=rebuild_laplacians)
        if trace is not None:
            trace.append(obj_data)
    return U, V, obj_data


def nmf_manifold_vec_update_tradeoff(X, U, V, active, n_steps=10, tradeoff=0.5,
                                     gamma=1.0, delta=1.0, trace=None):
    """prmf_runner.py:497-554: same update; after each step gamma = delta =
    (1-t)*recon/(t*manifold) (:542-548; x/0 := 1)."""
    obj_data = None
    for _ in range(n_steps):
        U, V = update_step(X, U, V, active, gamma, delta)
        obj_data = nmf_manifold_vec_obj(X, U, V, active, gamma, delta)
        if trace is not None:
            trace.append(obj_data)
        denom = tradeoff * obj_data["manifold"]
        gamma = 1 if denom == 0 else ((1 - tradeoff) * obj_data["recon"]) / denom
        delta = gamma
    return U, V, obj_data, gamma, delta


# --------------------------------------------------------------------------
# candidate scoring / restriction  (prmf_runner.py:115-127, 129-194)
# --------------------------------------------------------------------------
def score_latent_pathway_match(v, pw):
    """||vhat[S]||_2 + 1 - vhat^T Lhat vhat   (prmf_runner.py:120-126)."""
    vhat = v / np.linalg.norm(v)
    mass = np.sqrt(np.sum(vhat[pw.support] ** 2))
    man = 1 - pw.L_normal().dot(vhat).dot(vhat)
    return mass + man


def score_matrix(V, pathways):
    """k x P matrix of score_latent_pathway_match (what `restrict` evaluates for
    the still-candidate pairs)."""
    k = V.shape[1]
    S = np.zeros((k, len(pathways)))
    for kk in range(k):
        for p, pw in enumerate(pathways):
            S[kk, p] = score_latent_pathway_match(V[:, kk], pw)
    return S


def restrict(V, pathways, latent_to_pathway_data, percentile=19.9):
    """prmf_runner.py:129-194: per latent keep candidates whose score is above
    the 19.9th percentile of that latent's candidate scores."""
    rv = {}
    for k in range(V.shape[1]):
        data = latent_to_pathway_data[k]
        inds = [x[0] for x in data]
        if len(inds) > 1:
            scores = np.array([score_latent_pathway_match(V[:, k], pathways[p]) for p in inds])
            keep = np.where(scores > np.percentile(scores, percentile))[0]
            if len(keep) == 0:
                # :174-182 -- reachable only when all scores tie; the reference's
                # sample size is negative there and np.random.choice raises.
                size = math.ceil(len(inds) * (1 - percentile) / 100)
                chosen = np.random.choice(np.array(inds), size=size, replace=False)
                rv[k] = [(c, scores[c]) for c in chosen]
            else:
                rv[k] = [(inds[i], scores[i]) for i in keep]
        else:
            rv[k] = data
    return rv


def force_distinct_lapls(V, pathways, latent_to_pathway_data, active_support, gamma, delta):
    """prmf_runner.py:209-258: max-weight matching latent <-> pathway with edge
    weight 1/(gamma*v^T L v + delta*ignore); NOTE (:233-235) `ignore` is
    overwritten, not accumulated, over the active supports, so only the last
    latent's support counts -- kept as is."""
    import networkx as nx
    G = nx.Graph()
    ks = list(active_support.keys())
    for k, data in latent_to_pathway_data.items():
        for lapl, _score in data:
            L = pathways[lapl].L
            man = L.dot(V[:, k]).dot(V[:, k])
            ign = 0
            for k2 in ks:
                ign = np.sum(1.0 / (V[active_support[k2], k2] + 1))
            denom = gamma * man + delta * ign
            weight = 0 if denom == 0 else 1 / denom
            G.add_edge("k{}".format(k), "l{}".format(lapl), weight=weight)
    mate = nx.max_weight_matching(G)
    if isinstance(mate, dict):
        pairs = list(mate.items())
    else:
        pairs = []
        for a, b in mate:
            pairs.append((a, b))
            pairs.append((b, a))
    # the reference iterates a symmetric dict (networkx 1.x); the assignment is
    # idempotent so visiting each pair in both orders gives the same result
    for n1, n2 in pairs:
        if n1[0] == "k":
            latent_to_pathway_data[int(n1[1:])] = [(int(n2[1:]), 2)]
        else:
            latent_to_pathway_data[int(n2[1:])] = [(int(n1[1:]), 2)]
    return latent_to_pathway_data


# --------------------------------------------------------------------------
# outer solver  (prmf_runner.py:556-792)
# --------------------------------------------------------------------------
def nmf_pathway(X, pathways, gamma=1.0, delta=1.0, tradeoff=None, k_latent=6, tol=1e-3,
                max_iter=1000, modulus=10, U_init=None, V_init=None, trace=None,
                round_log=None, printer=None):
    """Restatement of nmf_pathway with `pathways` already converted to Pathway
    records (prmf_runner.py:670-696 is `pathway_from_graph`).

    Consumes the legacy global numpy RNG in the same order as the reference:
    rand(m,k), rand(n,k) (:651-653), then one multinomial.rvs per latent per
    round (:723)."""
    norm_X = np.linalg.norm(X)                       # :640
    if printer:
        printer("norm(X) = {}".format(norm_X))
    gamma = gamma * norm_X / k_latent                # :644
    delta = delta * 10 / norm_X                      # :645
    m, n = X.shape
    if U_init is None:
        U_init = 3 * (1 - np.random.rand(m, k_latent))
    if V_init is None:
        V_init = 3 * (1 - np.random.rand(n, k_latent))
    U, V = U_init, V_init
    if U.shape != (m, k_latent):
        raise ValueError("Invalid U_init with shape {} != (m_obs, k_latent) = {}".format(U.shape, (m, k_latent)))
    if V.shape != (n, k_latent):
        raise ValueError("Invalid V_init with shape {} != (n_feature, k_latent) = {}".format(V.shape, (n, k_latent)))

    P = len(pathways)
    l2p = {k: [(p, 1) for p in range(P)] for k in range(k_latent)}   # :700
    converged = False
    candidates_remain = True
    obj = math.inf
    obj_data = {}
    best = {"obj_data": {"obj": np.inf}}
    i = 0
    k_to_lapl = {}
    while i < max_iter and (candidates_remain or not converged):
        for k in range(k_latent):                                    # :717-729
            inds = [x[0] for x in l2p[k]]
            scores = np.array([x[1] for x in l2p[k]])
            prob = scores / np.sum(scores)
            sample = multinomial.rvs(1, prob)
            k_to_lapl[k] = inds[np.where(sample != 0)[0][0]]
        active = [pathways[k_to_lapl[k]] for k in range(k_latent)]
        if round_log is not None:
            round_log.append([int(k_to_lapl[k]) for k in range(k_latent)])
        step_trace = []
        if tradeoff is None:
            U, V, obj_data = nmf_manifold_vec_update(X, U, V, active, n_steps=modulus,
                                                     gamma=gamma, delta=delta, trace=step_trace)
        else:
            U, V, obj_data, gamma, delta = nmf_manifold_vec_update_tradeoff(
                X, U, V, active, n_steps=modulus, tradeoff=tradeoff, gamma=gamma, delta=delta,
                trace=step_trace)
        for s, od in enumerate(step_trace):
            if printer:
                printer("{} {}".format(i + s + 1, od["obj"]))
            if trace is not None:
                trace.append(od)
        i += modulus
        if obj_data["obj"] < best["obj_data"]["obj"]:                # :747-750
            best = {"U": U, "V": V, "obj_data": obj_data}
        if candidates_remain:                                        # :754-770
            if min(len(v) for v in l2p.values()) <= k_latent:
                sup = {k: active[k].support for k in range(k_latent)}
                l2p = force_distinct_lapls(V, pathways, l2p, sup, gamma, delta)
                candidates_remain = False
            else:
                l2p = restrict(V, pathways, l2p)
                candidates_remain = any(len(v) > 1 for v in l2p.values())
        prev_obj = obj
        obj = obj_data["obj"]
        converged = abs(obj - prev_obj) / obj < tol                  # :772-774
    if best["obj_data"]["obj"] < obj_data["obj"]:                    # :778-782
        if printer:
            printer("Local optima at convergence (or after max iterations) is not the best among all iterates; returning best instead")
        U, V, obj_data = best["U"], best["V"], best["obj_data"]
    obj_data = dict(obj_data)
    obj_data["latent_to_pathway_data"] = l2p
    return U, V, obj_data


# --------------------------------------------------------------------------
# helpers around the solve
# --------------------------------------------------------------------------
def embed_arr(all_col_names, some_col_names, arr):
    """prmf/__init__.py:331-352, vectorised."""
    if len(some_col_names) != arr.shape[1]:
        raise ValueError("some_col_names != #columns of arr")
    idx = {name: i for i, name in enumerate(all_col_names)}
    rv = np.zeros((arr.shape[0], len(all_col_names)))
    rv[:, [idx[c] for c in some_col_names]] = arr
    return rv


def measure_cv_performance(V, X_test):
    """prmf/__init__.py:768-798: per held-out sample nnls reconstruction error / ||x||."""
    err = np.zeros(X_test.shape[0])
    for s in range(X_test.shape[0]):
        _u, e = scipy.optimize.nnls(np.asarray(V), X_test[s, :])
        err[s] = e / np.linalg.norm(X_test[s, :])
    return err


# --------------------------------------------------------------------------
# synthetic inputs shared by tests / bench
# --------------------------------------------------------------------------
def random_tree_pathways(n, P, order, rs):
    """P random recursive trees on `order` distinct genes each (after
    script/gen_sim.py:17-51: KEGG-sized tree pathways), unit weights."""
    out = []
    for _ in range(P):
        nodes = rs.choice(n, size=order, replace=False)
        parent = np.array([rs.randint(0, i) for i in range(1, order)], dtype=np.int64)
        edges = np.stack([nodes[1:], nodes[parent]], axis=1)
        out.append(pathway_from_edges(n, edges, nodes))
    return out
