"""Host-side pathway library: graphs -> the flat arrays libprmf_hip.so consumes.

Replaces the per-graph W / D / L / feat_inds lists the reference builds with
networkx + scipy at script/prmf_runner.py:670-696 (and the NORMALIZED_LAPLACIANS
global) by one concatenated, support-local CSR that is uploaded to the GPU once
(`prmf_set_pathway_library`) and then addressed by pathway index.
"""
import numpy as np
import scipy.sparse as sp


class PathwayLibrary(object):
    """P pathways over a universe of n genes.

    For pathway p, support nodes are sup_idx[sup_ptr[p]:sup_ptr[p+1]] (gene
    indices, graph node order).  For the s-th support node overall: sup_deg[s] is
    the column sum of W (diagonal of D), sup_ldiag[s] = L[i,i], and its adjacency
    row is (col_local, w)[row_ptr[s]:row_ptr[s+1]] with col_local indexing the
    same pathway's support list."""

    def __init__(self, n, supports, local_adj, names=None):
        self.n = int(n)
        self.P = len(supports)
        if self.P == 0:
            raise ValueError("pathway library is empty")
        self.names = list(names) if names is not None else [str(i) for i in range(self.P)]
        self.supports = [np.asarray(s, dtype=np.int64) for s in supports]
        self.local_adj = [sp.csr_matrix(a, dtype=np.float64) for a in local_adj]
        sizes = np.array([len(s) for s in self.supports], dtype=np.int64)
        self.sup_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.sup_idx = (np.concatenate(self.supports) if sizes.sum() else np.zeros(0)).astype(np.int32)
        degs, ldiags, rows, cols, ws = [], [], [np.zeros(1, dtype=np.int64)], [], []
        off = 0
        for s, a in zip(self.supports, self.local_adj):
            if a.shape != (len(s), len(s)):
                raise ValueError("local adjacency shape %s does not match support size %d" % (a.shape, len(s)))
            if len(np.unique(s)) != len(s):
                raise ValueError("duplicate node in a pathway support")
            if len(s) and (s.min() < 0 or s.max() >= self.n):
                raise ValueError("support index outside [0, n)")
            a.sum_duplicates()
            a.sort_indices()
            deg = np.asarray(a.sum(axis=0)).ravel()      # column sums (prmf_runner.py:680)
            degs.append(deg)
            ldiags.append(deg - a.diagonal())            # L = D - W          (:683)
            rows.append(a.indptr[1:].astype(np.int64) + off)
            off += a.nnz
            cols.append(a.indices.astype(np.int32))
            ws.append(a.data.astype(np.float64))
        cat = lambda xs, dt: (np.concatenate(xs) if len(xs) else np.zeros(0)).astype(dt)
        self.degs = degs                                  # per pathway, reused by laplacian_quadratic
        self.sup_deg = cat(degs, np.float64)
        self.sup_ldiag = cat(ldiags, np.float64)
        self.row_ptr = cat(rows, np.int64)
        self.col_local = cat(cols, np.int32)
        self.w = cat(ws, np.float64)
        self.max_support = int(sizes.max())

    # ---- constructors -----------------------------------------------------
    @classmethod
    def from_edge_lists(cls, n, pathways, names=None):
        """pathways: iterable of (edges [E,2] gene indices, weights [E] or None,
        nodes [S] gene indices in graph order).  Undirected; a self-loop is
        counted once on the diagonal (networkx adjacency semantics)."""
        supports, adjs = [], []
        for edges, w, nodes in pathways:
            nodes = np.asarray(nodes, dtype=np.int64)
            edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
            w = np.ones(len(edges)) if w is None else np.asarray(w, dtype=np.float64)
            loc = -np.ones(n, dtype=np.int64)
            loc[nodes] = np.arange(len(nodes))
            u, v = loc[edges[:, 0]], loc[edges[:, 1]]
            if len(edges) and (u.min() < 0 or v.min() < 0):
                raise ValueError("edge endpoint outside the pathway's node list")
            off = u != v
            r = np.concatenate([u, v[off]])
            c = np.concatenate([v, u[off]])
            d = np.concatenate([w, w[off]])
            adjs.append(sp.coo_matrix((d, (r, c)), shape=(len(nodes), len(nodes))).tocsr())
            supports.append(nodes)
        return cls(n, supports, adjs, names)

    @classmethod
    def from_graphs(cls, Gs, nodelist, names=None):
        """networkx graphs -> library, as nmf_pathway does (prmf_runner.py:662-690):
        each graph is restricted to the nodelist, rows/columns follow the nodelist,
        the support is the restricted graph's node order, edge attribute 'weight'
        (default 1) is honoured."""
        index = {x: i for i, x in enumerate(nodelist)}
        items = []
        for G in Gs:
            nodes = [index[x] for x in G.nodes() if x in index]
            edges, w = [], []
            for a, b, d in G.edges(data=True):
                if a in index and b in index:
                    edges.append((index[a], index[b]))
                    w.append(d.get("weight", 1))
            items.append((np.array(edges, dtype=np.int64).reshape(-1, 2), np.array(w, dtype=np.float64), nodes))
        return cls.from_edge_lists(len(nodelist), items, names)

    @classmethod
    def from_sparse(cls, Ws, feat_inds, names=None):
        """From the reference's own inner-loop arguments: W (n x n scipy sparse)
        and the support list per pathway (k_to_W / k_to_feat_inds of
        nmf_manifold_vec_update, prmf_runner.py:374).  Every non-zero of W must
        lie on the support (true for everything nmf_pathway builds)."""
        supports, adjs = [], []
        n = None
        for W, s in zip(Ws, feat_inds):
            W = sp.csr_matrix(W)
            n = W.shape[0] if n is None else n
            s = np.asarray(list(s), dtype=np.int64)
            sub = W[s][:, s]
            if sub.nnz != W.nnz and abs(sub).sum() != abs(W).sum():
                raise ValueError("adjacency has entries outside the pathway support; not supported")
            supports.append(s)
            adjs.append(sub)
        return cls(n, supports, adjs, names)

    # ---- cache (the built library as one .npz) -------------------------
    @staticmethod
    def cache_path(path):
        """np.savez appends '.npz' to a name without it; `save` and `load` agree on the name through this."""
        return path if str(path).endswith(".npz") else str(path) + ".npz"

    def save(self, path, key=""):
        """The flat arrays (what the GPU gets) + names + a caller-chosen `key` naming the inputs they were built from.
        Written to a temporary file and renamed into place: a concurrent `load` (another rank, another job) sees the old
        file or the new one, never a truncated archive."""
        import os
        import tempfile
        path = self.cache_path(path)
        fd, tmp = tempfile.mkstemp(suffix=".npz", dir=os.path.dirname(os.path.abspath(path)) or ".")
        try:
            with os.fdopen(fd, "wb") as fh:
                np.savez_compressed(fh, n=np.int64(self.n), names=np.array([str(x) for x in self.names]), key=np.array(str(key)),
                                    sup_ptr=self.sup_ptr, sup_idx=self.sup_idx, row_ptr=self.row_ptr, col_local=self.col_local, w=self.w)
            os.replace(tmp, path)
        except BaseException:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise

    @classmethod
    def load(cls, path, key=None):
        """Library saved by `save`; None if `key` is given and the file was built from something else (or is unreadable)."""
        try:
            z = np.load(cls.cache_path(path))
            if key is not None and str(z["key"]) != str(key):
                return None
            sup_ptr, row_ptr = z["sup_ptr"], z["row_ptr"]
            supports, adjs = [], []
            for p in range(len(sup_ptr) - 1):
                s0, s1 = int(sup_ptr[p]), int(sup_ptr[p + 1])
                e0, e1 = int(row_ptr[s0]), int(row_ptr[s1])
                supports.append(z["sup_idx"][s0:s1].astype(np.int64))
                adjs.append(sp.csr_matrix((z["w"][e0:e1], z["col_local"][e0:e1], row_ptr[s0:s1 + 1] - e0), shape=(s1 - s0, s1 - s0)))
            return cls(int(z["n"]), supports, adjs, [str(x) for x in z["names"]])
        except Exception:      # missing, truncated, corrupt (zipfile.BadZipFile, EOFError, pickle errors ...): rebuild
            return None

    # ---- host-side helpers used by the outer solver -------------------------
    def laplacian_quadratic(self, p, v):
        """v^T L_p v for a full-length vector v (force_distinct_lapls, :232)."""
        s = self.supports[p]
        vs = np.asarray(v, dtype=np.float64)[s]
        a = self.local_adj[p]
        return float(vs.dot(self.degs[p] * vs - a.dot(vs)))

    def laplacian_quadratics(self, p, V, cols):
        """v^T L_p v for the columns `cols` of V (n x k): ONE sparse product per pathway instead of one per (latent, pathway)
        pair -- force_distinct_lapls asks for 64 x 61 pairs at the 50000 x 20000, k = 64 shape (3904 SciPy calls, 44 ms of
        a 0.33 s solve).  Per column the arithmetic is laplacian_quadratic's: the same row-wise accumulation in the sparse
        product, the same contiguous dot at the end (columns are made contiguous), so the values are identical."""
        s = self.supports[p]
        Vs = np.asfortranarray(np.asarray(V, dtype=np.float64)[np.ix_(s, np.asarray(cols, dtype=np.int64))])      # support rows only
        T = np.asfortranarray(self.degs[p][:, None] * Vs - self.local_adj[p].dot(Vs))
        return [float(Vs[:, j].dot(T[:, j])) for j in range(Vs.shape[1])]

    def arrays(self):
        return (self.sup_ptr, self.sup_idx, self.sup_deg, self.sup_ldiag, self.row_ptr, self.col_local, self.w)
