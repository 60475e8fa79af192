"""Harness that loads the *reference* PRMF implementation from /root/reference.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py (in the build container,
where /root/reference is mounted) to generate the golden vectors committed under
tests/golden/.  Nothing here is imported by the product (prmf_amd/) and nothing
here runs on the GPU box (the reference does not travel).

The reference was written against networkx 1.11 / numpy 1.18
(/root/reference/env/minimal-environment.yml:5-10, requirements.txt:5).  This
container has networkx 3.4 / numpy 2.2, so `nmf_pathway` and `main` need a small
compatibility shim installed *before* the runner module is imported (SURVEY.md
Appendix C).  The shim lives only in this harness; the reference is never edited
and none of its source is copied.  The inner loop (`nmf_manifold_vec_update`,
`nmf_manifold_vec_obj`, ...) needs none of it.
"""
import importlib.util
import os
import sys

import numpy as np
import scipy.sparse as sp

REFERENCE_ROOT = os.environ.get("PRMF_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "script", "prmf_runner.py"))


def _install_networkx_1x_shim():
    import networkx as nx

    if getattr(nx, "_prmf_shim_installed", False):
        return
    Graph = nx.Graph

    # nodes_iter / edges_iter / .node were removed in networkx 2.x
    def nodes_iter(self):
        return iter(self.nodes())

    def edges_iter(self, *a, **kw):
        return iter(self.edges(*a, **kw))

    Graph.nodes_iter = nodes_iter
    Graph.edges_iter = edges_iter
    Graph.node = property(lambda self: self._node)

    # networkx 1.x accepted a positional attribute dict
    _add_edge = Graph.add_edge

    def add_edge(self, u, v, attr_dict=None, **attr):
        if attr_dict is not None:
            attr = dict(attr_dict, **attr)
        return _add_edge(self, u, v, **attr)

    Graph.add_edge = add_edge
    _add_node = Graph.add_node

    def add_node(self, n, attr_dict=None, **attr):
        if attr_dict is not None:
            attr = dict(attr_dict, **attr)
        return _add_node(self, n, **attr)

    Graph.add_node = add_node

    # networkx 1.11 adjacency_matrix(G, nodelist): rows/cols follow nodelist,
    # nodes of nodelist that are absent from G give empty rows, edges with an
    # endpoint outside nodelist are dropped.
    def adjacency_matrix(G, nodelist=None, weight="weight", dtype=None):
        if nodelist is None:
            nodelist = list(G)
        n = len(nodelist)
        index = {u: i for i, u in enumerate(nodelist)}
        rows, cols, data = [], [], []
        for u, v, d in G.edges(data=True):
            if u in index and v in index:
                w = d.get(weight, 1)
                rows.append(index[u]); cols.append(index[v]); data.append(w)
                if u != v:
                    rows.append(index[v]); cols.append(index[u]); data.append(w)
        if not data:
            return sp.csr_matrix((n, n))
        return sp.coo_matrix((data, (rows, cols)), shape=(n, n)).tocsr()

    nx.adjacency_matrix = adjacency_matrix

    # networkx 1.x max_weight_matching returned a symmetric dict
    _mwm = nx.max_weight_matching

    def max_weight_matching(G, maxcardinality=False, weight="weight"):
        pairs = _mwm(G, maxcardinality=maxcardinality, weight=weight)
        mate = {}
        for a, b in pairs:
            mate[a] = b
            mate[b] = a
        return mate

    nx.max_weight_matching = max_weight_matching
    nx._prmf_shim_installed = True


def load_reference_runner():
    """Return the reference's script/prmf_runner.py as a module object."""
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    _install_networkx_1x_shim()
    name = "prmf_reference_runner"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REFERENCE_ROOT, "script", "prmf_runner.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reset_reference_globals(mod):
    """The reference appends to module globals on every nmf_pathway call
    (prmf_runner.py:691-696); reset them so that several solves can be run
    from one harness process."""
    mod.PATHWAY_TO_SUPPORT = None
    mod.LAPLACIANS = []
    mod.NORMALIZED_LAPLACIANS = []
