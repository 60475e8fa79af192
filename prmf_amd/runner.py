#!/usr/bin/env python
"""prmf_runner-compatible command line for the MI355X PRMF solver.

Same flags, exit codes, stdout lines and output files as the reference CLI
(script/prmf_runner.py:870-1093 with prmf/prmf_args.py:7-26):

    U.csv, V.csv        QUOTE_NONNUMERIC, header "LV0".., V indexed by the nodelist
    obj.txt             "key = %0.5f" lines then "k -> manifold_path" lines
    init_pathways.txt   with --manifolds-init
    test_error.csv      with --cross-validation

The solve itself is prmf_amd.solver.nmf_pathway (HIP kernels on the GPU).  Added
flags (the reference's are unchanged): --x-dtype {f32,bf16}; --data also accepts a .npy matrix
(samples x genes, no header: needs --nodelist).

Reference quirks that are kept on purpose: --delta,
--tolerence, --node-attribute and --m-samples are parsed but unused; graph nodes
are always relabelled by their "name" attribute; --high-dimensional is type=bool
(any non-empty string is True) and transposes X when m > n.
"""
import argparse
import copy
import csv
import os
import random
import sys
from argparse import RawTextHelpFormatter

import numpy as np


def add_prmf_arguments(parser):
    """Flag surface of prmf/prmf_args.py:7-26."""
    parser.add_argument("--data", type=str, required=True, help="n_obs x n_features matrix")
    parser.add_argument("--manifolds", nargs='+', help="graphml files to use as manifold. Node identifiers must appear in nodelist.")
    parser.add_argument("--manifolds-file", help="A file containing newline-delimited filepaths which are used as graphml files as in <manifolds>")
    parser.add_argument("--manifolds-init", nargs='*', help="If provided, use this list of manifolds to initialize PRMF (see the reference help).")
    parser.add_argument("--node-attribute", default=None, help="Relabel nodes in manifolds/graphs so that their node identifiers come from this node attribute.")
    parser.add_argument("--outdir", type=str, required=True, help="Directory containing results")
    parser.add_argument("--nodelist", type=str, help="Association of node identifier to matrix indexes. If not provided, inferred from the header in <--data>.")
    parser.add_argument("--k-latent", "-k", default=6, help="Number of latent factors", type=int)
    parser.add_argument("--tolerence", type=float, default=1e-3)
    parser.add_argument("--seed", default=None)
    parser.add_argument("--gamma", default=1.0, type=float, help="Tradeoff between reconstruction error and manifold regularization term; Default = 1.0")
    parser.add_argument("--delta", default=1.0, type=float, help="Regularization parameter for penalty for ignoring manifold; Default = 1.0")
    parser.add_argument("--tradeoff", default=-1, type=float, help="If set, automatically update gamma and delta each iteration. Must be in [0,1]; -1 disables. Default = -1.")
    parser.add_argument("--high-dimensional", default=True, type=bool, help="If True, ensure that <data> is of shape m x n with m < n ; otherwise ensure that X is m x n with m > n. Default = True.")
    parser.add_argument("--no-normalize", action='store_true', help="If flag is provided, don't quantile normalize the data")
    parser.add_argument("--delimiter", default=",", help="Field delimiter in <--data>")
    parser.add_argument("--m-samples", type=int, help="(parsed, unused -- as in the reference)")
    parser.add_argument("--cross-validation", "-c", type=float, help="Fraction of the samples to hold out and measure model performance with")
    parser.add_argument("--verbose", "-v", action='store_true', help="If flag is provided, report more information during each iteration")


def check_header(fpath, delim):
    """prmf_runner.py:794-808: the data file has a header iff some word of its first line is not a float."""
    with open(fpath, 'r') as fh:
        for line in fh:
            for word in line.rstrip().split(delim):
                try:
                    float(word)
                except ValueError:
                    return True
            break
    return False


def check_row_names(fpath, delim, has_header):
    """prmf_runner.py:810-831: row names iff the first word of the first data line is not a float."""
    target = 2 if has_header else 1
    data_line = None
    with open(fpath, 'r') as fh:
        for i, line in enumerate(fh, 1):
            data_line = line.rstrip()
            if i >= target:
                break
    try:
        float(data_line.split(delim)[0])
    except ValueError:
        return True
    return False


def relabel_nodes(G, node_attribute):
    """prmf/__init__.py:65-76."""
    import networkx as nx
    if node_attribute is not None:
        mapping = {n: d[node_attribute] for n, d in G.nodes(data=True) if node_attribute in d}
        G = nx.relabel_nodes(G, mapping)
    return G


def parse_pathways(manifold_fps, node_attribute="name"):
    """prmf_runner.py:833-868."""
    import networkx as nx
    return [(relabel_nodes(nx.read_graphml(fp).to_undirected(), node_attribute), fp) for fp in manifold_fps]


def parse_nodelist(fh):
    """prmf/__init__.py:247-257: whitespace-separated identifiers."""
    rv = []
    for line in fh:
        rv.extend(line.rstrip().split())
    return rv


def embed_arr(all_col_names, some_col_names, arr):
    """prmf/__init__.py:331-352 (vectorised: the reference loops over every element)."""
    if len(some_col_names) != arr.shape[1]:
        raise ValueError('some_col_names != #columns of arr: {} != {}'.format(len(some_col_names), arr.shape[1]))
    idx = {name: i for i, name in enumerate(all_col_names)}
    rv = np.zeros((arr.shape[0], len(all_col_names)))
    rv[:, [idx[c] for c in some_col_names]] = arr
    return rv


def pathway_to_vec(X, G, nodelist, rel_weight=5):
    """prmf_runner.py:272-306."""
    n = len(nodelist)
    index = {x: i for i, x in enumerate(nodelist)}
    v = np.zeros((n,))
    for node in G.nodes():
        v[index[node]] = 1
    on = (v == 1)
    off = np.invert(on)
    v[off] = np.mean(X.transpose()[off], axis=1)
    v[on] = (np.sum(v[off]) * rel_weight) / np.sum(on)
    v = v.reshape((n, 1))
    return v / np.linalg.norm(v), on


def nmf_init_u(X, v):
    """prmf_runner.py:308-317."""
    import scipy.optimize
    u, res = scipy.optimize.nnls(X.transpose(), v.flatten())
    u = u / np.linalg.norm(u) ** 2
    return u.reshape((X.shape[0], 1)), res


def nmf_init_v(X, u):
    """prmf_runner.py:319-334."""
    import scipy.optimize
    v, res = scipy.optimize.nnls(X, u.flatten())
    v = v / np.linalg.norm(v) ** 2
    return v.reshape(X.shape[1], 1), res


def main(argv=None):
    import pandas as pd
    from sklearn.model_selection import KFold

    parser = argparse.ArgumentParser(description="""
MI355X implementation of Pathway-Regularized NMF (same interface as prmf_runner.py).

Solve an optimization problem of the form
  min ||X - UV^T|| +
    gamma * sum_k min_i V[:,k]^T Ls[i] V[:,k] +
    delta * sum_k sum_{i | i in G_k} 1 / V[i,k] +
    ||U||_F^2
""", formatter_class=RawTextHelpFormatter)
    add_prmf_arguments(parser)
    parser.add_argument("--x-dtype", default="f32", choices=["f32", "bf16"],
                        help="storage type of X on the GPU (bf16 halves the HBM traffic of the sweeps; factors stay float32)")
    args = parser.parse_args(argv)

    tradeoff = None if args.tradeoff == -1 else args.tradeoff

    if args.manifolds is None and args.manifolds_file is None:
        sys.stderr.write("Exactly one of --manifolds or --manifolds-file is required.\n")
        sys.exit(22)
    elif args.manifolds is None:
        with open(args.manifolds_file, 'r') as fh:
            manifold_fps = [line.rstrip() for line in fh]
    elif args.manifolds_file is None:
        manifold_fps = args.manifolds
    else:
        sys.stderr.write("Exactly one of --manifolds or --manifolds-file is required.\n")
        sys.exit(23)
    G_fp_pairs = parse_pathways(manifold_fps)          # :915 (always the "name" attribute)
    fp_to_G = {fp: G for G, fp in G_fp_pairs}
    Gs = [G for G, _ in G_fp_pairs]

    if args.seed is not None:
        seed = int(args.seed)
        np.random.seed(seed)
        random.seed(seed)

    if args.data.endswith(".npy"):
        # extension: a binary matrix (samples x genes, no header, no row names -> --nodelist is required), for sizes
        # where parsing 1e9 delimited numbers would dominate the run; everything downstream is unchanged
        has_header = has_row_names = False
        X = pd.DataFrame(np.load(args.data, mmap_mode="r"))
    else:
        has_header = check_header(args.data, args.delimiter)
        has_row_names = check_row_names(args.data, args.delimiter, has_header)
        X = pd.read_csv(args.data, sep=args.delimiter, header='infer' if has_header else None, nrows=None,
                        index_col=0 if has_row_names else None)
    samples = list(X.index)

    m, n = X.shape                                      # :946-952
    if args.high_dimensional:
        if m > n:
            X = X.transpose()
    else:
        if m < n:
            X = X.transpose()

    if args.nodelist is not None:                       # :957-975
        with open(args.nodelist) as fh:
            nodelist = parse_nodelist(fh)
        X = X.to_numpy()
    elif has_header:
        nodelist = list(X.columns)
        seen = set(nodelist)
        for G in Gs:
            for node in G:
                if node not in seen:
                    nodelist.append(node)
                    seen.add(node)
        X = embed_arr(nodelist, list(X.columns), X.to_numpy())
    else:
        sys.stderr.write("--nodelist is not provided and there is no header in <--data>\n")
        sys.exit(25)

    nodelist_set = set(nodelist)                        # :977-996
    fracs, all_zero = [], True
    for G in Gs:
        count = sum(1 for node in G.nodes() if node in nodelist_set)
        fracs.append(count / G.order())
        if count != 0:
            all_zero = False
    if all_zero:
        sys.stderr.write("Invalid manifolds. Check that the node identifiers of the manifolds are present in the nodelist. Try setting --node-attribute if the node identifier is in a graphml attribute rather than the XML node attribute 'id'\n")
        sys.exit(24)
    sys.stdout.write("Printing manifold node representation in nodelist:\n")
    for (G, fp), frac in zip(G_fp_pairs, fracs):
        sys.stdout.write("{}: {:2.1f}%\n".format(fp, frac * 100))

    U_fp = os.path.join(args.outdir, "U.csv")
    V_fp = os.path.join(args.outdir, "V.csv")
    obj_fp = os.path.join(args.outdir, "obj.txt")

    X_test = None                                       # :1002-1013
    if args.cross_validation is not None:
        kf = KFold(n_splits=round(1 / args.cross_validation))
        for train_index, test_index in kf.split(X):
            X_test = X[test_index]
            X = X[train_index]
            samples = [samples[i] for i in train_index]
            break

    if not args.no_normalize:                           # :1019-1020, on the GPU (prmf_quantile_transform)
        from .preprocess import quantile_transform
        X = quantile_transform(X)

    U_init = V_init = None                              # :1022-1064
    if args.manifolds_init is not None:
        Gs_init = [fp_to_G[fp] for fp in args.manifolds_init]
        init_fps = []
        if len(args.manifolds_init) < args.k_latent:
            non_init = list(set(manifold_fps) - set(args.manifolds_init))
            chosen = random.sample(non_init, args.k_latent - len(args.manifolds_init))
            init_fps = copy.copy(args.manifolds_init)
            for fp in chosen:
                Gs_init.append(fp_to_G[fp])
                init_fps.append(fp)
        elif len(args.manifolds_init) == args.k_latent:
            init_fps = args.manifolds_init
        else:
            inds = np.random.choice(len(Gs_init), args.k_latent)
            Gs_init = [Gs_init[i] for i in inds]
            init_fps = [args.manifolds_init[i] for i in inds]
        vs, us = [], []
        for G in Gs_init:
            v, on = pathway_to_vec(X, G, nodelist)
            signal = v[on]
            u, _res = nmf_init_u(X, v)
            v_new, _res = nmf_init_v(X, u)
            v_new[on] = signal
            vs.append(v_new)
            us.append(u)
        V_init = np.concatenate(vs, axis=1)
        U_init = np.concatenate(us, axis=1)
        sys.stdout.write("Using the following manifolds for initialization:\n{}\n".format("\n".join(init_fps)))
        with open(os.path.join(args.outdir, 'init_pathways.txt'), 'w') as fh:
            fh.write("\n".join(init_fps))

    from .solver import nmf_pathway                     # the GPU solve (:1067)
    U, V, obj_data = nmf_pathway(X, Gs, nodelist=nodelist, gamma=args.gamma, tradeoff=tradeoff,
                                 k_latent=args.k_latent, U_init=U_init, V_init=V_init, verbose=args.verbose,
                                 x_dtype=args.x_dtype)
    cols = ["LV{}".format(x) for x in range(args.k_latent)]
    U = pd.DataFrame(U, index=samples, columns=cols)
    V = pd.DataFrame(V, index=nodelist, columns=cols)
    U.to_csv(U_fp, sep=",", index=has_row_names, quoting=csv.QUOTE_NONNUMERIC)
    V.to_csv(V_fp, sep=",", index=True, quoting=csv.QUOTE_NONNUMERIC)

    if args.cross_validation is not None:               # :1073-1079
        from .solver import measure_cv_performance     # GPU: one sweep + batched NNLS (prmf/__init__.py:768-798)
        errs = measure_cv_performance(V.to_numpy(), X_test, x_dtype=args.x_dtype)
        np.savetxt(os.path.join(args.outdir, 'test_error.csv'), errs, delimiter=",")
        obj_data['average_normalized_test_error'] = np.mean(errs)

    with open(obj_fp, 'w') as fh:                        # :1081-1093
        l2p = obj_data.pop('latent_to_pathway_data', {})
        for k, v in obj_data.items():
            fh.write("{} = {:0.5f}\n".format(k, v))
        for k in sorted(l2p.keys()):
            fh.write("{} -> {}\n".format(k, G_fp_pairs[l2p[k][0][0]][1]))


if __name__ == "__main__":
    main()
