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
(samples x genes, no header: needs --nodelist); --n-gpus N / --dist-mode.

Multi-GPU (one process per GPU, torch.distributed over RCCL): either start the script under the launcher yourself,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 script/prmf_runner.py ...

or pass --n-gpus 8, which does exactly that as a child process before this process has touched a GPU.  Every rank
parses the same inputs; X is row-sharded over the ranks inside nmf_pathway, V is replicated; rank 0 is the only one
that prints and writes U.csv / V.csv / obj.txt (U is gathered to it).  Without --seed rank 0 draws one and
broadcasts it, so that all ranks consume identical legacy RNG streams in the host stages.

Reference quirks that are kept on purpose: --delta,
--tolerence, --node-attribute and --m-samples are parsed but unused; graph nodes
are always relabelled by their "name" attribute; --high-dimensional is type=bool
(any non-empty string is True) and transposes X when m > n.
"""
import argparse
import csv
import os
import random
import sys
from argparse import RawTextHelpFormatter

import numpy as np

MAX_K_LATENT = 128      # libprmf_hip: 1 <= k <= 128 (include/prmf_hip.h, prmf_create)
EXIT_K_LATENT = 26      # exit code of a --k-latent beyond it (the reference's own codes are 22-25, script/prmf_runner.py:904-993)


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
    parser.add_argument("--tradeoff", default=-1, type=float, help="If set, automatically update gamma and delta each iteration. Must be in [0,1]; -1 disables. Default = -1. "
                             "(Here: with the feedback on, every iteration's reconstruction error comes from one more sweep over X -- the "
                             "materialised residual the reference computes -- so an iteration costs about twice a plain one.)")
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
    """prmf_runner.py:833-868: `relabel_nodes(nx.read_graphml(fp).to_undirected(), node_attribute)` per file -- built in one
    pass by prmf_amd.graphml.read_pathway (the same networkx.Graph; files that are not plain go through networkx itself)."""
    from .graphml import read_pathway
    return [(read_pathway(fp, node_attribute), fp) for fp in manifold_fps]


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


def nmf_init_u(X, v, solver=None):
    """prmf_runner.py:308-317: u = nnls(X^T, v), scaled.  solver: a prmf_amd.preprocess.DeviceNnls holding X (the GPU's
    Lawson-Hanson); None: scipy's on the host, as the reference calls it."""
    u = None
    if solver is not None:
        from .preprocess import NnlsDeclined
        try:
            u, res = solver.nnls(v.flatten(), transposed=True)
        except NnlsDeclined:
            pass                                     # conditioning beyond the GPU solver's factor: scipy's, as the reference
    if u is None:
        import scipy.optimize
        u, res = scipy.optimize.nnls(X.transpose(), v.flatten())
    u = u / np.linalg.norm(u) ** 2
    return u.reshape((X.shape[0], 1)), res


def nmf_init_v(X, u, solver=None):
    """prmf_runner.py:319-334: v = nnls(X, u), scaled."""
    v = None
    if solver is not None:
        from .preprocess import NnlsDeclined
        try:
            v, res = solver.nnls(u.flatten(), transposed=False)
        except NnlsDeclined:
            pass
    if v is None:
        import scipy.optimize
        v, res = scipy.optimize.nnls(X, u.flatten())
    v = v / np.linalg.norm(v) ** 2
    return v.reshape(X.shape[1], 1), res


# ---------------------------------------------------------------------------
# The command line, as a pipeline of small stages over one record of parsed inputs.
# What each stage must reproduce of the reference's `main` (script/prmf_runner.py:870-1093) is cited at the stage;
# flags, messages, exit codes 22-25, output files and the order in which the legacy RNG streams are consumed are the
# drop-in contract, the decomposition is this package's own.
# ---------------------------------------------------------------------------
class ManifoldSet(object):
    """The pathway graphs of a run in command-line order."""

    def __init__(self, paths):
        self.paths = list(paths)
        self.graphs = [G for G, _fp in parse_pathways(self.paths)]       # :915 (always the "name" attribute)
        self.by_path = dict(zip(self.paths, self.graphs))

    @classmethod
    def from_args(cls, args):
        """--manifolds xor --manifolds-file (:902-914; exit codes 22 / 23)."""
        if (args.manifolds is None) == (args.manifolds_file is None):
            sys.stderr.write("Exactly one of --manifolds or --manifolds-file is required.\n")
            sys.exit(22 if args.manifolds is None else 23)
        if args.manifolds is not None:
            return cls(args.manifolds)
        with open(args.manifolds_file, 'r') as fh:
            return cls([line.rstrip() for line in fh])

    def coverage(self, nodelist):
        """Fraction of every graph's nodes that the nodelist knows (:977-996); exit code 24 if none is known at all."""
        known = set(nodelist)
        fracs = [sum(1 for node in G.nodes() if node in known) / G.order() for G in self.graphs]
        if not any(fracs):
            sys.stderr.write("Invalid manifolds. Check that the node identifiers of the manifolds are present in the nodelist. Try setting --node-attribute if the node identifier is in a graphml attribute rather than the XML node attribute 'id'\n")
            sys.exit(24)
        sys.stdout.write("Printing manifold node representation in nodelist:\n")
        for fp, frac in zip(self.paths, fracs):
            sys.stdout.write("{}: {:2.1f}%\n".format(fp, frac * 100))


class DataMatrix(object):
    """--data as read from disk: a frame oriented samples x genes, whether it carried row names / a header."""

    def __init__(self, args, header_only=False):
        import pandas as pd
        if header_only:           # a .npy matrix whose rows the ranks read themselves (RowSource): only the sample index
            self.has_header = self.has_row_names = False
            shape = np.load(args.data, mmap_mode="r").shape
            self.samples = list(range(shape[0]))           # the frame's RangeIndex, taken before the orientation step as always
            self.frame = None
            return
        if args.data.endswith(".npy"):
            # extension: a binary matrix (samples x genes, no header, no row names -> --nodelist is required), for sizes
            # where parsing 1e9 delimited numbers would dominate the run; everything downstream is unchanged
            self.has_header = self.has_row_names = False
            frame = pd.DataFrame(np.load(args.data, mmap_mode="r"))
        else:
            self.has_header = check_header(args.data, args.delimiter)                                   # :937-942
            self.has_row_names = check_row_names(args.data, args.delimiter, self.has_header)
            frame = None
            from . import textio
            if textio.wanted(args.data):
                # the numeric body parsed on the GPU (prmf_text_*): the same frame, values bit for bit; a file that is not
                # plain (quotes, missing values, ragged lines, ...) is left to pandas, which says what it means
                import torch
                why = []
                frame = textio.read_frame(args.data, args.delimiter, self.has_header, self.has_row_names,
                                          device=torch.cuda.current_device(), why=why, mixed_as_float=True)
                if frame is None and getattr(args, "verbose", False):
                    sys.stderr.write("--data is read by pandas: %s\n" % why[0])
            if frame is None:
                frame = pd.read_csv(args.data, sep=args.delimiter, header='infer' if self.has_header else None, nrows=None,
                                    index_col=0 if self.has_row_names else None)
        self.samples = list(frame.index)                   # taken BEFORE the orientation step, as the reference does
        m, n = frame.shape                                  # :946-952: --high-dimensional wants m <= n
        if (m > n) if args.high_dimensional else (m < n):
            frame = frame.transpose()
        self.frame = frame

    def resolve_genes(self, args, manifolds):
        """-> (X ndarray, nodelist).  --nodelist names the columns; otherwise the header does and genes that only the
        graphs know are appended, X zero-embedded (:957-975, prmf.embed_arr); neither: exit code 25."""
        if args.nodelist is not None:
            with open(args.nodelist) as fh:
                return self.frame.to_numpy(), parse_nodelist(fh)
        if not self.has_header:
            sys.stderr.write("--nodelist is not provided and there is no header in <--data>\n")
            sys.exit(25)
        measured = list(self.frame.columns)
        nodelist, seen = list(measured), set(measured)
        for G in manifolds.graphs:
            for node in G:
                if node not in seen:
                    nodelist.append(node)
                    seen.add(node)
        return embed_arr(nodelist, measured, self.frame.to_numpy()), nodelist


class RowSource(object):
    """This rank's rows of the data matrix, for a multi-rank run started with a binary (.npy) --data: the file is
    memory-mapped and every rank converts ONLY what it needs -- its row shard for the solve and, unless --no-normalize, its
    column shard for the per-gene quantile normalisation (:1019-1020; genes are independent, so rank r normalises genes
    shard(n, N, r) over all samples in column sub-blocks and the ranks hand each other the rows they own, one sub-block at a
    time) -- instead of every rank reading, orienting and normalising the whole matrix (8 x the parse time and 8 x the
    host memory on one node).  nmf_pathway takes it in place of X: `.shape` and `.rows(lo, hi)`.
    `peak_bytes` = the largest total of float64 arrays this object held at once (tests/test_gpu_sharded_solver.py)."""

    SUB_BLOCK_BYTES = 64 << 20

    def __init__(self, path, high_dimensional, ranks, normalize, device):
        from .ranks import shard_of
        mm = np.load(path, mmap_mode="r")
        if mm.ndim != 2:
            raise ValueError("--data must be a 2-D matrix")
        m, n = mm.shape
        view = mm.T if ((m > n) if high_dimensional else (m < n)) else mm          # :946-952
        self.shape = view.shape
        m, n = self.shape
        self.lo, self.hi = shard_of(m, ranks.world, ranks.rank)
        held = lambda *arrs: sum(a.nbytes for a in arrs if a is not None)
        self.peak_bytes = 0
        if not normalize:
            self.local = np.ascontiguousarray(view[self.lo:self.hi], dtype=np.float64)
            self.peak_bytes = held(self.local)
            return
        from .preprocess import SUBSAMPLE, _subsample_rows, quantile_transform_block
        self.local = np.empty((self.hi - self.lo, n), dtype=np.float64)
        fit_rows = _subsample_rows(m) if m > SUBSAMPLE else None                  # the same draw on every rank (same seed): as one call would make it
        block_bytes = int(os.environ.get("PRMF_INPUT_BLOCK_BYTES", self.SUB_BLOCK_BYTES))     # (tests shrink it)
        step = max(1, block_bytes // max(1, 8 * m))                                # genes per sub-block
        spans = [shard_of(n, ranks.world, r) for r in range(ranks.world)]
        rounds = max((c1 - c0 + step - 1) // step for c0, c1 in spans)
        for i in range(rounds):
            mine = None
            c0, c1 = spans[ranks.rank]
            a0, a1 = min(c0 + i * step, c1), min(c0 + (i + 1) * step, c1)
            if a1 > a0:
                mine = quantile_transform_block(np.ascontiguousarray(view[:, a0:a1], dtype=np.float64), fit_rows, device=device)
            for src, (s0, s1) in enumerate(spans):
                b0, b1 = min(s0 + i * step, s1), min(s0 + (i + 1) * step, s1)
                if b1 <= b0:
                    continue
                piece = ranks.array_from_rank(src, mine if src == ranks.rank else None, (m, b1 - b0), "cuda:%d" % device)
                self.peak_bytes = max(self.peak_bytes, held(self.local, mine, piece if piece is not mine else None))
                self.local[:, b0:b1] = piece[self.lo:self.hi]
                del piece

    def rows(self, lo, hi):
        if (lo, hi) != (self.lo, self.hi):
            raise ValueError("this rank holds rows [%d, %d), asked for [%d, %d)" % (self.lo, self.hi, lo, hi))
        return self.local


def hold_out(X, samples, fraction):
    """--cross-validation f: the first of round(1/f) contiguous folds is held out (:1002-1013)."""
    n_splits, n = round(1 / fraction), len(X)
    if isinstance(n_splits, int) and 2 <= n_splits <= n:
        # KFold without shuffling cuts contiguous folds, the first n % n_splits of them one longer (scikit-learn's
        # _BaseKFold / KFold._iter_test_indices); importing scikit-learn for that costs 0.3 s of a command-line run
        first = n // n_splits + (1 if n % n_splits else 0)
        train, test = np.arange(first, n), np.arange(first)
    else:                      # not a valid split: scikit-learn says so in its own words
        from sklearn.model_selection import KFold
        train, test = next(KFold(n_splits=n_splits).split(X))
    return X[train], X[test], [samples[i] for i in train]


def initial_factors(args, X, manifolds, nodelist, write=True):
    """--manifolds-init: U0 / V0 from one pathway indicator per latent factor through two NNLS solves (:1022-1064);
    fewer than k given -> the rest drawn with random.sample, more -> np.random.choice (the reference's RNG calls)."""
    chosen = list(args.manifolds_init)
    if len(chosen) < args.k_latent:
        # the reference draws from list(set(...)), whose order depends on the process's string-hash seed; sorted here so
        # that a seed names one run -- in every process of a multi-rank launch too
        rest = sorted(set(manifolds.paths) - set(chosen))
        chosen += random.sample(rest, args.k_latent - len(chosen))
    elif len(chosen) > args.k_latent:
        chosen = [chosen[i] for i in np.random.choice(len(chosen), args.k_latent)]
    us, vs = [], []
    # the 2 k NNLS solves against X run on the GPU (prmf_nnls): 14.4 s -> 0.3 s at 500 x 20000, k = 6
    import torch
    from .preprocess import DeviceNnls
    solver = DeviceNnls(X, device=torch.cuda.current_device())
    for fp in chosen:
        v, on = pathway_to_vec(X, manifolds.by_path[fp], nodelist)
        signal = v[on]
        u, _res = nmf_init_u(X, v, solver)
        v_new, _res = nmf_init_v(X, u, solver)
        v_new[on] = signal
        us.append(u)
        vs.append(v_new)
    sys.stdout.write("Using the following manifolds for initialization:\n{}\n".format("\n".join(chosen)))
    if write:
        with open(os.path.join(args.outdir, 'init_pathways.txt'), 'w') as fh:
            fh.write("\n".join(chosen))
    return np.concatenate(us, axis=1), np.concatenate(vs, axis=1)


def write_outputs(args, data, manifolds, samples, nodelist, U, V, obj_data):
    """U.csv / V.csv (QUOTE_NONNUMERIC, row index only if the input had row names, :1068-1071) and obj.txt
    (`key = %.5f` lines, then `k -> manifold path`, :1081-1093)."""
    import pandas as pd
    cols = ["LV{}".format(x) for x in range(args.k_latent)]
    from . import textio
    # the two matrices: prmf_write_csv prints the same bytes (repr floats, QUOTE_NONNUMERIC labels) on all cores; labels that
    # are not plain strings / integers go through pandas as before
    # (the frame is built from ALL sample labels even when they are not written: labels that do not fit U raise pandas'
    # shape error there, as in the reference -- tests/test_gpu_solver.py::test_cli_genes_by_samples_input_fails_...)
    if len(samples) != np.shape(U)[0] or \
            not textio.write_frame(os.path.join(args.outdir, "U.csv"), U, samples if data.has_row_names else None, cols):
        pd.DataFrame(U, index=samples, columns=cols).to_csv(os.path.join(args.outdir, "U.csv"), sep=",",
                                                            index=data.has_row_names, quoting=csv.QUOTE_NONNUMERIC)
    if len(nodelist) != np.shape(V)[0] or not textio.write_frame(os.path.join(args.outdir, "V.csv"), V, nodelist, cols):
        pd.DataFrame(V, index=nodelist, columns=cols).to_csv(os.path.join(args.outdir, "V.csv"), sep=",", index=True,
                                                             quoting=csv.QUOTE_NONNUMERIC)
    with open(os.path.join(args.outdir, "obj.txt"), 'w') as fh:
        l2p = obj_data.pop('latent_to_pathway_data', {})
        for key, val in obj_data.items():
            fh.write("{} = {:0.5f}\n".format(key, val))
        for k in sorted(l2p.keys()):
            fh.write("{} -> {}\n".format(k, manifolds.paths[l2p[k][0][0]]))


def main(argv=None):
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
    parser.add_argument("--n-gpus", type=int, default=1,
                        help="shard the solve over N GPUs of this node: re-launches this command under torch.distributed.run "
                             "(one process per GPU); not needed when the command is already started by that launcher")
    parser.add_argument("--launcher", default="processes", choices=["processes", "threads"],
                        help="how --n-gpus N becomes N ranks: `processes` (default) = one process per GPU under "
                             "torch.distributed.run; `threads` = one host thread per GPU inside this process (inputs are "
                             "read and normalised once, the ranks only run the solve)")
    parser.add_argument("--library-cache", default=None,
                        help="keep the built pathway library (support lists + support-local adjacency of every manifold over this "
                             "nodelist) in this .npz file and reuse it when the same manifold files and nodelist come back")
    parser.add_argument("--dist-mode", default=None, choices=["colshard", "replicated"],
                        help="multi-GPU V update: sharded by latent columns (default) or replicated on every rank")
    args = parser.parse_args(argv)

    if args.k_latent > MAX_K_LATENT:
        # the reference takes any --k-latent (prmf/prmf_args.py:14); this build's kernels tile the latent columns into at most
        # four 32-column MFMA tiles (prmf_create: PRMF_ERR_UNSUPPORTED).  Said here, before any input is read.
        sys.stderr.write("prmf_runner (MI355X build): --k-latent %d is not supported: libprmf_hip handles at most %d latent factors "
                         "(the reference implementation has no such limit)\n" % (args.k_latent, MAX_K_LATENT))
        sys.exit(EXIT_K_LATENT)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.n_gpus > 1 and args.launcher == "threads":
        if args.seed is None:
            args.seed = random.randrange(2 ** 31)
        _run(args, 0, 1, local_ranks=args.n_gpus)
        return
    if world == 1 and args.n_gpus > 1:
        # launcher mode: nothing in this process has touched the GPU (or imported torch); the ranks are children
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.n_gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])]
        cmd += list(sys.argv[1:] if argv is None else argv)
        sys.exit(subprocess.call(cmd))
    rank, group_ready = 0, False
    # PRMF_FORCE_RANK_PATH=1 under the launcher with ONE process: the multi-process branch at world size 1 (prmf_amd/ranks.py)
    forced = bool(os.environ.get("PRMF_FORCE_RANK_PATH")) and "RANK" in os.environ
    if world > 1 or forced:
        import torch
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # PRMF_DIST_BACKEND=gloo PRMF_SHARE_GPU=1: rehearsal on a one-GPU box (tests), ranks share cuda:0
        backend = os.environ.get("PRMF_DIST_BACKEND", "nccl")
        dev_index = 0 if os.environ.get("PRMF_SHARE_GPU") else local_rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:%d" % dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        torch.cuda.set_device(dev_index)
        group_ready = True
        if args.seed is None:      # one seed for all ranks (the reference leaves the streams unseeded: any seed is as good)
            box = [random.randrange(2 ** 31) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            args.seed = box[0]
        if rank != 0:              # rank 0 speaks for the job
            sys.stdout = open(os.devnull, "w")
    try:
        _run(args, rank, world, process_ranks=group_ready)
    except BaseException:
        # no barrier on the way out of a failure: the other ranks may be inside a different collective (or gone); this
        # process ends non-zero and the launcher ends the job
        if group_ready:
            try:
                import torch.distributed as dist
                dist.destroy_process_group()
            except Exception:
                pass
        raise
    if group_ready:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _run(args, rank, world, local_ranks=0, process_ranks=False):
    manifolds = ManifoldSet.from_args(args)
    if args.seed is not None:                              # both legacy streams, as the reference seeds them
        np.random.seed(int(args.seed))
        random.seed(int(args.seed))
    rank_local = ((world > 1 or process_ranks) and args.data.endswith(".npy") and args.nodelist is not None and args.cross_validation is None
                  and args.manifolds_init is None)
    if rank_local:
        # several process ranks, binary input: nobody reads the whole matrix (RowSource)
        import torch
        from .ranks import TorchRanks
        with open(args.nodelist) as fh:
            nodelist = parse_nodelist(fh)
        data = DataMatrix(args, header_only=True)
        manifolds.coverage(nodelist)
        X = RowSource(args.data, args.high_dimensional, TorchRanks(None), not args.no_normalize, torch.cuda.current_device())
        if args.verbose:
            sys.stderr.write("rank %d: rows [%d, %d) of %d x %d, peak %.1f MB of float64 input held\n"
                             % (rank, X.lo, X.hi, X.shape[0], X.shape[1], X.peak_bytes / 1e6))
        samples, X_test = data.samples, None
    else:
        data = DataMatrix(args)
        X, nodelist = data.resolve_genes(args, manifolds)
        manifolds.coverage(nodelist)
        samples, X_test = data.samples, None
        if args.cross_validation is not None:
            X, X_test, samples = hold_out(X, samples, args.cross_validation)
        if not args.no_normalize:                              # :1019-1020, on the GPU (prmf_quantile_transform)
            from .preprocess import quantile_transform
            # one process, one GPU, nothing else that reads X on the host (--manifolds-init does): the normalised matrix
            # stays on the device and nmf_pathway takes it from there
            stay = world == 1 and not process_ranks and local_ranks <= 1 and args.manifolds_init is None
            if stay:
                import torch
                X = quantile_transform(X, device=torch.cuda.current_device(), on_device=True)
            else:
                X = quantile_transform(X)
    U_init = V_init = None
    if args.manifolds_init is not None:
        U_init, V_init = initial_factors(args, X, manifolds, nodelist, write=rank == 0)

    from .solver import measure_cv_performance, nmf_pathway
    library = None
    if args.library_cache:                                  # graphs -> W / D / L / supports once per (manifolds, nodelist)
        import hashlib
        from .pathways import PathwayLibrary
        h = hashlib.sha256("\n".join(nodelist).encode())
        for fp in manifolds.paths:
            st = os.stat(fp)
            h.update(("%s|%d|%d" % (os.path.abspath(fp), st.st_size, int(st.st_mtime))).encode())
        library = PathwayLibrary.load(args.library_cache, key=h.hexdigest())
        if library is None:
            library = PathwayLibrary.from_graphs(manifolds.graphs, nodelist)       # exactly what nmf_pathway would build
            if rank == 0:
                library.save(args.library_cache, key=h.hexdigest())
    solve = dict(nodelist=nodelist, gamma=args.gamma, tradeoff=None if args.tradeoff == -1 else args.tradeoff,
                 k_latent=args.k_latent, U_init=U_init, V_init=V_init, verbose=args.verbose, x_dtype=args.x_dtype,
                 dist_mode=args.dist_mode, library=library)
    if local_ranks > 1:                                     # --launcher threads: the ranks share everything read so far
        from .pathways import PathwayLibrary
        from .ranks import LocalRanks
        if solve["library"] is None:
            solve["library"] = PathwayLibrary.from_graphs(manifolds.graphs, nodelist)
        U, V, obj_data = LocalRanks(local_ranks).run(lambda view: nmf_pathway(X, manifolds.graphs, group=view, **solve))[0]
    else:
        X = [X]                                             # handed over: nmf_pathway drops the matrix once the library holds it
        U, V, obj_data = nmf_pathway(X, manifolds.graphs, **solve)                                # the GPU solve (:1067)
    if rank != 0:                                          # every rank holds the same U, V, obj_data; rank 0 writes them
        return
    if X_test is not None:                                 # :1073-1079: one sweep + batched NNLS (prmf/__init__.py:768-798)
        errs = measure_cv_performance(V, X_test, x_dtype=args.x_dtype)
        np.savetxt(os.path.join(args.outdir, 'test_error.csv'), errs, delimiter=",")
        obj_data['average_normalized_test_error'] = np.mean(errs)
    write_outputs(args, data, manifolds, samples, nodelist, U, V, obj_data)


if __name__ == "__main__":
    main()
