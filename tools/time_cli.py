#!/usr/bin/env python
"""End to end: the command line at the reference's typical size (samples << genes, delimited text in, U.csv / V.csv out),
wall clock by stage.  Writes its inputs under $TMPDIR first (not timed).
    M=500 N=20000 K=6 P=20 python tools/time_cli.py"""
import cProfile
import io
import os
import pstats
import shlex
import sys
import tempfile
import time

t_start = time.perf_counter()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

m, n, k, P = (int(os.environ.get(v, d)) for v, d in (("M", 500), ("N", 20000), ("K", 6), ("P", 20)))
d = tempfile.mkdtemp(prefix="prmf_cli_")
rs = np.random.RandomState(0)
X = rs.gamma(5.0, size=(m, k)).dot(rs.gamma(5.0, size=(n, k)).T)
genes = ["ENSP%d" % i for i in range(n)]
t0 = time.perf_counter()
as_npy = os.environ.get("FORMAT", "text") == "npy"          # FORMAT=npy: the binary input of large runs (+ --nodelist)
data_fp = os.path.join(d, "data.npy" if as_npy else "data.csv")
if as_npy:
    np.save(data_fp, X.astype(np.float32))
    with open(os.path.join(d, "nodelist.txt"), "w") as fh:
        fh.write("\n".join(genes))
else:
    with open(data_fp, "w") as fh:
        fh.write("," + ",".join(genes) + "\n")
        for i in range(m):
            fh.write("s%d," % i + ",".join(map(repr, X[i].tolist())) + "\n")
print("inputs: %d x %d matrix as %s, %.0f MB, written in %.1f s" % (m, n, "float32 .npy" if as_npy else "text", os.path.getsize(data_fp) / 1e6, time.perf_counter() - t0), flush=True)
fps = []
for p in range(P):
    inds = rs.choice(n, 300, replace=False)
    fp = os.path.join(d, "g%d.graphml" % p)
    with open(fp, "w") as fh:
        fh.write('<?xml version="1.0" encoding="utf-8"?><graphml xmlns="http://graphml.graphdrawing.org/xmlns"><key id="d0" for="node" attr.name="name" attr.type="string"/><graph edgedefault="undirected">\n')
        for j, g in enumerate(inds):
            fh.write('<node id="%d"><data key="d0">%s</data></node>\n' % (j, genes[g]))
        for j in range(1, len(inds)):
            fh.write('<edge source="%d" target="%d"/>\n' % (rs.randint(0, j), j))
        fh.write('</graph></graphml>\n')
    fps.append(fp)

if os.environ.get("SUBPROCESS"):          # the whole process as a user starts it: interpreter, imports, device, main()
    import subprocess
    argv = ["--data", data_fp, "--manifolds"] + fps + ["--outdir", os.path.join(d, "out"), "-k", str(k), "--seed", "1"] + \
        (["--nodelist", os.path.join(d, "nodelist.txt")] if as_npy else []) + shlex.split(os.environ.get("EXTRA", ""))
    os.makedirs(os.path.join(d, "out"), exist_ok=True)
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "script", "prmf_runner.py")
    for rep in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, script] + argv, capture_output=True, text=True)
        print("process %d: %.2f s wall clock (exit %d, %d lines printed)" % (rep, time.perf_counter() - t0, r.returncode, r.stdout.count("\n")), flush=True)
    r = subprocess.run([sys.executable, "-X", "importtime", "-c", "import sys; sys.path.insert(0, %r); import prmf_amd.runner, torch, pandas, networkx" %
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")], capture_output=True, text=True)
    tops = []
    for line in r.stderr.splitlines():
        parts = line.split("|")
        if len(parts) == 3 and not parts[2].startswith("   "):          # top-level imports only
            try:
                tops.append((int(parts[1]), parts[2].strip()))
            except ValueError:
                pass
    print("imports (cumulative ms): " + ", ".join("%s %.0f" % (nm, us / 1e3) for us, nm in sorted(tops, reverse=True)[:8]), flush=True)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    sys.exit(0)
t_imp = time.perf_counter()
from prmf_amd import runner  # noqa: E402
import torch  # noqa: E402
torch.cuda.init()
print("imports (prmf_amd, torch, device init): %.2f s" % (time.perf_counter() - t_imp), flush=True)
argv = ["--data", data_fp, "--manifolds"] + fps + ["--outdir", os.path.join(d, "out"), "-k", str(k), "--seed", "1"] + \
    (["--nodelist", os.path.join(d, "nodelist.txt")] if as_npy else []) + shlex.split(os.environ.get("EXTRA", "")) + \
    (["--manifolds-init"] + fps[:k] if os.environ.get("INIT") else [])          # INIT=1: one initial pathway per latent factor
os.makedirs(os.path.join(d, "out"), exist_ok=True)
# wall clock of the stages (no profiler in the way): wrappers around the runner's own stage functions
stage = {}


def _wrap(obj, name, label):
    orig = getattr(obj, name)

    def f(*a, **kw):
        t0 = time.perf_counter()
        try:
            return orig(*a, **kw)
        finally:
            stage[label] = stage.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, f)


import prmf_amd.preprocess as _pp  # noqa: E402
import prmf_amd.solver as _sv  # noqa: E402
_wrap(runner.ManifoldSet, "from_args", "manifolds")
_wrap(runner.DataMatrix, "__init__", "read --data")
_wrap(runner.DataMatrix, "resolve_genes", "genes / embed")
_wrap(runner.ManifoldSet, "coverage", "coverage")
_wrap(_pp, "quantile_transform", "quantile normalisation")
_wrap(_sv, "nmf_pathway", "nmf_pathway")
_wrap(runner, "write_outputs", "outputs")
_wrap(runner, "initial_factors", "--manifolds-init")
if os.environ.get("PROFILE", "1") == "0":
    for rep in range(int(os.environ.get("REPS", 3))):
        stage.clear()
        so, se = sys.stdout, sys.stderr
        sys.stdout, sys.stderr = io.StringIO(), io.StringIO()
        t0 = time.perf_counter()
        try:
            runner.main(argv)
        finally:
            sys.stdout, sys.stderr = so, se
        total = time.perf_counter() - t0
        print("run %d: main() %.3f s = " % (rep, total) + " | ".join("%s %.3f" % kv for kv in stage.items()) +
              " | rest %.3f" % (total - sum(stage.values())), flush=True)
    import shutil  # noqa: E402
    shutil.rmtree(d, ignore_errors=True)
    sys.exit(0)
for rep in range(1 if os.environ.get("FIRST") else 2):          # FIRST=1: profile the first run of the process (what a user's run is)
    out = io.StringIO()
    pr = cProfile.Profile()
    so, se = sys.stdout, sys.stderr
    sys.stdout, sys.stderr = io.StringIO(), io.StringIO()
    t0 = time.perf_counter()
    try:
        pr.enable()
        runner.main(argv)
        pr.disable()
    finally:
        lines = sys.stdout.getvalue().count("\n")
        sys.stdout, sys.stderr = so, se
    print("run %d: main() %.2f s (%d lines printed)" % (rep, time.perf_counter() - t0, lines), flush=True)
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(int(os.environ.get("TOP", 40)))
print(out.getvalue())
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
