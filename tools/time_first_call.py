#!/usr/bin/env python
"""What the FIRST nmf_pathway call of a process pays over a warm one (a command-line run is always a first call): wall clock of
every engine call inside call 0 and call 1, side by side, at a benchmark shape or SHAPE=m,n,k,P (float32 host matrix).
    WORKLOAD=c3 python tools/time_first_call.py        SHAPE=500,20000,6,20 XD=f32 ITERS=630 python tools/time_first_call.py"""
import contextlib
import io
import os
import sys
import time

t_import = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, solver, synthetic as syn  # noqa: E402
torch.cuda.init()
torch.zeros(1, device="cuda:0")
t_import = time.perf_counter() - t_import

if os.environ.get("SHAPE"):
    m, n, k, P = [int(x) for x in os.environ["SHAPE"].split(",")]
    xd = os.environ.get("XD", "f32")
else:
    m, n, k, P, xd = syn.CONFIGS[os.environ.get("WORKLOAD", "c3")]
rs = np.random.RandomState(0)
X = rs.rand(m, n).astype(np.float32)
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, min(300, n // 4), 0))
names = ("__init__", "set_x_async", "device_tensor", "set_library", "set_active", "step", "score_candidates", "match_terms", "close")
acc = {}


def wrap(name):
    orig = getattr(solver.PrmfEngine, name)

    def f(self, *a, **kw):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **kw)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(solver.PrmfEngine, name, f)


for nm in names:
    wrap(nm)
_start = solver.PrmfEngine.set_x_async


def start_and_time_join(self, *a, **kw):
    pend = _start(self, *a, **kw)
    join = pend.result

    def result():
        t0 = time.perf_counter()
        try:
            return join()
        finally:
            acc["upload joined"] = acc.get("upload joined", 0.0) + time.perf_counter() - t0
    pend.result = result
    return pend


solver.PrmfEngine.set_x_async = start_and_time_join
print("imports + device init %.2f s; %d x %d, k = %d, %s" % (t_import, m, n, k, xd))
rows = []
for i in range(3):
    acc.clear()
    np.random.seed(0)
    before = set(sys.modules)
    prof = None
    if i == 0 and os.environ.get("PROFILE") == "1":
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        solver.nmf_pathway(X, None, library=lib, k_latent=k, max_iter=int(os.environ.get("ITERS", 100)), x_dtype=xd, tol=0.0)
    total = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(40)
        print(out.getvalue())
    rows.append((total, dict(acc)))
    fresh = sorted(set(sys.modules) - before)
    if fresh:
        tops = sorted({f.split(".")[0] for f in fresh})
        print("call %d imported %d modules: %s" % (i, len(fresh), " ".join(tops)))
for nm in ("total",) + names + ("upload joined", "everything else"):
    vals = []
    for total, a in rows:
        v = total if nm == "total" else (total - sum(a.values()) if nm == "everything else" else a.get(nm, 0.0))
        vals.append(v)
    print("%-18s %s   first - warm %+7.1f ms" % (nm, "  ".join("%7.1f" % (1e3 * v) for v in vals), 1e3 * (vals[0] - vals[2])))
