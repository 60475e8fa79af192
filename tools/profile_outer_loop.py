#!/usr/bin/env python
"""Where the outer loop's wall clock goes: cProfile of a WARM nmf_pathway call (the second of the process) at a benchmark
shape -- 100 iterations = 10 rounds with pathway sampling, candidate scoring, restrict, the matching (bench.py's
`outer_loop` measurement).
    WORKLOAD=c3 python tools/profile_outer_loop.py"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, solver, synthetic as syn  # noqa: E402

wl = os.environ.get("WORKLOAD", "c3")
m, n, k, P, xd = syn.CONFIGS[wl]
Xd = syn.x_rows_torch(torch, "cuda:0", 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
X = np.empty((m, n), dtype=np.float32)
for r0 in range(0, m, 16384):
    X[r0:r0 + 16384] = Xd[r0:r0 + 16384].float().cpu().numpy()
del Xd
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))


def call():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        solver.nmf_pathway(X, None, library=lib, k_latent=k, max_iter=int(os.environ.get("ITERS", 100)), x_dtype=xd, tol=0.0)


import gc  # noqa: E402
for i in range(2):
    t0 = time.perf_counter()
    call()
    print("call %d: %.3f s" % (i, time.perf_counter() - t0), flush=True)
for i in range(3):   # how much of a warm call is the cyclic collector walking the host objects (library, candidate lists)?
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    call()
    t1 = time.perf_counter()
    gc.enable()
    call()
    print("warm call with the collector off %.1f ms, on %.1f ms" % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)), flush=True)
# wall clock of the engine's setup calls inside a warm call (no added synchronisation)
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
    return orig


names = ("__init__", "set_x_async", "device_tensor", "set_norm_x_sq", "set_library", "set_active", "step", "score_candidates", "match_terms", "close")
origs = {nm: wrap(nm) for nm in names}
_start = solver.PrmfEngine.set_x_async          # (already wrapped: the time to START the upload); its join is timed apart


def start_and_time_join(self, *a, **kw):
    pend = _start(self, *a, **kw)
    t_started = time.perf_counter()
    join = pend.result

    def result():
        t0 = time.perf_counter()
        try:
            return join()
        finally:
            acc["upload: waited for at the join"] = acc.get("upload: waited for at the join", 0.0) + time.perf_counter() - t0
            acc["host work under the upload"] = acc.get("host work under the upload", 0.0) + t0 - t_started
    pend.result = result
    return pend


solver.PrmfEngine.set_x_async = start_and_time_join
names = names + ("upload: waited for at the join", "host work under the upload")
t0 = time.perf_counter()
call()
total = time.perf_counter() - t0
print("warm call %.1f ms: " % (1e3 * total) + " | ".join("%s %.1f" % (nm, 1e3 * acc.get(nm, 0.0)) for nm in names) +
      " | everything else %.1f  (ms)" % (1e3 * (total - sum(acc.values()) + acc.get("device_tensor", 0) * 0)), flush=True)
for nm, o in origs.items():
    setattr(solver.PrmfEngine, nm, o)
# the upload alone: host threads narrowing vs the chunk narrowed on the device
eng = solver.PrmfEngine(m, n, k, x_dtype=xd)
for label, th, hy in (("narrowed by 16 host threads, direct lane as needed", -1, None), ("the same, direct lane off", -1, "0"), ("the same, direct lane always", -1, "1"),
                      ("8 threads", 8, None), ("4 threads", 4, None), ("4 threads, direct lane off", 4, "0"), ("narrowed on the device (pageable copies)", 0, None)):
    ts = []
    for _ in range(3):
        if hy is not None:
            os.environ["PRMF_SETX_HYBRID"] = hy
        t0 = time.perf_counter()
        eng.set_x(X, host_threads=th)
        ts.append(time.perf_counter() - t0)
        os.environ.pop("PRMF_SETX_HYBRID", None)
    print("set_x(host float32 %d x %d -> %s), %s: %s ms  (%.1f GB/s of source)" % (
        m, n, xd, label, " ".join("%.1f" % (1e3 * t) for t in ts), X.nbytes / min(ts) / 1e9), flush=True)
eng.close()
pr = cProfile.Profile()
pr.enable()
call()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(int(os.environ.get("TOP", 45)))
print(out.getvalue())
