#!/usr/bin/env python
"""Wall-clock of the outer solver (nmf_pathway: rounds of 10 iterations + candidate scoring / restrict / matching on
the host) at the C3 shape, next to the time its inner iterations take on the GPU."""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd import solver  # noqa: E402

m, n, k, P, xd = syn.CONFIGS[os.environ.get("WORKLOAD", "c3")]
iters = int(os.environ.get("ITERS", 100))
X = syn.x_rows_torch(torch, "cuda:0", 0, n, k, 0, m, torch.float32).cpu().numpy()
lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0))
np.random.seed(0)
marks = []
orig_step = solver.PrmfEngine.step


def timed_step(self, *a, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig_step(self, *a, **kw)
    torch.cuda.synchronize()
    marks.append(time.perf_counter() - t0)
    return r


solver.PrmfEngine.step = timed_step
buf = io.StringIO()
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
    U, V, od = solver.nmf_pathway(X, None, library=lib, k_latent=k, max_iter=iters, x_dtype=xd, tol=0.0)
total = time.perf_counter() - t0
print("nmf_pathway %dx%d k=%d P=%d, %d iterations: %.2f s total; inner steps %.3f s (%d rounds, %.1f ms per round of 10); "
      "everything else (upload, scoring, restrict, matching, downloads) %.2f s"
      % (m, n, k, P, iters, total, sum(marks), len(marks), 1e3 * np.mean(marks), total - sum(marks)))

if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    np.random.seed(0)
    marks.clear()
    pr = cProfile.Profile()
    pr.enable()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        solver.nmf_pathway(X, None, library=lib, k_latent=k, max_iter=iters, x_dtype=xd, tol=0.0)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(22)
