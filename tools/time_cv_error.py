#!/usr/bin/env python
"""Time prmf_cv_error at CV scale (20% of C3 held out): m_test x n x k = 10000 x 20000 x 64."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

m, n, k = int(os.environ.get("M", 10000)), int(os.environ.get("N", 20000)), int(os.environ.get("K", 64))
dev = "cuda:0"
X = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.float32)
V = syn.init_factor_torch(torch, dev, 2, n, k)
eng = PrmfEngine(m, n, k, x_dtype="f32")
eng.set_x(X)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    err = eng.cv_error(V)
    torch.cuda.synchronize()
    print("cv_error %dx%dx%d: %.1f ms  mean err %.4f" % (m, n, k, 1e3 * (time.perf_counter() - t0), float(np.nanmean(err))))
# scipy on a sample of rows for scale
import scipy.optimize
Vh, Xh = V.double().cpu().numpy(), X[:20].double().cpu().numpy()
t0 = time.perf_counter()
ref = np.array([scipy.optimize.nnls(Vh, Xh[i])[1] / np.linalg.norm(Xh[i]) for i in range(20)])
dt = time.perf_counter() - t0
print("scipy nnls: %.1f ms/sample -> %.0f s for %d samples; max rel diff on 20 rows %.2e"
      % (1e3 * dt / 20, dt / 20 * m, m, np.max(np.abs(err[:20] - ref) / ref)))
eng.close()
