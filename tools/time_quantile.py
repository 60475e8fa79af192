#!/usr/bin/env python
"""prmf_quantile_transform at the C3 shape (50000 x 20000 float64 = 8 GB) next to scikit-learn on a column sample."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd.preprocess import quantile_transform  # noqa: E402

m, n = int(os.environ.get("M", 50000)), int(os.environ.get("N", 20000))
rs = np.random.RandomState(0)
X = rs.gamma(2.0, size=(m, n))
t0 = time.perf_counter()
Y = quantile_transform(X)
t1 = time.perf_counter()
print("GPU quantile_transform %dx%d: %.2f s wall (float64 upload + transform + download)" % (m, n, t1 - t0))
from sklearn.preprocessing import quantile_transform as sk  # noqa: E402
nc = 200
t0 = time.perf_counter()
W = sk(X[:, :nc], copy=True)
dt = time.perf_counter() - t0
print("scikit-learn on %d of the columns: %.2f s -> %.0f s for all %d; max |diff| on them %.1e"
      % (nc, dt, dt * n / nc, n, np.abs(W - Y[:, :nc]).max()))
