#!/usr/bin/env python
"""prmf_nnls against scipy.optimize.nnls on random problems: both orientations, 1 .. 400 rows / columns, dense positive data
(the initialiser's case), signed data, low-rank and duplicated columns (minimiser not unique: residual and image compared),
zero columns, badly scaled columns, right-hand sides with zeros / negative entries / inside the cone.
    SEED=1 CASES=300 python tools/fuzz_nnls.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import scipy.optimize  # noqa: E402
from prmf_amd.preprocess import DeviceNnls, NnlsDeclined  # noqa: E402

rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 200))
worst = 0.0
declined = 0
not_optimal = 0
t0 = time.time()
for case in range(ncases):
    m, n = int(rs.randint(1, 400)), int(rs.randint(1, 400))
    kind = rs.randint(6)
    if kind == 0:
        X = rs.rand(m, n)
    elif kind == 1:
        X = rs.randn(m, n)
    elif kind == 2:                                   # low rank + noise
        r = int(rs.randint(1, 6))
        X = rs.gamma(2.0, size=(m, r)).dot(rs.gamma(2.0, size=(r, n))) + 0.01 * rs.rand(m, n)
    elif kind == 3:                                   # exactly low rank / duplicated columns: the minimiser need not be unique
        r = int(rs.randint(1, 4))
        X = rs.rand(m, r).dot(rs.rand(r, n))
    elif kind == 4:                                   # zero and badly scaled columns
        X = rs.rand(m, n) * 10.0 ** rs.randint(-6, 6, size=(1, n))
        X[:, rs.rand(n) < 0.1] = 0.0
    else:
        X = (np.argsort(np.argsort(rs.rand(m, n), axis=0), axis=0) + 0.5) / m          # rank-normalised like quantile_transform
    transposed = bool(rs.rand() < 0.5)
    A = X.T if transposed else X
    eq = A.shape[0]
    bk = rs.randint(4)
    b = rs.rand(eq) if bk == 0 else (rs.randn(eq) if bk == 1 else (A.dot(np.maximum(rs.randn(A.shape[1]), 0)) if bk == 2 else rs.rand(eq) * (rs.rand(eq) < 0.5)))
    try:
        want, wres = scipy.optimize.nnls(A, b)
    except RuntimeError:
        continue                                      # scipy gave up (maximum iterations): nothing to compare with
    try:
        got, gres = DeviceNnls(X).nnls(b, transposed=transposed)
    except NnlsDeclined:
        declined += 1                                 # said so instead of answering: the command line then calls scipy
        continue
    except RuntimeError as e:
        print("CASE %d: %s  kind %d, %d x %d, transposed %s, rhs kind %d, scipy residual %.3e" % (case, e, kind, m, n, transposed, bk, wres))
        sys.exit(1)
    scale = max(np.linalg.norm(b), 1e-300)
    if os.environ.get("ONLY") and case == int(os.environ["ONLY"]):
        os.environ["PRMF_NNLS_DEBUG"] = "1"
        DeviceNnls(X).nnls(b, transposed=transposed)
        print("case", case, "kind", kind, m, n, transposed, "rhs kind", bk, "scipy", wres, "mine", gres, "support", (want > 0).sum(), (got > 0).sum())
        np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "nnls_case.npz"), X=X, b=b, transposed=transposed, got=got, want=want)
        sys.exit(0)
    assert (got >= 0).all() and np.isfinite(got).all(), (case, kind)
    # The judge is the problem itself (its optimality conditions), not the other solver: scipy 1.15's nnls returns points that are
    # not optimal on some small underdetermined problems (its reported rnorm then differs from ||A x - b|| of its own x).
    r = b - A.dot(got)
    w = A.T.dot(r)
    colmag = np.abs(A).T.dot(np.abs(r)) + np.abs(A).T.dot(np.abs(b))          # what the inner products can resolve
    tolw = 1e-9 * np.maximum(colmag, 1e-300)
    assert abs(np.linalg.norm(r) - gres) <= 1e-9 * scale, (case, "reported residual")
    assert (w[got == 0] <= tolw[got == 0]).all(), (case, kind, m, n, transposed, bk, "a free unknown still descends", (w - tolw)[got == 0].max())
    assert (np.abs(w[got > 0]) <= tolw[got > 0]).all(), (case, kind, m, n, transposed, bk, "a passive unknown is not at its minimum")
    true_wres = np.linalg.norm(b - A.dot(want))
    assert np.linalg.norm(r) <= true_wres + 1e-9 * scale, (case, kind, m, n, transposed, bk, np.linalg.norm(r), true_wres)
    scipy_ok = abs(true_wres - wres) <= 1e-9 * scale and np.linalg.norm(r) >= true_wres - 1e-9 * scale
    not_optimal += not scipy_ok
    if scipy_ok:
        assert np.linalg.norm(A.dot(got) - A.dot(want)) <= 1e-5 * scale, (case, kind, m, n, transposed, bk)
        if kind in (0, 1, 2, 5) and (want > 0).any() and np.linalg.matrix_rank(A[:, want > 0]) == (want > 0).sum() and wres > 1e-9 * scale:
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
            worst = max(worst, err)
            assert err <= 1e-6, (case, kind, m, n, transposed, err)
print("all %d problems: optimality conditions hold, residual <= scipy's, image equal and solutions within %.1e where scipy's point is optimal and unique; "
      "%d declined as ill-conditioned, scipy not optimal in %d (%.0f s)" % (ncases, worst, declined, not_optimal, time.time() - t0))
