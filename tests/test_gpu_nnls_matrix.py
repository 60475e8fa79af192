"""prmf_nnls (prmf_amd/csrc/nnls_matrix.hip) against scipy.optimize.nnls -- the two solves per latent factor of
--manifolds-init (script/prmf_runner.py:308-334: nnls(X^T, v) and nnls(X, u))."""
import numpy as np
import pytest
import scipy.optimize

from prmf_amd.preprocess import DeviceNnls

pytestmark = pytest.mark.gpu


def both(X, b, transposed):
    A = X.T if transposed else X
    want, wres = scipy.optimize.nnls(A, b)
    got, gres = DeviceNnls(X).nnls(b, transposed=transposed)
    return got, gres, want, wres


def close(got, gres, want, wres, tol=1e-9):
    assert got.shape == want.shape and (got >= 0).all()
    assert np.linalg.norm(got - want) <= tol * max(np.linalg.norm(want), 1e-300), (np.linalg.norm(got - want), np.linalg.norm(want))
    assert abs(gres - wres) <= 1e-9 * max(wres, 1.0)
    assert np.array_equal(got > 0, want > 0) or np.abs(got - want).max() < 1e-12          # the same face


@pytest.mark.parametrize("m,n,seed", [(40, 300, 0), (200, 1500, 1), (7, 5000, 2), (300, 60, 3), (1, 50, 4), (120, 120, 5)])
def test_random_problems_both_orientations(m, n, seed):
    rs = np.random.RandomState(seed)
    X = rs.rand(m, n) + 0.05 * rs.gamma(2.0, size=(m, 4)).dot(rs.gamma(2.0, size=(4, n)))
    for transposed in (True, False):
        b = rs.rand(n if transposed else m)
        if seed % 2:
            b[rs.rand(b.shape[0]) < 0.3] = 0.0
        close(*both(X, b, transposed))


def test_the_initialiser_s_own_problems():
    """As --manifolds-init poses them: X quantile-normalised (values in [0, 1], no zeros), v a normalised pathway indicator with
    a raised plateau (pathway_to_vec), then u from the first solve as the right-hand side of the second."""
    from prmf_amd.runner import nmf_init_u, nmf_init_v
    rs = np.random.RandomState(11)
    m, n = 150, 4000
    raw = rs.gamma(5.0, size=(m, 6)).dot(rs.gamma(5.0, size=(n, 6)).T)
    X = (np.argsort(np.argsort(raw, axis=0), axis=0) + 0.5) / m          # rank-normalised per gene, like quantile_transform
    on = np.zeros(n, dtype=bool)
    on[rs.choice(n, 300, replace=False)] = True
    v = np.zeros(n)
    v[~on] = X.T[~on].mean(axis=1)
    v[on] = v[~on].sum() * 5 / on.sum()
    v = (v / np.linalg.norm(v)).reshape(n, 1)
    solver = DeviceNnls(X)
    u_ref, r1 = nmf_init_u(X, v)
    u_gpu, g1 = nmf_init_u(X, v, solver)
    assert np.linalg.norm(u_gpu - u_ref) <= 1e-9 * np.linalg.norm(u_ref) and abs(g1 - r1) <= 1e-9 * r1
    v_ref, r2 = nmf_init_v(X, u_ref)
    v_gpu, g2 = nmf_init_v(X, u_ref, solver)
    assert np.linalg.norm(v_gpu - v_ref) <= 1e-9 * np.linalg.norm(v_ref) and abs(g2 - r2) <= 1e-9 * max(r2, 1e-300)
    assert (u_ref == 0).any() and (v_ref == 0).sum() > n - m          # both solutions sit on a face
    again, _ = nmf_init_v(X, u_ref, solver)
    assert np.array_equal(again, v_gpu)                               # the same bits on a second call


def test_degenerate_inputs():
    rs = np.random.RandomState(1)
    X = rs.rand(20, 30)
    s = DeviceNnls(X)
    x, r = s.nnls(np.zeros(20))                          # b = 0: x = 0
    assert not x.any() and r == 0.0
    x, r = s.nnls(-np.ones(20))                          # no unknown can help: x = 0, residual = ||b||
    assert not x.any() and abs(r - np.sqrt(20.0)) < 1e-12
    b = X[:, 3] * 2.0 + X[:, 17] * 0.5                   # b inside the cone: residual 0, the generators found
    x, r = s.nnls(b)
    assert r < 1e-10 and np.linalg.norm(X.dot(x) - b) < 1e-10
    Xd = np.concatenate([X, X[:, :5]], axis=1)           # duplicated columns: dependent candidates are refused, not factored
    b = rs.rand(20) + 1.0
    want, wres = scipy.optimize.nnls(Xd, b)
    got, gres = DeviceNnls(Xd).nnls(b)                    # the minimiser is not unique here; its residual and its image are
    assert (got >= 0).all() and np.isfinite(got).all()
    assert abs(gres - wres) <= 1e-9 * max(wres, 1.0) and np.linalg.norm(Xd.dot(got) - Xd.dot(want)) <= 1e-8
    with pytest.raises(ValueError):
        s.nnls(np.ones(21))
    with pytest.raises(ValueError):
        s.nnls(np.full(20, np.nan))


def test_a_problem_beyond_the_normal_equations_is_declined_not_answered():
    """Rows scaled over twelve orders of magnitude with b inside the cone: the gradients that would still descend cannot be told
    from rounding in float64 inner products.  prmf_nnls says so (NnlsDeclined) instead of returning its stalled point, and the
    initialiser then solves that problem with scipy as the reference does."""
    from prmf_amd.preprocess import NnlsDeclined
    from prmf_amd.runner import nmf_init_v
    rs = np.random.RandomState(2)
    declined = 0
    for trial in range(12):
        m, n = 60 + 10 * trial, 90 + 7 * trial
        X = rs.rand(m, n) * 10.0 ** rs.randint(-6, 6, size=(m, 1))          # A = X: equations (rows) of wildly different scales
        b = X.dot(np.maximum(rs.randn(n), 0))
        solver = DeviceNnls(X)
        try:
            got, gres = solver.nnls(b)
        except NnlsDeclined:
            declined += 1
            try:                                                            # the command line's path: scipy takes over --
                want, wres = scipy.optimize.nnls(X, b)
            except RuntimeError as e:                                       # -- and may give up itself, in its own words
                with pytest.raises(RuntimeError, match=str(e)[:20]):
                    nmf_init_v(X, b.reshape(-1, 1), solver)
                continue
            v, res = nmf_init_v(X, b.reshape(-1, 1), solver)
            assert np.array_equal(v.ravel(), want / np.linalg.norm(want) ** 2) and res == wres
            continue
        r = b - X.dot(got)                                                  # answered: then it is optimal
        w = X.T.dot(r)
        tol = 1e-9 * (np.abs(X).T.dot(np.abs(r)) + np.abs(X).T.dot(np.abs(b)))
        assert (w[got == 0] <= tol[got == 0]).all() and (np.abs(w[got > 0]) <= tol[got > 0]).all()
        assert np.linalg.norm(r) <= np.linalg.norm(b - X.dot(scipy.optimize.nnls(X, b)[0])) + 1e-9 * np.linalg.norm(b)
    assert declined >= 1
