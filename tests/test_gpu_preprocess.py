"""prmf_quantile_transform (prmf_amd/csrc/prep.hip) against scikit-learn's quantile_transform -- the call the
reference makes at script/prmf_runner.py:1019-1020 -- on shapes that exercise ties, constant columns, fewer
samples than quantiles, NaNs, the row subsample (m > 1e5) and several column blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sk(X, **kw):
    from sklearn.preprocessing import quantile_transform
    return quantile_transform(X, copy=True, **kw)


@pytest.mark.parametrize("m,n,kind", [
    (100, 37, "gamma"),            # the CLI fixtures' size: fewer samples than the 1000 quantiles
    (1500, 64, "gamma"),           # more samples than quantiles: interpolated order statistics
    (2500, 33, "ties"),            # heavy ties and repeated quantiles (ascending / descending interp differ)
    (1200, 20, "const"),           # constant and two-valued columns
    (900, 17, "nan"),              # missing values are ignored by the fit and stay NaN
    (3000, 700, "gamma"),          # several 32-column tiles
])
def test_matches_sklearn(m, n, kind):
    from prmf_amd.preprocess import quantile_transform
    rs = np.random.RandomState(m + n)
    X = rs.gamma(2.0, size=(m, n)) * rs.rand(1, n) * 100
    if kind == "ties":
        X = np.round(X / 25.0) * 25.0
        X[:, 3] = np.floor(rs.rand(m) * 3)
    if kind == "const":
        X[:, 0] = 7.25
        X[:, 1] = (rs.rand(m) < 0.5) * 1.0
        X[:, 2] = 0.0
        X[: m // 2, 4] = -3.0
    if kind == "nan":
        X[rs.rand(m, n) < 0.05] = np.nan
    want = sk(X)
    got = quantile_transform(X)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.max(np.abs(got[ok] - want[ok])) <= 1e-15, np.max(np.abs(got[ok] - want[ok]))


def test_row_subsample_matches_sklearn():
    """m > 1e5: scikit-learn fits the quantiles on a shuffled row subset drawn from the global numpy RNG."""
    from prmf_amd.preprocess import quantile_transform
    m, n = 100500, 6
    rs = np.random.RandomState(5)
    X = rs.gamma(2.0, size=(m, n))
    np.random.seed(11)
    want = sk(X)
    np.random.seed(11)
    got = quantile_transform(X)
    assert np.max(np.abs(got - want)) <= 1e-15
    # both consumed the global stream identically
    np.random.seed(11)
    sk(X)
    a = np.random.rand()
    np.random.seed(11)
    quantile_transform(X)
    assert np.random.rand() == a
