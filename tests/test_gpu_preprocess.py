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


def test_result_left_on_the_device_and_narrow_inputs():
    """on_device=True returns the same float64 values as a tensor on the GPU; a float32 / integer host matrix is widened on
    the device (uploaded in its own dtype, in row chunks once it is large) to the same float64 values numpy would give."""
    import torch
    from prmf_amd.preprocess import quantile_transform
    rs = np.random.RandomState(2)
    X32 = (rs.gamma(2.0, size=(3000, 3100)) * 50).astype(np.float32)          # 37 MB: crosses the chunked-upload threshold
    want = quantile_transform(X32.astype(np.float64))
    got = quantile_transform(X32)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    dev = quantile_transform(X32, on_device=True)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda and dev.dtype == torch.float64 and np.array_equal(dev.cpu().numpy(), want)
    counts = rs.randint(0, 1000, size=(400, 30))
    assert np.array_equal(quantile_transform(counts), sk(counts.astype(np.float64)))
    assert np.array_equal(quantile_transform(counts.astype(np.int16)), sk(counts.astype(np.float64)))


def test_solver_takes_the_normalised_matrix_from_the_device():
    """nmf_pathway(X) with X a float64 tensor on the GPU (what the command line hands over after the normalisation) gives, bit
    for bit, what it gives for the same values as a host array -- both storages of X."""
    import synth
    from prmf_amd import PathwayLibrary
    from prmf_amd.preprocess import quantile_transform
    from prmf_amd.solver import nmf_pathway
    rs = np.random.RandomState(9)
    m, n, k = 90, 700, 5
    raw = rs.gamma(2.0, size=(m, n)) * 10
    lib = PathwayLibrary.from_edge_lists(n, synth.tree_pathways(n, 8, 40, 3))
    host = quantile_transform(raw)
    dev = quantile_transform(raw, on_device=True)
    for xd in ("f32", "bf16"):
        out = []
        for X in (host, dev):
            np.random.seed(4)
            out.append(nmf_pathway(X, None, library=lib, k_latent=k, max_iter=40, x_dtype=xd))
        (U0, V0, o0), (U1, V1, o1) = out
        assert np.array_equal(U0, U1) and np.array_equal(V0, V1) and o0["obj"] == o1["obj"]
