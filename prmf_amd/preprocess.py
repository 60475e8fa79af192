"""Pre-processing of the data matrix on the GPU.

`quantile_transform` is the per-gene quantile normalisation the reference's CLI applies unless --no-normalize is
given (script/prmf_runner.py:1019-1020: sklearn.preprocessing.quantile_transform with its defaults).  The HIP
implementation (prmf_amd/csrc/prep.hip, `prmf_quantile_transform`) reproduces scikit-learn 1.7 / numpy 2.2
operation for operation in float64; this module only moves the matrix to the device and, for more than 1e5
samples, draws the row subsample exactly as scikit-learn does (same global numpy RNG calls)."""
import ctypes as C

import numpy as np

from . import _capi

SUBSAMPLE = int(1e5)          # quantile_transform's default `subsample`


def _subsample_rows(m, random_state=None):
    """sklearn.utils.resample(X, replace=False, n_samples=SUBSAMPLE, random_state=random_state):
    indices = arange(m); random_state.shuffle(indices); indices[:SUBSAMPLE]."""
    if random_state is None or random_state is np.random:
        rs = np.random.mtrand._rand
    elif isinstance(random_state, (int, np.integer)):
        rs = np.random.RandomState(random_state)
    else:
        rs = random_state
    idx = np.arange(m)
    rs.shuffle(idx)
    return np.ascontiguousarray(idx[:SUBSAMPLE], dtype=np.int64)


def quantile_transform_block(Xb, fit_rows=None, n_quantiles=1000, device=0):
    """The same transform for a block of GENES (all samples x some columns, float64, C-contiguous) with the row
    subsample already drawn (None: all rows): columns are independent, so a matrix can be normalised block by block --
    what the rank-local input path of the multi-GPU CLI does (prmf_amd.runner.RowSource)."""
    import torch
    lib = _capi.load()
    m, nb = Xb.shape
    if m == 0 or nb == 0:
        return Xb
    Xd = torch.from_numpy(Xb).to("cuda:%d" % device)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    rc = lib.prmf_quantile_transform(int(device), C.c_void_p(Xd.data_ptr()), m, nb, nb,
                                     fit_rows.ctypes.data_as(C.c_void_p) if fit_rows is not None else None,
                                     len(fit_rows) if fit_rows is not None else m, int(n_quantiles), stream)
    if rc != _capi.PRMF_OK:
        raise _capi.PrmfError(rc, lib.prmf_last_error(None).decode())
    return Xd.cpu().numpy()


def _to_device_f64(torch, X, device):
    """A host matrix of any real dtype -> float64 on the device: uploaded in its OWN dtype in row chunks and widened there
    (float32 input: half the PCIe bytes and no float64 copy on the host; integers and float32 widen exactly)."""
    dev = "cuda:%d" % device
    if X.dtype == np.float64 or X.dtype.kind not in "fiu" or X.nbytes < (32 << 20):
        return torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    if X.dtype.kind in "iu" and X.dtype.itemsize == 8:
        return torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)      # 64-bit integers: numpy's own rounding
    if X.dtype not in (np.float32, np.float16, np.int8, np.int16, np.int32, np.uint8):
        # uint16 / uint32 / long double ...: torch.from_numpy does not take them; widened on the host
        return torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    out = torch.empty(X.shape, dtype=torch.float64, device=dev)
    rows = max(1, (160 << 20) // max(1, X.shape[1] * X.dtype.itemsize))
    for r0 in range(0, X.shape[0], rows):
        out[r0:r0 + rows] = torch.from_numpy(np.ascontiguousarray(X[r0:r0 + rows])).to(dev)
    return out


def quantile_transform(X, n_quantiles=1000, random_state=None, device=0, copy=True, on_device=False):
    """Drop-in for sklearn.preprocessing.quantile_transform(X) (uniform output, axis 0): numpy in, numpy out.
    on_device=True: the result stays where it was computed -- a float64 torch tensor on the GPU, which nmf_pathway takes
    as X -- instead of travelling to the host and back (the command line's single-GPU path)."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("prmf_amd.preprocess.quantile_transform needs a ROCm GPU; there is no CPU path here")
    lib = _capi.load()
    X = np.asarray(X)
    if X.dtype.kind not in "fiu":
        X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be 2-D (samples x genes)")
    m, n = X.shape
    if m == 0 or n == 0:
        X = np.asarray(X, dtype=np.float64)
        if on_device:
            return torch.from_numpy(np.ascontiguousarray(X)).to("cuda:%d" % device)
        return X.copy() if copy else X
    rows = _subsample_rows(m, random_state) if m > SUBSAMPLE else None
    Xd = _to_device_f64(torch, X, device)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    rc = lib.prmf_quantile_transform(int(device), C.c_void_p(Xd.data_ptr()), m, n, n,
                                     rows.ctypes.data_as(C.c_void_p) if rows is not None else None,
                                     len(rows) if rows is not None else m, int(n_quantiles), stream)
    if rc != _capi.PRMF_OK:
        raise _capi.PrmfError(rc, lib.prmf_last_error(None).decode())
    if on_device:
        return Xd
    out = Xd.cpu().numpy()
    if not copy and X.dtype == np.float64:
        X[...] = out
        return X
    return out


class NnlsDeclined(Exception):
    """prmf_nnls met a problem whose conditioning is beyond its normal-equations factor and returned nothing; the caller solves it
    with scipy.optimize.nnls (an orthogonal factorisation of the whole matrix) as the reference does."""


class DeviceNnls(object):
    """scipy.optimize.nnls(X, b) / nnls(X.T, b) for ONE data matrix and many right-hand sides, on the GPU (`prmf_nnls`,
    prmf_amd/csrc/nnls_matrix.hip): X is uploaded once as float64 and never copied or transposed again.  The two solves per
    latent factor of --manifolds-init (script/prmf_runner.py:308-334)."""

    def __init__(self, X, device=0):
        import torch
        self.lib = _capi.load()
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[0] < 1 or X.shape[1] < 1:
            raise ValueError("X must be a non-empty 2-D matrix")
        self.device = int(device)
        self.shape = X.shape
        self.Xd = _to_device_f64(torch, X, self.device).contiguous()
        self._torch = torch
        # scipy.optimize.nnls refuses a matrix with NaN / inf (ValueError); prmf_nnls would find no candidate in a NaN
        # gradient and return x = 0 with PRMF_OK.  One reduction on the device, once per matrix.
        if not bool(torch.isfinite(self.Xd).all().item()):
            raise ValueError("array must not contain infs or NaNs")

    def nnls(self, b, transposed=False):
        """-> (x, rnorm) like scipy.optimize.nnls(X.T if transposed else X, b)."""
        m, n = self.shape
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
        unknowns, equations = (m, n) if transposed else (n, m)
        if b.shape[0] != equations:
            raise ValueError("Expected a 1D array of length %d for the right-hand side, got %d" % (equations, b.shape[0]))
        if not np.all(np.isfinite(b)):
            raise ValueError("array must not contain infs or NaNs")
        x = np.zeros(unknowns, dtype=np.float64)
        rnorm, iters = C.c_double(), C.c_int()
        stream = C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)
        rc = self.lib.prmf_nnls(self.device, C.c_void_p(self.Xd.data_ptr()), m, n, n, 1 if transposed else 0,
                                b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.byref(rnorm), C.byref(iters), 0, stream)
        if rc == _capi.PRMF_ERR_DECLINED:          # a status of its own: every refusal, whatever its wording
            raise NnlsDeclined(self.lib.prmf_last_error(None).decode())
        if rc == _capi.PRMF_ERR_NUMERIC:
            raise RuntimeError("Maximum number of iterations reached.")          # scipy's words
        if rc != _capi.PRMF_OK:
            raise _capi.PrmfError(rc, self.lib.prmf_last_error(None).decode())
        self.last_iterations = iters.value
        return x, rnorm.value
