"""How X enters the library (prmf_set_x / prmf_set_x_device / prmf_set_x_host; replaces the `X` argument of
nmf_manifold_vec_update, script/prmf_runner.py:374, and `norm_X = np.linalg.norm(X)`, :640): every path must leave the SAME
bits in the two resident copies and report the same ||X||^2 -- the conversions are torch's (`Tensor.to(dtype)`), the norm is
the float64 sum of the squares of the stored values."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def resident(eng, which):
    """the padded resident copy (0: X, 1: X^T) as a uint16 / float32 array [rows][ld]"""
    rows, ld = C.c_int64(), C.c_int64()
    eng._check(eng.lib.prmf_debug_resident_shape(eng.ctx, which, C.byref(rows), C.byref(ld)))
    out = np.empty((rows.value, ld.value), dtype=np.uint16 if eng.x_dtype == 1 else np.float32)
    eng._check(eng.lib.prmf_debug_resident_x(eng.ctx, which, C.c_void_p(out.ctypes.data)))
    return out


def bits(a):
    return a.view(np.uint16 if a.dtype.itemsize == 2 else np.uint32)


def awkward_values(rs, m, n, dtype):
    """random BIT PATTERNS of the source type (denormals, infinities, NaNs, ties of the rounding) next to ordinary values"""
    if dtype == np.float32:
        raw = rs.randint(0, 2 ** 32, size=(m, n), dtype=np.uint64).astype(np.uint32).view(np.float32)
    else:
        hi = rs.randint(0, 2 ** 32, size=(m, n), dtype=np.uint64) << np.uint64(32)
        raw = (hi | rs.randint(0, 2 ** 32, size=(m, n), dtype=np.uint64)).view(np.float64)
        # most random doubles are outside float32's range: bring half of them into it
        scale = rs.rand(m, n) < 0.5
        raw = np.where(scale, rs.standard_normal((m, n)) * np.exp(rs.uniform(-90, 90, (m, n))), raw)
    plain = rs.rand(m, n).astype(dtype)
    pick = rs.rand(m, n) < 0.5
    out = np.where(pick, raw, plain).astype(dtype)
    # exact ties of f32 -> bf16 (the 16 dropped bits are 0x8000): round to even both ways
    if dtype == np.float32:
        t = (rs.randint(0, 2 ** 15, size=n, dtype=np.uint32) << np.uint32(16)) | np.uint32(0x8000) | np.uint32(0x3f000000)
        out[0, :] = t.view(np.float32)
    return out


@pytest.mark.parametrize("x_dtype", ["bf16", "f32"])
@pytest.mark.parametrize("src", [np.float32, np.float64])
def test_every_upload_path_stores_the_same_bits(x_dtype, src):
    import torch
    from prmf_amd.solver import PrmfEngine
    rs = np.random.RandomState(11)
    m, n, k = 777, 1301, 8          # ragged: 13 tile rows (the last 9 rows), 21 tile columns (the last 21 columns)
    X = awkward_values(rs, m, n, src)
    tdt = torch.bfloat16 if x_dtype == "bf16" else torch.float32
    with np.errstate(all="ignore"):
        want = torch.from_numpy(X).cuda().to(tdt)          # the conversion torch performs: the contract
    want_np = want.view(torch.int16 if x_dtype == "bf16" else torch.int32).cpu().numpy()
    want_bits = want_np.view(np.uint16 if x_dtype == "bf16" else np.uint32)
    stored = {}
    norms = {}
    def hybrid(e):
        # PRMF_SETX_HYBRID=1: every other chunk takes the direct lane (source type over the wire, narrowed by the layout kernel)
        # while the workers narrow its neighbour -- what the library does by itself when the host threads are starved
        import os
        os.environ["PRMF_SETX_HYBRID"] = "1"
        try:
            e.set_x(X, host_threads=3)
        finally:
            del os.environ["PRMF_SETX_HYBRID"]

    paths = {
        "legacy prmf_set_x (device, storage type)": lambda e: e._legacy_set_x(want),
        "host, both lanes (narrowed chunks next to direct ones)": hybrid,
        "device, source type": lambda e: e.set_x(torch.from_numpy(X).cuda()),
        "host, narrowed by 5 threads": lambda e: e.set_x(X, host_threads=5),
        "host, narrowed on the device": lambda e: e.set_x(X, host_threads=0),
        "host, strided rows": lambda e: e.set_x(np.ascontiguousarray(np.pad(X, ((0, 0), (0, 37))))[:, :n], host_threads=3),
    }
    for name, feed in paths.items():
        eng = PrmfEngine(m, n, k, x_dtype=x_dtype)
        try:
            feed(eng)
            stored[name] = (resident(eng, 0), resident(eng, 1))
            norms[name] = eng.local_sq if name != "legacy prmf_set_x (device, storage type)" else None
        finally:
            eng.close()
    for name, (Xr, XTr) in stored.items():
        assert np.array_equal(bits(Xr[:m, :n]), want_bits), name
        assert np.array_equal(bits(XTr[:n, :m]), want_bits.T), name
        # the padding reads as zero
        assert not bits(Xr[m:]).any() and not bits(Xr[:, n:]).any(), name
        assert not bits(XTr[n:]).any() and not bits(XTr[:, m:]).any(), name
    vals = [v for v in norms.values() if v is not None]
    assert all((v == vals[0]) or (np.isnan(v) and np.isnan(vals[0])) for v in vals), norms


@pytest.mark.parametrize("x_dtype,m,n", [("bf16", 4000, 9000), ("f32", 3000, 5000), ("bf16", 70, 33)])
def test_norm_is_chunking_invariant_and_exact(x_dtype, m, n):
    """||X||^2 from the one-pass upload: the same bits whatever the number of chunks the matrix arrives in (a 144 MB source
    is five chunks, a device tensor one launch), equal to prmf_sum_squares, and the float64 sum of the stored squares."""
    import torch
    from prmf_amd.solver import PrmfEngine
    rs = np.random.RandomState(3)
    X = rs.rand(m, n).astype(np.float32)
    tdt = torch.bfloat16 if x_dtype == "bf16" else torch.float32
    Xd = torch.from_numpy(X).cuda().to(tdt)
    ref = float((Xd.double() ** 2).sum().item())
    got = []
    def both_lanes(e):
        import os
        os.environ["PRMF_SETX_HYBRID"] = "1"
        try:
            e.set_x(X, host_threads=4)
        finally:
            del os.environ["PRMF_SETX_HYBRID"]

    for feed in (lambda e: e.set_x(X, host_threads=4), lambda e: e.set_x(X, host_threads=0), lambda e: e.set_x(Xd),
                 lambda e: e.set_x(torch.from_numpy(X).cuda()), both_lanes):
        eng = PrmfEngine(m, n, 4, x_dtype=x_dtype)
        try:
            feed(eng)
            got.append(eng.local_sq)
            assert eng.local_norm_sq(Xd) == eng.local_sq
        finally:
            eng.close()
    assert len(set(got)) == 1, got
    assert abs(got[0] - ref) <= 1e-12 * ref


def test_async_upload_overlaps_and_reports_errors():
    import torch
    from prmf_amd.solver import PrmfEngine
    rs = np.random.RandomState(5)
    X = rs.rand(300, 500)
    eng = PrmfEngine(300, 500, 4, x_dtype="f32")
    try:
        pend = eng.set_x_async(X)
        draws = np.random.rand(10)          # the caller's own work under the upload
        assert pend.result() == eng.local_sq and eng.norm_x_sq == eng.local_sq and draws.shape == (10,)
        with pytest.raises(ValueError):
            eng.set_x_async(X[:100])
        # a later upload replaces the matrix
        eng.set_x(torch.from_numpy(2 * X).cuda())
        assert abs(eng.local_sq - 4 * float(np.sum(np.float32(X).astype(np.float64) ** 2))) < 1e-6 * eng.local_sq
    finally:
        eng.close()
