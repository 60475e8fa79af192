#!/usr/bin/env python
"""Random shapes through every way X enters the library (prmf_set_x_host / prmf_set_x_device, round 6): 1 x 1 up to a few
chunks of 32 MB, float32 / float64 sources, bf16 / float32 storage, contiguous and strided host rows, 0 .. 16 narrowing
threads, the direct lane forced on and off, device tensors of the source and of the storage type -- and a fraction of the
cases made of awkward BIT PATTERNS (denormals, infinities, NaNs, rounding ties).  Every path must leave the bits torch's own
conversion produces in both resident copies, zeros in the padding, and the float64 sum of the stored squares (the same bits
from every path of one case).  Not part of the test suite; a failure prints the case and exits non-zero.
    SEED=1 CASES=60 python tools/fuzz_setx.py"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402
from test_gpu_setx import awkward_values, bits, resident  # noqa: E402

rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 60))


def draw():
    kind = rs.randint(5)
    if kind == 0:
        m, n = rs.randint(1, 70), rs.randint(1, 70)
    elif kind == 1:
        m, n = rs.randint(1, 40), rs.randint(1000, 30000)
    elif kind == 2:
        m, n = rs.randint(1000, 30000), rs.randint(1, 40)
    elif kind == 3:          # several 32 MB chunks of source
        m, n = rs.randint(2000, 6000), rs.randint(2000, 6000)
    else:
        m, n = rs.randint(30, 2000), rs.randint(30, 3000)
    return dict(m=int(m), n=int(n), src=str(rs.choice(["float32", "float64"])), x_dtype=str(rs.choice(["bf16", "f32"])),
                awkward=bool(rs.rand() < 0.3), pad=int(rs.choice([0, 1, 7, 64, 1000])), threads=int(rs.choice([1, 2, 3, 5, 8, 16])),
                seed=int(rs.randint(1 << 30)))


def with_hybrid(fn):
    def run(e):
        os.environ["PRMF_SETX_HYBRID"] = "1"
        try:
            fn(e)
        finally:
            del os.environ["PRMF_SETX_HYBRID"]
    return run


t_start = time.time()
for ci in range(ncases):
    c = draw()
    m, n = c["m"], c["n"]
    r2 = np.random.RandomState(c["seed"])
    src = np.float32 if c["src"] == "float32" else np.float64
    X = awkward_values(r2, m, n, src) if c["awkward"] else (r2.rand(m, n) * np.exp(r2.uniform(-3, 3))).astype(src)
    tdt = torch.bfloat16 if c["x_dtype"] == "bf16" else torch.float32
    with np.errstate(all="ignore"):
        want = torch.from_numpy(X).cuda().to(tdt)
        want_bits = want.view(torch.int16 if c["x_dtype"] == "bf16" else torch.int32).cpu().numpy().view(np.uint16 if c["x_dtype"] == "bf16" else np.uint32)
        ref = float((want.double() ** 2).sum().item())
    strided = np.ascontiguousarray(np.pad(X, ((0, 0), (0, c["pad"]))))[:, :n] if c["pad"] else X
    paths = {
        "host, default threads": lambda e: e.set_x(X),
        "host, %d threads, strided by %d" % (c["threads"], c["pad"]): lambda e: e.set_x(strided, host_threads=c["threads"]),
        "host, narrowed on the device": lambda e: e.set_x(strided, host_threads=0),
        "host, both lanes": with_hybrid(lambda e: e.set_x(X, host_threads=c["threads"])),
        "device, source type": lambda e: e.set_x(torch.from_numpy(X).cuda()),
        "device, storage type": lambda e: e.set_x(want),
        "async": lambda e: e.set_x_async(X, host_threads=c["threads"]).result(),
    }
    norms, bad = {}, []
    for name, feed in paths.items():
        eng = PrmfEngine(m, n, int(rs.choice([1, 6, 33])), x_dtype=c["x_dtype"])
        try:
            feed(eng)
            Xr, XTr = resident(eng, 0), resident(eng, 1)
            norms[name] = eng.local_sq
            if eng.local_norm_sq(want) != eng.local_sq and not (np.isnan(eng.local_sq) and np.isnan(eng.local_norm_sq(want))):
                bad.append(name + ": prmf_sum_squares differs from the upload's norm")
        finally:
            eng.close()
        if not np.array_equal(bits(Xr[:m, :n]), want_bits):
            bad.append(name + ": X")
        if not np.array_equal(bits(XTr[:n, :m]), want_bits.T):
            bad.append(name + ": X^T")
        if bits(Xr[m:]).any() or bits(Xr[:, n:]).any() or bits(XTr[n:]).any() or bits(XTr[:, m:]).any():
            bad.append(name + ": padding not zero")
    vals = list(norms.values())
    if not all((v == vals[0]) or (np.isnan(v) and np.isnan(vals[0])) for v in vals):
        bad.append("norms differ between paths: %r" % norms)
    if np.isfinite(ref) and not abs(vals[0] - ref) <= 1e-12 * max(ref, 1e-300):
        bad.append("norm %r != float64 sum of the stored squares %r" % (vals[0], ref))
    print("%3d %s  %s" % (ci, " ".join("%s=%s" % kv for kv in c.items()), ("FAIL: " + "; ".join(bad)) if bad else "ok (norm %.6g)" % vals[0]), flush=True)
    if bad:
        sys.exit(1)
print("all %d cases: every upload path stores torch's bits in X and X^T, zero padding, one norm (%.0f s)" % (ncases, time.time() - t_start))
