#!/usr/bin/env python
"""The GPU text reader by stage (wall clock, second pass of the process) next to pd.read_csv on the same file:
reading the file into the 64 MB buffer, prmf_text_append (host -> device), prmf_text_index (count, scan, positions),
prmf_text_parse (kernel + the matrix back to the host), the row names through pandas, the DataFrame.
    M=500 N=20000 python tools/time_text_reader.py          (run under rocprofv3 --kernel-trace --stats for the kernels)"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from prmf_amd import _capi, textio  # noqa: E402

m, n = int(os.environ.get("M", 500)), int(os.environ.get("N", 20000))
d = tempfile.mkdtemp(prefix="prmf_text_")
path = os.path.join(d, "data.csv")
rs = np.random.RandomState(0)
X = rs.gamma(5.0, size=(m, 6)).dot(rs.gamma(5.0, size=(n, 6)).T)
with open(path, "w") as fh:
    fh.write("," + ",".join("ENSP%d" % j for j in range(n)) + "\n")
    for i in range(m):
        fh.write("s%d," % i + ",".join(map(repr, X[i].tolist())) + "\n")
size = os.path.getsize(path)
print("%d x %d as text: %.1f MB, %.1f bytes per field" % (m, n, size / 1e6, size / (m * n)), flush=True)
lib = _capi.load()
for rep in range(2):
    t = {}
    t0 = time.perf_counter()
    with open(path, "rb") as fh:
        first = fh.readline()
        body_off = len(first)
        h = C.c_void_p()
        _capi.check(lib, None, lib.prmf_text_create(C.byref(h), 0, size - body_off))
        buf = bytearray(min(textio.CHUNK_BYTES, size - body_off))
        cbuf = (C.c_char * len(buf)).from_buffer(buf)
        t["create"] = time.perf_counter() - t0
        t["read"] = t["append"] = 0.0
        while True:
            a = time.perf_counter()
            got = fh.readinto(buf)
            b = time.perf_counter()
            t["read"] += b - a
            if not got:
                break
            _capi.check(lib, None, lib.prmf_text_append(h, cbuf, got))
            t["append"] += time.perf_counter() - b
    rows, flags = C.c_int64(), C.c_int()
    a = time.perf_counter()
    _capi.check(lib, None, lib.prmf_text_index(h, b",", n + 1, C.byref(rows), C.byref(flags)))
    t["index"] = time.perf_counter() - a
    body = np.empty((rows.value, n))
    spans = np.empty((rows.value, 2), dtype=np.int64)
    isint = np.empty(n, dtype=np.int32)
    a = time.perf_counter()
    _capi.check(lib, None, lib.prmf_text_parse(h, 1, body.ctypes.data_as(C.c_void_p), spans.ctypes.data_as(C.c_void_p),
                                               isint.ctypes.data_as(C.c_void_p), C.byref(flags), None))
    t["parse"] = time.perf_counter() - a
    lib.prmf_text_destroy(h)
    t["total by hand"] = time.perf_counter() - t0
    a = time.perf_counter()
    frame = textio.read_frame(path, ",", True, True)
    t["read_frame"] = time.perf_counter() - a
    a = time.perf_counter()
    want = pd.read_csv(path, sep=",", index_col=0)
    t["pd.read_csv"] = time.perf_counter() - a
    assert flags.value == 0 and (frame.to_numpy().view(np.int64) == want.to_numpy().view(np.int64)).all() and (frame.index == want.index).all()
    print("pass %d: " % rep + " | ".join("%s %.1f" % (k, 1e3 * v) for k, v in t.items()) + "  (ms)   -> %.2f GB/s of text, %.0fx pandas"
          % (size / t["read_frame"] / 1e9, t["pd.read_csv"] / t["read_frame"]), flush=True)
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
