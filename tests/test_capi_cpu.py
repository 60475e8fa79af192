"""CPU checks of the boundary: the header, the ctypes binding and the built library
agree on every entry point, and the library refuses to run without a GPU instead
of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="prmf_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(prmf_[a-z_]+)\s*\(", txt)))


def test_header_binding_and_library_export_the_same_symbols():
    from prmf_amd import _build, _capi
    _build.build()
    lib = _capi.load()
    names = header_functions()
    assert names, "no functions parsed from include/prmf_hip.h"
    assert sorted(_capi.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.prmf_abi_version() == _capi.PRMF_ABI_VERSION
    # the validation hooks have a header of their own; together the two headers name EVERY prmf_* export of the library
    dbg = header_functions("prmf_hip_debug.h")
    assert dbg and sorted(_capi.DEBUG_SIGNATURES) == dbg
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\b(prmf_[a-z_]+)$", nm, flags=re.M)))
    assert exported == sorted(names + dbg), set(exported) ^ set(names + dbg)
    txt = open(os.path.join(ROOT, "include", "prmf_hip.h")).read()
    assert "#define PRMF_OBJ_STRIDE %d" % _capi.PRMF_OBJ_STRIDE in txt and "#define PRMF_ABI_VERSION %d" % _capi.PRMF_ABI_VERSION in txt


def test_error_paths_without_compute():
    import torch
    from prmf_amd import _capi
    lib = _capi.load()
    ctx = C.c_void_p()
    # bad arguments are rejected before any device work
    assert lib.prmf_create(C.byref(ctx), 0, 10, 0, 4, 0, 1, 0, 0) == _capi.PRMF_ERR_ARG
    assert b"bad shape" in lib.prmf_last_error(None)
    assert lib.prmf_create(C.byref(ctx), 0, 10, 10, 4000, 0, 1, 0, 0) == _capi.PRMF_ERR_UNSUPPORTED
    assert lib.prmf_create(C.byref(ctx), 0, 10, 10, 4, 7, 1, 0, 0) == _capi.PRMF_ERR_ARG
    assert lib.prmf_create(C.byref(ctx), 0, 10, 10, 4, 0, 2, 2, 0) == _capi.PRMF_ERR_ARG
    if not torch.cuda.is_available():
        # no GPU: creation fails loudly, there is no CPU fallback
        assert lib.prmf_create(C.byref(ctx), 0, 10, 10, 4, 0, 1, 0, 0) == _capi.PRMF_ERR_HIP
        assert not ctx.value
        from prmf_amd.solver import PrmfEngine
        with pytest.raises(RuntimeError):
            PrmfEngine(10, 10, 4)
    assert lib.prmf_destroy(None) == 0
    assert lib.prmf_step(None, None, None, 1, 1.0, 1.0, -1.0, None, None, None) == _capi.PRMF_ERR_ARG


def test_headers_are_plain_c_and_cxx(tmp_path):
    """the boundary is a C ABI: both headers compile as strict C99 and as C++11, with nothing but <stddef.h> / <stdint.h> behind them"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "prmf_hip.h"\n#include "prmf_hip_debug.h"\nint use(void) { return prmf_abi_version() + (int)sizeof(prmf_comm_sizes) + PRMF_OBJ_STRIDE; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_debug_stamp_buffer_length_is_checked():
    """the sweep's debug stamps go to fixed, disjoint regions of the caller's buffer: a buffer shorter than the layout is refused"""
    from prmf_amd import _capi
    lib = _capi.load()
    where, phase, wgs = C.c_int(), C.c_int(), C.c_int()
    n = lib.prmf_debug_stamp_layout(C.byref(where), C.byref(phase), C.byref(wgs))
    assert wgs.value >= 512 and where.value >= 2 * wgs.value and phase.value >= where.value + wgs.value
    assert n >= phase.value + wgs.value * 8 * 8                  # 8 phase sums for each of 8 waves of every workgroup
    fake = C.c_void_p(0x1000)                                    # never dereferenced: the call only records the pointer
    assert lib.prmf_debug_set_stamps(fake, n - 1) == _capi.PRMF_ERR_ARG
    assert lib.prmf_debug_set_stamps(fake, n) == 0
    assert lib.prmf_debug_set_stamps(None, 0) == 0


def test_missing_library_is_an_error(tmp_path):
    from prmf_amd import _capi
    with pytest.raises(_capi.PrmfLibraryError):
        _capi.load(str(tmp_path / "libprmf_hip.so"))


def test_sweep_plans_of_the_benchmark_shapes():
    """The sweep planner is pure host arithmetic (prmf_debug_plan_sweep touches no GPU): pin what it decides for the
    shapes DESIGN.md section 7 quotes, so that a threshold edit that changes one of them is a deliberate edit.
    X V is the sweep over X (m rows, depth n), X^T U the one over X^T (n rows, depth m)."""
    from prmf_amd import _capi
    lib = _capi.load()
    BF16, F32 = _capi.PRMF_X_BF16, _capi.PRMF_X_F32

    def plan(xd, rows, depth, k):
        out = (C.c_int * 8)()
        assert lib.prmf_debug_plan_sweep(xd, rows, depth, k, 256, out) == _capi.PRMF_OK
        keys = ("waves", "tiles_per_wave", "panel_rows", "grid", "slices", "max_slabs", "double_buffered", "panels")
        return dict(zip(keys, list(out)))

    # C3 50000 x 20000, k = 64, bf16 (round 4): BOTH sweeps in the 8-wave form with the ping-pong chunk loop, unsliced -- X V (5.1 MB
    # factor: >= 4 MB and >= 16384 rows) and X^T U (12.8 MB); "double_buffered" carries the ping-pong flag in bit 1
    xv, xtu = plan(BF16, 50000, 20000, 64), plan(BF16, 20000, 50000, 64)
    assert (xv["waves"], xv["panel_rows"], xv["grid"], xv["slices"], xv["panels"], xv["double_buffered"]) == (8, 512, 256, 1, 98, 3)
    assert (xtu["waves"], xtu["panel_rows"], xtu["grid"], xtu["slices"], xtu["panels"], xtu["double_buffered"]) == (8, 512, 256, 1, 40, 3)
    # the same shape with f32 X: both sweeps 8-wave (round 5: X V too, from a 4 MB factor on >= 16384 rows), no ping-pong
    # (64-row wave tiles at k = 33..64); its four-rank share (12500 rows) keeps the 4-wave form for X V
    xv, xtu = plan(F32, 50000, 20000, 64), plan(F32, 20000, 50000, 64)
    assert (xv["waves"], xv["tiles_per_wave"], xv["slices"], xv["double_buffered"]) == (8, 2, 1, 1)
    assert (xtu["waves"], xtu["tiles_per_wave"], xtu["slices"], xtu["double_buffered"]) == (8, 2, 1, 1)
    assert plan(F32, 12500, 20000, 64)["waves"] == 4 and plan(F32, 50000, 20000, 128)["waves"] == 8
    # the two- and four-rank shares of C3: 25000 rows take the ping-pong form for X V, 12500 rows (< 16384) keep the 4-wave one
    assert (plan(BF16, 25000, 20000, 64)["waves"], plan(BF16, 25000, 20000, 64)["double_buffered"]) == (8, 3)
    assert (plan(BF16, 12500, 20000, 64)["waves"], plan(BF16, 12500, 20000, 64)["double_buffered"]) == (4, 1)
    # k > 64: no ping-pong (measured equal: profiles/r04_ab_k128_pingpong.txt)
    assert plan(BF16, 200000, 25000, 128)["double_buffered"] == 1
    # C5 200000 x 25000, k = 128, bf16: both sweeps 8-wave; X V (12.8 MB) unsliced, X^T U (102 MB) sliced
    xv, xtu = plan(BF16, 200000, 25000, 128), plan(BF16, 25000, 200000, 128)
    assert (xv["waves"], xv["slices"], xv["double_buffered"], xv["panels"]) == (8, 1, 1, 391)
    assert xtu["waves"] == 8 and xtu["slices"] > 1 and xtu["double_buffered"] == 1 and xtu["panels"] == 49
    # its N = 8 share (25000 x 25000): both 8-wave, unsliced; with f32 X: 32-row wave tiles
    for p in (plan(BF16, 25000, 25000, 128), plan(F32, 25000, 25000, 128)):
        assert (p["waves"], p["slices"]) == (8, 1)
    assert plan(F32, 25000, 25000, 128)["tiles_per_wave"] == 1 and plan(F32, 25000, 25000, 128)["panel_rows"] == 256
    # C2 5000 x 10000, k = 16, f32: small factors, 4 waves x one 32-row tile, unsliced, double-buffered
    for p in (plan(F32, 5000, 10000, 16), plan(F32, 10000, 5000, 16)):
        assert (p["waves"], p["tiles_per_wave"], p["panel_rows"], p["slices"], p["double_buffered"]) == (4, 1, 128, 1, 1)
    # the N = 8 share of C3 (6250 x 20000): X V's 5.1 MB factor takes two slices there, X^T U (1.6 MB) none
    assert plan(BF16, 6250, 20000, 64)["slices"] == 2 and plan(BF16, 20000, 6250, 64)["slices"] == 1
    # a tiny matrix launches few workgroups (round 5: at least 2 chunks each and then at most one per CU; at least 4 where that
    # still fills one per CU) and never more slabs than the 64-bit mask holds
    tiny = plan(F32, 100, 2000, 6)
    assert tiny["grid"] == 32 and tiny["max_slabs"] <= 64
    assert plan(F32, 2000, 2000, 6)["grid"] == 256 and plan(F32, 500, 2000, 6)["grid"] == 128
    assert plan(F32, 500, 20000, 6)["grid"] == 256      # the reference's typical size: X V (depth 20000) streams as before ...
    # ... and X^T U, a reduction over 500 samples, is the EXACT short-reduction product since round 6 (depth <= 512 after padding:
    # float32 MFMA against the unpacked factor; 256-row panels; one wave per (32-row tile, 32-column tile, reduction slice); bit 2)
    xtu = plan(F32, 20000, 500, 6)
    assert (xtu["double_buffered"] & 4, xtu["panel_rows"], xtu["max_slabs"], xtu["slices"], xtu["panels"]) == (4, 256, 4, 1, 79)
    assert xtu["grid"] == (79 * 8 * 1 * 4 + 3) // 4            # 632 (tile, column tile) pairs x 4 reduction slices, 4 waves per workgroup
    # the limit is (padded depth) x (32-column tiles) <= 512 (the float32 pipe's cost grows with both)
    assert plan(F32, 20000, 512, 32)["double_buffered"] & 4 and not plan(F32, 20000, 513, 32)["double_buffered"] & 4
    # one column tile and at most 4096 rows (X V of a few hundred samples over <= 1024 genes): depth up to 1024
    assert plan(F32, 4096, 1024, 32)["double_buffered"] & 4 and not plan(F32, 4097, 1024, 32)["double_buffered"] & 4
    assert not plan(F32, 4096, 1025, 32)["double_buffered"] & 4 and not plan(F32, 100, 1024, 64)["double_buffered"] & 4
    assert plan(BF16, 20000, 256, 64)["double_buffered"] & 4 and not plan(BF16, 20000, 257, 64)["double_buffered"] & 4
    assert plan(BF16, 20000, 128, 128)["double_buffered"] & 4 and not plan(BF16, 20000, 129, 128)["double_buffered"] & 4
    # the reference's test problems (100 samples x 1000 genes, k = 6): X^T U exact -- few rows, so 2 slices of its 4 chunks --, X V (1000 genes) too
    small = plan(F32, 1000, 100, 6)
    assert small["double_buffered"] & 4 and small["max_slabs"] == 2 and plan(F32, 100, 1000, 6)["double_buffered"] & 4
    for xd, rows, depth, k in ((BF16, 50000, 20000, 64), (BF16, 20000, 50000, 64), (F32, 10000, 5000, 16), (BF16, 20000, 6250, 64)):
        assert not plan(xd, rows, depth, k)["double_buffered"] & 4                                  # no benchmark shape is touched
    # bad arguments
    out = (C.c_int * 8)()
    assert lib.prmf_debug_plan_sweep(BF16, 10, 0, 4, 256, out) == _capi.PRMF_ERR_ARG
    assert lib.prmf_debug_plan_sweep(7, 10, 10, 4, 256, out) == _capi.PRMF_ERR_ARG
