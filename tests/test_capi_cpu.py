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


def test_missing_library_is_an_error(tmp_path):
    from prmf_amd import _capi
    with pytest.raises(_capi.PrmfLibraryError):
        _capi.load(str(tmp_path / "libprmf_hip.so"))
