"""ctypes binding of libprmf_hip.so (include/prmf_hip.h).

The product path has no CPU fallback: if the library is missing or does not
load, `load()` raises -- callers must not catch that to substitute other
arithmetic.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libprmf_hip.so")

PRMF_ABI_VERSION = 3
PRMF_X_F32, PRMF_X_BF16 = 0, 1
PRMF_SRC_F32, PRMF_SRC_BF16, PRMF_SRC_F64 = 0, 1, 2
PRMF_OBJ_STRIDE = 10
PRMF_OBJ_FLAGS, PRMF_OBJ_RECON_SQ_IDENTITY = 8, 9
PRMF_RECON_DIRECT, PRMF_RECON_IMPRECISE, PRMF_RECON_CLAMPED = 1, 2, 4
PRMF_FLAG_TIMING = 0x1
PRMF_FLAG_REPLICATED = 0x2
PRMF_FLAG_EAGER_OBJECTIVE = 0x4
PRMF_FLAG_DIRECT_RESIDUAL = 0x8
PRMF_DIST_COLUMN_SHARDED, PRMF_DIST_REPLICATED = 0, 1
PRMF_COMM_ID_BYTES = 128
PRMF_TEXT_RAGGED, PRMF_TEXT_QUOTED, PRMF_TEXT_NOT_NUMERIC, PRMF_TEXT_WIDE_INT, PRMF_TEXT_BARE_CR = 1, 2, 4, 8, 16
OBJ_KEYS = ("recon", "manifold", "ignore", "fro", "gamma", "delta", "obj")

PRMF_OK = 0
PRMF_ERR_ARG, PRMF_ERR_HIP, PRMF_ERR_STATE, PRMF_ERR_NUMERIC, PRMF_ERR_UNSUPPORTED, PRMF_ERR_DECLINED, PRMF_ERR_IO = -1, -2, -3, -4, -5, -6, -7


class CommSizes(C.Structure):
    _fields_ = [("gram_doubles", C.c_int64), ("q_send_floats", C.c_int64), ("q_recv_floats", C.c_int64),
                ("v_send_floats", C.c_int64), ("v_recv_floats", C.c_int64), ("kc", C.c_int32),
                ("mode", C.c_int32)]


class SweepInfo(C.Structure):
    _fields_ = [("bytes_x_sweep", C.c_int64), ("bytes_xt_sweep", C.c_int64),
                ("splits_xv", C.c_int32), ("splits_xtu", C.c_int32),
                ("grid_xv", C.c_int32), ("grid_xtu", C.c_int32), ("f32_split", C.c_int32), ("reserved", C.c_int32)]


class CommInfo(C.Structure):
    _fields_ = [("in_library", C.c_int32), ("comm_nranks", C.c_int32), ("comm_rank", C.c_int32), ("mode", C.c_int32),
                ("ms_phase", C.c_double * 4)]


# every symbol include/prmf_hip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _d, _u = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint
SIGNATURES = {
    "prmf_abi_version": (_i, []),
    "prmf_last_error": (C.c_char_p, [_vp]),
    "prmf_create": (_i, [C.POINTER(_vp), _i, _i64, _i64, _i, _i, _i, _i, _u]),
    "prmf_destroy": (_i, [_vp]),
    "prmf_set_x": (_i, [_vp, _vp, _i64, _d, _vp]),
    "prmf_set_x_device": (_i, [_vp, _vp, _i, _i64, C.POINTER(_d), _vp]),
    "prmf_set_x_host": (_i, [_vp, _vp, _i, _i64, _i, C.POINTER(_d), _vp]),
    "prmf_set_norm_x_sq": (_i, [_vp, _d]),
    "prmf_release_staging": (_i, []),
    "prmf_match_terms": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "prmf_set_pathway_library": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "prmf_set_active": (_i, [_vp, _vp, _vp]),
    "prmf_invalidate_factors": (_i, [_vp]),
    "prmf_step": (_i, [_vp, _vp, _vp, _i, _d, _d, _d, _vp, _vp, _vp]),
    "prmf_get_comm_sizes": (_i, [_vp, C.POINTER(CommSizes)]),
    "prmf_set_comm_buffers": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "prmf_begin": (_i, [_vp, _vp, _vp, _d, _d, _d, _vp]),
    "prmf_phase_u": (_i, [_vp, _vp]),
    "prmf_phase_q": (_i, [_vp, _vp]),
    "prmf_phase_v": (_i, [_vp, _vp]),
    "prmf_phase_post": (_i, [_vp, _vp]),
    "prmf_end": (_i, [_vp, _i, _vp, _vp, _vp]),
    "prmf_comm_unique_id": (_i, [_vp]),
    "prmf_comm_init": (_i, [_vp, _vp]),
    "prmf_step_dist": (_i, [_vp, _vp, _vp, _i, _d, _d, _d, _vp, _vp, _vp]),
    "prmf_set_global_rows": (_i, [_vp, _i64]),
    "prmf_set_comm_timeout": (_i, [_vp, _d]),
    "prmf_score_candidates": (_i, [_vp, _vp, _vp, _vp]),
    "prmf_cv_error": (_i, [_vp, _vp, _vp, _vp]),
    "prmf_residual_sq": (_i, [_vp, _vp, _vp, C.POINTER(_d), _vp]),
    "prmf_nnls": (_i, [_i, _vp, C.c_int64, C.c_int64, C.c_int64, _i, _vp, _vp, C.POINTER(_d), C.POINTER(_i), _i, _vp]),
    "prmf_sum_squares": (_i, [_i, _vp, _i, C.c_int64, C.c_int64, C.c_int64, C.POINTER(_d), _vp]),
    "prmf_quantile_transform": (_i, [_i, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int64, _i, _vp]),
    "prmf_text_create": (_i, [C.POINTER(_vp), _i, C.c_int64]),
    "prmf_text_append": (_i, [_vp, _vp, C.c_int64]),
    "prmf_text_index": (_i, [_vp, C.c_char, C.c_int64, C.POINTER(C.c_int64), C.POINTER(_i)]),
    "prmf_text_parse": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(C.c_int64)]),
    "prmf_text_destroy": (_i, [_vp]),
    "prmf_write_csv": (_i, [C.c_char_p, C.c_char_p, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_char, _i]),
    "prmf_get_sweep_info": (_i, [_vp, C.POINTER(SweepInfo)]),
    "prmf_set_timing": (_i, [_vp, _i]),
    "prmf_get_timing": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d)]),
    "prmf_comm_info": (_i, [_vp, C.POINTER(CommInfo)]),
}

# validation / measurement hooks (include/prmf_hip_debug.h); not used by the product path
DEBUG_SIGNATURES = {
    "prmf_debug_slab_sum": (_i, [_vp, _i, _vp, _vp]),
    "prmf_debug_recon_terms": (_i, [_vp, _i, C.POINTER(_d), _vp, _vp]),
    "prmf_debug_set_stamps": (_i, [_vp, C.c_size_t]),
    "prmf_debug_stamp_layout": (_i, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "prmf_debug_set_epi_stamps": (_i, [_vp]),
    "prmf_debug_bump_cursor": (_i, [_vp]),
    "prmf_debug_swap": (_i, [_vp, _vp, _i]),
    "prmf_debug_plan_sweep": (_i, [_i, C.c_longlong, C.c_longlong, _i, _i, C.POINTER(_i)]),
    "prmf_debug_stream": (_i, [_vp, _i, _i, C.POINTER(_d), C.POINTER(_i64), _vp]),
    "prmf_debug_sync_stream": (_i, [_vp, _vp]),
    "prmf_debug_resident_shape": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "prmf_debug_resident_x": (_i, [_vp, _i, _vp]),
    "prmf_debug_set_tail_probe": (_i, [_vp, _i, _vp, _i]),
}

_lib = None


class PrmfLibraryError(RuntimeError):
    pass


class PrmfError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libprmf_hip error %d: %s" % (code, message))
        self.code = code


def load(path=None):
    """Load libprmf_hip.so and declare every entry point; raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # PRMF_HIP_LIB: another build of the SAME library (tools/ab_sweep.py compares kernel variants in one GPU session);
    # it is still libprmf_hip -- there is no other implementation to fall back to
    p = path or os.environ.get("PRMF_HIP_LIB") or LIB_PATH
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  libprmf_hip.so
    # must share that HIP runtime instance (torch owns the device buffers and the
    # stream), so torch is always loaded first: the dynamic loader then resolves our
    # DT_NEEDED libamdhip64.so.7 to the copy that is already mapped.  Loading ours
    # first leaves two runtimes in the process and the second one finds no device.
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise PrmfLibraryError(
            "%s not found: build it with `python -m prmf_amd._build` (hipcc, gfx950). "
            "There is no CPU fallback for the PRMF hot path." % p)
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise PrmfLibraryError("cannot load %s: %s" % (p, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DEBUG_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.prmf_abi_version() != PRMF_ABI_VERSION:
        raise PrmfLibraryError("ABI version mismatch: library %d, binding %d" %
                               (lib.prmf_abi_version(), PRMF_ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != PRMF_OK:
        msg = lib.prmf_last_error(ctx)
        raise PrmfError(rc, msg.decode("utf-8", "replace") if msg else "")
