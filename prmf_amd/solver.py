"""Host side of the MI355X PRMF hot path: a device-resident engine plus drop-in
counterparts of the reference's solver functions

    nmf_manifold_vec_update            script/prmf_runner.py:374-451
    nmf_manifold_vec_update_tradeoff   script/prmf_runner.py:497-554
    nmf_pathway                        script/prmf_runner.py:556-792

with the same names, argument meaning, stdout trace and error behaviour.  All
arithmetic of the iteration runs in libprmf_hip.so (hand-written HIP for gfx950)
through the ctypes C-ABI of include/prmf_hip.h.  PyTorch is used only to own
device buffers and to name the current HIP stream.  There is no CPU fallback.
"""
import ctypes as C
import datetime
import math
import sys

import numpy as np

from . import _capi
from .pathways import PathwayLibrary

EPSILON = np.finfo(np.float32).eps     # prmf_runner.py:23


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("prmf_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU path for the PRMF inner loop")
    return torch


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class PrmfEngine(object):
    """One GPU's share of a PRMF solve: X (and X^T) resident in HBM, the pathway
    library resident, U/V updated in place by `step`.

    nranks > 1: X is row-sharded (this rank holds m_local samples) and the
    phase_* methods are driven by prmf_amd.distributed with collectives between
    them; see include/prmf_hip.h.  dist_mode: "colshard" (default: the V update is
    sharded by latent columns -- reduce-scatter of X^T U, all-gather of V) or
    "replicated" (every rank updates all of V from the all-reduced X^T U)."""

    def __init__(self, m_local, n, k, x_dtype="f32", device=None, nranks=1, rank=0, timing=False, dist_mode=None,
                 eager_objective=False, direct_residual=False):
        self.torch = _torch()
        self.lib = _capi.load()
        torch = self.torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.m, self.n, self.k = int(m_local), int(n), int(k)
        self.x_dtype = {"f32": _capi.PRMF_X_F32, "fp32": _capi.PRMF_X_F32, "float32": _capi.PRMF_X_F32,
                        "bf16": _capi.PRMF_X_BF16, "bfloat16": _capi.PRMF_X_BF16}[x_dtype]
        self.torch_x_dtype = torch.float32 if self.x_dtype == _capi.PRMF_X_F32 else torch.bfloat16
        self.nranks, self.rank = int(nranks), int(rank)
        self.ctx = C.c_void_p()
        flags = _capi.PRMF_FLAG_TIMING if timing else 0
        if dist_mode not in (None, "replicated", "colshard"):
            raise ValueError("dist_mode must be 'replicated' or 'colshard'")
        if dist_mode == "replicated":
            flags |= _capi.PRMF_FLAG_REPLICATED
        if eager_objective:       # per-iteration objective launches instead of the deferred evaluation (prmf_hip.h)
            flags |= _capi.PRMF_FLAG_EAGER_OBJECTIVE
        if direct_residual:       # recon of EVERY record from the materialised-residual form of prmf_runner.py:337
            flags |= _capi.PRMF_FLAG_DIRECT_RESIDUAL
        self.last_flags = []      # PRMF_RECON_* bits of the records the last step / end call returned
        rc = self.lib.prmf_create(C.byref(self.ctx), self.device, self.m, self.n, self.k, self.x_dtype,
                                  self.nranks, self.rank, flags)
        if rc != _capi.PRMF_OK:
            raise _capi.PrmfError(rc, self.lib.prmf_last_error(None).decode())
        self.library = None
        self._comm = None
        self.sizes = _capi.CommSizes()
        self._check(self.lib.prmf_get_comm_sizes(self.ctx, C.byref(self.sizes)))
        self.dist_mode = "replicated" if self.sizes.mode == _capi.PRMF_DIST_REPLICATED else "colshard"

    # -- plumbing -------------------------------------------------------------
    def _check(self, rc):
        _capi.check(self.lib, self.ctx, rc)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.prmf_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_tensor(self, a, dtype=None):
        """numpy / torch -> contiguous tensor on this engine's GPU.  A large host matrix travels in row chunks in its own
        dtype and is converted ON THE DEVICE: `from_numpy(X).to(device, dtype)` converts on one host thread first
        (50000 x 20000 f32 -> bf16: 0.45 s, f64 -> f32: 0.91 s; chunked: 0.07 s = 54 GB/s, profiles/r03_outer_solver.txt)."""
        torch = self.torch
        dev = "cuda:%d" % self.device
        if isinstance(a, torch.Tensor) and a.is_cuda and a.ndim == 2 and a.numel() * a.element_size() > (32 << 20) and dtype not in (None, a.dtype):
            out = torch.empty(a.shape, dtype=dtype, device=dev)          # a device matrix: narrowed in row chunks, no whole-size temporary
            rows = max(1, (160 << 20) // max(1, a.shape[1] * a.element_size()))
            for r0 in range(0, a.shape[0], rows):
                out[r0:r0 + rows] = a[r0:r0 + rows].to(device=dev)
            return out
        if isinstance(a, np.ndarray) and a.ndim == 2 and a.nbytes > (32 << 20) and a.dtype in (np.float32, np.float64):
            a = np.ascontiguousarray(a)
            out = torch.empty(a.shape, dtype=dtype or torch.from_numpy(a[:0]).dtype, device=dev)
            rows = max(1, (160 << 20) // max(1, a.shape[1] * a.itemsize))
            for r0 in range(0, a.shape[0], rows):
                out[r0:r0 + rows] = torch.from_numpy(a[r0:r0 + rows]).to(device=dev)
            return out
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=dev, dtype=dtype or a.dtype).contiguous()

    # -- inputs ---------------------------------------------------------------
    def _x_source(self, X):
        """(kind, object kept alive, pointer, PRMF_SRC_* code, leading dimension) of a matrix prmf_set_x_host / _device can
        read in place: a float32 / float64 ndarray with contiguous rows, or a 2-D float32 / bfloat16 / float64 tensor on
        this GPU; anything else is widened to float64 on the host first (exact for integers below 2^53 and for float16)."""
        torch = self.torch
        if isinstance(X, torch.Tensor):
            if X.is_cuda and X.ndim == 2 and X.dtype in (torch.float32, torch.bfloat16, torch.float64):
                if X.device.index != self.device:
                    X = X.to("cuda:%d" % self.device)
                if X.shape[1] > 0 and X.stride(1) != 1:
                    X = X.contiguous()
                code = {torch.float32: _capi.PRMF_SRC_F32, torch.bfloat16: _capi.PRMF_SRC_BF16, torch.float64: _capi.PRMF_SRC_F64}[X.dtype]
                ld = X.stride(0) if X.shape[0] > 1 else X.shape[1]
                return "device", X, X.data_ptr(), code, max(int(ld), X.shape[1])
            X = X.detach().cpu().to(torch.float64).numpy() if X.dtype != torch.float32 else X.detach().cpu().numpy()
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X must be a matrix, got shape {}".format(X.shape))
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        if ((X.shape[1] > 0 and X.strides[1] != X.itemsize) or (X.shape[0] > 1 and X.strides[0] < X.shape[1] * X.itemsize)
                or X.strides[0] % X.itemsize):
            X = np.ascontiguousarray(X)
        ld = X.strides[0] // X.itemsize if X.shape[0] > 1 else X.shape[1]
        return "host", X, X.ctypes.data, (_capi.PRMF_SRC_F32 if X.dtype == np.float32 else _capi.PRMF_SRC_F64), max(int(ld), X.shape[1])

    def set_x(self, X, norm_x_sq=None, host_threads=-1):
        """This rank's rows of X into the library: a numpy matrix (float32 / float64; it travels chunk by chunk, narrowed to
        the engine's storage type on the way -- `prmf_set_x_host`) or a tensor on the GPU (float32 / bfloat16 / float64 --
        `prmf_set_x_device`).  One pass writes the padded copy, the padded transpose and ||X||^2 of the stored values
        (`self.local_sq`: this rank's rows).  norm_x_sq = ||X||_F^2 of the GLOBAL matrix; None (single rank): the local
        value.  Returns the norm the engine now holds."""
        if norm_x_sq is None and self.nranks != 1:
            raise ValueError("norm_x_sq (global) is required with nranks > 1 (or: set_x_local, add the ranks' sums, set_norm_x_sq)")
        self.set_x_async(X, norm_x_sq, host_threads, thread=False).result()
        return self.norm_x_sq

    def set_x_local(self, X, host_threads=-1):
        """`set_x` of a multi-rank run's first half: the rows go in, the LOCAL sum of squares comes back; the caller adds
        the ranks' values and hands the total to `set_norm_x_sq`."""
        return self.set_x_async(X, None, host_threads, thread=False).result()

    def set_x_async(self, X, norm_x_sq=None, host_threads=-1, thread=True):
        """`set_x` on a helper thread (the library call releases the GIL: a 4 GB upload is 0.05-0.08 s during which the
        caller can draw its random factors and build the pathway library).  `.result()` joins, re-raises what the upload
        raised and returns the local sum of squares."""
        kind, keep, ptr, code, ld = self._x_source(X)
        if tuple(keep.shape) != (self.m, self.n):
            raise ValueError("X has shape {} != (m_local, n) = {}".format(tuple(keep.shape), (self.m, self.n)))
        stream = self._stream()
        box = {"keep": keep}                # the source stays alive until the copy has finished, and no longer
        del keep, X

        def work():
            try:
                out = C.c_double()
                if kind == "device":
                    rc = self.lib.prmf_set_x_device(self.ctx, C.c_void_p(ptr), code, ld, C.byref(out), stream)
                else:
                    rc = self.lib.prmf_set_x_host(self.ctx, C.c_void_p(ptr), code, ld, int(host_threads), C.byref(out), stream)
                self._check(rc)
                box["value"] = out.value
            except BaseException as e:      # re-raised by result()
                box["error"] = e

        class Pending(object):
            def result(_self):
                if th is not None:
                    th.join()
                box.pop("keep", None)
                if "error" in box:
                    raise box["error"]
                self.local_sq = box["value"]
                if norm_x_sq is not None:
                    self.set_norm_x_sq(norm_x_sq)
                else:
                    self.norm_x_sq = box["value"]
                return box["value"]

        th = None
        if not thread:
            work()
        else:
            import threading
            th = threading.Thread(target=work, name="prmf-set-x")
            th.start()
        return Pending()

    def _legacy_set_x(self, Xd, norm_x_sq=0.0):
        """`prmf_set_x` as rounds 1-5 called it: a device matrix already in the storage type, the norm supplied by the
        caller, asynchronous on the stream (kept in the ABI; tests compare it with the one-pass entry points)."""
        self._check(self.lib.prmf_set_x(self.ctx, C.c_void_p(Xd.data_ptr()), Xd.stride(0), float(norm_x_sq), self._stream()))
        self.torch.cuda.current_stream(self.device).synchronize()
        self.norm_x_sq = float(norm_x_sq)

    def set_norm_x_sq(self, norm_x_sq):
        """||X||_F^2 of the GLOBAL matrix (multi-rank: the sum of the ranks' local values)"""
        self._check(self.lib.prmf_set_norm_x_sq(self.ctx, float(norm_x_sq)))
        self.norm_x_sq = float(norm_x_sq)

    def local_norm_sq(self, Xd):
        """sum of squares of the rounded local X (a contiguous device tensor in the engine's dtype) in float64
        (`prmf_sum_squares`: one kernel pass, the summation tree of the upload: the value `set_x` returns)."""
        torch = self.torch
        if not (isinstance(Xd, torch.Tensor) and Xd.is_cuda and Xd.dtype == self.torch_x_dtype and Xd.ndim == 2 and Xd.is_contiguous()):
            raise ValueError("local_norm_sq takes a contiguous 2-D device tensor of the engine's x dtype")
        out = C.c_double()
        self._check(self.lib.prmf_sum_squares(self.device, C.c_void_p(Xd.data_ptr()), self.x_dtype, Xd.shape[0], Xd.shape[1], Xd.shape[1],
                                              C.byref(out), self._stream()))
        return out.value

    def set_library(self, library):
        if library.n != self.n:
            raise ValueError("pathway library is over %d genes, engine over %d" % (library.n, self.n))
        arrs = library.arrays()
        self._check(self.lib.prmf_set_pathway_library(self.ctx, library.P, *[_ptr(a) for a in arrs], self._stream()))
        self.library = library

    def set_active(self, k_to_lapl):
        a = np.ascontiguousarray(np.asarray(k_to_lapl, dtype=np.int32))
        if a.shape != (self.k,):
            raise ValueError("k_to_lapl must have k_latent = %d entries" % self.k)
        self._check(self.lib.prmf_set_active(self.ctx, _ptr(a), self._stream()))

    def invalidate_factors(self):
        self._factor_key = None
        self._check(self.lib.prmf_invalidate_factors(self.ctx))

    def product(self, which):
        """Dense copy of the product the last sweep left in the slabs (validation only):
        "xv" -> P = X V (m x k) between phase_u and phase_q, "xtu" -> Q = X^T U (n x k) after phase_q."""
        torch = self.torch
        rows = self.m if which == "xv" else self.n
        out = torch.empty((rows, self.k), dtype=torch.float32, device="cuda:%d" % self.device)
        self._check(self.lib.prmf_debug_slab_sum(self.ctx, 0 if which == "xv" else 1, C.c_void_p(out.data_ptr()),
                                                 self._stream()))
        return out

    # -- the hot path ----------------------------------------------------------
    def _check_factors(self, U, V):
        torch = self.torch
        for name, t, shape in (("U", U, (self.m, self.k)), ("V", V, (self.n, self.k))):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
            if tuple(t.shape) != shape:
                raise ValueError("%s has shape %s != %s" % (name, tuple(t.shape), shape))

    def _bind_factors(self, U, V):
        """The library caches the packed V and V^T V of the factors it last updated and keys that cache on
        the raw device pointers only.  A different tensor can live at the same address (torch's caching
        allocator hands freed blocks out again) and a caller can edit V in place, so the cache is kept only
        when U and V are the very tensor objects of the previous call and their version counters have not
        moved (the library writes through raw pointers: `_version` changes on caller edits only)."""
        import weakref
        key = (U.data_ptr(), U._version, V.data_ptr(), V._version)
        prev = getattr(self, "_factor_key", None)
        same = (prev is not None and prev[0] == key and prev[1]() is U and prev[2]() is V)
        if not same:
            self.invalidate_factors()
        self._factor_key = (key, weakref.ref(U), weakref.ref(V))
        self._bound = (U, V)      # keep the tensors alive while the library holds their pointers

    def step(self, U, V, n_steps, gamma, delta, tradeoff=None):
        """n_steps iterations in place on device tensors U (m x k), V (n x k).
        Returns (list of objective dicts, one per iteration; (gamma, delta) for
        the next call).  Contract: U and V are updated in place through their raw
        pointers; the factors cached inside the library (packed V, V^T V) are reused
        only when the same tensor objects come back unmodified (`_bind_factors`)."""
        self._check_factors(U, V)
        self._bind_factors(U, V)
        obj = np.zeros((max(n_steps, 1), _capi.PRMF_OBJ_STRIDE))
        gd = np.zeros(2)
        t = -1.0 if tradeoff is None else float(tradeoff)
        rc = self.lib.prmf_step(self.ctx, C.c_void_p(U.data_ptr()), C.c_void_p(V.data_ptr()), int(n_steps),
                                float(gamma), float(delta), t, _ptr(obj), _ptr(gd), self._stream())
        if rc == _capi.PRMF_ERR_NUMERIC:
            # the reference runs under np.seterr(divide='raise') (prmf_runner.py:22)
            raise FloatingPointError(self.lib.prmf_last_error(self.ctx).decode())
        self._check(rc)
        return self.records_to_dicts(obj[:n_steps]), (float(gd[0]), float(gd[1]))

    def records_to_dicts(self, obj):
        """Records -> the reference's obj_data dicts (prmf_runner.py:363-371: exactly its seven keys -- the CLI writes
        every key to obj.txt); how each recon was obtained (include/prmf_hip.h PRMF_RECON_*) goes to `last_flags`."""
        self.last_flags = [int(row[_capi.PRMF_OBJ_FLAGS]) for row in obj]
        return [{key: float(row[j]) for j, key in enumerate(_capi.OBJ_KEYS)} for row in obj]

    def residual_sq(self, U, V):
        """sum (X_g - U_g V^T)^2 over this rank's rows, evaluated directly (prmf_runner.py:337 before the root)."""
        self._check_factors(U, V)
        out = C.c_double()
        self._check(self.lib.prmf_residual_sq(self.ctx, C.c_void_p(U.data_ptr()), C.c_void_p(V.data_ptr()), C.byref(out),
                                              self._stream()))
        return out.value

    # -- phase interface (multi-rank; driven by prmf_amd.distributed) -------------
    def comm_buffers(self):
        """Allocate the collective buffers as torch tensors on this GPU and hand
        their pointers to the library.  With one rank, and in the replicated
        multi-rank mode (Q all-reduced in place), q_recv / v_recv alias the send buffers."""
        if self._comm is None:
            torch, s = self.torch, self.sizes
            dev = "cuda:%d" % self.device
            b = {"gram": torch.zeros(s.gram_doubles, dtype=torch.float64, device=dev),
                 "q_send": torch.zeros(s.q_send_floats, dtype=torch.float32, device=dev),
                 "v_send": torch.zeros(s.v_send_floats, dtype=torch.float32, device=dev)}
            if self.nranks == 1 or self.dist_mode == "replicated":
                b["q_recv"], b["v_recv"] = b["q_send"], b["v_send"]
            else:
                b["q_recv"] = torch.zeros(s.q_recv_floats, dtype=torch.float32, device=dev)
                b["v_recv"] = torch.zeros(s.v_recv_floats, dtype=torch.float32, device=dev)
            p = lambda t: C.c_void_p(t.data_ptr())
            self._check(self.lib.prmf_set_comm_buffers(self.ctx, p(b["gram"]), p(b["q_send"]), p(b["q_recv"]),
                                                       p(b["v_send"]), p(b["v_recv"])))
            self._comm = b
        return self._comm

    def begin(self, U, V, gamma, delta, tradeoff=None):
        self._check_factors(U, V)
        self._bind_factors(U, V)
        t = -1.0 if tradeoff is None else float(tradeoff)
        self._check(self.lib.prmf_begin(self.ctx, C.c_void_p(U.data_ptr()), C.c_void_p(V.data_ptr()),
                                        float(gamma), float(delta), t, self._stream()))

    def phase_u(self):
        self._check(self.lib.prmf_phase_u(self.ctx, self._stream()))

    def phase_q(self):
        self._check(self.lib.prmf_phase_q(self.ctx, self._stream()))

    def phase_v(self):
        self._check(self.lib.prmf_phase_v(self.ctx, self._stream()))

    def phase_post(self):
        self._check(self.lib.prmf_phase_post(self.ctx, self._stream()))

    def end(self, n_steps):
        obj = np.zeros((max(n_steps, 1), _capi.PRMF_OBJ_STRIDE))
        gd = np.zeros(2)
        rc = self.lib.prmf_end(self.ctx, int(n_steps), _ptr(obj), _ptr(gd), self._stream())
        if rc == _capi.PRMF_ERR_NUMERIC:
            raise FloatingPointError(self.lib.prmf_last_error(self.ctx).decode())
        self._check(rc)
        return self.records_to_dicts(obj[:n_steps]), (float(gd[0]), float(gd[1]))

    # -- collectives inside the library (RCCL on the stream; no per-iteration host work) ----
    def comm_init(self, group=None):
        """Create this rank's RCCL communicator inside the library: rank 0 draws the
        unique id, the rank group (prmf_amd.ranks: a torch.distributed group or thread ranks) hands the 128 bytes
        round, every rank joins.

        Every decision is taken COLLECTIVELY so that no rank ends up inside ncclCommInitRank (or later inside
        prmf_step_dist's nccl* calls) while another one has given up: (1) every rank checks that the library can
        bind RCCL at all and the flags are MIN-reduced before anything else; (2) after ncclCommInitRank the result
        flags are MIN-reduced again.  Any failure raises PrmfError on EVERY rank."""
        import os
        from .ranks import as_ranks
        idbuf = np.zeros(_capi.PRMF_COMM_ID_BYTES, dtype=np.uint8)
        t = os.environ.get("PRMF_COLLECTIVE_TIMEOUT_S")
        if t:
            self.set_comm_timeout(float(t))
        ranks = as_ranks(group)
        if self.nranks == 1 and not ranks.multi:
            self._check_create(self.lib.prmf_comm_unique_id(_ptr(idbuf)))
            self._check(self.lib.prmf_comm_init(self.ctx, _ptr(idbuf)))
            self._native_comm = True
            return
        if ranks.world != self.nranks or ranks.rank != self.rank:
            raise ValueError("engine is rank %d of %d, the group says %d of %d" % (self.rank, self.nranks, ranks.rank, ranks.world))
        cdev = "cuda:%d" % self.device

        def all_ok(local_ok, what):
            if ranks.min_int(1 if local_ok else 0, cdev) == 0:
                raise _capi.PrmfError(_capi.PRMF_ERR_UNSUPPORTED,
                                      "%s failed on at least one rank (this rank: %s); no rank uses the in-library collectives"
                                      % (what, "ok" if local_ok else self.lib.prmf_last_error(None).decode()))

        scratch = np.zeros(_capi.PRMF_COMM_ID_BYTES, dtype=np.uint8)
        rc = self.lib.prmf_comm_unique_id(_ptr(idbuf if ranks.rank == 0 else scratch))   # binds librccl
        all_ok(rc == _capi.PRMF_OK, "binding RCCL (prmf_comm_unique_id)")
        idbuf = np.ascontiguousarray(ranks.bytes_from_rank0(idbuf, cdev))
        rc = self.lib.prmf_comm_init(self.ctx, _ptr(idbuf))
        all_ok(rc == _capi.PRMF_OK, "prmf_comm_init (ncclCommInitRank)")
        self._native_comm = True

    def set_global_rows(self, m_global):
        """samples over all ranks (host-driven collectives; prmf_comm_init finds it out itself)"""
        self._check(self.lib.prmf_set_global_rows(self.ctx, int(m_global)))

    def set_comm_timeout(self, seconds):
        self._check(self.lib.prmf_set_comm_timeout(self.ctx, float(seconds)))

    def _check_create(self, rc):
        if rc != _capi.PRMF_OK:
            raise _capi.PrmfError(rc, self.lib.prmf_last_error(None).decode())

    def step_dist(self, U, V, n_steps, gamma, delta, tradeoff=None):
        """Multi-rank `step`: the library runs phases and collectives on the stream."""
        self._check_factors(U, V)
        self._bind_factors(U, V)
        obj = np.zeros((max(n_steps, 1), _capi.PRMF_OBJ_STRIDE))
        gd = np.zeros(2)
        t = -1.0 if tradeoff is None else float(tradeoff)
        rc = self.lib.prmf_step_dist(self.ctx, C.c_void_p(U.data_ptr()), C.c_void_p(V.data_ptr()), int(n_steps),
                                     float(gamma), float(delta), t, _ptr(obj), _ptr(gd), self._stream())
        if rc == _capi.PRMF_ERR_NUMERIC:
            raise FloatingPointError(self.lib.prmf_last_error(self.ctx).decode())
        self._check(rc)
        return self.records_to_dicts(obj[:n_steps]), (float(gd[0]), float(gd[1]))

    def score_candidates(self, V):
        """k x P matrix of restrict's match scores (prmf_runner.py:115-127)."""
        out = np.zeros((self.k, self.library.P))
        self._check(self.lib.prmf_score_candidates(self.ctx, C.c_void_p(V.data_ptr()), _ptr(out), self._stream()))
        return out

    def match_terms(self, V):
        """(quad, ignore), both k x P: v_k^T L_p v_k with the un-normalised Laplacian and sum_{i in S_p} 1 / (V[i, k] + 1)
        -- what force_distinct_lapls (prmf_runner.py:227-246) weighs its matching with (`prmf_match_terms`)."""
        quad = np.zeros((self.k, self.library.P))
        ign = np.zeros((self.k, self.library.P))
        self._check(self.lib.prmf_match_terms(self.ctx, C.c_void_p(V.data_ptr()), _ptr(quad), _ptr(ign), self._stream()))
        return quad, ign

    def cv_error(self, V):
        """Per-row NNLS reconstruction error of this engine's X (the held-out samples) against V
        (measure_cv_performance, prmf/__init__.py:768-798)."""
        if tuple(V.shape) != (self.n, self.k) or V.dtype != self.torch.float32 or not V.is_contiguous():
            raise ValueError("V must be a contiguous float32 tensor of shape {}".format((self.n, self.k)))
        out = np.zeros(self.m)
        rc = self.lib.prmf_cv_error(self.ctx, C.c_void_p(V.data_ptr()), _ptr(out), self._stream())
        if rc == _capi.PRMF_ERR_NUMERIC:
            raise RuntimeError(self.lib.prmf_last_error(self.ctx).decode())
        self._check(rc)
        return out

    def comm_info(self):
        """dict: in_library, comm_nranks / comm_rank as RCCL reports them, mode, ms_phase (see prmf_hip.h)."""
        info = _capi.CommInfo()
        self._check(self.lib.prmf_comm_info(self.ctx, C.byref(info)))
        return {"in_library": bool(info.in_library), "comm_nranks": int(info.comm_nranks), "comm_rank": int(info.comm_rank),
                "mode": "replicated" if info.mode == _capi.PRMF_DIST_REPLICATED else "colshard",
                "ms_phase": [float(x) for x in info.ms_phase]}

    def sweep_info(self):
        info = _capi.SweepInfo()
        self._check(self.lib.prmf_get_sweep_info(self.ctx, C.byref(info)))
        return info

    def stream_read(self, which="x", reps=5):
        """(ms per launch, bytes per launch) of a pure non-temporal stream of the resident X ("x") or X^T ("xt"): the
        calibration bench.py prices the sweeps against (include/prmf_hip_debug.h: prmf_debug_stream)."""
        ms, nbytes = C.c_double(), C.c_int64()
        self._check(self.lib.prmf_debug_stream(self.ctx, 0 if which == "x" else 1, int(reps), C.byref(ms), C.byref(nbytes),
                                               self._stream()))
        return ms.value, nbytes.value

    def set_timing(self, enabled):
        self._check(self.lib.prmf_set_timing(self.ctx, 1 if enabled else 0))

    def timing(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.prmf_get_timing(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


# ---------------------------------------------------------------------------
# drop-in inner loop  (prmf_runner.py:374-451 / :497-554)
# ---------------------------------------------------------------------------
def _run_inner(X, U, V, k_to_W, k_to_feat_inds, n_steps, gamma, delta, i, verbose, tradeoff, x_dtype, engine):
    m, k_latent = U.shape
    n, _ = V.shape
    own = engine is None
    if own:
        engine = PrmfEngine(m, n, k_latent, x_dtype=x_dtype)
        engine.set_x(X)
    try:
        library = PathwayLibrary.from_sparse([k_to_W[k] for k in range(k_latent)],
                                             [k_to_feat_inds[k] for k in range(k_latent)])
        engine.set_library(library)
        engine.set_active(np.arange(k_latent))
        torch = engine.torch
        Ud = engine.device_tensor(np.asarray(U), torch.float32)
        Vd = engine.device_tensor(np.asarray(V), torch.float32)
        engine.invalidate_factors()
        objs, gd = engine.step(Ud, Vd, n_steps, gamma, delta, tradeoff)
        for s, od in enumerate(objs):
            print(i + s + 1, od["obj"])               # :447
            if verbose:
                print(od)
        U_new = Ud.cpu().numpy().astype(np.float64)
        V_new = Vd.cpu().numpy().astype(np.float64)
    finally:
        if own:
            engine.close()
    return U_new, V_new, (objs[-1] if objs else None), gd


def nmf_manifold_vec_update(X, U, V, k_to_W, k_to_D, k_to_L, k_to_feat_inds, n_steps=10, gamma=1.0, delta=1.0,
                            i=0, verbose=False, norm_X=None, x_dtype="f32", engine=None):
    """Same call as the reference (prmf_runner.py:374): `n_steps` update steps with a
    fixed Laplacian per latent factor; returns (U, V, obj_data).  k_to_D / k_to_L
    are accepted for signature compatibility (D and L are derived from W)."""
    U2, V2, od, _ = _run_inner(X, U, V, k_to_W, k_to_feat_inds, n_steps, gamma, delta, i, verbose, None, x_dtype, engine)
    return U2, V2, od


def nmf_manifold_vec_update_tradeoff(X, U, V, k_to_W, k_to_D, k_to_L, k_to_feat_inds, n_steps=10, i=0, verbose=False,
                                     norm_X=None, tradeoff=0.5, gamma=1, delta=1, x_dtype="f32", engine=None):
    """Same call as the reference (prmf_runner.py:497): returns (U, V, obj_data, gamma, delta)."""
    U2, V2, od, gd = _run_inner(X, U, V, k_to_W, k_to_feat_inds, n_steps, gamma, delta, i, verbose, tradeoff, x_dtype, engine)
    return U2, V2, od, gd[0], gd[1]


# ---------------------------------------------------------------------------
# outer solver  (prmf_runner.py:556-792)
# ---------------------------------------------------------------------------
def restrict(scores_kP, latent_to_pathway_data, percentile=19.9):
    """prmf_runner.py:129-194 on a precomputed k x P score matrix (the GPU scores
    every pair; only the still-candidate ones are read here).  Dict in, dict out; the solver keeps the candidates as arrays
    between the rounds (`_Candidates.restrict`, the same code)."""
    c = _Candidates(0, 0)
    c.take_dict(latent_to_pathway_data)
    c.restrict(scores_kP, percentile)
    return c.as_dict()


def print_latent_to_pathway_data(latent_to_pathway_data, out=None):
    """The `--verbose` candidate table (prmf_runner.py:67-88): per latent factor its remaining candidate pathways,
    best score first, `<pathway id>\t<score>` indented by two spaces."""
    out = out or sys.stdout
    lines = ["-----------------------------", "Latent to pathway match data:", "-----------------------------"]
    for latent_id, data in latent_to_pathway_data.items():
        lines.append("Latent vector: {}".format(latent_id))
        lines.extend("  {}\t{}".format(pid, score) for pid, score in sorted(data, key=lambda x: x[1], reverse=True))
    out.write("\n".join(lines) + "\n")


def _max_weight_matching(per_latent):
    """Maximum-weight matching between latent factors and candidate pathways: [(k, [(pathway, weight >= 0), ...]), ...]
    -> [(k, pathway)], the matching networkx.max_weight_matching returns for the reference (prmf_runner.py:246).

    networkx's blossom algorithm is pure Python (64 latents x 61 candidates each: 0.80 s, 700 000 calls of its `slack`;
    cubic in the number of latents).  The graph is bipartite, so the same optimum is an assignment problem: solved by
    scipy's linear_sum_assignment (C), and then PROVEN to be the only optimum -- another optimal matching would have to
    leave out one of the matched edges, so the problem is re-solved once per matched edge with that edge removed and
    every one of those optima must be strictly lighter.  A unique optimum is what any correct algorithm returns, networkx
    included.  If the proof fails (exact ties, zero-weight edges) networkx itself is run: its tie-breaking stays the
    contract."""
    from scipy.optimize import linear_sum_assignment
    lat = [k for k, _ws in per_latent]
    paths = sorted({p for _k, ws in per_latent for p, _w in ws})
    col = {p: j for j, p in enumerate(paths)}
    W = np.zeros((len(lat), len(paths)))
    generic = True
    for i, (_k, ws) in enumerate(per_latent):
        for p, w in ws:
            W[i, col[p]] = w
            generic = generic and w > 0
    if generic and W.size:
        def solve(M):
            r, c = linear_sum_assignment(M, maximize=True)
            keep = M[r, c] > 0
            return r[keep], c[keep], math.fsum(M[r, c][keep])
        r, c, best = solve(W)
        unique = best > 0
        for i, j in zip(r, c):
            if not unique:
                break
            w = W[i, j]
            W[i, j] = 0.0
            unique = solve(W)[2] < best * (1.0 - 1e-9)
            W[i, j] = w
        if unique:
            return [(lat[i], paths[j]) for i, j in zip(r, c)]
    import networkx as nx
    G = nx.Graph()
    for k, ws in per_latent:
        for p, w in ws:
            G.add_edge("k{}".format(k), "l{}".format(p), weight=w)
    out = []
    for a, b in nx.max_weight_matching(G):
        if a[0] != "k":
            a, b = b, a
        out.append((int(a[1:]), int(b[1:])))
    return out


def force_distinct_lapls(V, library, latent_to_pathway_data, k_to_lapl, gamma, delta):
    """prmf_runner.py:209-258: max-weight matching between latent factors and their
    remaining candidate pathways, weight 1/(gamma v^T L v + delta*ignore).  As in the
    reference the ignore term is overwritten (not summed) over the active supports
    (:233-235), so it is the last latent's value.  Host form (V: n x k ndarray); the solver itself takes the two terms
    from the GPU (`PrmfEngine.match_terms`) and calls `force_distinct_from_terms`."""
    k_last = max(k_to_lapl.keys())
    ign = float(np.sum(1.0 / (V[library.supports[k_to_lapl[k_last]], k_last] + 1)))
    # v_k^T L_p v_k for every remaining (latent, pathway) pair, one sparse product per pathway (PathwayLibrary.laplacian_quadratics)
    by_pathway = {}
    for k, data in latent_to_pathway_data.items():
        for lapl, _score in data:
            by_pathway.setdefault(lapl, []).append(k)
    quad = {}
    for lapl, ks in by_pathway.items():
        for k, q in zip(ks, library.laplacian_quadratics(lapl, V, ks)):
            quad[(k, lapl)] = q
    return _match_from(lambda k, lapl: quad[(k, lapl)], ign, latent_to_pathway_data, gamma, delta)


def force_distinct_from_terms(quad_kP, ignore_kP, latent_to_pathway_data, k_to_lapl, gamma, delta):
    """The same with the two terms of every (latent, pathway) pair already evaluated (k x P arrays, `prmf_match_terms`):
    no download of V, no sparse products on the host."""
    k_last = max(k_to_lapl.keys())
    ign = float(ignore_kP[k_last, k_to_lapl[k_last]])
    return _match_from(lambda k, lapl: float(quad_kP[k, lapl]), ign, latent_to_pathway_data, gamma, delta)


def _match_from(quad_of, ign, latent_to_pathway_data, gamma, delta):
    per_latent = []
    for k, data in latent_to_pathway_data.items():
        ws = []
        for lapl, _score in data:
            denom = gamma * quad_of(k, lapl) + delta * ign
            ws.append((lapl, 0 if denom == 0 else 1 / denom))
        per_latent.append((k, ws))
    for k, p in _max_weight_matching(per_latent):
        latent_to_pathway_data[k] = [(p, 2)]
    return latent_to_pathway_data


class _Candidates(object):
    """latent_to_pathway_data (prmf_runner.py:700: latent -> [(pathway, score), ...]) held as one index array and one score
    array per latent between the rounds -- restrict cuts them with array operations -- and turned into the reference's
    dict of tuple lists only where that form is the contract (the verbose table, the matching, the returned obj_data).
    Entries keep the reference's types: the initial score is the int 1, a matched pair's the int 2, restrict's are floats."""

    def __init__(self, k_latent, P):
        self.inds = {k: np.arange(P, dtype=np.int64) for k in range(k_latent)}
        self.scores = {k: np.ones(P, dtype=np.int64) for k in range(k_latent)}

    def min_len(self):
        return min(len(v) for v in self.inds.values())

    def any_open(self):
        return any(len(v) > 1 for v in self.inds.values())

    def as_dict(self):
        out = {}
        for k, inds in self.inds.items():
            sc = self.scores[k]
            conv = int if sc.dtype.kind == "i" else float
            out[k] = [(int(p), conv(x)) for p, x in zip(inds.tolist(), sc.tolist())]
        return out

    def take_dict(self, l2p):
        for k, data in l2p.items():
            self.inds[k] = np.array([x[0] for x in data], dtype=np.int64)
            sc = [x[1] for x in data]
            self.scores[k] = np.array(sc, dtype=np.int64 if all(isinstance(x, int) for x in sc) else np.float64)

    def restrict(self, scores_kP, percentile=19.9):
        """prmf_runner.py:129-194 on the k x P score matrix of the GPU.  Latents with equally many candidates are cut
        together: np.percentile along an axis gives every row the value the per-row call gives, bit for bit (tested)."""
        by_len = {}
        for k, inds in self.inds.items():
            if len(inds) > 1:
                by_len.setdefault(len(inds), []).append(k)
        for _n, ks in by_len.items():
            IND = np.stack([self.inds[k] for k in ks])
            S = scores_kP[np.asarray(ks)[:, None], IND]
            keep = S > np.percentile(S, percentile, axis=1)[:, None]
            for r, k in enumerate(ks):
                if not keep[r].any():
                    # :174-182 all scores tie: the reference asks np.random.choice for a negative sample size here and
                    # raises ValueError; so do we.
                    size = math.ceil(len(IND[r]) * (1 - percentile) / 100)
                    chosen = np.random.choice(IND[r], size=size, replace=False)
                    self.inds[k] = np.asarray(chosen, dtype=np.int64)
                    self.scores[k] = scores_kP[k, self.inds[k]].astype(np.float64)
                else:
                    self.inds[k] = IND[r][keep[r]]
                    self.scores[k] = S[r][keep[r]].astype(np.float64)


def multinomial_one(p):
    """`scipy.stats.multinomial.rvs(1, p)` (the draw at prmf_runner.py:723) on the legacy global stream WITHOUT importing
    scipy.stats, which costs 0.26 s of a command-line run that takes 0.9 s.  What scipy (1.15: `_process_parameters`, then
    `random_state.multinomial`) does with it: the last probability becomes 1 - sum(p[:-1]) unless that differs from 1 by no
    more than 1e-15, and numpy's legacy global RandomState draws -- the same two steps here, so the same stream positions and
    the same one-hot vector (tests/test_host_logic_cpu.py::test_multinomial_draw_equals_scipy compares them draw by draw)."""
    p = np.array(p, dtype=np.float64, copy=True)
    adjusted = 1.0 - p[:-1].sum(axis=-1)
    if np.abs(adjusted) > 1e-15:
        p[-1] = adjusted
    return np.random.mtrand._rand.multinomial(1, p)


def nmf_pathway(X, Gs, gamma=1.0, delta=1.0, tradeoff=None, k_latent=6, tol=1e-3, max_iter=1000, nodelist=None,
                modulus=10, U_init=None, V_init=None, verbose=False, x_dtype="f32", library=None, round_log=None,
                trace=None, group=None, dist_mode=None):
    """Same call and return value as the reference's nmf_pathway (prmf_runner.py:556).

    Extra keyword-only arguments: x_dtype ("f32" | "bf16" storage of X on the
    GPU), library (a prebuilt PathwayLibrary instead of Gs), round_log / trace
    (lists that receive the per-round active assignment / per-iteration records).
    Randomness comes from the legacy global numpy stream in the reference's order:
    rand(m,k), rand(n,k) (:651-653), one multinomial draw per latent per round (:723).

    Multi-GPU: called by every rank of an initialised torch.distributed process group (one process per GPU; `group`,
    default the world) -- or of a prmf_amd.ranks.LocalRanks group (one thread per GPU; `group` = the rank's view) --
    with the same arguments, the solve is sharded -- rank r keeps rows shard_rows(m, N, r) of X and
    U on its GPU, V is replicated, each iteration exchanges the k x k Gram U^T U and the n x k product X^T U over RCCL
    (prmf_amd.distributed; dist_mode "colshard" (default) or "replicated").  Rank 0's RNG stream is the solve's: its
    draws are broadcast.  Every rank returns the same (U, V, obj_data) -- U gathered to all m rows; only rank 0 prints."""
    from .ranks import as_ranks
    ranks = as_ranks(group)
    say = print if ranks.rank == 0 else (lambda *a, **kw: None)
    # X: the reference's ndarray; a RowSource (rank-local input: .shape, .rows(lo, hi)); or a 2-D tensor already on the GPU
    # (prmf_amd.preprocess.quantile_transform(on_device=True): the normalised matrix never visits the host)
    if isinstance(X, list) and len(X) == 1:
        X = X.pop()      # ownership handed over: the matrix is dropped as soon as the library holds its copies (the CLI's
        #                  normalised float64 matrix on the GPU is 8 m n bytes next to the engine's own 2 x 4 m n)
    on_gpu = hasattr(X, "is_cuda") and X.is_cuda
    X = X if (hasattr(X, "rows") or on_gpu) else np.asarray(X)
    m, n = X.shape
    lo, hi = 0, m
    if ranks.multi:
        from .synthetic import shard_rows
        lo, hi = shard_rows(m, ranks.world, ranks.rank)
    # The cyclic collector is held off for the duration of the solve and put back as it was: a full collection walks every
    # container of the caller's process (the pathway library alone is thousands of arrays) and lands wherever an allocation
    # happens to trigger it -- 10-25 ms of a 150 ms solve at 50000 x 20000 (profiles/r06_outer_loop_host_time.txt).  Nothing
    # here builds reference cycles; reference counting frees everything else as always.
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    engine = None
    try:
        engine = PrmfEngine(hi - lo, n, k_latent, x_dtype=x_dtype, nranks=ranks.world, rank=ranks.rank,
                            dist_mode=dist_mode if ranks.multi else None, device=ranks.device_index)
        torch = engine.torch
        dev = "cuda:%d" % engine.device
        # X goes up on a helper thread (one pass in the library: narrowing, padded copy, padded transpose, ||X||^2 -- a host
        # matrix chunk by chunk under its own PCIe copy); what does not need X yet runs under it: the reference's random
        # draws (:651-653: the ORDER of the draws is the contract, not when they happen) and the pathway library (:670-696)
        pending = engine.set_x_async(X.rows(lo, hi) if hasattr(X, "rows") else X[lo:hi])
        if "scipy.optimize" not in sys.modules:
            # the matching at the end of the candidate rounds needs scipy.optimize (171 modules, 35-40 ms in a fresh process: every
            # command-line run): imported on a thread of its own it runs while this one waits, interpreter lock released, for the
            # upload and for the first batches of iterations
            import threading
            threading.Thread(target=__import__, args=("scipy.optimize",), name="prmf-import").start()      # (not a daemon: never cut off mid-import at exit)
        early = None
        try:
            # rank 0's factors are THE factors, drawn here (:651-653) or handed in: every rank takes rank 0's (a caller's
            # per-rank initialisers need not agree, e.g. set iteration order under different hash seeds); as tensors
            given = ranks.from_rank0((U_init is not None, V_init is not None)) if ranks.multi else (U_init is not None, V_init is not None)
            if ranks.rank == 0 or not ranks.multi:
                # 3 * (1 - rand(..))  (:651-653), written with explicit outputs: as one expression numpy asks, for each of its two
                # temporaries, whether it may reuse it in place -- a check that walks the C stack (glibc backtrace), and the first
                # walk of a process with torch and the ROCm libraries loaded costs 80 ms (profiles/r06_first_call.txt; every
                # command-line run is a first call).  The same two operations per element, the same values.
                if not given[0]:
                    U_init = np.random.rand(m, k_latent)
                    np.subtract(1, U_init, out=U_init)
                    np.multiply(3, U_init, out=U_init)
                if not given[1]:
                    V_init = np.random.rand(n, k_latent)
                    np.subtract(1, V_init, out=V_init)
                    np.multiply(3, V_init, out=V_init)
            if library is None:
                try:
                    library = PathwayLibrary.from_graphs(Gs, nodelist)    # :670-696
                except Exception as e:      # the reference meets its graphs after the shape checks below: so does this error
                    early = e
        finally:
            local_sq = pending.result()
            del pending
            X = None
        norm_sq = ranks.sum_float(local_sq, dev)
        engine.set_norm_x_sq(norm_sq)
        norm_X = math.sqrt(norm_sq)                                 # :640 (of the X the GPUs hold)
        say("norm(X) = {}".format(norm_X))                          # :641
        gamma = gamma * norm_X / k_latent                           # :644
        delta = delta * 10 / norm_X                                 # :645
        if ranks.multi:
            shp = ranks.from_rank0((np.shape(U_init), np.shape(V_init)) if ranks.rank == 0 else None)
            U_init = ranks.array_from_rank0(U_init, shp[0], dev)
            V_init = ranks.array_from_rank0(V_init, shp[1], dev)
        if U_init.shape != (m, k_latent):
            raise ValueError("Invalid U_init with shape {} != (m_obs, k_latent) = {}".format(U_init.shape, (m, k_latent)))
        if V_init.shape != (n, k_latent):
            raise ValueError("Invalid V_init with shape {} != (n_feature, k_latent) = {}".format(V_init.shape, (n, k_latent)))
        if early is not None:
            raise early
        engine.set_library(library)
        U = engine.device_tensor(U_init[lo:hi], torch.float32)
        V = engine.device_tensor(V_init, torch.float32)
        if ranks.multi:
            import os
            from .distributed import DistributedPrmf
            # PRMF_RCCL_LIB (the RCCL test double): the in-library step although the process group is not "nccl"
            step = DistributedPrmf(engine, group=group, native=True if os.environ.get("PRMF_RCCL_LIB") else None).step
        else:
            step = engine.step

        cands = _Candidates(k_latent, library.P)                    # :700
        converged, candidates_remain = False, True
        obj, obj_data = math.inf, {}
        best = {"obj_data": {"obj": np.inf}}
        i = 0
        k_to_lapl = {}
        while i < max_iter and (candidates_remain or not converged):
            if ranks.rank == 0:
                for k in range(k_latent):                          # :717-729
                    scores = cands.scores[k]
                    sample = multinomial_one(scores / np.sum(scores))
                    k_to_lapl[k] = int(cands.inds[k][np.where(sample != 0)[0][0]])
            k_to_lapl = ranks.from_rank0(k_to_lapl)
            if round_log is not None:
                round_log.append([int(k_to_lapl[k]) for k in range(k_latent)])
            if verbose:
                say('--------------------------------------------')
                say('Latent/Pathway association at this iteration')
                say('--------------------------------------------')
                for k, lapl_ind in k_to_lapl.items():
                    say(k, lapl_ind)
                say('--------------------------------------------')
            engine.set_active([k_to_lapl[k] for k in range(k_latent)])
            objs, (g2, d2) = step(U, V, modulus, gamma, delta, tradeoff)                 # :740 / :742
            if tradeoff is not None:
                gamma, delta = g2, d2
            for s, od in enumerate(objs):
                say(i + s + 1, od["obj"])
                if verbose:
                    say(od)
                if trace is not None:
                    trace.append(od)
            obj_data = objs[-1]
            i += modulus
            if obj_data["obj"] < best["obj_data"]["obj"]:           # :747-750
                # best iterate (:747-750): two device-to-device copies into buffers allocated once (31 MB at
                # 50000 x 20000, k = 64: ~10 us per improving round; the reference copies on the host)
                if "U" not in best:
                    best["U"], best["V"] = torch.empty_like(U), torch.empty_like(V)
                best["U"].copy_(U)
                best["V"].copy_(V)
                best["obj_data"] = obj_data
            if candidates_remain:                                   # :754-770  (V is replicated: every rank decides alike)
                if cands.min_len() <= k_latent:
                    # the two terms of every (latent, pathway) pair from the GPU (k x P each): V stays where it is
                    quad, ign = engine.match_terms(V)
                    cands.take_dict(force_distinct_from_terms(quad, ign, cands.as_dict(), dict(k_to_lapl), gamma, delta))
                    candidates_remain = False
                else:
                    if ranks.rank == 0:
                        sys.stderr.write('Before restrict: ' + str(datetime.datetime.now()) + '\n')     # :762
                    cands.restrict(engine.score_candidates(V))
                    if ranks.rank == 0:
                        sys.stderr.write('After restrict: ' + str(datetime.datetime.now()) + '\n')      # :764
                    candidates_remain = cands.any_open()
                if verbose and ranks.rank == 0:
                    print_latent_to_pathway_data(cands.as_dict())                                  # :759, :770
            prev_obj, obj = obj, obj_data["obj"]
            converged = abs(obj - prev_obj) / obj < tol              # :772-774
        if best["obj_data"]["obj"] < obj_data["obj"]:                # :778-782
            say('Local optima at convergence (or after max iterations) is not the best among all iterates; returning best instead')
            U, V, obj_data = best["U"], best["V"], best["obj_data"]
        U_out = ranks.gather_rows(torch, U, m).astype(np.float64)
        V_out = V.cpu().numpy().astype(np.float64)
        obj_data = dict(obj_data)
        obj_data["latent_to_pathway_data"] = cands.as_dict()
        return U_out, V_out, obj_data
    finally:
        if engine is not None:
            engine.close()
        if gc_was_on:
            gc.enable()


def measure_cv_performance(gene_by_latent_train, data_test, x_dtype="f32", device=0):
    """Drop-in for prmf.measure_cv_performance (prmf/__init__.py:768-798): normalised NNLS
    reconstruction error of every held-out sample, computed on the GPU (float64 X_test V and
    V^T V, batched active-set solve; `prmf_cv_error`)."""
    import torch
    V = np.asarray(gene_by_latent_train)
    Xt = np.asarray(data_test)
    m, n = Xt.shape
    if V.shape[0] != n:
        raise ValueError("gene_by_latent_train has {} rows, data_test has {} columns".format(V.shape[0], n))
    if m == 0:
        return np.zeros(0)
    eng = PrmfEngine(m, n, V.shape[1], x_dtype=x_dtype, device=device)
    try:
        eng.set_x(Xt)
        return eng.cv_error(eng.device_tensor(V, torch.float32))
    finally:
        eng.close()

