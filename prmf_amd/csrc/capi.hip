// C-ABI of libprmf_hip.so (include/prmf_hip.h): context, HBM workspace and the
// sequencing of one PRMF iteration into four device phases.
#include "../../include/prmf_hip.h"
#include "../../include/prmf_hip_debug.h"
#include "kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <vector>

using namespace prmf;

namespace {
thread_local char g_create_error[512] = "";
constexpr int MAX_RECORDS = 256;   // objective records kept on the device between begin/end
static_assert(PRMF_OBJ_STRIDE == REC_STRIDE && PRMF_OBJ_FLAGS == REC_FLAGS && PRMF_OBJ_RECON_SQ_IDENTITY == REC_RECON_SQ_IDENTITY &&
              PRMF_RECON_DIRECT == REC_FLAG_DIRECT && PRMF_RECON_IMPRECISE == REC_FLAG_IMPRECISE && PRMF_RECON_CLAMPED == REC_FLAG_CLAMPED,
              "record layout of the header and of the kernels");
static_assert(PRMF_TEXT_RAGGED == TEXT_FLAG_RAGGED && PRMF_TEXT_QUOTED == TEXT_FLAG_QUOTED && PRMF_TEXT_NOT_NUMERIC == TEXT_FLAG_NOT_NUMERIC &&
              PRMF_TEXT_WIDE_INT == TEXT_FLAG_WIDE_INT && PRMF_TEXT_BARE_CR == TEXT_FLAG_BARE_CR, "text flags of the header and of the kernels");
}

struct prmf_ctx {
  int device = 0;
  int64_t m = 0, n = 0;
  int k = 0, KP = 0, NT = 0;
  int x_dtype = 0;
  int pack_dtype = 0;                 // fragment layout of the packed factors (= x_dtype unless split_f32)
  bool split_f32 = false;             // f32 X multiplied as three bf16 pieces on the bf16 MFMA (see prmf_create)
  bool w16 = false;                   // packed factors in the fragment order of the 16x16x32 MFMA (bf16 X, k > 64: gemm.hip)
  int nranks = 1, rank = 0;
  int64_t m_global = -1;              // samples over all ranks (prmf_comm_init all-reduces it, or prmf_set_global_rows); -1: unknown
  // Layout ranks: how many column blocks the V update is split into.  Column-sharded mode (default for nranks > 1):
  // lr = nranks (each rank updates kc = ceil(k / nranks) columns; reduce-scatter of Q, all-gather of V).  Replicated mode
  // (PRMF_FLAG_REPLICATED): lr = 1 -- every rank updates all of V from the all-reduced Q with the single-rank fused
  // kernels; exchanges: all-reduce of U^T U (k x k float64) and all-reduce of Q (n x k float32).
  int lr = 1;
  bool replicated = false;
  int kc = 0, col0 = 0, ncols = 0;
  unsigned flags = 0;
  char err[512] = "";

  // X (m x n) and X^T (n x m), zero padded, resident
  void* X = nullptr;
  void* XT = nullptr;
  size_t x_bytes = 0, xt_bytes = 0;
  int64_t ldx = 0, ldxt = 0;          // elements
  int64_t Kpad_x = 0, Kpad_xt = 0;    // reduction lengths (n resp. m, padded to 64)
  double norm_x_sq = 0.0;
  bool have_x = false;
  // scratch of the one-pass layout kernels (setx.hip): tile sums, tile-row sums, the total
  double* setx_partial = nullptr;
  double* setx_rowsum = nullptr;
  double* setx_total = nullptr;

  // packed factors + sweep slabs
  void* Vp = nullptr;
  void* Up = nullptr;
  GemmPlan plan_xv{}, plan_xtu{};
  segmask_t* nseg_xv = nullptr;        // device copies of the plans' per-panel slab masks
  segmask_t* nseg_xtu = nullptr;
  float* slabs = nullptr;             // shared by both sweeps
  int64_t slab_stride_p = 0, slab_stride_q = 0;

  // Gram
  double* gram_partial = nullptr;
  double* Gv = nullptr;               // k*k
  double* gram_buf = nullptr;         // all-reduce buffer (U^T U), caller's or own
  bool own_comm = false;

  // comm buffers
  float *q_send = nullptr, *q_recv = nullptr, *v_send = nullptr, *v_recv = nullptr;
  // Layout of a rank's all-gather block (v_send): [n * kc floats: own columns of V_new][pad to 8 B]
  //   column-sharded mode only: [vblocks][kc + 1] doubles -- per-workgroup sums of the own-column V update (partial_off)
  //   [4 doubles: <Q, V_new>, manifold, ignore of the own columns -- per-iteration objective only] (extras_off)
  int64_t partial_off = 0, extras_off = 0, v_send_floats = 0;

  // pathway library
  PathwayLibDev lib{};
  std::vector<void*> lib_allocs;
  bool have_lib = false, have_active = false;
  int32_t* active_dev = nullptr;
  int32_t* pos = nullptr;             // [n][kc]: position of gene j in the active support of own column c, or -1
  int32_t* pos_all = nullptr;         // column-sharded mode: the same for all k columns [n][k] (deferred objective)
  float *gnum = nullptr, *gden = nullptr;
  double* obj_scratch = nullptr;      // [kc][max_support]
  double* obj_scratch_all = nullptr;  // [k][max_support] (deferred objective, supports too large for LDS); = obj_scratch if kc == k
  double* colres = nullptr;           // [kc][2]
  double* vpartial = nullptr;         // [vblocks][kc+1]  (column-sharded mode: lives in the tail of v_send, see vpartial_of)
  int vblocks = 0;
  int wpt_u = 1, wpt_v = 1;           // waves per tile in the fused epilogues (cooperative slab sum)
  bool fused_v = false, gram_v_done = false;
  // fused_v: the V epilogue straight from the sweep's slabs (one column block: single rank or replicated mode)
  bool graph_done = false;            // the pathway terms of this iteration were already computed in phase_u (merged launch)

  // Deferred objective + sweep prologues (tradeoff off; `lazy` is decided per batch in prmf_begin).  One rank: the
  // iteration is 4 launches: sweep X V [prologue: V^T V reduce + column norms / <Q,V> of the previous iteration] ->
  // U epilogue (+ pathway terms) -> sweep X^T U [prologue: U^T U reduce] -> V epilogue; what the objective record
  // needs is kept per iteration in the rings below and evaluated for the whole batch in prmf_end.  Several ranks: the
  // same, with the V side split by the mode (replicated: dense-Q sum + fused V epilogue; column-sharded: dense-Q sum,
  // own-column update, assemble / pack / Gram of the gathered V) -- see prmf_step_dist.
  bool lazy = false;
  bool gv_pending = false;            // the V side's V^T V partials / per-workgroup sums are not reduced yet
  bool batch_open = false;
  bool tradeoff_on = false;           // this batch runs the gamma / delta feedback (prmf_begin)
  bool dist_driver = false;           // inside prmf_step_dist: the library issues the collectives itself
  int nparts_u = 0, nparts_v = 0;     // Gram partials the U / V side writes per iteration
  double* Gu_keep = nullptr;          // [MAX_RECORDS][k*k]
  double* Gv_keep = nullptr;          // [MAX_RECORDS][k*k]
  double* colnorm2_ring = nullptr;    // [MAX_RECORDS][k]
  double* cross_ring = nullptr;       // [MAX_RECORDS]
  double* colres_ring = nullptr;      // [MAX_RECORDS][k][2]
  float* vsup_ring = nullptr;         // [MAX_RECORDS][k][max_support]   (allocated with the pathway library)

  // hyper-parameters + records
  double* hyper = nullptr;            // device: gamma, delta, tradeoff
  double* hyper_host = nullptr;       // pinned
  double* records = nullptr;          // device [MAX_RECORDS][8]
  double* records_host = nullptr;     // pinned
  int* need_host = nullptr;           // pinned: the direct-residual request of the batch's last record (prmf_end)
  int cursor = 0;

  // bound factors
  float* U = nullptr;
  float* V = nullptr;
  bool prepared = false;

  // timing
  std::vector<hipEvent_t> ev;          // [MAX_RECORDS][4] with PRMF_FLAG_TIMING
  double ms_xv = 0, ms_xtu = 0, ms_total = 0;
  int timed_iters = 0;
  std::vector<hipEvent_t> ev_phase;    // [MAX_RECORDS][5]: iteration start, after phase_u, phase_q (+ exchange), phase_v (+ exchange), phase_post
  double ms_phase[4] = {0, 0, 0, 0};
  int phase_timed_iters = 0;

  // RCCL (resolved at run time, see prmf_comm_init)
  ncclComm_t comm = nullptr;
  bool comm_broken = false;           // a collective failed or timed out: the communicator was aborted (comm_fail)
  double comm_timeout_s = 300.0;      // multi-rank: how long a stream synchronisation may wait for a collective (prmf_set_comm_timeout)

  // direct residual (residual.hip): device request flag, per-wave partials, the (all-reduced) sum
  int n_cus = 256;
  int* need_dev = nullptr;
  double* resid_partial = nullptr;
  double* resid_sq = nullptr;

  // scoring scratch
  double* colnorm = nullptr;
  double* scores_dev = nullptr;
  int64_t scores_cap = 0;
};

namespace {

// RCCL entry points, bound with dlopen so that the library itself has no link-time
// dependency (PyTorch-ROCm ships librccl.so.1; the copy already mapped by torch is reused).
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;        // optional (reporting only)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;     // optional
  ncclResult_t (*GroupStart)() = nullptr;                             // optional: without them the two exchanges after
  ncclResult_t (*GroupEnd)() = nullptr;                               //   phase_q are two launches instead of one
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                    // optional: after a failed collective (comm_fail)
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    // PRMF_RCCL_LIB: another library with the same seven entry points (tests/native/mock_rccl.cpp runs the
    // multi-rank step with two ranks on one GPU, which RCCL itself refuses)
    if (const char* e = getenv("PRMF_RCCL_LIB")) a.handle = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    if (!a.handle) a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle) a.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle) return a;
#define PRMF_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name))
    PRMF_SYM(GetUniqueId, "ncclGetUniqueId");
    PRMF_SYM(CommInitRank, "ncclCommInitRank");
    PRMF_SYM(CommDestroy, "ncclCommDestroy");
    PRMF_SYM(AllReduce, "ncclAllReduce");
    PRMF_SYM(ReduceScatter, "ncclReduceScatter");
    PRMF_SYM(AllGather, "ncclAllGather");
    PRMF_SYM(GetErrorString, "ncclGetErrorString");
    PRMF_SYM(CommCount, "ncclCommCount");
    PRMF_SYM(CommUserRank, "ncclCommUserRank");
    PRMF_SYM(GroupStart, "ncclGroupStart");
    PRMF_SYM(GroupEnd, "ncclGroupEnd");
    PRMF_SYM(CommAbort, "ncclCommAbort");
#undef PRMF_SYM
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.ReduceScatter && a.AllGather && a.GetErrorString;
    return a;
  }();
  return api;
}

int fail(prmf_ctx* c, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(c ? c->err : g_create_error, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return fail(ctx, PRMF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

template <typename T>
int dev_alloc(prmf_ctx* c, T** p, size_t count, bool zero = true) {
  size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) return fail(c, PRMF_ERR_HIP, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
  if (zero) {
    // hipMemset runs on the NULL stream and need not have finished when it returns; the caller's stream may be a
    // non-blocking one (torch's side streams; one per rank thread), which the null stream is not ordered with: without
    // the wait a kernel of the caller's stream could write the buffer first and be zeroed afterwards
    e = hipMemset(*p, 0, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(c, PRMF_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
  }
  return PRMF_OK;
}

// Setup copies (hipMemcpy from pageable host memory, null stream) may return before the DMA has reached the device;
// nothing the caller's stream launches afterwards may overtake them.
int null_stream_done(prmf_ctx* c) {
  HIP_TRY(c, hipStreamSynchronize(nullptr));
  return PRMF_OK;
}

void dev_free(void* p) {
  if (p) (void)hipFree(p);
}

int64_t packed_bytes(const prmf_ctx* c, int64_t Kpad) {
  return Kpad / chunk_elems(c->x_dtype) * packed_chunk_bytes(c->x_dtype, c->NT);
}

int set_device(prmf_ctx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  return PRMF_OK;
}

// pack V and build V^T V from the bound V (first step, or after the caller changed V)
int prepare(prmf_ctx* c, hipStream_t s) {
  FactorView fv{c->V, (int)c->n, c->k, c->k, 0};
  HIP_TRY(c, launch_pack_factor(fv, c->pack_dtype, c->NT, c->Kpad_x, c->Vp, s, c->w16));
  HIP_TRY(c, launch_gram(fv, c->gram_partial, c->Gv, s));
  c->prepared = true;
  c->gv_pending = false;
  return PRMF_OK;
}

int check_ready(prmf_ctx* c) {
  if (!c) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  if (!c->have_lib) return fail(c, PRMF_ERR_STATE, "prmf_set_pathway_library has not been called");
  if (!c->have_active) return fail(c, PRMF_ERR_STATE, "prmf_set_active has not been called");
  if ((c->m > 0 && !c->U) || !c->V) return fail(c, PRMF_ERR_STATE, "factors are not bound (prmf_begin / prmf_step)");
  if (!c->gram_buf || !c->q_send || !c->q_recv || !c->v_send || !c->v_recv)
    return fail(c, PRMF_ERR_STATE, "communication buffers are not set (prmf_set_comm_buffers)");
  return PRMF_OK;
}

int ensure_own_comm(prmf_ctx* c) {
  if (c->gram_buf) return PRMF_OK;
  if (c->nranks != 1 && !c->comm)
    return fail(c, PRMF_ERR_STATE, "nranks > 1 needs prmf_set_comm_buffers (host-driven collectives) or prmf_comm_init");
  int rc;
  if ((rc = dev_alloc(c, &c->gram_buf, (size_t)c->k * c->k))) return rc;
  if ((rc = dev_alloc(c, &c->q_send, (size_t)c->n * c->kc * c->lr))) return rc;
  if ((rc = dev_alloc(c, &c->v_send, (size_t)c->v_send_floats))) return rc;
  if (c->lr == 1) {
    c->q_recv = c->q_send;
    c->v_recv = c->v_send;
  } else {
    if ((rc = dev_alloc(c, &c->q_recv, (size_t)c->n * c->kc))) return rc;
    if ((rc = dev_alloc(c, &c->v_recv, (size_t)c->v_send_floats * c->lr))) return rc;
  }
  c->own_comm = true;
  return PRMF_OK;
}

// the pathway terms of this rank's columns as extra workgroups of the U epilogue launch
static GraphJob graph_job(const prmf_ctx* c) {
  GraphJob gj;
  gj.lib = c->lib;
  gj.active = c->active_dev;
  gj.col0 = c->col0;
  gj.ncols = c->ncols;

  gj.nsplit = graph_job_split(c->lib.max_support);
  gj.V = c->V;
  gj.hyper = c->hyper;
  gj.gnum = c->gnum;
  gj.gden = c->gden;
  return gj;
}

// prologue that closes iteration `it` of the batch on the V side (it < 0: results go to the spare ring slot)
SweepPrologue post_prologue(prmf_ctx* c, int it) {
  const int slot = it >= 0 ? it : MAX_RECORDS;
  SweepPrologue p;
  p.kind = PROLOGUE_POST;
  p.gram_partial = (const float*)c->gram_partial;
  p.nparts = c->nparts_v;
  p.k = c->k;
  p.G = c->Gv;
  p.G_keep = c->Gv_keep + (size_t)slot * c->k * c->k;
  p.colnorm2 = c->colnorm2_ring + (size_t)slot * c->k;
  p.cross = c->cross_ring + slot;
  if (c->fused_v) {
    p.vpartial = c->vpartial;
    p.nwaves = c->vblocks;
  } else {                             // column-sharded: every rank's sums arrived in the tail of its all-gather block
    p.gath = c->v_recv;
    p.gath_stride = c->v_send_floats;
    p.gath_off = c->partial_off;
    p.gath_nblk = c->vblocks;
    p.gath_kc = c->kc;
    p.gath_nranks = c->lr;
  }
  if (c->nranks > 1) {                 // multi-rank: the kept U^T U is the all-reduced one
    p.gu_src = c->gram_buf;
    p.gu_keep = c->Gu_keep + (size_t)slot * c->k * c->k;
  }
  return p;
}

SweepPrologue mid_prologue(prmf_ctx* c, int it) {
  SweepPrologue p;
  p.kind = PROLOGUE_MID;
  p.gram_partial = (const float*)c->gram_partial;
  p.nparts = c->nparts_u;
  p.k = c->k;
  p.G = c->gram_buf;
  p.G_keep = c->Gu_keep + (size_t)it * c->k * c->k;
  p.lib = c->lib;
  p.active = c->active_dev;
  p.col0 = 0;
  p.ncols = 0;                         // the pathway terms were computed next to the U epilogue (GraphJob)
  p.V = c->V;
  p.hyper = c->hyper;
  p.gnum = c->gnum;
  p.gden = c->gden;
  return p;
}

// A collective failed on this rank (or the stream did not come back from one in time): this rank's peers may be inside
// the same collective waiting for it.  The communicator is aborted -- ncclCommAbort releases what it holds and, with the
// test double, wakes the peers -- and the context refuses further multi-rank steps; the caller gets PRMF_ERR_HIP and is
// expected to end the process non-zero so that the launcher ends the job (with RCCL a peer's kernel can only be ended that
// way; the peers' own stream timeouts, sync_stream below, turn their wait into the same error).
int comm_fail(prmf_ctx* c, const char* what, const char* why) {
  // Without ncclCommAbort the communicator is leaked on purpose: ncclCommDestroy waits for the outstanding collective
  // this call is trying to get away from.
  if (c->comm && rccl().CommAbort) (void)rccl().CommAbort(c->comm);
  c->comm = nullptr;
  c->comm_broken = true;
  c->batch_open = false;
  return fail(c, PRMF_ERR_HIP, "%s: %s (rank %d of %d; the communicator was aborted)", what, why, c->rank, c->nranks);
}

#define NCCL_TRY(ctx, expr)                                                       \
  do {                                                                            \
    ncclResult_t _r = (expr);                                                     \
    if (_r != ncclSuccess) return comm_fail(ctx, #expr, rccl().GetErrorString(_r)); \
  } while (0)

// hipStreamSynchronize for a stream that may carry collectives: with several ranks the wait is bounded -- a peer that
// died leaves this rank's collective kernel spinning for ever, and a library call must not hang with it.  The bound is
// c->comm_timeout_s PLUS an allowance per iteration the stream still carries (`pending_iters`: the batch prmf_end waits
// for), so that a long healthy batch -- many steps, or eight thread ranks sharing one GPU through the test double -- is
// not taken for a dead peer.  The poll spins for the first 2 ms (yielding now and then), after that it sleeps between
// queries for 1 % of the time already waited (5 .. 200 us): N rank threads do not burn N cores on a long wait, and the
// call returns at most 1 % after the stream has finished -- a timed multi-rank batch is not stretched by the polling.
// One rank: the plain call.
constexpr double SYNC_ALLOWANCE_PER_ITERATION_S = 2.0;
int sync_stream(prmf_ctx* c, hipStream_t s, int pending_iters = 0) {
  if (c->nranks == 1 || !c->comm || !(c->comm_timeout_s > 0.0)) {
    HIP_TRY(c, hipStreamSynchronize(s));
    return PRMF_OK;
  }
  const double limit_s = c->comm_timeout_s + SYNC_ALLOWANCE_PER_ITERATION_S * (pending_iters > 0 ? pending_iters : 0);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned polls = 0;; ++polls) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return PRMF_OK;
    if (e != hipErrorNotReady) return fail(c, PRMF_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
    if ((polls & 15) != 15) continue;
    const double waited_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited_s > limit_s)
      return comm_fail(c, "stream synchronisation", "timed out waiting for a collective -- a peer rank may have failed");
    if (waited_s < 2e-3) {
      std::this_thread::yield();
    } else {
      const double nap_us = waited_s * 1e4 < 5.0 ? 5.0 : (waited_s * 1e4 > 200.0 ? 200.0 : waited_s * 1e4);      // 1 % of the wait so far
      std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(nap_us * 1e3)));
    }
  }
}

// Can this context replace a record's identity-based recon by the direct residual?  One rank: always.  Several ranks:
// the local sums have to be added up, which the library can only do itself (RCCL communicator, inside prmf_step_dist).
bool direct_available(const prmf_ctx* c) { return c->nranks == 1 || (c->comm != nullptr && c->dist_driver); }
// Every record from the direct residual: on request (PRMF_FLAG_DIRECT_RESIDUAL), and whenever the tradeoff feedback is on --
// gamma = delta = (1 - t) recon / (t manifold) (prmf_runner.py:542-548) feeds recon back into the update every step, and
// the identity's error (on the reference's 100 x 1000 test problem 2e-5: the 16-bit packed factors do not average out
// over 600 entries of U) was amplified to 1.7e-3 on the objective trace over 1000 iterations, where a float64 run whose
// only float32 part is the state drifts 5e-7 (tests/test_oracle_golden.py::test_tradeoff_trace_conditioning).
bool direct_forced(const prmf_ctx* c) { return (c->flags & PRMF_FLAG_DIRECT_RESIDUAL) != 0 || c->tradeoff_on; }
// recon^2 (identity) below this value is too small to trust: kernels.h RECON_DIRECT_THRESHOLD*
// plus the part of the error that comes from the 16-bit packed factors: it averages over the min(m, n) k entries of the
// smaller factor only (2.6e-7 ||X||^2 measured at 100 x 1000, k = 6: profiles/r03_recon_identity_terms.txt)
// Several ranks: the GLOBAL sample count (every rank must raise the same flags and take the same collectives: shard sizes
// differ by one when nranks does not divide m, and a rank may hold no rows at all); unknown -> the gene count alone.
double direct_below(const prmf_ctx* c) {
  const int64_t m_all = c->nranks == 1 ? c->m : (c->m_global >= 0 ? c->m_global : c->n);
  const double entries = (double)(m_all < c->n ? (m_all > 0 ? m_all : 1) : c->n) * c->k;
  // k > 64 (one accumulator chain for the hi and lo products): the error of <X^T U, V> grows with the length of that chain --
  // the samples one rank reduces over.  Measured: 8.4e-8 ||X||^2 at 25000 samples, 4.5e-7 at 200000 (the whole of BASELINE
  // configs[4] on one GPU: recon 4.6e-5 off at a 7 % fit, profiles/r05_parity_measured.jsonl), so the threshold scales with it
  // (rank-invariant: the global count over the ranks, rounded up).
  double chain = 1.0;
  if (c->NT > 2) {
    const double rows = std::ceil((double)(m_all > 0 ? m_all : 1) / (double)c->nranks);
    if (rows > RECON_DIRECT_CHAIN_ROWS) chain = rows / RECON_DIRECT_CHAIN_ROWS;
  }
  return ((c->NT <= 2 ? RECON_DIRECT_THRESHOLD_LO : RECON_DIRECT_THRESHOLD * chain) + RECON_DIRECT_FACTOR_TERM / std::sqrt(entries)) * c->norm_x_sq;
}

// The launches behind a record whose kernel may have raised c->need_dev: sum (X - U V^T)^2 over this rank's rows (the
// kernels return at once when the flag is clear), the sum over ranks, and the rewrite of the record.
// `on_device_flag`: the kernels look at c->need_dev themselves (per-iteration form: no host round trip between the
// iterations); false: the host has already seen the flag raised (prmf_end) and the kernels run unconditionally.
int direct_residual(prmf_ctx* c, double* record, bool feedback, bool on_device_flag, hipStream_t s) {
  int nparts = 0;
  const int* need = on_device_flag ? c->need_dev : c->need_dev + 1;     // need_dev[1] is a constant 1
  if (c->m > 0) {
    HIP_TRY(c, launch_residual(c->x_dtype, c->X, c->ldx, c->m, c->n, c->k, c->U, c->V, c->n_cus, c->resid_partial, &nparts, need, s));
  }
  if (c->nranks == 1) {
    HIP_TRY(c, launch_residual_sum(c->resid_partial, nparts, need, c->resid_sq, record, c->hyper, feedback, s));
    return PRMF_OK;
  }
  HIP_TRY(c, launch_residual_sum(c->resid_partial, nparts, need, c->resid_sq, nullptr, nullptr, false, s));
  NCCL_TRY(c, rccl().AllReduce(c->resid_sq, c->resid_sq, 1, ncclFloat64, ncclSum, c->comm, s));
  HIP_TRY(c, launch_record_fix(need, c->resid_sq, record, c->hyper, feedback, s));
  return PRMF_OK;
}

// per-iteration form: the record of iteration c->cursor, its direct residual if asked for, the tradeoff feedback
int finalize_record(prmf_ctx* c, const double* loc_partial, int loc_nblocks, const double* loc_colres, hipStream_t s) {
  double* record = c->records + (size_t)c->cursor * PRMF_OBJ_STRIDE;
  const bool direct = direct_available(c);
  HIP_TRY(c, launch_finalize(c->gram_buf, c->Gv, c->k, c->v_recv, c->lr, c->v_send_floats, c->extras_off, c->norm_x_sq, c->hyper,
                             record, loc_partial, loc_nblocks, loc_colres, direct ? c->need_dev : nullptr, direct_forced(c), direct_below(c), s));
  if (direct) return direct_residual(c, record, true, true, s);
  return PRMF_OK;
}

}  // namespace

extern "C" {

int prmf_abi_version(void) { return PRMF_ABI_VERSION; }

const char* prmf_last_error(const prmf_ctx* ctx) { return ctx ? ctx->err : g_create_error; }

int prmf_create(prmf_ctx** out, int device, int64_t m_local, int64_t n, int k, int x_dtype, int nranks, int rank,
                unsigned flags) {
  if (!out) return fail(nullptr, PRMF_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (m_local < 0 || n <= 0 || k <= 0) return fail(nullptr, PRMF_ERR_ARG, "bad shape m=%lld n=%lld k=%d", (long long)m_local, (long long)n, k);
  if (k > MAX_KP) return fail(nullptr, PRMF_ERR_UNSUPPORTED, "k=%d > %d latent columns", k, MAX_KP);
  if (x_dtype != PRMF_X_F32 && x_dtype != PRMF_X_BF16) return fail(nullptr, PRMF_ERR_ARG, "bad x_dtype %d", x_dtype);
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, PRMF_ERR_ARG, "bad rank %d of %d", rank, nranks);
  if (m_local * (int64_t)k > 0x7fffffff00LL || n * (int64_t)k > 0x7fffffff00LL) return fail(nullptr, PRMF_ERR_UNSUPPORTED, "factor too large");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, PRMF_ERR_HIP, "no HIP device %d (count %d): %s", device, ndev, hipGetErrorString(e));
  prmf_ctx* c = new (std::nothrow) prmf_ctx();
  if (!c) return fail(nullptr, PRMF_ERR_HIP, "out of host memory");
  c->device = device;
  c->m = m_local;
  c->n = n;
  c->k = k;
  c->KP = (int)round_up(k, 32);
  c->NT = c->KP / 32;
  c->x_dtype = x_dtype;
  c->nranks = nranks;
  c->rank = rank;
  c->flags = flags;
  c->replicated = nranks > 1 && (flags & PRMF_FLAG_REPLICATED);
  c->lr = c->replicated ? 1 : nranks;
  c->kc = (k + c->lr - 1) / c->lr;
  c->col0 = (c->replicated ? 0 : rank) * c->kc;
  c->ncols = c->col0 >= k ? 0 : (c->col0 + c->kc <= k ? c->kc : k - c->col0);
  c->fused_v = c->lr == 1;
  c->partial_off = round_up(n * c->kc, 4);     // blocks and their parts stay 16-B aligned (the staged loads of v_post)
  c->extras_off = c->partial_off + (c->fused_v ? 0 : 2 * (int64_t)v_update_blocks(n) * (c->kc + 1));
  c->v_send_floats = round_up(c->extras_off + 8, 4);
  int rc = PRMF_OK;
  auto bail = [&](int code) {
    std::snprintf(g_create_error, sizeof g_create_error, "%s", c->err);
    prmf_destroy(c);
    return code;
  };
  if ((rc = set_device(c))) return bail(rc);

  c->Kpad_x = round_up(n, K_PAD);
  c->Kpad_xt = round_up(m_local > 0 ? m_local : 1, K_PAD);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(c, PRMF_ERR_HIP, "hipGetDeviceProperties failed"); return bail(PRMF_ERR_HIP); }
  const int n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->n_cus = n_cus;
  SweepTuning tune;                                     // the library's only tuning hook, for the decomposition test
  if (const char* e = getenv("PRMF_SWEEP_XCD")) tune.force_slices = atoi(e);
  if (const char* e = getenv("PRMF_SWEEP_NO_EXACT")) tune.no_exact = atoi(e) != 0;
  c->plan_xv = plan_gemm(x_dtype, m_local > 0 ? m_local : 1, c->Kpad_x, c->NT, n_cus, tune);
  c->plan_xtu = plan_gemm(x_dtype, n, c->Kpad_xt, c->NT, n_cus, tune);
  // f32 X: the sweep splits every f32 value of X exactly into three bf16 pieces in registers and runs the
  // bf16 MFMA against the hi/lo-packed factor (5 products, cross terms in their own accumulators) instead of the f32
  // MFMA at 1/16 of the rate: 0.75 vs 1.05 ms per sweep at 50000 x 20000, and a smaller error (P = X V: rms 2.9e-7
  // against 7.4e-7, no bias).  k > 64: the same with 32-row wave tiles (two accumulator sets of a 64-row tile do not fit
  // the registers): 0.80 vs 1.17 ms per sweep at 25000 x 25000, k = 128.
  c->split_f32 = x_dtype == PRMF_X_F32;
  c->pack_dtype = c->split_f32 ? PRMF_X_BF16 : x_dtype;
  c->w16 = c->NT > 2 || (x_dtype == PRMF_X_F32 && c->NT == 2);    // gemm.hip SweepW16
  // the 8-wave sweeps need more than 64 KiB of dynamic LDS: raised once per context (= per device), no state in the launcher
  if (launch_gemm(x_dtype, c->plan_xv, nullptr, 0, 0, nullptr, c->NT, nullptr, 0, nullptr, c->split_f32, nullptr, true) != hipSuccess ||
      launch_gemm(x_dtype, c->plan_xtu, nullptr, 0, 0, nullptr, c->NT, nullptr, 0, nullptr, c->split_f32, nullptr, true) != hipSuccess) {
    fail(c, PRMF_ERR_HIP, "hipFuncSetAttribute(sweep kernel, dynamic LDS) failed");
    return bail(PRMF_ERR_HIP);
  }
  // slabs are written as whole 32-row tiles of whole panels (fragment order, no row masking)
  c->slab_stride_p = (int64_t)c->plan_xv.panels * c->plan_xv.panel_rows * c->KP;
  c->slab_stride_q = (int64_t)c->plan_xtu.panels * c->plan_xtu.panel_rows * c->KP;
  int64_t slab_floats = c->slab_stride_p * c->plan_xv.max_seg;
  if (c->slab_stride_q * c->plan_xtu.max_seg > slab_floats) slab_floats = c->slab_stride_q * c->plan_xtu.max_seg;
  if ((rc = dev_alloc(c, &c->nseg_xv, c->plan_xv.segmask.size(), false))) return bail(rc);
  if ((rc = dev_alloc(c, &c->nseg_xtu, c->plan_xtu.segmask.size(), false))) return bail(rc);
  if (hipMemcpy(c->nseg_xv, c->plan_xv.segmask.data(), c->plan_xv.segmask.size() * sizeof(segmask_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->nseg_xtu, c->plan_xtu.segmask.data(), c->plan_xtu.segmask.size() * sizeof(segmask_t), hipMemcpyHostToDevice) != hipSuccess) {
    fail(c, PRMF_ERR_HIP, "hipMemcpy(plan) failed");
    return bail(PRMF_ERR_HIP);
  }

  if ((rc = dev_alloc(c, (char**)&c->Vp, (size_t)packed_bytes(c, c->Kpad_x)))) return bail(rc);
  if ((rc = dev_alloc(c, (char**)&c->Up, (size_t)packed_bytes(c, c->Kpad_xt)))) return bail(rc);
  if ((rc = dev_alloc(c, &c->slabs, (size_t)slab_floats, false))) return bail(rc);
  int gb = gram_blocks(m_local > n ? m_local : n);
  if (gb < 1024) gb = 1024;                    // the fused epilogues write up to GRAM_MAX_PARTS per-workgroup partials
  if ((rc = dev_alloc(c, (float**)&c->gram_partial, (size_t)gb * k * k))) return bail(rc);
  if ((rc = dev_alloc(c, &c->Gv, (size_t)k * k))) return bail(rc);
  if ((rc = dev_alloc(c, &c->active_dev, (size_t)k))) return bail(rc);
  if ((rc = dev_alloc(c, &c->pos, (size_t)n * c->kc, false))) return bail(rc);
  c->wpt_u = epilogue_wpt(c->Kpad_xt, c->plan_xv.max_seg, c->NT);
  c->wpt_v = epilogue_wpt(c->Kpad_x, c->plan_xtu.max_seg, c->NT);
  c->vblocks = c->fused_v ? v_epilogue_waves(n, c->wpt_v) : v_update_blocks(n);
  if (c->fused_v && (rc = dev_alloc(c, &c->vpartial, (size_t)c->vblocks * (c->kc + 1)))) return bail(rc);
  if (!c->fused_v && (rc = dev_alloc(c, &c->pos_all, (size_t)n * k, false))) return bail(rc);
  if ((rc = dev_alloc(c, &c->colres, (size_t)c->kc * 2))) return bail(rc);
  if ((rc = dev_alloc(c, &c->hyper, 4))) return bail(rc);
  if ((rc = dev_alloc(c, &c->records, (size_t)MAX_RECORDS * PRMF_OBJ_STRIDE))) return bail(rc);
  if ((rc = dev_alloc(c, &c->colnorm, (size_t)k))) return bail(rc);
  if ((rc = dev_alloc(c, &c->need_dev, 2))) return bail(rc);
  {
    const int one = 1;                 // need_dev[0]: the request flag the record kernels raise; need_dev[1]: always 1
    if (hipMemcpy(c->need_dev + 1, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { fail(c, PRMF_ERR_HIP, "hipMemcpy failed"); return bail(PRMF_ERR_HIP); }
  }
  if ((rc = dev_alloc(c, &c->resid_partial, (size_t)4 * RESIDUAL_MAX_BLOCKS))) return bail(rc);
  if ((rc = dev_alloc(c, &c->resid_sq, 2))) return bail(rc);
  if ((rc = dev_alloc(c, &c->Gu_keep, (size_t)(MAX_RECORDS + 1) * k * k, false))) return bail(rc);
  if ((rc = dev_alloc(c, &c->Gv_keep, (size_t)(MAX_RECORDS + 1) * k * k, false))) return bail(rc);
  if ((rc = dev_alloc(c, &c->colnorm2_ring, (size_t)(MAX_RECORDS + 1) * k))) return bail(rc);
  if ((rc = dev_alloc(c, &c->cross_ring, (size_t)MAX_RECORDS + 1))) return bail(rc);
  if ((rc = dev_alloc(c, &c->colres_ring, (size_t)MAX_RECORDS * k * 2))) return bail(rc);
  if (hipHostMalloc((void**)&c->need_host, 2 * sizeof(int)) != hipSuccess ||
      hipHostMalloc((void**)&c->hyper_host, 4 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&c->records_host, (size_t)MAX_RECORDS * PRMF_OBJ_STRIDE * sizeof(double)) != hipSuccess) {
    fail(c, PRMF_ERR_HIP, "hipHostMalloc failed");
    return bail(PRMF_ERR_HIP);
  }
  if ((flags & PRMF_FLAG_TIMING) && (rc = prmf_set_timing(c, 1))) return bail(rc);
  if ((rc = null_stream_done(c))) return bail(rc);
  *out = c;
  return PRMF_OK;
}

int prmf_destroy(prmf_ctx* c) {
  if (!c) return PRMF_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  dev_free(c->X);
  dev_free(c->XT);
  dev_free(c->setx_partial);
  dev_free(c->setx_rowsum);
  dev_free(c->setx_total);
  dev_free(c->Vp);
  dev_free(c->Up);
  dev_free(c->slabs);
  dev_free(c->nseg_xv);
  dev_free(c->nseg_xtu);
  dev_free(c->gram_partial);
  dev_free(c->Gv);
  if (c->own_comm) {
    dev_free(c->gram_buf);
    dev_free(c->q_send);
    dev_free(c->v_send);
    if (c->lr > 1) {
      dev_free(c->q_recv);
      dev_free(c->v_recv);
    }
  }
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  for (void* p : c->lib_allocs) dev_free(p);
  dev_free(c->active_dev);
  dev_free(c->pos);
  dev_free(c->pos_all);
  dev_free(c->gnum);
  dev_free(c->gden);
  if (c->obj_scratch_all != c->obj_scratch) dev_free(c->obj_scratch_all);
  dev_free(c->obj_scratch);
  dev_free(c->colres);
  dev_free(c->vpartial);
  dev_free(c->hyper);
  dev_free(c->records);
  dev_free(c->colnorm);
  dev_free(c->need_dev);
  dev_free(c->resid_partial);
  dev_free(c->resid_sq);
  dev_free(c->Gu_keep);
  dev_free(c->Gv_keep);
  dev_free(c->colnorm2_ring);
  dev_free(c->cross_ring);
  dev_free(c->colres_ring);
  dev_free(c->vsup_ring);
  dev_free(c->scores_dev);
  if (c->hyper_host) (void)hipHostFree(c->hyper_host);
  if (c->need_host) (void)hipHostFree(c->need_host);
  if (c->records_host) (void)hipHostFree(c->records_host);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_phase)
    if (ev) (void)hipEventDestroy(ev);
  delete c;
  return PRMF_OK;
}

namespace {

// ---- X enters the library (setx.hip) -------------------------------------------------------------------------------
int src_elem_bytes(int src_dtype) { return src_dtype == PRMF_SRC_F64 ? 8 : (src_dtype == PRMF_SRC_BF16 ? 2 : 4); }

// buffers of the resident copies and the scratch of the layout kernels; the padding is zeroed on `s`
int setx_prepare(prmf_ctx* c, hipStream_t s) {
  int rc;
  const int eb = elem_bytes(c->x_dtype);
  const int64_t rows_x = round_up(c->m > 0 ? c->m : 1, M_PAD), rows_xt = round_up(c->n, M_PAD);
  // row stride = padded reduction length.  Extra padding between the rows (128 B ... 2 KiB, either copy) was measured again in
  // round 4 against the ping-pong sweeps: no effect (profiles/r04_ab_row_stride_padding.txt; one 3-round A/B that showed -2.7 % did
  // not repeat over 5 rounds in separate processes -- buffer placement, not the stride)
  c->ldx = c->Kpad_x;
  c->ldxt = c->Kpad_xt;
  if (!c->X) {
    c->x_bytes = (size_t)(rows_x * c->ldx * eb);
    c->xt_bytes = (size_t)(rows_xt * c->ldxt * eb);
    if ((rc = dev_alloc(c, (char**)&c->X, c->x_bytes, false))) return rc;
    if ((rc = dev_alloc(c, (char**)&c->XT, c->xt_bytes, false))) return rc;
  }
  if (!c->setx_partial) {
    if ((rc = dev_alloc(c, &c->setx_partial, (size_t)setx_partial_doubles(c->m, c->n), false))) return rc;
    if ((rc = dev_alloc(c, &c->setx_rowsum, (size_t)((c->m + 63) / 64), false))) return rc;
    if ((rc = dev_alloc(c, &c->setx_total, 1))) return rc;
  }
  HIP_TRY(c, hipMemsetAsync(c->X, 0, c->x_bytes, s));       // the padding (rows >= m, columns >= n) must read as zero
  HIP_TRY(c, hipMemsetAsync(c->XT, 0, c->xt_bytes, s));
  return PRMF_OK;
}

// `rows` rows of the source, starting at row `row0` (a multiple of 64) of this rank's matrix, into X, X^T and the tile-row sums
int setx_rows(prmf_ctx* c, int src_dtype, const void* src, int64_t ld, int64_t row0, int64_t rows, hipStream_t s) {
  const int eb = elem_bytes(c->x_dtype);
  HIP_TRY(c, launch_setx_rows(src_dtype, c->x_dtype, src, ld, rows, c->n, (char*)c->X + row0 * c->ldx * eb, c->ldx,
                              (char*)c->XT + row0 * eb, c->ldxt, c->setx_partial, c->setx_rowsum + row0 / 64, s));
  return PRMF_OK;
}

// Staging of prmf_set_x_host: two pinned host buffers (narrowed chunks), two device buffers (chunks as they arrive), a copy
// stream and a kernel stream (non-blocking: the pageable copies of the direct path run on the null stream and must not
// wait for the layout kernels of the chunk before).  Pinning memory costs milliseconds, a solve's setup has none to
// spare, so ONE set per device is kept by the process between calls (prmf_release_staging frees them); concurrent callers
// (thread ranks) get sets of their own, which are freed on release.
struct XStage {
  int device = -1;
  void* hb[2] = {nullptr, nullptr};
  void* ds[2] = {nullptr, nullptr};
  size_t hcap = 0, dcap = 0;
  void* dd[2] = {nullptr, nullptr};   // device buffers of the DIRECT lane (chunks that travel in their source type next to the narrowed ones)
  size_t ddcap = 0;
  hipStream_t sc = nullptr, sk = nullptr;
  hipEvent_t evc[2] = {nullptr, nullptr}, evk[2] = {nullptr, nullptr}, evd[2] = {nullptr, nullptr};
};
std::mutex g_stage_mu;
std::vector<XStage*> g_stage_idle;

void stage_free(XStage* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  for (int b = 0; b < 2; ++b) {
    if (st->hb[b]) (void)hipHostFree(st->hb[b]);
    if (st->ds[b]) (void)hipFree(st->ds[b]);
    if (st->dd[b]) (void)hipFree(st->dd[b]);
    if (st->evc[b]) (void)hipEventDestroy(st->evc[b]);
    if (st->evk[b]) (void)hipEventDestroy(st->evk[b]);
    if (st->evd[b]) (void)hipEventDestroy(st->evd[b]);
  }
  if (st->sc) (void)hipStreamDestroy(st->sc);
  if (st->sk) (void)hipStreamDestroy(st->sk);
  delete st;
}

XStage* stage_acquire(prmf_ctx* c, size_t hbytes, size_t dbytes, size_t ddbytes = 0) {
  XStage* st = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_stage_mu);
    for (size_t i = 0; i < g_stage_idle.size(); ++i)
      if (g_stage_idle[i]->device == c->device) {
        st = g_stage_idle[i];
        g_stage_idle.erase(g_stage_idle.begin() + (long)i);
        break;
      }
  }
  if (!st) {
    st = new (std::nothrow) XStage();
    if (!st) { fail(c, PRMF_ERR_HIP, "out of host memory"); return nullptr; }
    st->device = c->device;
    bool ok = hipStreamCreateWithFlags(&st->sc, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&st->sk, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b)
      ok = hipEventCreateWithFlags(&st->evc[b], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&st->evk[b], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&st->evd[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) { fail(c, PRMF_ERR_HIP, "staging streams / events could not be created"); stage_free(st); return nullptr; }
  }
  for (int b = 0; b < 2; ++b) {
    if (hbytes > st->hcap) {
      if (st->hb[b]) (void)hipHostFree(st->hb[b]);
      st->hb[b] = nullptr;
      if (hipHostMalloc(&st->hb[b], hbytes, hipHostMallocDefault) != hipSuccess) {
        fail(c, PRMF_ERR_HIP, "hipHostMalloc(%zu bytes of staging) failed", hbytes);
        st->hcap = 0;
        stage_free(st);
        return nullptr;
      }
    }
    if (dbytes > st->dcap) {
      if (st->ds[b]) (void)hipFree(st->ds[b]);
      st->ds[b] = nullptr;
      if (hipMalloc(&st->ds[b], dbytes) != hipSuccess) {
        fail(c, PRMF_ERR_HIP, "hipMalloc(%zu bytes of staging) failed", dbytes);
        st->dcap = 0;
        stage_free(st);
        return nullptr;
      }
    }
  }
  for (int b = 0; b < 2; ++b)
    if (ddbytes > st->ddcap) {
      if (st->dd[b]) (void)hipFree(st->dd[b]);
      st->dd[b] = nullptr;
      if (hipMalloc(&st->dd[b], ddbytes) != hipSuccess) {
        fail(c, PRMF_ERR_HIP, "hipMalloc(%zu bytes of staging) failed", ddbytes);
        st->ddcap = 0;
        stage_free(st);
        return nullptr;
      }
    }
  if (hbytes > st->hcap) st->hcap = hbytes;
  if (dbytes > st->dcap) st->dcap = dbytes;
  if (ddbytes > st->ddcap) st->ddcap = ddbytes;
  return st;
}

void stage_release(XStage* st) {
  {
    std::lock_guard<std::mutex> lock(g_stage_mu);
    bool have = false;
    for (XStage* o : g_stage_idle) have = have || o->device == st->device;
    if (!have) {
      g_stage_idle.push_back(st);
      return;
    }
  }
  stage_free(st);
}

// n worker threads that run one job at a time, each on its share (the narrowing of a chunk's rows)
class RowWorkers {
 public:
  explicit RowWorkers(int n) : n_(n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
  }
  ~RowWorkers() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    go_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(const std::function<void(int, int)>& job) {
    start(job);
    wait();
  }
  // start: the job must stay alive until wait() has returned; wait() returns when every worker has finished it and gives the
  // time the LAST one finished (what the narrowing of a chunk took, whatever the caller did meanwhile)
  void start(const std::function<void(int, int)>& job) {
    std::lock_guard<std::mutex> lock(mu_);
    job_ = &job;
    pending_ = n_;
    ++gen_;
    go_.notify_all();
  }
  std::chrono::steady_clock::time_point wait() {
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [this] { return pending_ == 0; });
    return finished_;
  }

 private:
  void loop(int i) {
    unsigned seen = 0;
    for (;;) {
      const std::function<void(int, int)>* job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        go_.wait(lock, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        job = job_;
      }
      (*job)(i, n_);
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--pending_ == 0) {
          finished_ = std::chrono::steady_clock::now();
          done_.notify_all();
        }
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable go_, done_;
  const std::function<void(int, int)>* job_ = nullptr;
  unsigned gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
  std::chrono::steady_clock::time_point finished_;
};

int setx_finish(prmf_ctx* c, double* local_out, hipStream_t s) {
  double total = 0.0;
  if (c->m > 0) {
    HIP_TRY(c, launch_setx_total(c->setx_rowsum, (c->m + 63) / 64, c->setx_total, s));
    HIP_TRY(c, hipMemcpyAsync(&total, c->setx_total, sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  c->norm_x_sq = total;
  c->have_x = true;
  if (local_out) *local_out = total;
  return PRMF_OK;
}

}  // namespace

int prmf_set_x(prmf_ctx* c, const void* dX, int64_t ld, double norm_x_sq, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (c->m > 0 && !dX) return fail(c, PRMF_ERR_ARG, "dX is NULL");
  if (ld < c->n) return fail(c, PRMF_ERR_ARG, "ld=%lld < n=%lld", (long long)ld, (long long)c->n);
  if (!(norm_x_sq >= 0.0) || !std::isfinite(norm_x_sq)) return fail(c, PRMF_ERR_ARG, "norm_x_sq must be finite and >= 0");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = setx_prepare(c, s))) return rc;
  if (c->m > 0 && (rc = setx_rows(c, c->x_dtype, dX, ld, 0, c->m, s))) return rc;
  c->norm_x_sq = norm_x_sq;
  c->have_x = true;
  return PRMF_OK;
}

int prmf_set_norm_x_sq(prmf_ctx* c, double norm_x_sq) {
  if (!c) return PRMF_ERR_ARG;
  if (!(norm_x_sq >= 0.0) || !std::isfinite(norm_x_sq)) return fail(c, PRMF_ERR_ARG, "norm_x_sq must be finite and >= 0");
  c->norm_x_sq = norm_x_sq;
  return PRMF_OK;
}

int prmf_set_x_device(prmf_ctx* c, const void* dX, int src_dtype, int64_t ld, double* local_norm_sq_out, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (c->m > 0 && !dX) return fail(c, PRMF_ERR_ARG, "dX is NULL");
  if (ld < c->n) return fail(c, PRMF_ERR_ARG, "ld=%lld < n=%lld", (long long)ld, (long long)c->n);
  if (src_dtype != PRMF_SRC_F32 && src_dtype != PRMF_SRC_BF16 && src_dtype != PRMF_SRC_F64) return fail(c, PRMF_ERR_ARG, "bad src_dtype %d", src_dtype);
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = setx_prepare(c, s))) return rc;
  if (c->m > 0 && (rc = setx_rows(c, src_dtype, dX, ld, 0, c->m, s))) return rc;
  return setx_finish(c, local_norm_sq_out, s);
}

int prmf_set_x_host(prmf_ctx* c, const void* hX, int src_dtype, int64_t ld, int n_threads, double* local_norm_sq_out, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (c->m > 0 && !hX) return fail(c, PRMF_ERR_ARG, "hX is NULL");
  if (ld < c->n) return fail(c, PRMF_ERR_ARG, "ld=%lld < n=%lld", (long long)ld, (long long)c->n);
  if (src_dtype != PRMF_SRC_F32 && src_dtype != PRMF_SRC_BF16 && src_dtype != PRMF_SRC_F64) return fail(c, PRMF_ERR_ARG, "bad src_dtype %d", src_dtype);
  int rc;
  if ((rc = set_device(c))) return rc;
  HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));      // nothing queued by the caller still reads the old X
  const int eb = elem_bytes(c->x_dtype), seb = src_elem_bytes(src_dtype);
  // Narrow on the host when that halves (or quarters) what crosses PCIe: worker threads convert chunk i + 1 into pinned memory
  // while the DMA engine moves chunk i and the layout kernels take chunk i - 1.  n_threads = 0 (or a source no wider than
  // the storage type): the chunk travels as it is (pageable copy) and is narrowed by the layout kernel.
  int workers = n_threads;
  if (workers < 0) {
    const unsigned hw = std::thread::hardware_concurrency();
    workers = hw > 16 ? 16 : (hw > 1 ? (int)hw : 1);
  }
  const bool narrow = workers > 0 && seb > eb && c->m > 0;
  const int64_t target = 32LL << 20;                          // bytes of SOURCE per chunk
  int64_t rows_chunk = target / (c->n * (int64_t)seb) / 64 * 64;
  if (rows_chunk < 64) rows_chunk = 64;
  if (rows_chunk > round_up(c->m > 0 ? c->m : 1, 64)) rows_chunk = round_up(c->m > 0 ? c->m : 1, 64);
  const int trav_dtype = narrow ? (c->x_dtype == PRMF_X_BF16 ? PRMF_SRC_BF16 : PRMF_SRC_F32) : src_dtype;   // type on the wire
  const int teb = src_elem_bytes(trav_dtype);
  const size_t chunk_bytes = (size_t)(rows_chunk * c->n * teb);
  // The DIRECT lane of the narrowing path.  The worker threads share the host with whatever else runs there; when they cannot
  // keep the link busy (measured per chunk: narrowing takes more than 1.25 x the chunk's own time on the wire) the calling
  // thread, which would only wait for them, sends the NEXT chunk as it is -- a pageable copy in the source type, narrowed by the
  // layout kernel -- while they work: a starved host then loses 10-25 % against a quiet one instead of 50-100 %
  // (profiles/r06_outer_loop_host_time.txt).  Which lane a chunk took changes no stored bit and no bit of ||X||^2 (the tile
  // tree is per 64 rows).  PRMF_SETX_HYBRID = 0 / 1 pins the decision (a test hook, like PRMF_SWEEP_XCD).
  int hybrid = -1;
  if (const char* e = getenv("PRMF_SETX_HYBRID")) hybrid = atoi(e) != 0 ? 1 : 0;
  const size_t direct_bytes = narrow && hybrid != 0 ? (size_t)(rows_chunk * c->n * seb) : 0;
  XStage* st = stage_acquire(c, narrow ? chunk_bytes : 0, chunk_bytes, direct_bytes);
  if (!st) return PRMF_ERR_HIP;
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(st->sc);
    (void)hipStreamSynchronize(st->sk);
    stage_release(st);
    return code;
  };
#define SX_TRY(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return bail(fail(c, PRMF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)));      \
  } while (0)
  if ((rc = setx_prepare(c, st->sk))) return bail(rc);
  if (c->m > 0) {
    std::unique_ptr<RowWorkers> pool;
    if (narrow) pool.reset(new RowWorkers(workers));
    auto copy_rows = [&](void* dst, int64_t r0, int64_t nr) -> hipError_t {          // pageable rows [r0, r0 + nr) in the source type
      const char* srow = (const char*)hX + r0 * ld * seb;
      if (ld == c->n) return hipMemcpy(dst, srow, (size_t)(nr * c->n * seb), hipMemcpyHostToDevice);
      return hipMemcpy2D(dst, (size_t)(c->n * seb), srow, (size_t)(ld * seb), (size_t)(c->n * seb), (size_t)nr, hipMemcpyHostToDevice);
    };
    int64_t i = 0, j = 0;                      // chunks sent through the narrowing lane / the direct lane
    bool direct_on = hybrid == 1;
    double ema = 0.0;                          // narrowing time per byte on the wire, relative to 50 GB/s
    for (int64_t r0 = 0; r0 < c->m; ++i) {
      const int b = (int)(i & 1);
      const int64_t nr = c->m - r0 < rows_chunk ? c->m - r0 : rows_chunk;
      if (narrow) {
        if (i >= 2) SX_TRY(hipEventSynchronize(st->evc[b]));            // the DMA of chunk i - 2 has left this pinned buffer
        void* dst = st->hb[b];
        const int64_t ra = r0;
        const std::function<void(int, int)> job = [&](int w, int nw) {
          const int64_t lo = ra + nr * w / nw, hi = ra + nr * (w + 1) / nw;
          host_narrow_rows(src_dtype, c->x_dtype, hX, ld, lo, hi, c->n, (char*)dst + (lo - ra) * c->n * teb);
        };
        const auto t0 = std::chrono::steady_clock::now();
        pool->start(job);
        r0 += nr;
        if (direct_on && r0 < c->m) {          // the direct lane: the next chunk as it is, while the workers narrow this one
          const int bd = (int)(j & 1);
          const int64_t nrd = c->m - r0 < rows_chunk ? c->m - r0 : rows_chunk;
          hipError_t e = j >= 2 ? hipEventSynchronize(st->evd[bd]) : hipSuccess;
          if (e == hipSuccess) e = copy_rows(st->dd[bd], r0, nrd);
          if (e != hipSuccess) {
            (void)pool->wait();
            return bail(fail(c, PRMF_ERR_HIP, "direct lane: %s", hipGetErrorString(e)));
          }
          if ((rc = setx_rows(c, src_dtype, st->dd[bd], c->n, r0, nrd, st->sk))) {
            (void)pool->wait();
            return bail(rc);
          }
          e = hipEventRecord(st->evd[bd], st->sk);
          if (e != hipSuccess) {
            (void)pool->wait();
            return bail(fail(c, PRMF_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e)));
          }
          r0 += nrd;
          ++j;
        }
        const double t_narrow = std::chrono::duration<double>(pool->wait() - t0).count();
        if (hybrid < 0) {
          const double rel = t_narrow / ((double)nr * (double)c->n * (double)teb / 50e9);
          ema = i == 0 ? rel : 0.5 * ema + 0.5 * rel;
          direct_on = ema > 1.25;
        }
        if (i >= 2) SX_TRY(hipStreamWaitEvent(st->sc, st->evk[b], 0));  // the kernels of chunk i - 2 have read this device buffer
        SX_TRY(hipMemcpyAsync(st->ds[b], dst, (size_t)(nr * c->n * teb), hipMemcpyHostToDevice, st->sc));
        SX_TRY(hipEventRecord(st->evc[b], st->sc));
        SX_TRY(hipStreamWaitEvent(st->sk, st->evc[b], 0));
        if ((rc = setx_rows(c, trav_dtype, st->ds[b], c->n, ra, nr, st->sk))) return bail(rc);
        SX_TRY(hipEventRecord(st->evk[b], st->sk));
      } else {
        if (i >= 2) SX_TRY(hipEventSynchronize(st->evk[b]));
        SX_TRY(copy_rows(st->ds[b], r0, nr));
        if ((rc = setx_rows(c, trav_dtype, st->ds[b], c->n, r0, nr, st->sk))) return bail(rc);
        SX_TRY(hipEventRecord(st->evk[b], st->sk));
        r0 += nr;
      }
    }
  }
#undef SX_TRY
  rc = setx_finish(c, local_norm_sq_out, st->sk);
  return bail(rc);
}

int prmf_release_staging(void) {
  std::vector<XStage*> all;
  {
    std::lock_guard<std::mutex> lock(g_stage_mu);
    all.swap(g_stage_idle);
  }
  for (XStage* st : all) stage_free(st);
  return PRMF_OK;
}

int prmf_set_pathway_library(prmf_ctx* c, int P, const int64_t* sup_ptr, const int32_t* sup_idx, const double* sup_deg,
                             const double* sup_ldiag, const int64_t* row_ptr, const int32_t* col_local,
                             const double* w, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (P <= 0 || !sup_ptr || !sup_idx || !sup_deg || !sup_ldiag || !row_ptr) return fail(c, PRMF_ERR_ARG, "NULL library array or P <= 0");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t ts = sup_ptr[P], nnz = row_ptr[ts];
  if (sup_ptr[0] != 0 || row_ptr[0] != 0 || ts < 0 || nnz < 0) return fail(c, PRMF_ERR_ARG, "malformed pointer arrays");
  if (nnz > 0 && (!col_local || !w)) return fail(c, PRMF_ERR_ARG, "NULL adjacency arrays");
  int max_sup = 1;
  for (int p = 0; p < P; ++p) {
    int64_t ns = sup_ptr[p + 1] - sup_ptr[p];
    if (ns < 0) return fail(c, PRMF_ERR_ARG, "sup_ptr not monotone at pathway %d", p);
    if (ns > max_sup) max_sup = (int)ns;
    for (int64_t i = sup_ptr[p]; i < sup_ptr[p + 1]; ++i) {
      if (sup_idx[i] < 0 || sup_idx[i] >= c->n) return fail(c, PRMF_ERR_ARG, "support index %d out of range in pathway %d", sup_idx[i], p);
      if (row_ptr[i + 1] < row_ptr[i]) return fail(c, PRMF_ERR_ARG, "row_ptr not monotone");
      for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e)
        if (col_local[e] < 0 || col_local[e] >= ns) return fail(c, PRMF_ERR_ARG, "col_local out of range in pathway %d", p);
    }
  }
  // (re)allocate
  HIP_TRY(c, hipStreamSynchronize(s));
  for (void* p : c->lib_allocs) dev_free(p);
  c->lib_allocs.clear();
  dev_free(c->gnum); c->gnum = nullptr;
  dev_free(c->gden); c->gden = nullptr;
  if (c->obj_scratch_all != c->obj_scratch) dev_free(c->obj_scratch_all);
  c->obj_scratch_all = nullptr;
  dev_free(c->obj_scratch); c->obj_scratch = nullptr;
  dev_free(c->vsup_ring); c->vsup_ring = nullptr;
  c->have_lib = false;
  c->have_active = false;

  std::vector<double> dis((size_t)(ts ? ts : 1));
  for (int64_t i = 0; i < ts; ++i) dis[i] = sup_ldiag[i] != 0.0 ? std::pow(sup_ldiag[i], -0.5) : 0.0;

  auto up = [&](const void* host, size_t bytes, void** dev) -> int {
    char* d = nullptr;
    int r = dev_alloc(c, &d, bytes ? bytes : 1, false);
    if (r) return r;
    c->lib_allocs.push_back(d);
    if (bytes) {
      hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(c, PRMF_ERR_HIP, "hipMemcpy(library): %s", hipGetErrorString(e));
    }
    *dev = d;
    return PRMF_OK;
  };
  PathwayLibDev L{};
  L.P = P;
  L.max_support = max_sup;
  int64_t max_nnz = 0;
  for (int pp = 0; pp < P; ++pp) {
    const int64_t e = row_ptr[sup_ptr[pp + 1]] - row_ptr[sup_ptr[pp]];
    if (e > max_nnz) max_nnz = e;
  }
  L.max_nnz = (int)max_nnz;
  L.total_support = ts;
  L.total_nnz = nnz;
  if ((rc = up(sup_ptr, (size_t)(P + 1) * 8, (void**)&L.sup_ptr))) return rc;
  if ((rc = up(sup_idx, (size_t)ts * 4, (void**)&L.sup_idx))) return rc;
  if ((rc = up(sup_deg, (size_t)ts * 8, (void**)&L.sup_deg))) return rc;
  if ((rc = up(dis.data(), (size_t)ts * 8, (void**)&L.sup_dis))) return rc;
  if ((rc = up(row_ptr, (size_t)(ts + 1) * 8, (void**)&L.row_ptr))) return rc;
  if ((rc = up(col_local, (size_t)nnz * 4, (void**)&L.col_local))) return rc;
  if ((rc = up(w, (size_t)nnz * 8, (void**)&L.w))) return rc;
  c->lib = L;
  if ((rc = dev_alloc(c, &c->gnum, (size_t)c->kc * max_sup))) return rc;
  if ((rc = dev_alloc(c, &c->gden, (size_t)c->kc * max_sup))) return rc;
  if ((rc = dev_alloc(c, &c->obj_scratch, (size_t)c->kc * max_sup))) return rc;
  c->obj_scratch_all = c->obj_scratch;
  if (c->kc != c->k && (rc = dev_alloc(c, &c->obj_scratch_all, (size_t)c->k * max_sup))) return rc;
  if ((rc = dev_alloc(c, &c->vsup_ring, (size_t)MAX_RECORDS * c->k * max_sup))) return rc;
  if ((rc = null_stream_done(c))) return rc;
  c->have_lib = true;
  return PRMF_OK;
}

int prmf_set_active(prmf_ctx* c, const int32_t* k_to_lapl, void* stream) {
  if (!c || !k_to_lapl) return PRMF_ERR_ARG;
  if (!c->have_lib) return fail(c, PRMF_ERR_STATE, "prmf_set_pathway_library has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (c->batch_open && c->cursor > 0)
    return fail(c, PRMF_ERR_STATE, "prmf_set_active between prmf_begin and prmf_end: the objective records of the %d iterations run so far "
                                   "are evaluated in prmf_end with the active pathways", c->cursor);
  for (int i = 0; i < c->k; ++i)
    if (k_to_lapl[i] < 0 || k_to_lapl[i] >= c->lib.P) return fail(c, PRMF_ERR_ARG, "latent %d: pathway %d out of range", i, k_to_lapl[i]);
  // the k indices travel as kernel arguments: no staging buffer, no stream synchronisation
  HIP_TRY(c, launch_set_active(c->active_dev, k_to_lapl, c->k, s));
  HIP_TRY(c, launch_build_pos(c->lib, c->active_dev, c->col0, c->ncols, c->kc, c->n, c->pos, s));
  if (c->pos_all) HIP_TRY(c, launch_build_pos(c->lib, c->active_dev, 0, c->k, c->k, c->n, c->pos_all, s));
  c->have_active = true;
  return PRMF_OK;
}

int prmf_invalidate_factors(prmf_ctx* c) {
  if (!c) return PRMF_ERR_ARG;
  c->prepared = false;
  return PRMF_OK;
}

int prmf_get_comm_sizes(const prmf_ctx* c, prmf_comm_sizes* out) {
  if (!c || !out) return PRMF_ERR_ARG;
  out->gram_doubles = (int64_t)c->k * c->k;
  out->q_recv_floats = c->n * c->kc;
  out->q_send_floats = out->q_recv_floats * c->lr;
  out->v_send_floats = c->v_send_floats;
  out->v_recv_floats = c->v_send_floats * c->lr;
  out->kc = c->kc;
  out->mode = c->replicated ? PRMF_DIST_REPLICATED : PRMF_DIST_COLUMN_SHARDED;
  return PRMF_OK;
}

int prmf_set_comm_buffers(prmf_ctx* c, double* gram, float* q_send, float* q_recv, float* v_send, float* v_recv) {
  if (!c) return PRMF_ERR_ARG;
  if (!gram || !q_send || !q_recv || !v_send || !v_recv) return fail(c, PRMF_ERR_ARG, "NULL communication buffer");
  if (c->own_comm) {
    dev_free(c->gram_buf);
    dev_free(c->q_send);
    dev_free(c->v_send);
    if (c->lr > 1) {
      dev_free(c->q_recv);
      dev_free(c->v_recv);
    }
    c->own_comm = false;
  }
  if (c->lr == 1 && (q_recv != q_send || v_recv != v_send))
    return fail(c, PRMF_ERR_ARG, "with one column block (single rank or replicated mode) q_recv / v_recv must alias q_send / v_send");
  c->gram_buf = gram;
  c->q_send = q_send;
  c->q_recv = q_recv;
  c->v_send = v_send;
  c->v_recv = v_recv;
  return PRMF_OK;
}

int prmf_begin(prmf_ctx* c, float* dU, float* dV, double gamma, double delta, double tradeoff, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if ((c->m > 0 && !dU) || !dV) return fail(c, PRMF_ERR_ARG, "NULL factor pointer");
  if (tradeoff > 1.0) return fail(c, PRMF_ERR_ARG, "tradeoff must be < 0 (off) or in [0,1]");
  int rc;
  if ((rc = set_device(c))) return rc;
  if ((rc = ensure_own_comm(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (c->U != dU || c->V != dV) c->prepared = false;
  c->U = dU;
  c->V = dV;
  if ((rc = check_ready(c))) return rc;
  HIP_TRY(c, launch_set_hyper(c->hyper, gamma, delta, tradeoff, s));   // kernel arguments: no staging, no sync
  c->cursor = 0;
  c->graph_done = false;
  if (!c->prepared && (rc = prepare(c, s))) return rc;
  // Objective deferred to prmf_end + reductions folded into the sweeps' prologues: whenever the tradeoff feedback is
  // off (with it on, gamma / delta of step t + 1 need the record of step t) -- one rank and both multi-rank modes.
  // Decided per rank; the exchanges do not depend on it.
  c->tradeoff_on = tradeoff >= 0.0;
  c->lazy = !c->tradeoff_on && c->vsup_ring &&
            !(c->flags & (PRMF_FLAG_EAGER_OBJECTIVE | PRMF_FLAG_DIRECT_RESIDUAL));
  c->batch_open = true;
  return PRMF_OK;
}

namespace {
double* vpartial_of(const prmf_ctx* c) {     // per-workgroup sums of the V update: own buffer, or the tail of v_send
  return c->fused_v ? c->vpartial : reinterpret_cast<double*>(c->v_send + c->partial_off);
}
// Does the X^T U sweep's prologue reduce this rank's U^T U partials?  One rank: always (deferred form).  Several ranks:
// when the library issues the collectives itself -- the k x k all-reduce then follows the sweep, in one RCCL group with
// the exchange of Q.  Host-driven phase calls all-reduce U^T U between phase_u and phase_q, so phase_u reduces it.
bool gram_in_sweep(const prmf_ctx* c) { return c->lazy && c->m > 0 && (c->nranks == 1 || c->dist_driver); }
}  // namespace

int prmf_phase_u(prmf_ctx* c, void* stream) {
  int rc;
  if ((rc = check_ready(c))) return rc;
  if (!c->prepared) return fail(c, PRMF_ERR_STATE, "prmf_begin has not been called");
  if (c->cursor >= MAX_RECORDS) return fail(c, PRMF_ERR_STATE, "more than %d iterations between prmf_begin and prmf_end", MAX_RECORDS);
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (c->m <= 0) {                     // a rank without samples: no sweep to host the V-side reductions
    if (c->lazy && c->gv_pending) {
      HIP_TRY(c, launch_prologue_alone(post_prologue(c, c->cursor - 1), s));
      c->gv_pending = false;
    }
    HIP_TRY(c, hipMemsetAsync(c->gram_buf, 0, (size_t)c->k * c->k * sizeof(double), s));
    return PRMF_OK;
  }
  const bool timing = (c->flags & PRMF_FLAG_TIMING) != 0;
  if (c->lazy && c->gv_pending && c->cursor == 0) {   // cannot happen through prmf_begin/prmf_end; keep V^T V right anyway
    HIP_TRY(c, launch_prologue_alone(post_prologue(c, -1), s));
    c->gv_pending = false;
  }
  SweepPrologue pro;
  if (c->lazy && c->gv_pending) pro = post_prologue(c, c->cursor - 1);
  if (timing) HIP_TRY(c, hipEventRecord(c->ev[(size_t)c->cursor * 4 + 0], s));
  // P = X V                                             (prmf_runner.py:420)
  HIP_TRY(c, launch_gemm(c->x_dtype, c->plan_xv, c->X, c->ldx, c->m, c->Vp, c->NT, c->slabs, c->slab_stride_p, s, c->split_f32,
                         pro.kind != PROLOGUE_NONE ? &pro : nullptr, false, c->V, c->n, c->k));
  if (timing) HIP_TRY(c, hipEventRecord(c->ev[(size_t)c->cursor * 4 + 1], s));
  c->gv_pending = false;
  // U <- U * P / (U Gv + U)                              (:421-422); the pathway terms of the own columns (they depend on
  // the old V and gamma / delta only) ride in the same launch as extra workgroups
  FactorView fu{c->U, (int)c->m, c->k, c->k, 0};
  bool gram_done = false;
  const GraphJob gj = graph_job(c);
  HIP_TRY(c, launch_u_epilogue(c->U, c->m, c->k, c->slabs, c->nseg_xv, c->plan_xv.panel_rows, c->slab_stride_p, c->KP,
                               c->Gv, c->pack_dtype, c->Up, c->Kpad_xt, (float*)c->gram_partial, &gram_done, c->wpt_u, s, &gj, c->w16));
  c->graph_done = true;
  c->nparts_u = epilogue_blocks(c->Kpad_xt, c->wpt_u);
  if (!gram_done) {
    HIP_TRY(c, launch_gram_partials(fu, (float*)c->gram_partial, s));
    c->nparts_u = gram_blocks(fu.rows);
  }
  if (!gram_in_sweep(c))               // U^T U (this rank's) now: the per-iteration objective, or a host-driven all-reduce, reads it next
    HIP_TRY(c, launch_mid((const float*)c->gram_partial, c->nparts_u, c->k, c->gram_buf, c->lib, c->active_dev, c->col0, 0, c->V,
                          c->hyper, c->gnum, c->gden, s));
  return PRMF_OK;
}

int prmf_phase_q(prmf_ctx* c, void* stream) {
  int rc;
  if ((rc = check_ready(c))) return rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const bool timing = (c->flags & PRMF_FLAG_TIMING) != 0;
  if (c->m > 0) {
    if (timing) HIP_TRY(c, hipEventRecord(c->ev[(size_t)c->cursor * 4 + 2], s));
    // Q = X^T U_new                                       (:424)
    SweepPrologue pro;
    const bool in_sweep = gram_in_sweep(c);
    if (in_sweep) pro = mid_prologue(c, c->cursor);
    HIP_TRY(c, launch_gemm(c->x_dtype, c->plan_xtu, c->XT, c->ldxt, c->n, c->Up, c->NT, c->slabs, c->slab_stride_q, s, c->split_f32,
                           in_sweep ? &pro : nullptr, false, c->U, c->m, c->k));
    if (timing) HIP_TRY(c, hipEventRecord(c->ev[(size_t)c->cursor * 4 + 3], s));
    // both multi-rank modes exchange Q, so they need its summed dense form (by owner column block)
    if (!c->fused_v || c->replicated)
      HIP_TRY(c, launch_slab_sum_blocks(c->slabs, c->nseg_xtu, c->plan_xtu.panel_rows, c->slab_stride_q, c->KP, c->n, c->k, c->kc,
                                        c->lr, c->q_send, c->plan_xtu.max_seg, s));
  } else {
    HIP_TRY(c, hipMemsetAsync(c->q_send, 0, (size_t)c->n * c->kc * c->lr * sizeof(float), s));
  }
  return PRMF_OK;
}

int prmf_phase_v(prmf_ctx* c, void* stream) {
  int rc;
  if ((rc = check_ready(c))) return rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  // sparse pathway terms on the OLD V                    (:431-438) -- normally done next to the U epilogue
  if (!c->graph_done)
    HIP_TRY(c, launch_graph_terms(c->lib, c->active_dev, c->col0, c->ncols, c->V, c->k, c->hyper, c->gnum, c->gden, s));
  c->graph_done = false;
  // V_own <- V * (Q + gnum) / max(V Gu + gden, eps)      (:425, :440-444)
  if (c->fused_v) {
    // one column block (single rank / replicated): V updated in place straight from the sweep's slabs (or the all-reduced
    // dense Q), packed for the next sweep in the same pass; deferred objective: V_new on the supports goes to the ring
    HIP_TRY(c, launch_v_epilogue(c->V, c->n, c->k, c->slabs, c->nseg_xtu, c->plan_xtu.panel_rows, c->slab_stride_q, c->KP,
                                 c->gram_buf, c->pos, c->gnum, c->gden, c->lib.max_support, c->pack_dtype, c->Vp, c->Kpad_x,
                                 c->vpartial, (float*)c->gram_partial, &c->gram_v_done, c->replicated ? c->q_recv : nullptr,
                                 c->lazy ? c->vsup_ring + (size_t)c->cursor * c->k * c->lib.max_support : nullptr, c->wpt_v, s, c->w16));
    c->nparts_v = epilogue_blocks(c->Kpad_x, c->wpt_v);
    if (!c->gram_v_done) {
      FactorView fv{c->V, (int)c->n, c->k, c->k, 0};
      HIP_TRY(c, launch_gram_partials(fv, (float*)c->gram_partial, s));
      c->nparts_v = gram_blocks(fv.rows);
      c->gram_v_done = true;
    }
    if (c->lazy) {
      c->gv_pending = true;
      return PRMF_OK;
    }
    return PRMF_OK;                    // per-iteration form: phase_post reduces V^T V and writes the record
  }
  // column-sharded: the own columns into this rank's all-gather block, its per-workgroup sums into the block's tail
  double* vp = vpartial_of(c);
  HIP_TRY(c, launch_v_update(c->V, c->n, c->k, c->col0, c->ncols, c->kc, c->q_recv, c->gram_buf, c->pos, c->gnum, c->gden,
                             c->lib.max_support, c->v_send, vp, s));
  if (!c->lazy)                        // per-iteration objective: this rank's <Q,V>, manifold, ignore travel too
    HIP_TRY(c, launch_objective_terms(c->lib, c->active_dev, c->col0, c->ncols, c->kc, c->n, c->v_send, vp, c->vblocks, c->colres,
                                      c->obj_scratch, reinterpret_cast<double*>(c->v_send + c->extras_off), s));
  return PRMF_OK;
}

int prmf_phase_post(prmf_ctx* c, void* stream) {
  int rc;
  if ((rc = check_ready(c))) return rc;
  if ((rc = set_device(c))) return rc;
  if (c->cursor >= MAX_RECORDS) return fail(c, PRMF_ERR_STATE, "more than %d iterations between prmf_begin and prmf_end", MAX_RECORDS);
  hipStream_t s = (hipStream_t)stream;
  FactorView fv{c->V, (int)c->n, c->k, c->k, 0};
  if (!c->fused_v) {
    // column-sharded: the gathered column blocks -> replicated V, packed fragments, V^T V partials, and (deferred
    // objective) V_new on the active supports of all k columns
    bool gram_done = false;
    HIP_TRY(c, launch_v_post(c->V, c->n, c->k, c->v_recv, c->v_send_floats, c->kc, c->KP, c->pack_dtype, c->Vp, c->Kpad_x,
                             (float*)c->gram_partial, &gram_done, c->lazy ? c->pos_all : nullptr,
                             c->lazy ? c->vsup_ring + (size_t)c->cursor * c->k * c->lib.max_support : nullptr, c->lib.max_support, s,
                             c->w16));
    c->nparts_v = epilogue_blocks(c->Kpad_x);
    if (!gram_done) {
      HIP_TRY(c, launch_gram_partials(fv, (float*)c->gram_partial, s));
      c->nparts_v = gram_blocks(fv.rows);
    }
    if (!c->lazy) {
      HIP_TRY(c, launch_gram_reduce((const float*)c->gram_partial, c->nparts_v, c->k, c->Gv, s));
      if ((rc = finalize_record(c, nullptr, 0, nullptr, s))) return rc;
    } else {
      c->gv_pending = true;            // V^T V + the gathered sums: the next X V sweep's prologue, or prmf_end
    }
    c->cursor++;
    return PRMF_OK;
  }
  if (!c->lazy) {
    // per-iteration form: V^T V, the objective pieces of all columns and the record (+ tradeoff feedback)
    HIP_TRY(c, launch_post((const float*)c->gram_partial, c->nparts_v, c->k, c->Gv, c->lib, c->active_dev, c->V,
                           c->vpartial, c->vblocks, c->colres, c->obj_scratch, s));
    if ((rc = finalize_record(c, c->vpartial, c->vblocks, c->colres, s))) return rc;
  }
  c->cursor++;                         // deferred form: nothing to launch, the next X V sweep (or prmf_end) closes the iteration
  return PRMF_OK;
}

int prmf_end(prmf_ctx* c, int n_steps, double* obj_out, double* gamma_delta_out, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (n_steps < 0 || n_steps > c->cursor) return fail(c, PRMF_ERR_ARG, "n_steps=%d but %d iterations were run", n_steps, c->cursor);
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  bool lazy_direct = false;
  if (c->lazy) {
    if (c->gv_pending) {               // no sweep follows the last iteration: its V side is closed by a launch of its own
      HIP_TRY(c, launch_prologue_alone(post_prologue(c, c->cursor - 1), s));
      c->gv_pending = false;
    }
    // the objective records of the whole batch (prmf_runner.py:336-372), two launches
    HIP_TRY(c, launch_objective_all(c->lib, c->active_dev, c->k, c->cursor, c->colnorm2_ring, c->vsup_ring, c->colres_ring,
                                    c->obj_scratch_all, s));
    const bool direct = direct_available(c) && c->cursor > 0;
    HIP_TRY(c, launch_records_all(c->Gu_keep, c->Gv_keep, c->k, c->cursor, c->cross_ring, c->colres_ring, c->norm_x_sq, c->hyper,
                                  c->records, direct ? c->need_dev : nullptr, false, direct_below(c), s));
    lazy_direct = direct;
  }
  c->batch_open = false;
  if (n_steps > 0)
    HIP_TRY(c, hipMemcpyAsync(c->records_host, c->records + (size_t)(c->cursor - n_steps) * PRMF_OBJ_STRIDE,
                              (size_t)n_steps * PRMF_OBJ_STRIDE * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(c->hyper_host, c->hyper, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (lazy_direct) HIP_TRY(c, hipMemcpyAsync(c->need_host, c->need_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  if ((rc = sync_stream(c, s, c->cursor))) return rc;
  // The last record of the batch is the one the outer loop acts on: if its identity-based recon^2 was too small to trust
  // (the record kernel raised the flag -- identically on every rank: the records are computed from all-reduced / replicated
  // quantities and the threshold from global sizes, direct_below), the direct residual is evaluated now and that
  // record is fetched again.  The common case costs nothing: no launch, no second synchronisation.
  if (lazy_direct && c->need_host[0] != 0) {
    double* last = c->records + (size_t)(c->cursor - 1) * PRMF_OBJ_STRIDE;
    if ((rc = direct_residual(c, last, false, false, s))) return rc;
    if (n_steps > 0)
      HIP_TRY(c, hipMemcpyAsync(c->records_host + (size_t)(n_steps - 1) * PRMF_OBJ_STRIDE, last, PRMF_OBJ_STRIDE * sizeof(double),
                                hipMemcpyDeviceToHost, s));
    if ((rc = sync_stream(c, s))) return rc;
  }
  if ((c->flags & PRMF_FLAG_TIMING) && c->m > 0) {
    c->ms_xv = c->ms_xtu = 0;
    c->timed_iters = 0;
    for (int i = 0; i < c->cursor; ++i) {
      float a = 0, b = 0;
      HIP_TRY(c, hipEventElapsedTime(&a, c->ev[(size_t)i * 4 + 0], c->ev[(size_t)i * 4 + 1]));
      HIP_TRY(c, hipEventElapsedTime(&b, c->ev[(size_t)i * 4 + 2], c->ev[(size_t)i * 4 + 3]));
      c->ms_xv += a;
      c->ms_xtu += b;
      c->timed_iters++;
    }
  }
  if (obj_out && n_steps > 0) std::memcpy(obj_out, c->records_host, (size_t)n_steps * PRMF_OBJ_STRIDE * sizeof(double));
  if (gamma_delta_out) {
    gamma_delta_out[0] = c->hyper_host[0];
    gamma_delta_out[1] = c->hyper_host[1];
  }
  for (int i = 0; i < n_steps; ++i)
    for (int j = 0; j < 7; ++j)
      if (!std::isfinite(c->records_host[i * PRMF_OBJ_STRIDE + j]))
        return fail(c, PRMF_ERR_NUMERIC, "non-finite objective component %d at iteration %d of this call", j, i);
  return PRMF_OK;
}

int prmf_step(prmf_ctx* c, float* dU, float* dV, int n_steps, double gamma, double delta, double tradeoff,
              double* obj_out, double* gamma_delta_out, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (c->nranks != 1) return fail(c, PRMF_ERR_STATE, "prmf_step is the single-rank form; use prmf_step_dist or the phase calls with nranks > 1");
  if (n_steps < 0) return fail(c, PRMF_ERR_ARG, "n_steps < 0");
  int done = 0;
  int rc;
  while (done < n_steps || (n_steps == 0 && done == 0)) {
    int batch = n_steps - done < MAX_RECORDS ? n_steps - done : MAX_RECORDS;
    if ((rc = prmf_begin(c, dU, dV, gamma, delta, tradeoff, stream))) return rc;
    for (int i = 0; i < batch; ++i) {
      if ((rc = prmf_phase_u(c, stream))) return rc;
      if ((rc = prmf_phase_q(c, stream))) return rc;
      if ((rc = prmf_phase_v(c, stream))) return rc;
      if ((rc = prmf_phase_post(c, stream))) return rc;
    }
    double gd[2] = {gamma, delta};
    if ((rc = prmf_end(c, batch, obj_out ? obj_out + (size_t)done * PRMF_OBJ_STRIDE : nullptr, gd, stream))) return rc;
    gamma = gd[0];
    delta = gd[1];
    done += batch;
    if (n_steps == 0) break;
  }
  if (gamma_delta_out) {
    gamma_delta_out[0] = gamma;
    gamma_delta_out[1] = delta;
  }
  return PRMF_OK;
}

int prmf_comm_unique_id(void* id_out) {
  if (!id_out) return fail(nullptr, PRMF_ERR_ARG, "id_out is NULL");
  if (!rccl().ok) return fail(nullptr, PRMF_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror());
  static_assert(sizeof(ncclUniqueId) == PRMF_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, PRMF_ERR_HIP, "ncclGetUniqueId: %s", rccl().GetErrorString(r));
  std::memcpy(id_out, &id, sizeof id);
  return PRMF_OK;
}

int prmf_comm_init(prmf_ctx* c, const void* id_in) {
  if (!c || !id_in) return PRMF_ERR_ARG;
  if (!rccl().ok) return fail(c, PRMF_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
  if (c->comm) return fail(c, PRMF_ERR_STATE, "communicator already initialised");
  int rc;
  if ((rc = set_device(c))) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof id);
  if (c->comm_broken) return fail(c, PRMF_ERR_STATE, "this context's communicator was aborted after a failed collective; create a new context");
  ncclResult_t r = rccl().CommInitRank(&c->comm, c->nranks, id, c->rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(c, PRMF_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", c->rank, c->nranks, rccl().GetErrorString(r));
  }
  // the global sample count, once: the direct-residual threshold (direct_below) must be the same number on every rank
  if (c->nranks > 1) {
    const double mine = (double)c->m;
    HIP_TRY(c, hipMemcpy(c->resid_sq + 1, &mine, sizeof(double), hipMemcpyHostToDevice));
    NCCL_TRY(c, rccl().AllReduce(c->resid_sq + 1, c->resid_sq + 1, 1, ncclFloat64, ncclSum, c->comm, (hipStream_t) nullptr));
    if ((rc = sync_stream(c, (hipStream_t) nullptr))) return rc;
    double all = 0.0;
    HIP_TRY(c, hipMemcpy(&all, c->resid_sq + 1, sizeof(double), hipMemcpyDeviceToHost));
    c->m_global = (int64_t)(all + 0.5);
  }
  return PRMF_OK;
}

int prmf_set_global_rows(prmf_ctx* c, int64_t m_global) {
  if (!c) return PRMF_ERR_ARG;
  if (m_global < c->m) return fail(c, PRMF_ERR_ARG, "m_global=%lld < this rank's %lld rows", (long long)m_global, (long long)c->m);
  c->m_global = m_global;
  return PRMF_OK;
}

int prmf_set_comm_timeout(prmf_ctx* c, double seconds) {
  if (!c || !(seconds >= 0.0)) return PRMF_ERR_ARG;
  c->comm_timeout_s = seconds;
  return PRMF_OK;
}

int prmf_step_dist(prmf_ctx* c, float* dU, float* dV, int n_steps, double gamma, double delta, double tradeoff,
                   double* obj_out, double* gamma_delta_out, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  if (c->comm_broken) return fail(c, PRMF_ERR_STATE, "the communicator was aborted after a failed collective (rank %d); create a new context", c->rank);
  if (!c->comm) return fail(c, PRMF_ERR_STATE, "prmf_comm_init has not been called");
  if (n_steps < 0) return fail(c, PRMF_ERR_ARG, "n_steps < 0");
  hipStream_t s = (hipStream_t)stream;
  const RcclApi& N = rccl();
  const bool grouped = N.GroupStart && N.GroupEnd;
  int done = 0, rc;
  struct Internal {
    prmf_ctx* c;
    explicit Internal(prmf_ctx* c_) : c(c_) { c->dist_driver = true; }
    ~Internal() { c->dist_driver = false; }
  } internal(c);
  while (done < n_steps) {
    const int batch = n_steps - done < MAX_RECORDS ? n_steps - done : MAX_RECORDS;
    if ((rc = prmf_begin(c, dU, dV, gamma, delta, tradeoff, stream))) return rc;
    const bool ptime = (c->flags & PRMF_FLAG_TIMING) != 0 && !c->ev_phase.empty();
    for (int i = 0; i < batch; ++i) {
      const size_t pe = (size_t)c->cursor * 5;
      if (ptime) HIP_TRY(c, hipEventRecord(c->ev_phase[pe + 0], s));
      if ((rc = prmf_phase_u(c, stream))) return rc;
      if (ptime) HIP_TRY(c, hipEventRecord(c->ev_phase[pe + 1], s));
      if ((rc = prmf_phase_q(c, stream))) return rc;
      // Exchange 1, ONE RCCL group (one launch): the k x k all-reduce of U^T U (reduced locally by the X^T U sweep's
      // prologue; first read by the V update) together with the exchange of Q -- reduce-scatter by owner column block,
      // or all-reduce in place in the replicated mode.  The latency-bound 32 KB message rides with the bandwidth-bound
      // one instead of waiting on a side stream for the persistent sweep grid to leave it a CU
      // (profiles/r03_side_stream_cosched.txt).
      // A member that fails must not leave the group open (ncclGroupEnd is still called) before the communicator is aborted.
      {
        const char* what = "ncclGroupStart";
        ncclResult_t r = grouped ? N.GroupStart() : ncclSuccess;
        const bool opened = grouped && r == ncclSuccess;
        if (r == ncclSuccess) {
          what = "ncclAllReduce(U^T U)";
          r = N.AllReduce(c->gram_buf, c->gram_buf, (size_t)c->k * c->k, ncclFloat64, ncclSum, c->comm, s);
        }
        if (r == ncclSuccess && c->replicated) {
          what = "ncclAllReduce(X^T U)";
          r = N.AllReduce(c->q_send, c->q_send, (size_t)c->n * c->k, ncclFloat32, ncclSum, c->comm, s);
        } else if (r == ncclSuccess && (c->nranks > 1 || c->q_recv != c->q_send)) {
          what = "ncclReduceScatter(X^T U)";
          r = N.ReduceScatter(c->q_send, c->q_recv, (size_t)c->n * c->kc, ncclFloat32, ncclSum, c->comm, s);
        }
        if (opened) {
          const ncclResult_t re = N.GroupEnd();
          if (r == ncclSuccess && re != ncclSuccess) {
            what = "ncclGroupEnd";
            r = re;
          }
        }
        if (r != ncclSuccess) return comm_fail(c, what, N.GetErrorString(r));
      }
      if (ptime) HIP_TRY(c, hipEventRecord(c->ev_phase[pe + 2], s));
      if ((rc = prmf_phase_v(c, stream))) return rc;
      // Exchange 2 (column-sharded mode): all-gather of the updated own columns (+ their sums in the tail)
      if (!c->replicated && (c->nranks > 1 || c->v_recv != c->v_send))
        NCCL_TRY(c, N.AllGather(c->v_send, c->v_recv, (size_t)c->v_send_floats, ncclFloat32, c->comm, s));
      if (ptime) HIP_TRY(c, hipEventRecord(c->ev_phase[pe + 3], s));
      if ((rc = prmf_phase_post(c, stream))) return rc;
      if (ptime) HIP_TRY(c, hipEventRecord(c->ev_phase[pe + 4], s));
    }
    double gd[2] = {gamma, delta};
    if ((rc = prmf_end(c, batch, obj_out ? obj_out + (size_t)done * PRMF_OBJ_STRIDE : nullptr, gd, stream))) return rc;
    if (ptime) {                         // prmf_end has synchronised the stream
      for (int j = 0; j < 4; ++j) c->ms_phase[j] = 0;
      for (int i = 0; i < batch; ++i)
        for (int j = 0; j < 4; ++j) {
          float ms = 0;
          HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_phase[(size_t)i * 5 + j], c->ev_phase[(size_t)i * 5 + j + 1]));
          c->ms_phase[j] += ms;
        }
      c->phase_timed_iters = batch;
    }
    gamma = gd[0];
    delta = gd[1];
    done += batch;
  }
  if (gamma_delta_out) {
    gamma_delta_out[0] = gamma;
    gamma_delta_out[1] = delta;
  }
  return PRMF_OK;
}

int prmf_comm_info(const prmf_ctx* c, prmf_comm_info_t* out) {
  if (!c || !out) return PRMF_ERR_ARG;
  out->in_library = c->comm ? 1 : 0;
  out->comm_nranks = out->comm_rank = -1;
  if (c->comm && rccl().ok) {
    if (rccl().CommCount) (void)rccl().CommCount(c->comm, &out->comm_nranks);
    if (rccl().CommUserRank) (void)rccl().CommUserRank(c->comm, &out->comm_rank);
  }
  out->mode = c->replicated ? PRMF_DIST_REPLICATED : PRMF_DIST_COLUMN_SHARDED;
  for (int j = 0; j < 4; ++j) out->ms_phase[j] = c->phase_timed_iters > 0 ? c->ms_phase[j] / c->phase_timed_iters : -1.0;
  return PRMF_OK;
}

int prmf_score_candidates(prmf_ctx* c, const float* dV, double* scores_out, void* stream) {
  if (!c || !dV || !scores_out) return PRMF_ERR_ARG;
  if (!c->have_lib) return fail(c, PRMF_ERR_STATE, "prmf_set_pathway_library has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t need = (int64_t)c->k * c->lib.P;
  if (need > c->scores_cap) {
    HIP_TRY(c, hipStreamSynchronize(s));
    dev_free(c->scores_dev);
    c->scores_dev = nullptr;
    if ((rc = dev_alloc(c, &c->scores_dev, (size_t)need))) return rc;
    c->scores_cap = need;
  }
  HIP_TRY(c, launch_score_candidates(c->lib, dV, c->n, c->k, c->colnorm, c->scores_dev, s));
  HIP_TRY(c, hipMemcpyAsync(scores_out, c->scores_dev, (size_t)need * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  return PRMF_OK;
}

int prmf_match_terms(prmf_ctx* c, const float* dV, double* quad_out, double* ignore_out, void* stream) {
  if (!c || !dV || !quad_out || !ignore_out) return PRMF_ERR_ARG;
  if (!c->have_lib) return fail(c, PRMF_ERR_STATE, "prmf_set_pathway_library has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t need = (int64_t)c->k * c->lib.P;
  if (2 * need > c->scores_cap) {
    HIP_TRY(c, hipStreamSynchronize(s));
    dev_free(c->scores_dev);
    c->scores_dev = nullptr;
    if ((rc = dev_alloc(c, &c->scores_dev, (size_t)(2 * need)))) return rc;
    c->scores_cap = 2 * need;
  }
  HIP_TRY(c, launch_match_terms(c->lib, dV, c->k, c->scores_dev, c->scores_dev + need, s));
  HIP_TRY(c, hipMemcpyAsync(quad_out, c->scores_dev, (size_t)need * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(ignore_out, c->scores_dev + need, (size_t)need * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  return PRMF_OK;
}

int prmf_cv_error(prmf_ctx* c, const float* dV, double* err_out, void* stream) {
  if (!c || !dV || !err_out) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  if (c->m <= 0) return PRMF_OK;
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  // one scratch block: H [m][k] | row norms^2 [m] | errors [m] | G [k][k] | Gram partials | failure counter
  const size_t kk2 = (size_t)c->k * c->k;
  const size_t nb = (size_t)gram_f64_blocks(c->n);
  double* buf = nullptr;
  struct Guard {
    double*& p;
    ~Guard() { dev_free(p); }
  } guard{buf};
  if ((rc = dev_alloc(c, &buf, (size_t)c->m * c->k + 2 * (size_t)c->m + kk2 + nb * kk2 + 1, false))) return rc;
  double* H = buf;
  double* xx = H + (size_t)c->m * c->k;
  double* err_dev = xx + c->m;
  double* G = err_dev + c->m;
  double* partial = G + kk2;
  int* fail_dev = reinterpret_cast<int*>(partial + nb * kk2);
  HIP_TRY(c, hipMemsetAsync(fail_dev, 0, sizeof(double), s));
  HIP_TRY(c, launch_gram_f64(dV, c->n, c->k, partial, G, s));
  HIP_TRY(c, launch_xv_f64(c->x_dtype, c->X, c->ldx, c->m, c->n, dV, c->k, H, s));
  HIP_TRY(c, launch_row_normsq(c->x_dtype, c->X, c->ldx, c->m, c->n, xx, s));
  HIP_TRY(c, launch_nnls(G, H, xx, c->m, c->k, c->n, err_dev, fail_dev, s));
  int nfail = 0;
  HIP_TRY(c, hipMemcpyAsync(err_out, err_dev, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(&nfail, fail_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (nfail > 0) return fail(c, PRMF_ERR_NUMERIC, "nnls: %d of %lld samples did not converge in %d iterations", nfail, (long long)c->m, 3 * c->k);
  return PRMF_OK;
}

/* validation / measurement hooks: include/prmf_hip_debug.h */
int prmf_debug_slab_sum(prmf_ctx* c, int which, float* out_dev, void* stream) {
  if (!c || !out_dev || (which != 0 && which != 1)) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (which == 0)
    HIP_TRY(c, launch_slab_sum_blocks(c->slabs, c->nseg_xv, c->plan_xv.panel_rows, c->slab_stride_p, c->KP, c->m, c->k, c->k, 1, out_dev, c->plan_xv.max_seg, s));
  else
    HIP_TRY(c, launch_slab_sum_blocks(c->slabs, c->nseg_xtu, c->plan_xtu.panel_rows, c->slab_stride_q, c->KP, c->n, c->k, c->k, 1, out_dev, c->plan_xtu.max_seg, s));
  return PRMF_OK;
}

int prmf_debug_set_stamps(void* dev_buf, size_t n_entries) {
  if (dev_buf && n_entries < (size_t)prmf::STAMP_ENTRIES) return PRMF_ERR_ARG;      // the kernels index fixed regions of the buffer
  prmf::g_sweep_stamps = (unsigned long long*)dev_buf;
  return 0;
}
int prmf_debug_stamp_layout(int* where_offset, int* phase_offset, int* max_workgroups) {
  if (where_offset) *where_offset = prmf::STAMP_WHERE;
  if (phase_offset) *phase_offset = prmf::STAMP_PHASE;
  if (max_workgroups) *max_workgroups = prmf::STAMP_MAX_WGS;
  return prmf::STAMP_ENTRIES;
}
int prmf_debug_set_epi_stamps(void* dev_buf) { prmf::g_epi_stamps = (unsigned long long*)dev_buf; return 0; }

int prmf_debug_recon_terms(prmf_ctx* c, int it, double* cross, double* gu, double* gv) {
  if (!c || !cross || !gu || !gv || it < 0 || it >= c->cursor) return PRMF_ERR_ARG;
  if (!c->lazy || c->batch_open) return fail(c, PRMF_ERR_STATE, "no finished deferred-objective batch");
  int rc;
  if ((rc = set_device(c))) return rc;
  const size_t kk2 = (size_t)c->k * c->k;
  HIP_TRY(c, hipMemcpy(cross, c->cross_ring + it, sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(gu, c->Gu_keep + (size_t)it * kk2, kk2 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(gv, c->Gv_keep + (size_t)it * kk2, kk2 * sizeof(double), hipMemcpyDeviceToHost));
  return PRMF_OK;
}

// ---------------------------------------------------------------------------
// NNLS against the data matrix (nnls_matrix.hip): Lawson-Hanson, passive-set algebra on the host
// ---------------------------------------------------------------------------
namespace {
struct NnlsDev {
  double *b = nullptr, *r = nullptr, *q = nullptr, *w = nullptr, *s = nullptr, *out = nullptr, *wabs = nullptr, *babs = nullptr, *scratch = nullptr;
  unsigned char* state = nullptr;
  long long *P = nullptr, *arg = nullptr;
  ~NnlsDev() {
    (void)hipFree(b); (void)hipFree(r); (void)hipFree(q); (void)hipFree(w); (void)hipFree(s); (void)hipFree(out); (void)hipFree(wabs); (void)hipFree(babs); (void)hipFree(scratch);
    (void)hipFree(state); (void)hipFree(P); (void)hipFree(arg);
  }
};

// G and its Cholesky factor L as lower triangles, one row per passive unknown (row i holds i + 1 entries): storage follows the
// passive set, not the number of unknowns
typedef std::vector<std::vector<double>> Tri;

// row i of the factor from row i of G (rows before i are final); false if column i is numerically dependent on them
bool chol_row(const Tri& G, Tri& L, int i) {
  L[(size_t)i].assign((size_t)i + 1, 0.0);
  double ss = 0.0;
  for (int j = 0; j < i; ++j) {
    double v = G[(size_t)i][(size_t)j];
    for (int t = 0; t < j; ++t) v -= L[(size_t)i][(size_t)t] * L[(size_t)j][(size_t)t];
    v /= L[(size_t)j][(size_t)j];
    L[(size_t)i][(size_t)j] = v;
    ss += v * v;
  }
  const double d2 = G[(size_t)i][(size_t)i] - ss;
  if (!(d2 > 1e-13 * G[(size_t)i][(size_t)i])) return false;
  L[(size_t)i][(size_t)i] = std::sqrt(d2);
  return true;
}

void chol_solve(const Tri& L, int np, const double* rhs, double* sol) {
  for (int i = 0; i < np; ++i) {
    double v = rhs[i];
    for (int t = 0; t < i; ++t) v -= L[(size_t)i][(size_t)t] * sol[t];
    sol[i] = v / L[(size_t)i][(size_t)i];
  }
  for (int i = np - 1; i >= 0; --i) {
    double v = sol[i];
    for (int t = i + 1; t < np; ++t) v -= L[(size_t)t][(size_t)i] * sol[t];
    sol[i] = v / L[(size_t)i][(size_t)i];
  }
}
}  // namespace

int prmf_nnls(int device, const double* dX, int64_t rows, int64_t cols, int64_t ld, int transposed, const double* b, double* x,
              double* rnorm_out, int* iterations_out, int max_iter, void* stream) {
  if (rows < 1 || cols < 1 || ld < cols || !dX || !b || !x) return fail(nullptr, PRMF_ERR_ARG, "prmf_nnls: bad arguments");
  transposed = transposed ? 1 : 0;
  const int64_t nvec = transposed ? rows : cols, len = transposed ? cols : rows;
  const int64_t st_t = transposed ? 1 : ld, st_j = transposed ? ld : 1;
  if (nvec > ((int64_t)1 << 30) || len > ((int64_t)1 << 30)) return fail(nullptr, PRMF_ERR_UNSUPPORTED, "prmf_nnls: more than 2^30 unknowns or equations");
  const int cap = (int)((nvec < len ? nvec : len) + 1);          // a passive set holds linearly independent columns
  if (max_iter <= 0) max_iter = (int)(3 * nvec < 2000000000 ? 3 * nvec : 2000000000);      // scipy's default
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t s = (hipStream_t)stream;
  for (int64_t j = 0; j < nvec; ++j) x[j] = 0.0;
  if (iterations_out) *iterations_out = 0;
  NnlsDev d;
#define NN_TRY(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "prmf_nnls: %s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)
  NN_TRY(hipMalloc((void**)&d.b, (size_t)len * 8));
  NN_TRY(hipMalloc((void**)&d.r, (size_t)len * 8));
  NN_TRY(hipMalloc((void**)&d.q, (size_t)len * 8));
  NN_TRY(hipMalloc((void**)&d.w, (size_t)nvec * 8));
  NN_TRY(hipMalloc((void**)&d.wabs, (size_t)nvec * 8));
  NN_TRY(hipMalloc((void**)&d.babs, (size_t)nvec * 8));
  NN_TRY(hipMalloc((void**)&d.scratch, (size_t)nvec * 8));
  NN_TRY(hipMalloc((void**)&d.state, (size_t)nvec));
  NN_TRY(hipMalloc((void**)&d.s, (size_t)cap * 8));
  NN_TRY(hipMalloc((void**)&d.out, (size_t)cap * 8));
  NN_TRY(hipMalloc((void**)&d.P, (size_t)cap * 8));
  NN_TRY(hipMalloc((void**)&d.arg, 16));
  NN_TRY(hipMemcpyAsync(d.b, b, (size_t)len * 8, hipMemcpyHostToDevice, s));
  NN_TRY(hipMemcpyAsync(d.r, b, (size_t)len * 8, hipMemcpyHostToDevice, s));          // x = 0: r = b
  NN_TRY(hipMemsetAsync(d.state, 0, (size_t)nvec, s));
  NN_TRY(nnlsm_gradient(dX, ld, rows, cols, transposed != 0, d.b, d.scratch, d.babs, s));      // babs[j] = sum_t |a_tj b_t|, once
  Tri G, L;                                 // lower triangles of A_P^T A_P and of its Cholesky factor, in list order
  std::vector<double> c((size_t)cap), sv((size_t)cap), xs((size_t)cap, 0.0), tmp((size_t)cap), corr((size_t)cap);
  std::vector<long long> P;                 // passive list, in order of entry
  std::vector<long long> rejected;          // refused in this round (state 2)
  int np = 0, iters = 0;
  // An unknown may enter while its gradient a_j . r stands out of the rounding noise of that inner product and of the residual
  // it is taken with (nnlsm_argmax): Lawson-Hanson's "w_j > 0" in floating point, per column, free of row and column scales.
  const double c1 = 8.0 * 2.220446049250313e-16 * std::sqrt((double)len), c2 = 64.0 * 2.220446049250313e-16;
  std::vector<double> rhost((size_t)len);
  double rnorm_now = 0.0;
  bool suspect = false;                     // a refusal in the current round that exact arithmetic cannot produce
  auto residual_norm = [&]() -> int {
    NN_TRY(hipMemcpyAsync(rhost.data(), d.r, (size_t)len * 8, hipMemcpyDeviceToHost, s));
    NN_TRY(hipStreamSynchronize(s));
    double ss = 0.0;
    for (double v : rhost) ss += v * v;
    rnorm_now = std::sqrt(ss);
    return PRMF_OK;
  };
  // s = argmin || A_P s - b || on the current list: Cholesky solve + one refinement step with the residual from A itself
  auto solve_passive = [&]() -> int {
    chol_solve(L, np, c.data(), sv.data());
    NN_TRY(hipMemcpyAsync(d.s, sv.data(), (size_t)np * 8, hipMemcpyHostToDevice, s));
    NN_TRY(nnlsm_residual(dX, st_t, st_j, len, d.P, np, d.s, d.b, d.q, s));
    NN_TRY(nnlsm_dots(dX, st_t, st_j, len, d.P, np, d.q, d.out, s));
    NN_TRY(hipMemcpyAsync(tmp.data(), d.out, (size_t)np * 8, hipMemcpyDeviceToHost, s));
    NN_TRY(hipStreamSynchronize(s));
    chol_solve(L, np, tmp.data(), corr.data());
    for (int a = 0; a < np; ++a) sv[(size_t)a] += corr[(size_t)a];
    return PRMF_OK;
  };
  auto upload_list = [&]() -> int {
    if (np) NN_TRY(hipMemcpyAsync(d.P, P.data(), (size_t)np * 8, hipMemcpyHostToDevice, s));
    return PRMF_OK;
  };
  auto set_state = [&](long long j, unsigned char v) -> int {
    NN_TRY(hipMemcpyAsync(d.state + j, &v, 1, hipMemcpyHostToDevice, s));
    NN_TRY(hipStreamSynchronize(s));          // v is a stack byte
    return PRMF_OK;
  };
  int rc;
  bool need_gradient = true;
  for (;;) {
    if (need_gradient) {
      if ((rc = residual_norm())) return rc;
      NN_TRY(nnlsm_gradient(dX, ld, rows, cols, transposed != 0, d.r, d.w, d.wabs, s));
      need_gradient = false;
    }
    NN_TRY(nnlsm_argmax(d.w, d.state, d.wabs, d.babs, c1, c2, nvec, d.arg, s));
    long long arg[2];
    NN_TRY(hipMemcpyAsync(arg, d.arg, 16, hipMemcpyDeviceToHost, s));
    NN_TRY(hipStreamSynchronize(s));
    double wj;
    std::memcpy(&wj, &arg[1], 8);
    const long long j = arg[0];
    if (j < 0 || !(wj > 0.0) || np >= cap - 1) {
      if (getenv("PRMF_NNLS_DEBUG"))
        std::fprintf(stderr, "prmf_nnls stop: j=%lld wj=%g np=%d cap=%d iters=%d rnorm=%g rejected=%zu suspect=%d\n", j, wj, np, cap, iters,
                     rnorm_now, rejected.size(), (int)suspect);
      break;
    }
    if (++iters > max_iter) return fail(nullptr, PRMF_ERR_NUMERIC, "prmf_nnls: maximum number of iterations reached");
    // column j joins the list: its inner products with the list (one row of A_P^T A_P) and with b
    NN_TRY(nnlsm_column(dX, st_t, st_j, len, j, d.q, s));
    P.push_back(j);
    ++np;
    if ((rc = upload_list())) return rc;
    NN_TRY(nnlsm_dots(dX, st_t, st_j, len, d.P, np, d.q, d.out, s));
    NN_TRY(hipMemcpyAsync(tmp.data(), d.out, (size_t)np * 8, hipMemcpyDeviceToHost, s));
    NN_TRY(nnlsm_dots(dX, st_t, st_j, len, d.P + (np - 1), 1, d.b, d.out + cap - 1, s));
    NN_TRY(hipMemcpyAsync(&c[(size_t)np - 1], d.out + cap - 1, 8, hipMemcpyDeviceToHost, s));
    NN_TRY(hipStreamSynchronize(s));
    G.resize((size_t)np);
    L.resize((size_t)np);
    G[(size_t)np - 1].assign(tmp.begin(), tmp.begin() + np);
    bool ok = chol_row(G, L, np - 1);
    if (ok) {
      xs[(size_t)np - 1] = 0.0;
      if ((rc = solve_passive())) return rc;
      ok = sv[(size_t)np - 1] > 0.0;                  // Lawson-Hanson: the entering unknown must come out positive
    }
    if (!ok) {                                        // refuse it for this round and look at the next largest gradient
      // In exact arithmetic a column with a positive gradient is independent of the list and comes out positive; a refusal of a
      // column whose gradient is a real descent direction (cosine with the residual above 1e-6) means the Gram factor has lost
      // the problem (conditioning beyond what normal equations carry): the answer would be wrong, so there is none.
      const double aj = std::sqrt(tmp[(size_t)np - 1] > 0.0 ? tmp[(size_t)np - 1] : 0.0);
      if (aj > 0.0 && rnorm_now > 0.0 && wj > 1e-6 * aj * rnorm_now) suspect = true;
      P.pop_back();
      --np;
      G.resize((size_t)np);
      L.resize((size_t)np);
      rejected.push_back(j);
      if ((rc = set_state(j, 2))) return rc;
      if ((rc = upload_list())) return rc;
      continue;
    }
    if ((rc = set_state(j, 1))) return rc;
    // feasibility: walk from the current point towards the unconstrained solution until the first unknown reaches zero
    for (;;) {
      int worst = -1;
      double alpha = 2.0;
      for (int a = 0; a < np; ++a)
        if (sv[(size_t)a] <= 0.0) {
          const double t = xs[(size_t)a] / (xs[(size_t)a] - sv[(size_t)a]);
          if (t < alpha) {
            alpha = t;
            worst = a;
          }
        }
      if (worst < 0) break;
      if (++iters > max_iter) return fail(nullptr, PRMF_ERR_NUMERIC, "prmf_nnls: maximum number of iterations reached");
      for (int a = 0; a < np; ++a) xs[(size_t)a] += alpha * (sv[(size_t)a] - xs[(size_t)a]);
      xs[(size_t)worst] = 0.0;
      // every unknown that reached zero leaves the list; the Gram block, c and the point are compacted in list order
      // (sources lie at or after their targets in both coordinates, so the in-place copy is safe)
      std::vector<int> keep;
      for (int a = 0; a < np; ++a) {
        if (xs[(size_t)a] > 0.0) keep.push_back(a);
        else if ((rc = set_state(P[(size_t)a], 0))) return rc;
      }
      const int kept = (int)keep.size();
      for (int a = 0; a < kept; ++a) {
        std::vector<double> row((size_t)a + 1);
        for (int t = 0; t <= a; ++t) row[(size_t)t] = G[(size_t)keep[(size_t)a]][(size_t)keep[(size_t)t]];      // keep is increasing
        G[(size_t)a].swap(row);
        P[(size_t)a] = P[(size_t)keep[(size_t)a]];
        xs[(size_t)a] = xs[(size_t)keep[(size_t)a]];
        c[(size_t)a] = c[(size_t)keep[(size_t)a]];
      }
      G.resize((size_t)kept);
      L.resize((size_t)kept);
      P.resize((size_t)kept);
      np = kept;
      if ((rc = upload_list())) return rc;
      if (np == 0) break;
      for (int a = 0; a < np; ++a)
        if (!chol_row(G, L, a)) return fail(nullptr, PRMF_ERR_DECLINED, "prmf_nnls: ill-conditioned passive set (the passive columns became dependent to 1e-13); not solved");
      if ((rc = solve_passive())) return rc;
    }
    for (int a = 0; a < np; ++a) xs[(size_t)a] = sv[(size_t)a];
    // the new point: its residual, and a fresh gradient; refusals of this round are forgotten
    if (np) NN_TRY(hipMemcpyAsync(d.s, xs.data(), (size_t)np * 8, hipMemcpyHostToDevice, s));
    NN_TRY(nnlsm_residual(dX, st_t, st_j, len, d.P, np, d.s, d.b, d.r, s));
    for (long long rj : rejected)
      if ((rc = set_state(rj, 0))) return rc;
    rejected.clear();
    suspect = false;                        // the round found another way down: the refused column gets its chance again
    need_gradient = true;
  }
  if (suspect)
    return fail(nullptr, PRMF_ERR_DECLINED, "prmf_nnls: ill-conditioned passive set (a descent direction was refused); not solved");
  // "No free unknown descends" is only as good as the inner products that say so.  Where the residual has become so small next to b
  // AS A COLUMN SEES IT (sum |a_tj b_t| against sum |a_tj r_t|: rows of wildly different scales) that a gradient below 1e-8 of its
  // own magnitude could not be told from rounding, an orthogonal factorisation can still make progress and this method cannot know:
  // it declines.  (Well-scaled data: the ratio is ||b|| / ||r||, and the test only fires for fits exact to 1e-6.)
  {
    double bb = 0.0;
    for (int64_t t = 0; t < len; ++t) bb += b[t] * b[t];
    if ((rc = residual_norm())) return rc;
    if (rnorm_now > 1e-10 * std::sqrt(bb)) {
      std::vector<double> wa((size_t)nvec), ba((size_t)nvec);
      std::vector<unsigned char> st((size_t)nvec);
      NN_TRY(nnlsm_gradient(dX, ld, rows, cols, transposed != 0, d.r, d.w, d.wabs, s));
      NN_TRY(hipMemcpyAsync(wa.data(), d.wabs, (size_t)nvec * 8, hipMemcpyDeviceToHost, s));
      NN_TRY(hipMemcpyAsync(ba.data(), d.babs, (size_t)nvec * 8, hipMemcpyDeviceToHost, s));
      NN_TRY(hipMemcpyAsync(st.data(), d.state, (size_t)nvec, hipMemcpyDeviceToHost, s));
      NN_TRY(hipStreamSynchronize(s));
      for (int64_t jj = 0; jj < nvec; ++jj)
        if (st[(size_t)jj] != 1 && ba[(size_t)jj] > 0.0 && c2 * ba[(size_t)jj] > 1e-8 * wa[(size_t)jj])
          return fail(nullptr, PRMF_ERR_DECLINED, "prmf_nnls: ill-conditioned problem (the gradient of unknown %lld cannot be resolved to 1e-8); not solved", (long long)jj);
    }
  }
  for (int a = 0; a < np; ++a) x[P[(size_t)a]] = xs[(size_t)a];
  if (rnorm_out) {
    NN_TRY(hipMemcpyAsync(rhost.data(), d.r, (size_t)len * 8, hipMemcpyDeviceToHost, s));
    NN_TRY(hipStreamSynchronize(s));
    double ss = 0.0;
    for (double v : rhost) ss += v * v;
    *rnorm_out = std::sqrt(ss);
  }
  if (iterations_out) *iterations_out = iters;
#undef NN_TRY
  return PRMF_OK;
}

int prmf_sum_squares(int device, const void* dX, int x_dtype, int64_t m, int64_t n, int64_t ld, double* out, void* stream) {
  if (!out || m < 0 || n < 0 || ld < n || (m > 0 && n > 0 && !dX) || (x_dtype != PRMF_X_F32 && x_dtype != PRMF_X_BF16))
    return fail(nullptr, PRMF_ERR_ARG, "prmf_sum_squares: bad arguments");
  *out = 0.0;
  if (m == 0 || n == 0) return PRMF_OK;
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t s = (hipStream_t)stream;
  // the summation tree of the layout pass (setx.hip): the same bits as the norm prmf_set_x_device / _host report
  const int64_t np = setx_partial_doubles(m, n), tr = (m + 63) / 64;
  double* scratch = nullptr;
  double total = 0.0;
  hipError_t e = hipMalloc((void**)&scratch, (size_t)(np + tr + 1) * sizeof(double));
  if (e == hipSuccess) e = launch_setx_rows(x_dtype, x_dtype, dX, ld, m, n, nullptr, 0, nullptr, 0, scratch, scratch + np, s);
  if (e == hipSuccess) e = launch_setx_total(scratch + np, tr, scratch + np + tr, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&total, scratch + np + tr, sizeof(double), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(scratch);
  if (e != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "prmf_sum_squares: %s", hipGetErrorString(e));
  *out = total;
  return PRMF_OK;
}

int prmf_quantile_transform(int device, double* dX, int64_t m, int64_t n, int64_t ld, const int64_t* fit_rows,
                            int64_t n_fit, int n_quantiles, void* stream) {
  if (m < 0 || n < 0 || (m > 0 && n > 0 && !dX) || ld < n || n_quantiles < 1)
    return fail(nullptr, PRMF_ERR_ARG, "prmf_quantile_transform: bad arguments");
  if (m == 0 || n == 0) return PRMF_OK;
  if (!fit_rows) n_fit = m;
  if (n_fit < 1 || n_fit > m) return fail(nullptr, PRMF_ERR_ARG, "prmf_quantile_transform: n_fit=%lld outside [1, m]", (long long)n_fit);
  if (fit_rows)
    for (int64_t i = 0; i < n_fit; ++i)
      if (fit_rows[i] < 0 || fit_rows[i] >= m) return fail(nullptr, PRMF_ERR_ARG, "prmf_quantile_transform: fit row outside [0, m)");
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t s = (hipStream_t)stream;
  const int nq = (int)(n_quantiles < n_fit ? n_quantiles : n_fit);
  // references = np.linspace(0, 1, nq): i * (1 / (nq - 1)), last element exactly 1; percentile's q = (references * 100) / 100
  std::vector<double> host((size_t)2 * nq);
  const double step = nq > 1 ? 1.0 / (double)(nq - 1) : 0.0;
  for (int i = 0; i < nq; ++i) host[(size_t)i] = (double)i * step;
  if (nq > 1) host[(size_t)nq - 1] = 1.0;
  for (int i = 0; i < nq; ++i) host[(size_t)nq + i] = (host[(size_t)i] * 100.0) / 100.0;
  double* dref = nullptr;
  int64_t* drows = nullptr;
  hipError_t e = hipMalloc((void**)&dref, host.size() * 8);
  if (e == hipSuccess && fit_rows) e = hipMalloc((void**)&drows, (size_t)n_fit * 8);
  if (e == hipSuccess) e = hipMemcpyAsync(dref, host.data(), host.size() * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && fit_rows) e = hipMemcpyAsync(drows, fit_rows, (size_t)n_fit * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = quantile_transform_device(dX, m, n, ld, drows, n_fit, dref, dref + nq, nq, s);
  (void)hipFree(dref);
  (void)hipFree(drows);
  if (e != hipSuccess) return fail(nullptr, PRMF_ERR_HIP, "prmf_quantile_transform: %s", hipGetErrorString(e));
  return PRMF_OK;
}

// ---------------------------------------------------------------------------
// delimited text -> float64 matrix (textio.hip)
// ---------------------------------------------------------------------------
struct prmf_text {
  int device = 0;
  unsigned char* text = nullptr;      // capacity + 1 bytes (room for a final '\n')
  int64_t capacity = 0, filled = 0;
  int last_byte = -1;                 // of the text appended so far
  int64_t* sep_pos = nullptr;
  int64_t fields_per_line = 0, rows = 0;
  bool indexed = false;
};

namespace {
int text_fail(prmf_text* t, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_create_error, 512, fmt, ap);      // no context: prmf_last_error(NULL), as for prmf_create
  va_end(ap);
  (void)t;
  return code;
}
#define TEXT_TRY(t, expr)                                                                                \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));     \
  } while (0)
}  // namespace

int prmf_text_create(prmf_text** out, int device, int64_t nbytes) {
  if (!out) return PRMF_ERR_ARG;
  *out = nullptr;
  if (nbytes < 0 || nbytes > ((int64_t)1 << 40)) return text_fail(nullptr, PRMF_ERR_ARG, "prmf_text_create: nbytes=%lld outside [0, 2^40]", (long long)nbytes);
  if (hipSetDevice(device) != hipSuccess) return text_fail(nullptr, PRMF_ERR_HIP, "hipSetDevice(%d) failed", device);
  prmf_text* t = new (std::nothrow) prmf_text();
  if (!t) return text_fail(nullptr, PRMF_ERR_HIP, "out of host memory");
  t->device = device;
  t->capacity = nbytes;
  hipError_t e = hipMalloc((void**)&t->text, (size_t)nbytes + 16);
  if (e != hipSuccess) {
    delete t;
    return text_fail(nullptr, PRMF_ERR_HIP, "hipMalloc(%lld bytes of text): %s", (long long)nbytes + 16, hipGetErrorString(e));
  }
  *out = t;
  return PRMF_OK;
}

int prmf_text_append(prmf_text* t, const void* bytes, int64_t nbytes) {
  if (!t || nbytes < 0 || (nbytes > 0 && !bytes)) return PRMF_ERR_ARG;
  if (t->indexed) return text_fail(t, PRMF_ERR_STATE, "prmf_text_append after prmf_text_index");
  if (t->filled + nbytes > t->capacity)
    return text_fail(t, PRMF_ERR_ARG, "prmf_text_append: %lld + %lld bytes exceed the %lld the text was created for", (long long)t->filled,
                     (long long)nbytes, (long long)t->capacity);
  if (nbytes == 0) return PRMF_OK;
  if (hipSetDevice(t->device) != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "hipSetDevice(%d) failed", t->device);
  TEXT_TRY(t, hipMemcpy(t->text + t->filled, bytes, (size_t)nbytes, hipMemcpyHostToDevice));
  t->filled += nbytes;
  t->last_byte = ((const unsigned char*)bytes)[nbytes - 1];
  return PRMF_OK;
}

int prmf_text_index(prmf_text* t, char delimiter, int64_t fields_per_line, int64_t* rows_out, int* flags_out) {
  if (!t || !rows_out || !flags_out) return PRMF_ERR_ARG;
  const unsigned char d = (unsigned char)delimiter;
  if (d == 0 || d > 127 || d == '\n' || d == '\r' || d == '"' || d == '.' || d == '+' || d == '-' || d == 'e' || d == 'E' ||
      (d >= '0' && d <= '9'))
    return text_fail(t, PRMF_ERR_ARG, "prmf_text_index: delimiter 0x%02x cannot separate decimal numbers", (unsigned)d);
  if (fields_per_line < 1) return text_fail(t, PRMF_ERR_ARG, "prmf_text_index: fields_per_line < 1");
  if (t->indexed) return text_fail(t, PRMF_ERR_STATE, "prmf_text_index called twice");
  if (hipSetDevice(t->device) != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "hipSetDevice(%d) failed", t->device);
  *rows_out = 0;
  *flags_out = 0;
  t->fields_per_line = fields_per_line;
  if (t->filled == 0) {
    t->indexed = true;
    return PRMF_OK;
  }
  if (t->last_byte != '\n') {                     // the last line ends like the others
    const unsigned char nl = '\n';
    TEXT_TRY(t, hipMemcpy(t->text + t->filled, &nl, 1, hipMemcpyHostToDevice));
    t->filled += 1;
    t->last_byte = '\n';
  }
  hipStream_t s = nullptr;
  const int64_t nb = text_blocks(t->filled);
  int64_t *counts = nullptr, *base = nullptr;
  int* flags = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipMalloc((void**)&counts, (size_t)(nb + 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&base, (size_t)(nb + 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&flags, sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(flags, 0, sizeof(int), s);
  if (e == hipSuccess) e = hipMemsetAsync(counts + nb, 0, 8, s);       // one value more: its exclusive sum is the total
  if (e == hipSuccess) e = text_count(t->text, t->filled, d, counts, s);
  if (e == hipSuccess) e = text_scan(counts, base, nb + 1, nullptr, &tmp_bytes, s);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess) e = text_scan(counts, base, nb + 1, tmp, &tmp_bytes, s);
  int64_t n_sep = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_sep, base + nb, 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  int host_flags = 0;
  if (e == hipSuccess) {
    if (n_sep % fields_per_line != 0) {
      host_flags = PRMF_TEXT_RAGGED;
    } else {
      e = hipMalloc((void**)&t->sep_pos, (size_t)(n_sep ? n_sep : 1) * 8);
      if (e == hipSuccess) e = text_positions(t->text, t->filled, d, base, fields_per_line, t->sep_pos, flags, s);
      if (e == hipSuccess) e = hipMemcpyAsync(&host_flags, flags, sizeof(int), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
  }
  (void)hipFree(counts);
  (void)hipFree(base);
  (void)hipFree(flags);
  (void)hipFree(tmp);
  if (e != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "prmf_text_index: %s", hipGetErrorString(e));
  t->indexed = true;
  t->rows = host_flags ? 0 : n_sep / fields_per_line;
  *rows_out = t->rows;
  *flags_out = host_flags;
  return PRMF_OK;
}

int prmf_text_parse(prmf_text* t, int skip_first_field, double* out, int64_t* name_spans, int32_t* col_is_int, int* flags_out,
                    int64_t* first_bad_offset_out) {
  if (!t || !flags_out) return PRMF_ERR_ARG;
  if (!t->indexed) return text_fail(t, PRMF_ERR_STATE, "prmf_text_parse before prmf_text_index");
  skip_first_field = skip_first_field ? 1 : 0;
  const int64_t width = t->fields_per_line - skip_first_field, rows = t->rows;
  if (width < 1) return text_fail(t, PRMF_ERR_ARG, "prmf_text_parse: no data column (fields_per_line=%lld, skip_first_field=%d)", (long long)t->fields_per_line, skip_first_field);
  if (rows > 0 && (!out || !col_is_int || (skip_first_field && !name_spans))) return text_fail(t, PRMF_ERR_ARG, "prmf_text_parse: NULL output");
  *flags_out = 0;
  if (first_bad_offset_out) *first_bad_offset_out = -1;
  if (rows == 0) return PRMF_OK;
  if (hipSetDevice(t->device) != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "hipSetDevice(%d) failed", t->device);
  hipStream_t s = nullptr;
  double* d_out = nullptr;
  int64_t* d_spans = nullptr;
  int* d_small = nullptr;              // [0] flags, [2..3] first bad offset (8-byte aligned), [4 .. 4 + width) col_not_int
  hipError_t e = hipMalloc((void**)&d_out, (size_t)rows * width * 8);
  if (e == hipSuccess && skip_first_field) e = hipMalloc((void**)&d_spans, (size_t)rows * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&d_small, (size_t)(4 + width) * sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(d_small, 0, (size_t)(4 + width) * sizeof(int), s);
  if (e == hipSuccess) e = hipMemsetAsync(d_small + 2, 0xff, 8, s);
  if (e == hipSuccess)
    e = text_parse(t->text, t->sep_pos, rows * t->fields_per_line, t->fields_per_line, skip_first_field, d_out, d_spans, d_small + 4,
                   d_small, reinterpret_cast<unsigned long long*>(d_small + 2), s);
  std::vector<int> small((size_t)(4 + width));
  if (e == hipSuccess) e = hipMemcpyAsync(small.data(), d_small, small.size() * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess && small[0] == 0) {
    e = hipMemcpy(out, d_out, (size_t)rows * width * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && skip_first_field) e = hipMemcpy(name_spans, d_spans, (size_t)rows * 16, hipMemcpyDeviceToHost);
    for (int64_t c = 0; c < width; ++c) col_is_int[c] = small[(size_t)4 + c] ? 0 : 1;
  }
  (void)hipFree(d_out);
  (void)hipFree(d_spans);
  (void)hipFree(d_small);
  if (e != hipSuccess) return text_fail(t, PRMF_ERR_HIP, "prmf_text_parse: %s", hipGetErrorString(e));
  *flags_out = small[0];
  if (small[0] && first_bad_offset_out) {
    unsigned long long fb;
    std::memcpy(&fb, &small[2], 8);
    *first_bad_offset_out = (int64_t)fb;
  }
  return PRMF_OK;
}

int prmf_write_csv(const char* path, const char* header_line, const char* row_prefixes, const int64_t* row_prefix_offsets,
                   const double* values, int64_t rows, int64_t cols, int64_t ld, char separator, int n_threads) {
  if (!path || rows < 0 || cols < 1 || ld < cols || (rows > 0 && !values) || ((row_prefixes == nullptr) != (row_prefix_offsets == nullptr)))
    return text_fail(nullptr, PRMF_ERR_ARG, "prmf_write_csv: bad arguments");
  if (n_threads < 1) {
    const unsigned hw = std::thread::hardware_concurrency();
    n_threads = hw ? (int)(hw < 16 ? hw : 16) : 1;
  }
  if (rows * cols < 65536) n_threads = 1;
  const int err = text_write_matrix(path, header_line, row_prefixes, row_prefix_offsets, values, rows, cols, ld, separator, n_threads);
  if (err) return text_fail(nullptr, PRMF_ERR_IO, "prmf_write_csv(%s): errno %d: %s", path, err > 0 ? err : 5, err > 0 ? std::strerror(err) : "write failed");
  return PRMF_OK;
}

int prmf_text_destroy(prmf_text* t) {
  if (!t) return PRMF_OK;
  (void)hipSetDevice(t->device);
  (void)hipFree(t->text);
  (void)hipFree(t->sep_pos);
  delete t;
  return PRMF_OK;
}

int prmf_debug_stream(prmf_ctx* c, int which, int reps, double* ms_out, int64_t* bytes_out, void* stream) {
  if (!c || (which != 0 && which != 1) || reps < 1 || !ms_out) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const void* p = which == 0 ? c->X : c->XT;
  const size_t bytes = which == 0 ? c->x_bytes : c->xt_bytes;
  hipEvent_t a = nullptr, b = nullptr;
  HIP_TRY(c, hipEventCreate(&a));
  HIP_TRY(c, hipEventCreate(&b));
  hipError_t e = launch_stream_read(p, bytes, c->n_cus, c->resid_partial, s);      // warm-up (code object, clocks)
  if (e == hipSuccess) e = hipEventRecord(a, s);
  for (int r = 0; r < reps && e == hipSuccess; ++r) e = launch_stream_read(p, bytes, c->n_cus, c->resid_partial, s);
  if (e == hipSuccess) e = hipEventRecord(b, s);
  if (e == hipSuccess) e = hipEventSynchronize(b);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (e != hipSuccess) return fail(c, PRMF_ERR_HIP, "prmf_debug_stream: %s", hipGetErrorString(e));
  *ms_out = (double)ms / reps;
  if (bytes_out) *bytes_out = (int64_t)bytes;
  return PRMF_OK;
}

int prmf_debug_sync_stream(prmf_ctx* c, void* stream) {
  if (!c) return PRMF_ERR_ARG;
  int rc;
  if ((rc = set_device(c))) return rc;
  return sync_stream(c, (hipStream_t)stream);
}

int prmf_debug_set_tail_probe(prmf_ctx* c, int which, int* dev_buf, int tiles_per_group) {
  if (!dev_buf) { prmf::g_tail_probe = nullptr; prmf::g_tail_plan = nullptr; return PRMF_OK; }
  if (!c || (which != 0 && which != 1) || tiles_per_group < 1) return PRMF_ERR_ARG;
  const GemmPlan& p = which == 0 ? c->plan_xv : c->plan_xtu;
  if (p.panels > 4096 || p.nx != 1) return fail(c, PRMF_ERR_UNSUPPORTED, "tail probe: unsliced plans with <= 4096 panels only");
  int rc;
  if ((rc = set_device(c))) return rc;
  std::vector<int> host((size_t)64 + 2 * 4096, 0);
  const int64_t tiles = (int64_t)p.panels * p.panel_rows / 32;
  host[1] = (int)((tiles + tiles_per_group - 1) / tiles_per_group);
  host[2] = tiles_per_group;
  for (int q = 0; q < p.panels; ++q) host[(size_t)64 + 4096 + q] = __builtin_popcountll((unsigned long long)p.segmask[(size_t)q]);
  HIP_TRY(c, hipMemcpy(dev_buf, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(c, hipDeviceSynchronize());
  prmf::g_tail_launch = 0;
  prmf::g_tail_probe = dev_buf;
  prmf::g_tail_plan = &p;
  return PRMF_OK;
}

int prmf_debug_bump_cursor(prmf_ctx* c) {
  if (!c || c->cursor + 1 >= MAX_RECORDS) return PRMF_ERR_ARG;
  c->cursor++;
  return PRMF_OK;
}

int prmf_debug_swap(prmf_ctx* a, prmf_ctx* b, int which) {
  if (!a || !b || a->m != b->m || a->n != b->n || a->k != b->k || a->x_dtype != b->x_dtype) return PRMF_ERR_ARG;
  switch (which) {
    case 0: std::swap(a->X, b->X); break;
    case 1: std::swap(a->XT, b->XT); break;
    case 2: std::swap(a->slabs, b->slabs); break;
    case 3: std::swap(a->Vp, b->Vp); a->prepared = b->prepared = false; break;
    case 4: std::swap(a->Up, b->Up); break;
    default: return PRMF_ERR_ARG;
  }
  return PRMF_OK;
}

int prmf_debug_resident_x(prmf_ctx* c, int which, void* out_host) {
  if (!c || !out_host || (which != 0 && which != 1)) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  HIP_TRY(c, hipDeviceSynchronize());
  const size_t bytes = which == 0 ? c->x_bytes : c->xt_bytes;
  HIP_TRY(c, hipMemcpy(out_host, which == 0 ? c->X : c->XT, bytes, hipMemcpyDeviceToHost));
  return PRMF_OK;
}

int prmf_debug_resident_shape(const prmf_ctx* c, int which, int64_t* rows, int64_t* ld) {
  if (!c || !rows || !ld || (which != 0 && which != 1)) return PRMF_ERR_ARG;
  *rows = which == 0 ? round_up(c->m > 0 ? c->m : 1, M_PAD) : round_up(c->n, M_PAD);
  *ld = which == 0 ? c->Kpad_x : c->Kpad_xt;
  return PRMF_OK;
}

int prmf_residual_sq(prmf_ctx* c, const float* dU, const float* dV, double* out_local, void* stream) {
  if (!c || !dV || !out_local || (c->m > 0 && !dU)) return PRMF_ERR_ARG;
  if (!c->have_x) return fail(c, PRMF_ERR_STATE, "prmf_set_x has not been called");
  int rc;
  if ((rc = set_device(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  *out_local = 0.0;
  if (c->m <= 0) return PRMF_OK;
  int nparts = 0;
  HIP_TRY(c, launch_residual(c->x_dtype, c->X, c->ldx, c->m, c->n, c->k, dU, dV, c->n_cus, c->resid_partial, &nparts, nullptr, s));
  HIP_TRY(c, launch_residual_sum(c->resid_partial, nparts, nullptr, c->resid_sq + 1, nullptr, nullptr, false, s));
  HIP_TRY(c, hipMemcpyAsync(out_local, c->resid_sq + 1, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  return PRMF_OK;
}

int prmf_debug_plan_sweep(int x_dtype, long long rows, long long depth, int k, int n_cus, int* out) {
  if (!out || rows < 0 || depth < 1 || k < 1 || k > 128 || n_cus < 1 || (x_dtype != PRMF_X_F32 && x_dtype != PRMF_X_BF16)) return PRMF_ERR_ARG;
  const int NT = (k + 31) / 32;
  const GemmPlan p = plan_gemm(x_dtype, rows > 0 ? rows : 1, round_up(depth, K_PAD), NT, n_cus, SweepTuning{});
  const int v[8] = {p.nw, p.mt, p.panel_rows, p.grid, p.nx, p.max_seg, (p.bdb ? 1 : 0) | (p.pp ? 2 : 0) | (p.exact ? 4 : 0), p.panels};
  for (int i = 0; i < 8; ++i) out[i] = v[i];
  return PRMF_OK;
}

int prmf_get_sweep_info(const prmf_ctx* c, prmf_sweep_info* out) {
  if (!c || !out) return PRMF_ERR_ARG;
  const int eb = elem_bytes(c->x_dtype);
  out->bytes_x_sweep = round_up(c->m > 0 ? c->m : 1, 128) * c->Kpad_x * eb;
  out->bytes_xt_sweep = round_up(c->n, 128) * c->Kpad_xt * eb;
  out->splits_xv = c->plan_xv.max_seg;
  out->splits_xtu = c->plan_xtu.max_seg;
  out->grid_xv = c->plan_xv.grid;
  out->grid_xtu = c->plan_xtu.grid;
  out->f32_split = c->split_f32 ? 1 : 0;
  out->reserved = (c->plan_xv.exact ? 1 : 0) | (c->plan_xtu.exact ? 2 : 0);      // bit 0 / 1: X V / X^T U is the exact short-reduction product
  return PRMF_OK;
}

int prmf_set_timing(prmf_ctx* c, int enabled) {
  if (!c) return PRMF_ERR_ARG;
  if (enabled && c->ev.empty()) {
    int rc;
    if ((rc = set_device(c))) return rc;
    c->ev.assign((size_t)MAX_RECORDS * 4, nullptr);
    for (auto& ev : c->ev)
      if (hipEventCreate(&ev) != hipSuccess) return fail(c, PRMF_ERR_HIP, "hipEventCreate failed");
    c->ev_phase.assign((size_t)MAX_RECORDS * 5, nullptr);
    for (auto& ev : c->ev_phase)
      if (hipEventCreate(&ev) != hipSuccess) return fail(c, PRMF_ERR_HIP, "hipEventCreate failed");
  }
  if (enabled) c->flags |= PRMF_FLAG_TIMING; else c->flags &= ~PRMF_FLAG_TIMING;
  c->timed_iters = 0;
  return PRMF_OK;
}

int prmf_get_timing(const prmf_ctx* c, double* ms_xv, double* ms_xtu, double* ms_total) {
  if (!c) return PRMF_ERR_ARG;
  if (!(c->flags & PRMF_FLAG_TIMING) || c->timed_iters == 0) return PRMF_ERR_STATE;
  if (ms_xv) *ms_xv = c->ms_xv / c->timed_iters;
  if (ms_xtu) *ms_xtu = c->ms_xtu / c->timed_iters;
  if (ms_total) *ms_total = c->ms_total;
  return PRMF_OK;
}

}  // extern "C"
