/*
 * prmf_hip.h -- C-ABI of libprmf_hip.so: the MI355X (gfx950) implementation of
 * PRMF's alternating multiplicative-update inner loop.
 *
 * The reference (gitter-lab/prmf) has no FFI; its boundary for this path is the
 * Python function
 *     nmf_manifold_vec_update(X, U, V, k_to_W, k_to_D, k_to_L, k_to_feat_inds,
 *                             n_steps, gamma, delta, i, verbose, norm_X)
 *                                             script/prmf_runner.py:374-451
 * (+ its _tradeoff twin :497-554 and the objective :336-372), called once per
 * round of `modulus` iterations by nmf_pathway (:740/:742).  The entry points
 * below are what a ctypes binding for that call needs; INTEGRATION.md shows the
 * stub.  Conventions: plain pointers and sizes, no torch types; every function
 * returns 0 or a negative prmf_status and never throws; no hidden global state
 * (one ctx per device/rank, a ctx is not thread-safe); all device work is
 * enqueued on the caller's HIP stream (`stream` = hipStream_t as void*, NULL =
 * default stream); only the calls documented as synchronising wait for it.
 *
 * Device buffers passed in (U, V, comm buffers) are owned by the caller (the
 * Python host allocates them as torch tensors); everything else is owned by ctx.
 */
#ifndef PRMF_HIP_H
#define PRMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRMF_ABI_VERSION 3   /* 3: additions only (prmf_set_x_device / _host, prmf_set_norm_x_sq, prmf_match_terms) */

typedef struct prmf_ctx prmf_ctx;

enum prmf_status {
  PRMF_OK = 0,
  PRMF_ERR_ARG = -1,         /* bad argument / shape / alignment */
  PRMF_ERR_HIP = -2,         /* a HIP runtime call failed (see prmf_last_error) */
  PRMF_ERR_STATE = -3,       /* call order violated (e.g. step before set_x) */
  PRMF_ERR_NUMERIC = -4,     /* NaN/Inf in the objective: the reference would raise
                                FloatingPointError (np.seterr(divide='raise'), prmf_runner.py:22) */
  PRMF_ERR_UNSUPPORTED = -5,
  PRMF_ERR_IO = -7,          /* prmf_write_csv: the file could not be opened / written; prmf_last_error(NULL) = "...: errno N: text"
                                (the Python host raises OSError(N, ...) as DataFrame.to_csv would) */
  PRMF_ERR_DECLINED = -6     /* prmf_nnls: the problem's conditioning is beyond the method used here; NO answer was written
                                (the caller solves it another way -- the Python host calls scipy, as the reference does) */
};

enum prmf_x_dtype {
  PRMF_X_F32 = 0,            /* X float32: exact (every value split into three bf16 pieces on the bf16 MFMA) */
  PRMF_X_BF16 = 1            /* X bfloat16, bf16 MFMA with hi+lo split factors, fp32 acc */
};

/* objective record written per iteration: the keys of the reference's obj_data
 * dict (prmf_runner.py:363-371) in this order, + recon^2, + how recon was obtained.
 *
 * recon normally comes from the Gram identity ||X||^2 - 2<X^T U, V> + <U^T U, V^T V> (the sweeps' products are there
 * anyway), which is exact to ~1e-8 ||X||^2 at k <= 64 and ~1e-7 ||X||^2 at k > 64 (float32 accumulation of X^T U)
 * -- i.e. recon to that fraction times (||X|| / recon)^2 / 2.  The reference materialises X - U V^T (:337) and never
 * cancels, so when the identity value drops below 2e-4 ||X||^2 (k <= 64: recon / ||X|| < 1.4%) resp. 2e-3 ||X||^2
 * (k > 64: 4.5%; times samples-per-rank / 32768 beyond 32768 samples per rank: the error of that accumulation chain grows
 * with its length) -- plus 0.13 / sqrt(min(m, n) k) of ||X||^2 for small factors, whose 16-bit packing for the sweeps
 * does not average out -- the library evaluates sum (X - U V^T)^2 directly with one more sweep over X: for the LAST record of a
 * prmf_step / prmf_end batch (the one the outer loop's convergence test reads, :772-774).  With the tradeoff feedback
 * on (recon feeds gamma every step, :542-548) or PRMF_FLAG_DIRECT_RESIDUAL set, EVERY record takes the direct residual,
 * whatever its size.  PRMF_OBJ_FLAGS says which:
 *   PRMF_RECON_DIRECT     recon / obj / recon_sq of this record come from the direct residual
 *                         (PRMF_OBJ_RECON_SQ_IDENTITY keeps the identity value)
 *   PRMF_RECON_IMPRECISE  identity value below the threshold and NOT replaced (an earlier record of a batch, or a
 *                         multi-rank step driven through the phase calls without prmf_comm_init): relative error of
 *                         recon up to ~1e-7 (||X|| / recon)^2 (k > 64; a tenth of that at k <= 64)
 *   PRMF_RECON_CLAMPED    the identity value was negative; recon was reported as 0 (never without this bit) */
#define PRMF_OBJ_STRIDE 10
enum { PRMF_OBJ_RECON = 0, PRMF_OBJ_MANIFOLD, PRMF_OBJ_IGNORE, PRMF_OBJ_FRO,
       PRMF_OBJ_GAMMA, PRMF_OBJ_DELTA, PRMF_OBJ_OBJ, PRMF_OBJ_RECON_SQ,
       PRMF_OBJ_FLAGS /* PRMF_RECON_* bits as a double */, PRMF_OBJ_RECON_SQ_IDENTITY };
#define PRMF_RECON_DIRECT 1
#define PRMF_RECON_IMPRECISE 2
#define PRMF_RECON_CLAMPED 4

int prmf_abi_version(void);

/* Human-readable text of the last failure on ctx (ctx == NULL: last failure of
 * prmf_create in this thread).  Valid until the next call on the same ctx. */
const char* prmf_last_error(const prmf_ctx* ctx);

/*
 * One context per GPU / rank.  X is row-sharded over `nranks` ranks: this rank
 * holds `m_local` of the samples; n (genes) and k (latent) are global.  With
 * nranks > 1 the k latent columns (and their active pathway Laplacians) are sharded
 * for the V-update (default): rank r owns columns [r*kc, min(k,(r+1)*kc)),
 * kc = ceil(k/nranks) -- see prmf_comm_sizes; with PRMF_FLAG_REPLICATED in `flags`
 * every rank updates all of V from the all-reduced X^T U instead.
 * Replaces: the shapes implied by X/U/V at prmf_runner.py:417-418.
 * Limits: 1 <= k <= 128 (PRMF_ERR_UNSUPPORTED above); x_dtype PRMF_X_F32 or PRMF_X_BF16.
 */
int prmf_create(prmf_ctx** out, int device, int64_t m_local, int64_t n, int k,
                int x_dtype, int nranks, int rank, unsigned flags);
int prmf_destroy(prmf_ctx* ctx);

/*
 * Upload the data matrix.  dX: device pointer, m_local x n row-major with row
 * stride `ld` elements, dtype as given to prmf_create.  The library keeps its
 * own zero-padded copy of X and of X^T in HBM (both sweeps then stream a
 * row-major operand), so the caller may free dX after the stream has passed
 * this call.  norm_x_sq = ||X||_F^2 of the GLOBAL matrix (all ranks), needed by
 * the Gram-identity form of the reconstruction error.
 * Replaces: the `X` argument of nmf_manifold_vec_update (prmf_runner.py:374).
 */
int prmf_set_x(prmf_ctx* ctx, const void* dX, int64_t ld, double norm_x_sq, void* stream);

/*
 * The same upload from a source of ANY float type, with ||X||^2 computed on the way (ABI 3).  One pass over the source
 * narrows every value to the context's storage type (round to nearest even at every step: f64 -> f32, f32 -> bf16,
 * f64 -> f32 -> bf16 -- what `Tensor.to(dtype)` does), writes the padded copy and the padded transpose, and sums the
 * squares of the STORED values in float64 in a fixed tree that does not depend on how the rows arrive (64 x 64 tiles
 * anchored at the first row; prmf_sum_squares uses the same tree).  *local_norm_sq_out (nullable) = that sum over this
 * rank's rows; it also becomes the context's ||X||^2 -- with nranks > 1 the caller adds the ranks' values and passes the
 * total to prmf_set_norm_x_sq.  Both calls synchronise.
 *   prmf_set_x_device: dX is a DEVICE matrix (m_local x n, leading dimension ld elements of src_dtype).
 *   prmf_set_x_host:   hX is a HOST matrix (pageable memory is fine): it travels in chunks of ~32 MB; with n_threads > 0
 *       (n_threads < 0: as many as the host has, at most 16) and a source wider than the storage type, worker threads
 *       narrow chunk i + 1 into pinned memory while chunk i crosses PCIe and the layout kernels take chunk i - 1 -- half
 *       (f32 -> bf16, f64 -> f32) or a quarter (f64 -> bf16) of the bytes on the wire; n_threads = 0: every chunk travels as
 *       it is and is narrowed on the device.  With workers that cannot keep the link busy (a busy host) the calling thread
 *       sends every other chunk as it is while they narrow its neighbour (decided per chunk from the measured narrowing time;
 *       PRMF_SETX_HYBRID=0/1 in the environment pins it -- a test hook).  The stored bits are the same either way.  `stream` is synchronised on entry
 *       (nothing of the caller's may still read the old X); the work runs on streams of the library's own.  The staging
 *       buffers (2 x 32 MB pinned, 2 x 32 MB on the device) are kept by the process for the next call;
 *       prmf_release_staging() frees them.
 * Replaces: the `X` argument of nmf_manifold_vec_update (prmf_runner.py:374) and `norm_X = np.linalg.norm(X)` (:640).
 */
enum prmf_src_dtype { PRMF_SRC_F32 = 0, PRMF_SRC_BF16 = 1, PRMF_SRC_F64 = 2 };
int prmf_set_x_device(prmf_ctx* ctx, const void* dX, int src_dtype, int64_t ld, double* local_norm_sq_out, void* stream);
int prmf_set_x_host(prmf_ctx* ctx, const void* hX, int src_dtype, int64_t ld, int n_threads, double* local_norm_sq_out,
                    void* stream);
int prmf_set_norm_x_sq(prmf_ctx* ctx, double norm_x_sq);
int prmf_release_staging(void);

/*
 * Upload the whole pathway library once (host pointers; copied to the device).
 * Pathway p has support nodes sup_idx[sup_ptr[p] .. sup_ptr[p+1]) (gene indices,
 * in G.nodes() order -- prmf_runner.py:689-690).  For the s-th support node
 * overall: sup_deg[s] = column sum of W (diagonal of D, :680-682), sup_ldiag[s]
 * = L[i,i] = deg - W[i,i] (normalize_laplacian reads it, :60), and its
 * adjacency row is entries row_ptr[s] .. row_ptr[s+1) of (col_local, w), where
 * col_local indexes the SAME pathway's support list.
 * Replaces: the Ws/Ds/Ls/lapl_to_feat_inds lists built at prmf_runner.py:673-690
 * and the NORMALIZED_LAPLACIANS global (:693-696).
 */
int prmf_set_pathway_library(prmf_ctx* ctx, int P,
                             const int64_t* sup_ptr, const int32_t* sup_idx,
                             const double* sup_deg, const double* sup_ldiag,
                             const int64_t* row_ptr, const int32_t* col_local,
                             const double* w, void* stream);

/*
 * Choose the active pathway of every latent column for the next steps
 * (k host ints, indices into the library).
 * Replaces: map_k_to_lapls (prmf_runner.py:260-270, called at :730).
 */
int prmf_set_active(prmf_ctx* ctx, const int32_t* k_to_lapl, void* stream);

/*
 * Tell the library that the caller changed U or V (or their pointers) outside
 * prmf_step, so cached derived state (packed V, V^T V) is rebuilt on the next step.
 */
int prmf_invalidate_factors(prmf_ctx* ctx);

/*
 * n_steps iterations of U-update, V-update and objective, single rank
 * (nranks == 1).  dU (m x k) and dV (n x k): float32 row-major device buffers,
 * updated in place.  gamma/delta are the already-scaled values the reference
 * passes (prmf_runner.py:644-645).  tradeoff < 0: fixed gamma/delta (:374-451);
 * tradeoff in [0,1]: after each step gamma = delta = (1-t)*recon/(t*manifold)
 * (:542-548).  obj_out (host, n_steps x PRMF_OBJ_STRIDE doubles) receives one
 * record per iteration; gamma_delta_out (host, 2 doubles, may be NULL) the
 * values to use for the next call.  Synchronises `stream` before returning.
 * Returns PRMF_ERR_NUMERIC if any record is not finite.
 * Replaces: nmf_manifold_vec_update / _tradeoff (prmf_runner.py:374-451, 497-554).
 */
int prmf_step(prmf_ctx* ctx, float* dU, float* dV, int n_steps,
              double gamma, double delta, double tradeoff,
              double* obj_out, double* gamma_delta_out, void* stream);

/* ------------------------------------------------------------------------
 * Multi-rank form of one iteration: four device phases with collectives
 * between them, driven by the host over torch.distributed (RCCL), or issued by
 * the library itself (prmf_step_dist).  Column-sharded mode (default):
 *
 *   prmf_phase_u    P = X_g V ; U_g update ; partial U_g^T U_g      -> gram buffer
 *        all_reduce(gram, SUM)                       k*k doubles
 *   prmf_phase_q    Q_g = X_g^T U_g laid out by owner column block  -> q_send
 *        reduce_scatter(q_send -> q_recv, SUM)       n*kc floats per rank
 *   prmf_phase_v    graph terms + divide + clamps for the OWN columns,
 *                   their per-workgroup sums in the tail            -> v_send
 *        all_gather(v_send -> v_recv)                v_send_floats per rank
 *   prmf_phase_post rebuild replicated V, packed V, V^T V partials, objective state
 *
 * Replicated mode (PRMF_FLAG_REPLICATED):
 *
 *   prmf_phase_u    as above;  all_reduce(gram, SUM)
 *   prmf_phase_q    Q_g = X_g^T U_g (n x k)                         -> q_send
 *        all_reduce(q_send, SUM, in place)           n*k floats
 *   prmf_phase_v    V update (all columns) from the reduced Q
 *   prmf_phase_post objective state
 *
 * (prmf_step_dist issues the all_reduce of the gram buffer AFTER prmf_phase_q, in one RCCL group with the exchange of
 * Q: the X^T U sweep's prologue reduces the local partials, so prmf_phase_u has nothing to hand over.)
 *
 * prmf_step is exactly this sequence with the collectives elided.
 * ---------------------------------------------------------------------- */

/* element counts of the caller-allocated communication buffers */
typedef struct prmf_comm_sizes {
  int64_t gram_doubles;     /* all_reduce buffer (float64)                    */
  int64_t q_send_floats;    /* nranks * q_recv_floats                         */
  int64_t q_recv_floats;    /* n * kc                                         */
  int64_t v_send_floats;    /* n * kc, then the tail: column-sharded mode: the own-column update's per-workgroup
                               sums ([<= 256][kc + 1] doubles); always: 4 doubles of objective pieces */
  int64_t v_recv_floats;    /* nranks * v_send_floats                         */
  int32_t kc;               /* latent columns updated per rank                */
  int32_t mode;             /* PRMF_DIST_*: which exchanges the caller performs */
} prmf_comm_sizes;

/* Exchanges between the phases (host-driven form; prmf_step_dist issues the same ones itself):
 *   PRMF_DIST_COLUMN_SHARDED (default for nranks > 1): rank r updates latent columns
 *       [r*kc, (r+1)*kc) and their pathway Laplacians.
 *       phase_u; all_reduce(gram, float64 sum); phase_q; reduce_scatter(q_send -> q_recv, float32 sum); phase_v;
 *       all_gather(v_send -> v_recv); phase_post.
 *   PRMF_DIST_REPLICATED (PRMF_FLAG_REPLICATED at creation): every rank updates all of V.
 *       phase_u; all_reduce(gram); phase_q; all_reduce(q_send, float32 sum, in place; q_recv aliases q_send,
 *       n*k floats); phase_v; phase_post.  v_send / v_recv are not exchanged. */
#define PRMF_DIST_COLUMN_SHARDED 0
#define PRMF_DIST_REPLICATED 1

int prmf_get_comm_sizes(const prmf_ctx* ctx, prmf_comm_sizes* out);

/* Device buffers for the collectives (float64 gram, float32 others).  With
 * nranks == 1 this call is optional (ctx allocates them; q_recv aliases q_send
 * and v_recv aliases v_send). */
int prmf_set_comm_buffers(prmf_ctx* ctx, double* gram, float* q_send, float* q_recv,
                          float* v_send, float* v_recv);

/* Bind the factor buffers and hyper-parameters for a sequence of phase calls
 * (same meaning as in prmf_step); resets the objective-record cursor. */
int prmf_begin(prmf_ctx* ctx, float* dU, float* dV, double gamma, double delta,
               double tradeoff, void* stream);
int prmf_phase_u(prmf_ctx* ctx, void* stream);
int prmf_phase_q(prmf_ctx* ctx, void* stream);
int prmf_phase_v(prmf_ctx* ctx, void* stream);
int prmf_phase_post(prmf_ctx* ctx, void* stream);
/* Copy the records of the iterations run since prmf_begin to the host
 * (synchronises `stream`). */
int prmf_end(prmf_ctx* ctx, int n_steps, double* obj_out, double* gamma_delta_out, void* stream);

/* ------------------------------------------------------------------------
 * The same multi-rank iteration with the collectives issued by the library itself
 * (RCCL, resolved with dlopen("librccl.so.1") at prmf_comm_init; enqueued on the
 * caller's stream between the phases), so that n_steps iterations run without any
 * host round trip.  Bootstrap: rank 0 calls prmf_comm_unique_id and sends the 128
 * bytes to the other ranks by any host channel (the Python host broadcasts them with
 * torch.distributed); every rank then calls prmf_comm_init.  Communication buffers
 * are allocated by the library unless prmf_set_comm_buffers was called.
 * ---------------------------------------------------------------------- */
#define PRMF_COMM_ID_BYTES 128
int prmf_comm_unique_id(void* id_out /* PRMF_COMM_ID_BYTES */);
int prmf_comm_init(prmf_ctx* ctx, const void* id /* PRMF_COMM_ID_BYTES */);
/* The number of samples over ALL ranks.  The threshold below which a record's identity-based recon is replaced by the
 * direct residual (PRMF_RECON_* above) contains min(m, n): every rank has to evaluate it with the same m, or ranks whose
 * shards differ by a row would raise different flags and enter different collectives.  prmf_comm_init all-reduces the
 * local counts itself; a caller that drives the collectives (prmf_phase_*) calls this instead.  Unknown (never set,
 * nranks > 1): the library uses n alone -- rank-invariant as well. */
int prmf_set_global_rows(prmf_ctx* ctx, int64_t m_global);
/* Failure behaviour of the in-library collectives.  If an nccl* call fails on this rank, the open RCCL group is closed,
 * the communicator is ABORTED (ncclCommAbort), the call returns PRMF_ERR_HIP and every later prmf_step_dist on this
 * context returns PRMF_ERR_STATE: the caller is expected to exit non-zero so that its launcher ends the job.  A rank whose
 * peer has died is left inside a collective kernel; its wait for the stream (prmf_end inside prmf_step_dist) is therefore
 * bounded: after `seconds` (default 300; 0 = wait for ever) plus 2 s per iteration of the batch it is waiting for (a long
 * healthy batch is not a dead peer) it aborts its communicator and returns PRMF_ERR_HIP too.  No call hangs.  (An RCCL
 * build without ncclCommAbort: the communicator is leaked rather than destroyed -- ncclCommDestroy would wait for the
 * collective that is being escaped.) */
int prmf_set_comm_timeout(prmf_ctx* ctx, double seconds);
/* multi-rank prmf_step: same arguments and results (every rank gets the same
 * objective records); requires prmf_comm_init.  Also valid with nranks == 1. */
int prmf_step_dist(prmf_ctx* ctx, float* dU, float* dV, int n_steps,
                   double gamma, double delta, double tradeoff,
                   double* obj_out, double* gamma_delta_out, void* stream);

/*
 * Candidate scoring for `restrict` (prmf_runner.py:129-194 / :115-127):
 * scores_out[kk*P + p] = ||vhat_kk[S_p]||_2 + 1 - vhat_kk^T Lhat_p vhat_kk for
 * every latent column kk and every library pathway p (host buffer, k*P doubles).
 * dV: n x k float32 device buffer.  Synchronises `stream`.
 */
int prmf_score_candidates(prmf_ctx* ctx, const float* dV, double* scores_out, void* stream);

/*
 * The terms of the final latent <-> pathway matching, `force_distinct_lapls` (prmf_runner.py:209-258), for every latent
 * column kk and every library pathway p (host buffers, k*P doubles each; float64 arithmetic on the float32 V):
 *     quad_out[kk*P + p]   = v_kk^T L_p v_kk          the un-normalised Laplacian L = D - W (:232: L.dot(v).dot(v))
 *     ignore_out[kk*P + p] = sum_{i in S_p} 1 / (V[i, kk] + 1)      (:235 for k2 = kk and kk's pathway p)
 * so that V need not be downloaded for the k x (remaining candidates) sparse products of that function; the caller forms
 * the weights 1 / (gamma quad + delta ignore) and runs the matching.  Synchronises `stream`.  (ABI 3)
 */
int prmf_match_terms(prmf_ctx* ctx, const float* dV, double* quad_out, double* ignore_out, void* stream);

/*
 * Held-out reconstruction error, replaces measure_cv_performance
 * (prmf/__init__.py:768-798; called at script/prmf_runner.py:1084-1089): for every
 * row x of this context's X (the held-out samples, prmf_set_x)
 *     err_out[s] = min_{u >= 0} || x_s - u V^T ||_2 / || x_s ||_2
 * (scipy.optimize.nnls per sample in the reference; here X*V and V^T V accumulated in
 * float64 and a batched active-set solve on the normal equations, one wavefront per sample).  dV: n x k float32
 * device buffer; err_out: host buffer, m_local doubles; an all-zero sample gives
 * NaN (0/0) as in the reference.  No pathway library is needed.  Synchronises
 * `stream`.  PRMF_ERR_NUMERIC if a sample did not converge in 3k iterations (the
 * reference raises RuntimeError).
 */
int prmf_cv_error(prmf_ctx* ctx, const float* dV, double* err_out, void* stream);

/*
 * Non-negative least squares against the data matrix: x = argmin_{x >= 0} || A x - b ||_2 with A = X (transposed = 0:
 * rows equations, cols unknowns) or A = X^T (transposed = 1: cols equations, rows unknowns), X a float64 row-major DEVICE
 * matrix (rows x cols, leading dimension ld) that is neither copied nor transposed.  Replaces `scipy.optimize.nnls(X, u)`
 * and `scipy.optimize.nnls(X.transpose(), v)` of `--manifolds-init` (script/prmf_runner.py:308-334: nmf_init_u / nmf_init_v)
 * -- the Lawson-Hanson active-set method as there; gradient, column inner products and residual on the device, the
 * passive-set factor on the host, one refinement step per solve.  b (equations) and x (unknowns) are HOST arrays;
 * *rnorm_out (nullable) = || A x - b ||_2, *iterations_out (nullable) = active-set steps taken; max_iter <= 0: 3 x the
 * number of unknowns (scipy's default); PRMF_ERR_NUMERIC when it is exceeded ("maximum number of iterations reached");
 * PRMF_ERR_DECLINED when the passive-set factor (normal equations) loses a problem whose conditioning is beyond it -- the
 * passive columns became dependent, a descent direction was refused, or a gradient cannot be resolved: no answer is
 * returned rather than a wrong one; the caller falls back to scipy's orthogonal factorisation).
 * Both problems have a unique minimiser for data in general position, so the result equals scipy's up to rounding
 * (tests/test_gpu_nnls_matrix.py: 1e-9).  Synchronises `stream`; no context needed.
 */
int prmf_nnls(int device, const double* dX, int64_t rows, int64_t cols, int64_t ld, int transposed, const double* b, double* x,
              double* rnorm_out, int* iterations_out, int max_iter, void* stream);

/*
 * *out = sum of squares of a device matrix (m x n, leading dimension ld, x_dtype PRMF_X_F32 or PRMF_X_BF16) in float64:
 * ||X||_F^2 of the values the sweeps will read, what `norm_X = np.linalg.norm(X)` (script/prmf_runner.py:640) squares to
 * and what prmf_set_x takes as norm_x_sq (a rank of a multi-GPU run passes its rows and adds the ranks' results).
 * Deterministic (fixed summation tree).  No context needed; synchronises `stream`.
 */
int prmf_sum_squares(int device, const void* dX, int x_dtype, int64_t m, int64_t n, int64_t ld, double* out, void* stream);

/*
 * Per-gene quantile normalisation, replaces `X = quantile_transform(X)` (script/prmf_runner.py:1019-1020;
 * sklearn.preprocessing.quantile_transform with its defaults: n_quantiles = 1000, uniform output, axis 0).
 * dX: m x n float64 row-major device buffer (leading dimension ld >= n), transformed IN PLACE.  The quantiles of
 * every column are numpy's `linear` percentiles at linspace(0, 1, nq), nq = min(n_quantiles, rows used for the
 * fit), made monotone, and every value is mapped with the mean of an ascending and a descending np.interp --
 * float64, operation for operation as scikit-learn 1.7 / numpy 2.2 do it.  fit_rows (host, nullable): the row
 * subset the quantiles are fitted on (scikit-learn subsamples rows when m > 1e5; the caller draws them with the
 * reference's RNG), n_fit of them; NULL: all m rows.  NaNs are ignored by the fit and stay NaN.  No context
 * needed; synchronises `stream`.
 */
int prmf_quantile_transform(int device, double* dX, int64_t m, int64_t n, int64_t ld, const int64_t* fit_rows,
                            int64_t n_fit, int n_quantiles, void* stream);

/*
 * The numeric body of a delimited text matrix, parsed on the GPU: replaces the parse inside
 * `pd.read_csv(args.data, sep=args.delimiter, header=..., index_col=...)` (script/prmf_runner.py:942) -- at the
 * reference's typical size (500 x 20000, 187 MB of text) 1.7 s of a 2.0 s run.  pandas is a dependency of the reference
 * (any pandas >= 1.2: float_precision "high", converter `precise_xstrtod`); every value produced here equals pandas' bit
 * for bit because it is computed by the same float64 operations in the same order (prmf_amd/csrc/xstrtod.h).  What is
 * parsed: the lines AFTER the header (the caller reads header and row names itself), every line `fields_per_line` fields
 * separated by `delimiter`, ended by "\n" or "\r\n" (a missing final newline is supplied), every data field
 *     [ \t]* [+-]? digits* ( . digits* )? ( [eE] [+-]? digits{1,4} )? [ \t]*      with at least one mantissa digit.
 * Everything else is not interpreted but REPORTED in *flags_out, and the caller gives the file to pandas:
 *     PRMF_TEXT_RAGGED       a line with another number of fields (blank lines, quoted separators, lone "\r" line ends)
 *     PRMF_TEXT_QUOTED       a quote in a data field
 *     PRMF_TEXT_NOT_NUMERIC  an empty field, NA / nan / inf, hex, an exponent beyond 4 digits, an overflowing value
 *     PRMF_TEXT_WIDE_INT     an integer spelled with more than 15 digits (exact in pandas' int64, not in a double), or an
 *                            integer part of 19 or more digits (pandas' int64 / uint64 probes overflow on it and the column's
 *                            type then depends on what stood above that field)
 *     PRMF_TEXT_BARE_CR      a carriage return that is not followed by a line feed, anywhere (row names included): pandas
 *                            ends a line there
 * Calls: prmf_text_create(&t, device, nbytes) reserves device memory for the body; prmf_text_append copies the next piece
 * of it from host memory (any piece size: a 64 MB read buffer keeps the host footprint flat); prmf_text_index finds the
 * fields and checks the line structure (*rows_out lines); prmf_text_parse converts: out = rows x (fields_per_line -
 * skip_first_field) float64, row-major, HOST memory; with skip_first_field the first field of every line is not parsed
 * and name_spans[2 r], name_spans[2 r + 1] are the byte offsets [start, end) of row r's name within the appended text
 * (host, 2 * rows); col_is_int[c] = 1 iff no field of data column c has a '.' or an exponent (pandas makes such a column
 * int64).  *first_bad_offset_out (nullable): offset of the first refused field, -1 if none.  All four synchronise the
 * device's null stream; no prmf_ctx is needed; errors are read with prmf_last_error(NULL).
 */
typedef struct prmf_text prmf_text;
enum { PRMF_TEXT_RAGGED = 1, PRMF_TEXT_QUOTED = 2, PRMF_TEXT_NOT_NUMERIC = 4, PRMF_TEXT_WIDE_INT = 8, PRMF_TEXT_BARE_CR = 16 };
int prmf_text_create(prmf_text** out, int device, int64_t nbytes);
int prmf_text_append(prmf_text* text, const void* bytes, int64_t nbytes);
int prmf_text_index(prmf_text* text, char delimiter, int64_t fields_per_line, int64_t* rows_out, int* flags_out);
int prmf_text_parse(prmf_text* text, int skip_first_field, double* out, int64_t* name_spans, int32_t* col_is_int, int* flags_out,
                    int64_t* first_bad_offset_out);
int prmf_text_destroy(prmf_text* text);

/*
 * A float64 matrix as delimited text, byte for byte what `pd.DataFrame(U, index=..., columns=...).to_csv(path, sep=",",
 * quoting=csv.QUOTE_NONNUMERIC)` writes for U.csv / V.csv (script/prmf_runner.py:1068-1071): every float as Python's
 * repr() prints it (shortest digits that read back as the same double; positional when the decimal point falls in
 * (-4, 16], else d[.ddd]e[+-]XX; "inf" / "-inf"; NaN as the quoted empty string), fields joined by `separator`, lines
 * ended by "\n".  header_line (nullable): written first as it is (the caller quotes the labels and ends the line);
 * row_prefixes / row_prefix_offsets (both or neither): the bytes that start row r -- its quoted label and the separator
 * -- are row_prefixes[row_prefix_offsets[r] .. row_prefix_offsets[r + 1]).  values: rows x cols, leading dimension ld,
 * HOST memory.  n_threads < 1: one per core, at most 16.  Host work only (no device, no context).  PRMF_ERR_IO when the
 * file cannot be created or written (errno in the message).
 */
int prmf_write_csv(const char* path, const char* header_line, const char* row_prefixes, const int64_t* row_prefix_offsets,
                   const double* values, int64_t rows, int64_t cols, int64_t ld, char separator, int n_threads);

/*
 * Direct reconstruction error of this rank's rows: *out_local = sum_ij (X_g - U_g V^T)_ij^2 over the stored X
 * (float64 sum; X - U V^T formed on the bf16 MFMA with both factors split into three exact bf16 pieces, X exact), what the reference computes at
 * prmf_runner.py:337 before the square root.  One sweep over X; needs prmf_set_x only.  dU (m_local x k), dV (n x k):
 * float32 device buffers.  With nranks > 1 the caller sums *out_local over the ranks.  Synchronises `stream`.
 */
int prmf_residual_sq(prmf_ctx* ctx, const float* dU, const float* dV, double* out_local, void* stream);

/* Introspection used by bench.py for the roofline block: launch geometry and
 * per-launch algorithmic bytes of the dominant kernel (the X / X^T sweep). */
typedef struct prmf_sweep_info {
  int64_t bytes_x_sweep;    /* bytes of X streamed by the X*V sweep  (padded copy)  */
  int64_t bytes_xt_sweep;   /* bytes of X^T streamed by the X^T*U sweep             */
  int32_t splits_xv, splits_xtu;
  int32_t grid_xv, grid_xtu;
  int32_t f32_split;        /* 1: f32 X, multiplied as three bf16 pieces on the bf16 MFMA; 0: bf16 X */
  int32_t reserved;         /* bit 0 / bit 1: the X*V / X^T*U product is the EXACT short-reduction form (a reduction of at most
                               512 / (32-column tiles) elements: float32 MFMA against the unpacked factor instead of the hi/lo bf16 sweep) */
} prmf_sweep_info;
int prmf_get_sweep_info(const prmf_ctx* ctx, prmf_sweep_info* out);

/* Average duration (ms, HIP events recorded on the launch stream around each
 * launch) of the X*V sweep and of the X^T*U sweep over the iterations of the last
 * prmf_begin..prmf_end span (prmf_step = one span per <=256 iterations); needs
 * timing enabled (PRMF_FLAG_TIMING at creation or prmf_set_timing). */
#define PRMF_FLAG_TIMING 0x1u
#define PRMF_FLAG_REPLICATED 0x2u       /* nranks > 1: every rank updates all of V from the all-reduced X^T U instead of
                                           sharding the V update by latent columns (see prmf_comm_sizes) */
/* Single rank, tradeoff off: by default the iteration is 4 launches (the k x k reductions and the pathway terms run as
 * prologues of the two sweeps) and the objective records of all iterations since prmf_begin are evaluated in prmf_end
 * from per-iteration state kept on the device -- same arithmetic, bit-identical records.  This flag keeps the
 * per-iteration evaluation (three more launches per iteration): the form the tradeoff feedback (prmf_runner.py:542-548)
 * always uses.  Multi-rank: the replicated mode defers in the same way; the column-sharded mode keeps its update
 * sequence but reduces V^T V in the next sweep's prologue and writes the records in prmf_end from the kept Grams and
 * the gathered per-rank sums (two launches fewer per iteration, bit-identical records); the flag restores the
 * per-iteration form there too. */
#define PRMF_FLAG_EAGER_OBJECTIVE 0x4u
/* Every record's recon from the direct residual (one extra sweep over X per iteration; implies the per-iteration
 * objective): the reference's :337 for the whole trace, whatever the fit. */
#define PRMF_FLAG_DIRECT_RESIDUAL 0x8u
/* Switch the per-launch event timing on or off after creation (bench.py times its
 * throughput run without events and a second short run with them). */
int prmf_set_timing(prmf_ctx* ctx, int enabled);
int prmf_get_timing(const prmf_ctx* ctx, double* ms_xv, double* ms_xtu, double* ms_total);

/* What the multi-rank step actually ran with (bench.py reports it): whether the collectives are issued inside the
 * library, the size / rank RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank; -1 if the
 * collectives are host-driven), the V-update mode, and -- with timing enabled -- the average device time per iteration
 * of the four phases of the last prmf_step_dist batch, each including the exchange that follows it on the caller's
 * stream: [0] phase_u, [1] phase_q + the grouped exchange (k x k all-reduce of U^T U with the reduce-scatter /
 * all-reduce of Q), [2] phase_v + all-gather, [3] phase_post.  -1 when not measured. */
typedef struct prmf_comm_info_t {
  int32_t in_library;
  int32_t comm_nranks;
  int32_t comm_rank;
  int32_t mode;              /* PRMF_DIST_* */
  double ms_phase[4];
} prmf_comm_info_t;
int prmf_comm_info(const prmf_ctx* ctx, prmf_comm_info_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PRMF_HIP_H */
