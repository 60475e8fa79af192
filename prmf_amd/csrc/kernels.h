// Internal interface between the translation units of libprmf_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace prmf {

// ---------------------------------------------------------------------------
// Geometry shared by host and device code
// ---------------------------------------------------------------------------
// Both sweeps (P = X V and Q = X^T U) are the same kernel: C[M x KP] = A[M x K] * B[K x KP]
// with A a zero-padded row-major matrix resident in HBM (X, resp. the X^T copy)
// and B a factor (V resp. U) pre-packed into MFMA B-fragment order.
constexpr int ROW_BYTES_PER_CHUNK = 128;   // one reduction chunk = 128 B of every A row
constexpr int K_PAD = 128;                 // reduction length padded to 128 elements: an even number of chunks (both dtypes)
constexpr int M_PAD = 512;                 // rows padded to the largest workgroup panel (8 waves x 64 rows)
constexpr int MAX_KP = 128;                // latent columns padded to 32*NT, NT <= 4

// Objective record on the device and across the ABI (include/prmf_hip.h: PRMF_OBJ_*): 10 doubles per iteration
constexpr int REC_STRIDE = 10;
constexpr int REC_FLAGS = 8;               // PRMF_RECON_* bits, stored as a double
constexpr int REC_RECON_SQ_IDENTITY = 9;   // the Gram-identity value of recon^2, kept when the direct residual replaced it
constexpr int REC_FLAG_DIRECT = 1, REC_FLAG_IMPRECISE = 2, REC_FLAG_CLAMPED = 4;
// recon^2 from the Gram identity ||X||^2 - 2 <X^T U, V> + <U^T U, V^T V> is exact to a fraction of ||X||^2 set by the
// float32 accumulation of X^T U in the sweep (the Gram terms are ~1e-10; profiles/r03_recon_identity_terms.txt,
// r03_ab_bf16_low_piece_accumulators.txt): ~1e-9..1e-8 with the low-piece products in accumulators of their own
// (every f32 X sweep; bf16 X at k <= 64), ~1e-7 at k > 64.  Below these fractions of ||X||^2 the relative error of recon
// passes ~2.5e-5 and the direct residual takes over (recon / ||X|| < 1.4% resp. 4.5%).
constexpr double RECON_DIRECT_THRESHOLD_LO = 2e-4;      // k <= 64
constexpr double RECON_DIRECT_THRESHOLD = 2e-3;         // k > 64, reductions of up to RECON_DIRECT_CHAIN_ROWS samples per rank;
constexpr double RECON_DIRECT_CHAIN_ROWS = 32768.0;     //   longer ones: x (samples per rank / 32768) -- see direct_below (capi.hip)
// + RECON_DIRECT_FACTOR_TERM / sqrt(min(m, n) k): the rounding of the sweeps' hi/lo-packed factors (16 mantissa bits,
// 4e-6 per entry) enters <X^T U, V> through the min(m, n) k entries of the smaller factor and only averages over those
constexpr double RECON_DIRECT_FACTOR_TERM = 0.13;
constexpr int RESIDUAL_MAX_BLOCKS = 1024;

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int chunk_elems(int x_dtype) { return x_dtype == 1 ? 64 : 32; }   // bf16 : f32
static inline int elem_bytes(int x_dtype) { return x_dtype == 1 ? 2 : 4; }
// bytes of packed B per chunk: chunk_elems x KP values, bf16 hi+lo (4 B) or f32 (4 B)
static inline int64_t packed_chunk_bytes(int x_dtype, int NT) { return (int64_t)chunk_elems(x_dtype) * 32 * NT * 4; }

// Factor value accessor: F(row, col) = src[(col / kc) * blk_stride + row * kc + col % kc].
// Row-major n x k is kc = k (blk_stride unused); the all-gather receive buffer
// is kc = columns per rank, blk_stride = floats per rank block.
struct FactorView {
  const float* src;
  int rows;
  int cols;          // valid columns (k)
  int kc;
  int64_t blk_stride;
};

// Pathway library on the device (prmf_set_pathway_library): all P pathways concatenated
struct PathwayLibDev {
  int P;
  int max_support;
  int max_nnz;              // most adjacency entries of one pathway
  int64_t total_support, total_nnz;
  const int64_t* sup_ptr;   // [P+1]
  const int32_t* sup_idx;   // [total_support] gene index
  const double* sup_deg;    // D_ii
  const double* sup_dis;    // L_ii^-1/2 or 0     (normalize_laplacian, prmf_runner.py:56-65)
  const int64_t* row_ptr;   // [total_support+1]
  const int32_t* col_local; // [nnz] support-local column
  const double* w;          // [nnz]
};

// Work a sweep launch does in its first microseconds for the epilogues around it (prologue.h).
enum { PROLOGUE_NONE = 0, PROLOGUE_POST = 1, PROLOGUE_MID = 2 };
struct SweepPrologue {
  int kind = PROLOGUE_NONE;
  // both kinds: float64 reduction of `nparts` per-workgroup Gram partials [k*k] -> G (and the iteration's kept copy)
  const float* gram_partial = nullptr;
  int nparts = 0, k = 0;
  double* G = nullptr;
  double* G_keep = nullptr;
  // PROLOGUE_POST (before X V; closes the PREVIOUS iteration): column norms^2 of V_new and <Q, V_new>
  const double* vpartial = nullptr;     // [nwaves][k+1] from the V epilogue
  int nwaves = 0;
  double* colnorm2 = nullptr;           // [k]
  double* cross = nullptr;              // [1]
  // column-sharded multi-rank mode: the V update's per-workgroup sums arrive with the all-gather instead -- rank r's block
  // of the gathered buffer ends in [gath_nblk][gath_kc + 1] doubles (own-column sums of squares, <Q, V_new> last)
  const float* gath = nullptr;          // v_recv
  int64_t gath_stride = 0;              // floats per rank block
  int64_t gath_off = 0;                 // float offset of the partials inside a block
  int gath_nblk = 0, gath_kc = 0, gath_nranks = 0;
  const double* gu_src = nullptr;       // multi-rank: the ALL-REDUCED U^T U of the closed iteration (still in the exchange
  double* gu_keep = nullptr;            //   buffer) is copied to its kept slot here; null with one rank
  // PROLOGUE_MID (before X^T U): pathway terms of the V update on the old V
  PathwayLibDev lib{};
  const int32_t* active = nullptr;
  int col0 = 0, ncols = 0;
  const float* V = nullptr;
  const double* hyper = nullptr;
  float* gnum = nullptr;
  float* gden = nullptr;
};

// ---------------------------------------------------------------------------
// gemm.hip
// ---------------------------------------------------------------------------
// w16: the fragment order of the 16x16x32 MFMA (bf16 X at k > 64), see gemm.hip; the epilogues that pack take the same flag
hipError_t launch_pack_factor(const FactorView& f, int x_dtype, int NT, int64_t Kpad, void* packed, hipStream_t s, bool w16 = false);
// which slabs hold a segment of a row panel: one bit per slab, at most 64 slabs per panel (wide, short matrices -- the
// reference's typical samples << genes shape -- have 2-3 panels and would otherwise run the X V sweep on 32 workgroups per panel)
typedef uint64_t segmask_t;
constexpr int MAX_SLABS = 64;
struct GemmPlan {
  int nw = 8;               // waves per workgroup
  int mt = 1;               // 32-row tiles per wave
  bool bdb = true;          // packed factor double-buffered in LDS
  bool pp = false;          // 8-wave form: the two waves of a SIMD half a chunk apart (ping-pong chunk loop, gemm.hip)
  int panel_rows = 0;       // rows per panel (= rows of one workgroup tile)
  int panels = 0;           // row panels
  int nchunks = 0;          // reduction chunks per row (even)
  int ngroups = 0;          // work units per row panel: the unit is (panel, chunk)
  int grid = 0;             // workgroups launched: one per resident slot, equal contiguous unit ranges
  int nx = 1;               // slices of the reduction range (8 = one per XCD)
  int pmax = 1;             // most workgroups that touch one (panel, slice)
  int max_seg = 1;          // slabs allocated: nx * pmax
  bool slab_nt = true;      // partial slabs leave as non-temporal stores (false: plain stores, small slab volumes)
  bool exact = false;       // a short reduction (padded depth x column tiles <= 512): exact float32 products against the unpacked factor (gemm.hip,
                            //   exact_product_kernel), one slab per panel; the fields above then describe that launch
  std::vector<segmask_t> segmask;  // [panels] bit s set <=> slab s holds a segment of this panel
};
// Test hooks, read from the environment ONCE per prmf_create (capi.hip) -- the planner itself takes no knobs.
struct SweepTuning {
  int force_slices = 0;     // PRMF_SWEEP_XCD = 1 | 2 | 4 | 8: force the number of reduction slices (decomposition-equivalence test)
  bool no_exact = false;    // PRMF_SWEEP_NO_EXACT = 1: short reductions through the streaming sweep too (A/B and equivalence tests)
};
GemmPlan plan_gemm(int x_dtype, int64_t M, int64_t Kpad, int NT, int n_cus, const SweepTuning& tune);
// debug only (prmf_debug_set_stamps): layout of the sweep's stamp buffer, in uint64 entries.  Fixed regions that do not
// overlap at any grid (<= STAMP_MAX_WGS workgroups of <= 8 waves):
//   [0, 2 W)                    start / end wall clock of workgroup b at [2 b], [2 b + 1]
//   [STAMP_WHERE, + W)          where workgroup b ran: XCC_ID << 32 | HW_ID
//   [STAMP_PHASE, + W * 8 * 8)  -DPRMF_PHASE_PROF=1 builds: 8 phase sums per wave, wave-major per workgroup (8 waves' room)
constexpr int STAMP_MAX_WGS = 1024, STAMP_WHERE = 2 * STAMP_MAX_WGS, STAMP_PHASE = 3 * STAMP_MAX_WGS;
constexpr int STAMP_ENTRIES = STAMP_PHASE + STAMP_MAX_WGS * 8 * 8;
extern unsigned long long* g_sweep_stamps;   // debug only
extern unsigned long long* g_epi_stamps;     // debug only (column-split epilogue phases)
extern int* g_tail_probe;                     // debug only (tools/tail_probe.py)
extern int g_tail_launch;
extern const GemmPlan* g_tail_plan;
hipError_t launch_gemm(int x_dtype, const GemmPlan& p, const void* A, int64_t lda, int64_t M_store,
                       const void* packedB, int NT, float* slabs, int64_t slab_stride, hipStream_t s,
                       bool split = false /* f32 X as three bf16 pieces, packedB in the bf16 hi/lo layout */,
                       const SweepPrologue* prologue = nullptr /* runs inside the sweep when its jobs fit the grid, else as a launch of its own */,
                       bool attr_only = false /* launch nothing: raise the kernel's dynamic-LDS limit on the current device (prmf_create) */,
                       const float* factor = nullptr /* exact plans: the float32 factor itself [k_valid][k], row-major */,
                       int64_t k_valid = 0, int k = 0);
hipError_t launch_stream_read(const void* p, size_t bytes, int n_cus, void* sink /* >= 2 n_cus ints, never written */, hipStream_t s);
// Pathway terms of the V update as extra workgroups of the U epilogue launch (they depend on the OLD V and gamma / delta
// only; the epilogue leaves a quarter of the chip's slots idle and runs longer than the terms' dependent-load chain, so
// they cost nothing there -- as a prologue of the X^T U sweep they delayed k of its workgroups by 6-9 us).  ncols = 0: none.
struct GraphJob {
  PathwayLibDev lib{};
  const int32_t* active = nullptr;
  int col0 = 0, ncols = 0;
  int nsplit = 1;           // workgroups per column (large supports): the launch adds ncols * nsplit workgroups
  const float* V = nullptr;
  const double* hyper = nullptr;
  float* gnum = nullptr;
  float* gden = nullptr;
};
static inline int graph_job_split(int max_support) {      // one workgroup per 1024 support rows, at most 16
  const int s = (max_support + 1023) / 1024;
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}
// the same prologue work as a stand-alone launch (end of a batch: no sweep follows)
hipError_t launch_prologue_alone(const SweepPrologue& p, hipStream_t s);

// ---------------------------------------------------------------------------
// setx.hip: X enters the library in one pass -- narrow to the storage type, padded copy, padded transpose, ||X||^2
// ---------------------------------------------------------------------------
// src_dtype: 0 f32, 1 bf16, 2 f64 (include/prmf_hip.h PRMF_SRC_*); x_dtype: the storage type (0 f32, 1 bf16).
// `rows` rows of the source starting at a multiple of 64 rows of the matrix; X / XT point at that first row / column
// (both null: only the sums); partial: setx_partial_doubles(rows, cols) doubles of scratch; rowsum: this launch's tile rows
// ((rows + 63) / 64 doubles, float64 sums of squares of the STORED values), summed by launch_setx_total.
int64_t setx_partial_doubles(int64_t rows, int64_t cols);
hipError_t launch_setx_rows(int src_dtype, int x_dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols,
                            void* X, int64_t ldx, void* XT, int64_t ldxt, double* partial, double* rowsum, hipStream_t s);
hipError_t launch_setx_total(const double* rowsum, int64_t tile_rows, double* out, hipStream_t s);
// host: rows [r0, r1) of a host matrix narrowed to the storage type (same roundings), compact (leading dimension = cols)
void host_narrow_rows(int src_dtype, int x_dtype, const void* src, int64_t ld, int64_t r0, int64_t r1, int64_t cols, void* dst);

// ---------------------------------------------------------------------------
// update.hip
// ---------------------------------------------------------------------------

// fused U epilogue (any k; k % 4 == 0 takes float4 loads of the factor rows): slab sum + U (V^T V) on the f32 MFMA + update + fragment packing
// (*gram_done: the kernel also wrote epilogue_blocks(Kpad) per-workgroup partials of F_new^T F_new)
int epilogue_wpt(int64_t rows, int max_seg, int NT);   // waves sharing a tile's slab sum (1, 2, 4)
int epilogue_blocks(int64_t rows, int wpt = 1);
hipError_t launch_u_epilogue(float* U, int64_t m, int k, const float* slabs, const segmask_t* segmask, int panel_rows,
                             int64_t slab_stride, int KP, const double* Gv, int x_dtype, void* packed, int64_t Kpad,
                             float* gram_partial, bool* gram_done, int wpt, hipStream_t s, const GraphJob* graph = nullptr,
                             bool w16 = false);
hipError_t launch_gram_reduce(const float* partial, int nparts, int k, double* G, hipStream_t s);
// fused single-rank V epilogue (k % 4 == 0): in place, + packing + per-wave partials [waves][k+1]
int v_epilogue_waves(int64_t n, int wpt = 1);
hipError_t launch_v_epilogue(float* V, int64_t n, int k, const float* slabs, const segmask_t* segmask, int panel_rows,
                             int64_t slab_stride, int KP, const double* Gu, const int32_t* pos, const float* gnum,
                             const float* gden, int max_support, int x_dtype, void* packed, int64_t Kpad,
                             double* partial, float* gram_partial, bool* gram_done,
                             const float* qdense /* null: sum the slabs; else Q [n][k], already summed */,
                             float* vsup /* null, or [k][max_support]: V_new on the active supports (deferred objective) */,
                             int wpt, hipStream_t s, bool w16 = false);
// Deferred objective (single rank, tradeoff off): all iterations of a batch in two launches at prmf_end.
//   colnorm2 [iters][k], vsup [iters][k][max_support] -> colres [iters][k][2] (manifold, ignore of the normalised column)
hipError_t launch_objective_all(const PathwayLibDev& lib, const int32_t* active_dev, int k, int iters, const double* colnorm2,
                                const float* vsup, double* colres, double* scratch /* [k][max_support], large supports only */,
                                hipStream_t s);
//   Gu_keep / Gv_keep [iters][k*k], cross [iters], colres -> records [iters][8]   (gamma, delta constant over the batch)
//   `need` (nullable): raised by the LAST record of the batch, see launch_finalize
hipError_t launch_records_all(const double* Gu_keep, const double* Gv_keep, int k, int iters, const double* cross,
                              const double* colres, double norm_x_sq, const double* hyper, double* records, int* need, bool force,
                              double direct_below /* recon^2 below this: direct residual */, hipStream_t s);
// partial Gram: per-wave partials -> partial[gram_blocks][k*k]; reduce -> G (double)
int gram_blocks(int64_t rows);
hipError_t launch_gram(const FactorView& f, double* partial, double* G, hipStream_t s);
// only the gram_blocks(f.rows) per-workgroup partials [k*k] floats (the reduction happens in a sweep's prologue)
hipError_t launch_gram_partials(const FactorView& f, float* partial, hipStream_t s);
// Q slabs -> q_send[nranks][n][kc]
hipError_t launch_slab_sum_blocks(const float* slabs, const segmask_t* segmask, int panel_rows, int64_t slab_stride, int KP,
                                  int64_t n, int k, int kc, int nranks, float* q_send,
                                  int max_seg /* most slabs a panel receives (plan.max_seg): sizes the cooperative sum */, hipStream_t s);
hipError_t launch_set_hyper(double* hyper, double gamma, double delta, double tradeoff, hipStream_t s);
hipError_t launch_set_active(int32_t* active_dev, const int32_t* host_vals, int k, hipStream_t s);
// pos[n][kc] = index in the active pathway's support of gene j for own column c, or -1
hipError_t launch_build_pos(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols, int kc,
                            int64_t n, int32_t* pos, hipStream_t s);
// gnum/gden[ncols][max_support]: gamma*W v + delta*(v+1)^-2, gamma*deg*v on the support  (:431-438)
hipError_t launch_graph_terms(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols,
                              const float* V, int k, const double* hyper, float* gnum, float* gden, hipStream_t s);
int v_update_blocks(int64_t n);
// V_own <- V * (Q + gnum) / max(V Gu + gden, eps), clamp; partial column sums of squares and <Q,V_new>
hipError_t launch_v_update(const float* V, int64_t n, int k, int col0, int ncols, int kc, const float* q_recv,
                           const double* Gu, const int32_t* pos, const float* gnum, const float* gden,
                           int max_support, float* v_send, double* partial /*[blocks][kc+1]*/, hipStream_t s);
// per own column: manifold / ignore of the normalised column (:344-352); then rank extras into v_send tail
hipError_t launch_objective_terms(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols, int kc,
                                  int64_t n, const float* v_send, const double* partial, int nblocks,
                                  double* colres /*[kc][2]*/, double* scratch /*[kc][max_support]*/,
                                  double* extras_out /*3 doubles in v_send tail*/, hipStream_t s);
// V[n][k] row-major from the gathered blocks
// objective record + tradeoff feedback
// `need` (nullable): raised when this record's identity-based recon^2 is below RECON_DIRECT_THRESHOLD ||X||^2 (or `force`);
// the tradeoff feedback is then left to record_fix (residual.hip), which runs on the direct value
hipError_t launch_finalize(const double* Gu, const double* Gv, int k, const float* v_recv, int nranks,
                           int64_t blk_stride, int64_t extras_off, double norm_x_sq, double* hyper /*gamma,delta,tradeoff*/,
                           double* record /*REC_STRIDE doubles*/, const double* loc_partial, int loc_nblocks,
                           const double* loc_colres, int* need, bool force, double direct_below, hipStream_t s);
// merged small launches of the single-rank path
hipError_t launch_mid(const float* gram_partial, int nparts, int k, double* Gu, const PathwayLibDev& lib,
                      const int32_t* active_dev, int col0, int ncols, const float* V, const double* hyper, float* gnum,
                      float* gden, hipStream_t s);
// multi-rank phase_post in one pass: gathered blocks -> V, packed fragments, V^T V partials
//   pos_all / vsup (nullable): V_new on the active supports of ALL k columns for the deferred objective (pos_all: [n][k])
hipError_t launch_v_post(float* V, int64_t n, int k, const float* v_recv, int64_t blk_stride, int kc, int KP, int x_dtype,
                         void* packed, int64_t Kpad, float* gram_partial, bool* gram_done, const int32_t* pos_all, float* vsup,
                         int max_support, hipStream_t s, bool w16 = false);
hipError_t launch_post(const float* gram_partial, int nparts, int k, double* Gv, const PathwayLibDev& lib,
                       const int32_t* active_dev, const float* V, const double* partial, int nblocks, double* colres,
                       double* scratch, hipStream_t s);
// candidate scores k x P
hipError_t launch_score_candidates(const PathwayLibDev& lib, const float* V, int64_t n, int k,
                                   double* colnorm_scratch, double* scores, hipStream_t s);
// the weights of the final matching: quad[kk * P + p] = v_kk^T L_p v_kk, ignore[kk * P + p] = sum_{i in S_p} 1 / (v_i,kk + 1)
hipError_t launch_match_terms(const PathwayLibDev& lib, const float* V, int k, double* quad, double* ignore, hipStream_t s);

// ---------------------------------------------------------------------------
// residual.hip: sum (X - U V^T)^2 directly (prmf_runner.py:337), one more sweep over the resident X
// ---------------------------------------------------------------------------
// per-wave float64 partials [*nparts_out] (<= 4 * RESIDUAL_MAX_BLOCKS); `need` (nullable device flag): run only if set
hipError_t launch_residual(int x_dtype, const void* X, int64_t ldx, int64_t m, int64_t n, int k, const float* U, const float* V,
                           int n_cus, double* partial, int* nparts_out, const int* need, hipStream_t s);
// out[0] = sum of the partials (0 if `need` is clear); record != null: the record is rewritten from it in the same launch
hipError_t launch_residual_sum(const double* partial, int nparts, const int* need, double* out, double* record, double* hyper,
                               bool feedback, hipStream_t s);
// recon / obj / recon_sq / flags of `record` from the (all-reduced) direct value; the tradeoff feedback if `feedback`
hipError_t launch_record_fix(const int* need, const double* resid_sq, double* record, double* hyper, bool feedback, hipStream_t s);

// ---------------------------------------------------------------------------
// nnls.hip
// ---------------------------------------------------------------------------
// out[row] = sum_j X[row][j]^2 (float64) over the stored copy of X
hipError_t launch_row_normsq(int x_dtype, const void* X, int64_t ld, int64_t m, int64_t n, double* out, hipStream_t s);
// H = X V and G = V^T V accumulated in float64 (partial: gram_f64_blocks(n) * k * k doubles)
hipError_t launch_xv_f64(int x_dtype, const void* X, int64_t ld, int64_t m, int64_t n, const float* V, int k, double* H,
                         hipStream_t s);
int gram_f64_blocks(int64_t n);
hipError_t launch_gram_f64(const float* V, int64_t n, int k, double* partial, double* G, hipStream_t s);
// err[s] = min_{c>=0} ||x_s - V c|| / ||x_s|| from G = V^T V, H = X V, xx = row norms^2   (prmf/__init__.py:768-798)
hipError_t launch_nnls(const double* G, const double* H, const double* xx, int64_t m, int k, int64_t n, double* err,
                       int* n_not_converged, hipStream_t s);

// ---------------------------------------------------------------------------
// prep.hip
// ---------------------------------------------------------------------------
// sklearn.preprocessing.quantile_transform (uniform output, axis 0) in place on a float64 row-major device matrix
hipError_t quantile_transform_device(double* dX, int64_t m, int64_t n, int64_t ld, const int64_t* fit_rows_dev, int64_t n_fit,
                                     const double* refs, const double* qs, int nq, hipStream_t s);

// ---------------------------------------------------------------------------
// nnls_matrix.hip: the device steps of Lawson-Hanson NNLS on A = X or X^T (float64 row-major X, never copied).
// Element t of column j of A is X[t * st_t + j * st_j]  (A = X: st_t = ld, st_j = 1;  A = X^T: st_t = 1, st_j = ld).
// ---------------------------------------------------------------------------
hipError_t nnlsm_gradient(const double* X, int64_t ld, int64_t rows, int64_t cols, bool transposed, const double* r, double* w,
                          double* wabs, hipStream_t s);                          // w = A^T r and wabs[j] = sum_t |a_tj r_t|, all unknowns
//   the free unknown with the largest w among those with w[j] > c1 * wabs[j] + c2 * babs[j]
hipError_t nnlsm_argmax(const double* w, const unsigned char* state, const double* wabs, const double* babs, double c1, double c2,
                        int64_t n, long long* out /* index, bits of w */, hipStream_t s);
hipError_t nnlsm_column(const double* X, int64_t st_t, int64_t st_j, int64_t len, int64_t j, double* q, hipStream_t s);
hipError_t nnlsm_dots(const double* X, int64_t st_t, int64_t st_j, int64_t len, const long long* P, int np, const double* q,
                      double* out, hipStream_t s);                               // out[a] = A(:, P[a]) . q
hipError_t nnlsm_residual(const double* X, int64_t st_t, int64_t st_j, int64_t len, const long long* P, int np, const double* sv,
                          const double* b, double* r, hipStream_t s);            // r = b - A_P s

// ---------------------------------------------------------------------------
// textio.hip: the numeric body of a delimited text matrix (pd.read_csv at script/prmf_runner.py:942)
// ---------------------------------------------------------------------------
constexpr int TEXT_FLAG_RAGGED = 1;        // = PRMF_TEXT_RAGGED ... (include/prmf_hip.h; checked by a static_assert in capi.hip)
constexpr int TEXT_FLAG_QUOTED = 2;
constexpr int TEXT_FLAG_NOT_NUMERIC = 4;
constexpr int TEXT_FLAG_WIDE_INT = 8;
constexpr int TEXT_FLAG_BARE_CR = 16;
int64_t text_blocks(int64_t nbytes);       // workgroups (= per-workgroup counts) for nbytes of text
//   counts[text_blocks(nbytes)]: separators (delim or '\n') per 4 KiB of text
hipError_t text_count(const unsigned char* text, int64_t nbytes, unsigned char delim, int64_t* counts, hipStream_t s);
//   exclusive sum of n_values counts into `base` (tmp == nullptr: only *tmp_bytes is set)
hipError_t text_scan(const int64_t* counts, int64_t* base, int64_t n_values, void* tmp, size_t* tmp_bytes, hipStream_t s);
//   sep_pos[j] = byte offset of separator j; TEXT_FLAG_RAGGED unless separator j is '\n' exactly when (j + 1) % fields_per_line == 0
hipError_t text_positions(const unsigned char* text, int64_t nbytes, unsigned char delim, const int64_t* block_base,
                          int64_t fields_per_line, int64_t* sep_pos, int* flags, hipStream_t s);
//   out[row][col - skip_first] of every field; name_spans[row] = [start, end) of field 0 when it is skipped;
//   col_not_int[c] = 1 if some field of data column c has a '.' or an exponent; first_bad = lowest offset of a refused field
hipError_t text_parse(const unsigned char* text, const int64_t* sep_pos, int64_t n_fields, int64_t fields_per_line, int skip_first,
                      double* out, int64_t* name_spans, int* col_not_int, int* flags, unsigned long long* first_bad, hipStream_t s);
//   host: a float64 matrix as DataFrame.to_csv(quoting=QUOTE_NONNUMERIC) writes it (repr floats); 0 or the errno of the failure
int text_write_matrix(const char* path, const char* header, const char* prefixes, const int64_t* prefix_off, const double* values,
                      int64_t rows, int64_t cols, int64_t ld, char sep, int n_threads);

}  // namespace prmf
