// Device code shared by gemm.hip (the sweep kernel's prologue) and update.hip (the stand-alone
// small kernels): the float64 reduction of per-workgroup Gram partials, the sparse pathway terms
// of the V update, and the per-column reductions the objective needs.
//
// Why a sweep has a prologue: between the two sweeps of an iteration sit a few microseconds of
// reduction work (U^T U / V^T V partials -> k x k float64, pathway terms gamma W v + delta (v+1)^-2,
// column norms of V).  As separate launches (`mid`, `post`, `finalize` in round 1) each of them cost
// its own latency chain plus a kernel boundary on the critical path: 26 us + 3 boundaries per
// iteration at 50000 x 20000, a third of the step at 5000 x 10000.  The work depends only on the
// epilogue BEFORE the sweep and is first read by the epilogue AFTER it, so the sweep's persistent
// workgroups do it in their first microseconds (each workgroup a small, fixed share; LDS scratch
// is the factor stage buffer the main loop has not touched yet) and the launches disappear.
//
// All reductions run in a fixed order (bit-identical run to run, and identical whichever kernel
// hosts them).
#pragma once
#include "kernels.h"

namespace prmf {

// deterministic block-wide sum of one double per thread (256 threads)
__device__ __forceinline__ double block_sum_256(double v, double* red /* [256] shared */) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  double r = red[0];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------
// G[e] = sum over the per-workgroup float partials, in float64.
// Job j (256 threads) finishes entries [8 j, 8 j + 8): thread (sub = tid >> 3, el = tid & 7) loads
// partials sub, sub + 32, ... of entry 8 j + el -- all of them in flight at once (nparts <= 1024:
// at most 32 independent loads, one memory round trip) -- and sums them in index order; the 32
// sub-sums are then added in index order.  The order depends on (nparts, entry) only.
// ---------------------------------------------------------------------------
constexpr int GRAM_JOB_ENTRIES = 8;
constexpr int GRAM_MAX_PARTS = 1024;
__host__ __device__ inline int gram_reduce_jobs(int kk2) { return (kk2 + GRAM_JOB_ENTRIES - 1) / GRAM_JOB_ENTRIES; }

__device__ __forceinline__ void gram_reduce_job(const float* __restrict__ partial, int nparts, int kk2,
                                                double* __restrict__ G, double* __restrict__ G_keep, int job,
                                                double* red /* [256] shared */) {
  const int tid = threadIdx.x, el = tid & 7, sub = tid >> 3;
  const int e = job * GRAM_JOB_ENTRIES + el;
  float v[GRAM_MAX_PARTS / 32];
#pragma unroll
  for (int j = 0; j < GRAM_MAX_PARTS / 32; ++j) {
    const int b = sub + 32 * j;
    v[j] = (e < kk2 && b < nparts) ? partial[(int64_t)b * kk2 + e] : 0.f;
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < GRAM_MAX_PARTS / 32; ++j) s += (double)v[j];
  red[sub * 8 + el] = s;
  __syncthreads();
  if (sub == 0 && e < kk2) {
    double t = red[el];
#pragma unroll
    for (int j = 1; j < 32; ++j) t += red[j * 8 + el];
    G[e] = t;
    if (G_keep) G_keep[e] = t;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// sparse pathway terms of the V update for one own column   (prmf_runner.py:431-438)
//   gnum = gamma * (W_k v_k) + delta * (v_k + 1)^-2   on the support S_k
//   gden = gamma * deg_k * v_k
// The column's support values are gathered once into LDS (`vs`), the CSR rows (support-local
// column indices) then read LDS only.
// ---------------------------------------------------------------------------
// `part` of `nsplit`: a large pathway (thousands of support genes) is shared by several workgroups -- each gathers the
// whole support into its LDS and walks every nsplit-th block of 256 rows (one workgroup per column walked 39 dependent
// rows per thread at |S| = 10000: +0.25 ms per iteration).
template <bool USE_LDS>
__device__ __forceinline__ void graph_terms_body(const PathwayLibDev& lib, const int32_t* __restrict__ active, int col0, int c,
                                                 const float* __restrict__ V, int k, const double* __restrict__ hyper,
                                                 float* __restrict__ gnum, float* __restrict__ gden, float* vs,
                                                 int part = 0, int nsplit = 1) {
  const int kk = col0 + c;
  const int p = active[kk];
  const int64_t s0 = lib.sup_ptr[p];
  const int ns = (int)(lib.sup_ptr[p + 1] - s0);
  const double gamma = hyper[0], delta = hyper[1];
  if (USE_LDS) {
    for (int s = threadIdx.x; s < ns; s += 256) vs[s] = V[(int64_t)lib.sup_idx[s0 + s] * k + kk];
    __syncthreads();
  }
  for (int s = part * 256 + threadIdx.x; s < ns; s += 256 * nsplit) {
    const double v = USE_LDS ? (double)vs[s] : (double)V[(int64_t)lib.sup_idx[s0 + s] * k + kk];
    double wv = 0.0;
    for (int64_t e = lib.row_ptr[s0 + s]; e < lib.row_ptr[s0 + s + 1]; ++e) {
      const int cl = lib.col_local[e];
      const double vc = USE_LDS ? (double)vs[cl] : (double)V[(int64_t)lib.sup_idx[s0 + cl] * k + kk];
      wv += lib.w[e] * vc;
    }
    const double vp1 = v + 1.0;
    gnum[(int64_t)c * lib.max_support + s] = (float)(gamma * wv + delta / (vp1 * vp1));
    gden[(int64_t)c * lib.max_support + s] = (float)(gamma * (lib.sup_deg[s0 + s] * v));
  }
  if (USE_LDS) __syncthreads();      // vs is reused by the caller's next job
}

// column c < k: sum of squares of column c of V_new;  c == k: <Q, V_new> -- from the V epilogue's
// per-wave float64 partials [nwaves][k + 1], thread-strided then a fixed tree
__device__ __forceinline__ void colcross_job(const double* __restrict__ vpartial, int nwaves, int k, int c,
                                             double* __restrict__ colnorm2, double* __restrict__ cross, double* red) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nwaves; b += 256) s += vpartial[(int64_t)b * (k + 1) + c];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    if (c < k) colnorm2[c] = s;
    else *cross = s;
  }
}

// The same two sums in the column-sharded multi-rank mode, from the partials every rank's own-column V update appended
// to its block of the all-gather: column c belongs to rank c / kc; <Q, V_new> is summed rank-major.
__device__ __forceinline__ void colcross_gathered_job(const SweepPrologue& p, int c, double* red) {
  if (c < p.k) {
    const double* part = reinterpret_cast<const double*>(p.gath + (int64_t)(c / p.gath_kc) * p.gath_stride + p.gath_off);
    double s = 0.0;
    for (int b = threadIdx.x; b < p.gath_nblk; b += 256) s += part[(int64_t)b * (p.gath_kc + 1) + (c % p.gath_kc)];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) p.colnorm2[c] = s;
    return;
  }
  double tot = 0.0;                    // per rank first, then the ranks in order: the order of the per-iteration form
  for (int r = 0; r < p.gath_nranks; ++r) {
    const double* part = reinterpret_cast<const double*>(p.gath + (int64_t)r * p.gath_stride + p.gath_off);
    double s = 0.0;
    for (int b = threadIdx.x; b < p.gath_nblk; b += 256) s += part[(int64_t)b * (p.gath_kc + 1) + p.gath_kc];
    tot += block_sum_256(s, red);
  }
  if (threadIdx.x == 0) *p.cross = tot;
}

// The prologue a sweep launch runs before its main loop (see the header comment).  `lds` is scratch
// of at least 2 KiB + 4 * max_support bytes for the LDS form of the pathway terms (else global gathers).
__device__ __forceinline__ void run_sweep_prologue(const SweepPrologue& p, char* lds, int lds_bytes) {
  double* red = reinterpret_cast<double*>(lds);
  const int kk2 = p.k * p.k;
  const int nj = gram_reduce_jobs(kk2);
  for (int j = blockIdx.x; j < nj; j += gridDim.x) gram_reduce_job(p.gram_partial, p.nparts, kk2, p.G, p.G_keep, j, red);
  if (p.kind == PROLOGUE_POST) {
    // workgroups from the far end of the grid first: at small k they are the ones without a Gram job
    if (p.vpartial)
      for (int c = (int)(gridDim.x - 1 - blockIdx.x); c <= p.k; c += gridDim.x)
        colcross_job(p.vpartial, p.nwaves, p.k, c, p.colnorm2, p.cross, red);
    else if (p.gath)                 // column-sharded multi-rank mode: the sums arrived with the all-gather
      for (int c = (int)(gridDim.x - 1 - blockIdx.x); c <= p.k; c += gridDim.x) colcross_gathered_job(p, c, red);
    if (p.gu_keep)
      for (int e = (int)(blockIdx.x * 256 + threadIdx.x); e < kk2; e += (int)(gridDim.x * 256)) p.gu_keep[e] = p.gu_src[e];
  } else if (p.kind == PROLOGUE_MID) {
    float* vs = reinterpret_cast<float*>(lds + 2048);
    const bool fits = lds_bytes >= 2048 + (int)sizeof(float) * p.lib.max_support;
    for (int c = (int)(gridDim.x - 1 - blockIdx.x); c < p.ncols; c += gridDim.x) {
      if (fits) graph_terms_body<true>(p.lib, p.active, p.col0, c, p.V, p.k, p.hyper, p.gnum, p.gden, vs);
      else graph_terms_body<false>(p.lib, p.active, p.col0, c, p.V, p.k, p.hyper, p.gnum, p.gden, vs);
    }
  }
}

}  // namespace prmf
