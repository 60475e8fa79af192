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

// deterministic sum of one double per thread over a group of 256 threads (`tid` = index in the group, `red` = the
// group's 256 doubles of LDS).  Every thread of the WORKGROUP has to call it: the barriers are workgroup barriers (a
// 512-thread sweep workgroup runs two groups side by side, prologue below).
__device__ __forceinline__ double block_sum_256(double v, double* red /* [256] shared */, int tid = (int)threadIdx.x) {
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
// Job j (256 threads) finishes entries [E j, E j + E): thread (sub = tid / E, el = tid % E) loads partials sub,
// sub + 256 / E, ... of entry E j + el -- 32 of them in flight at once -- and sums them in index order; the 256 / E
// sub-sums are then added in index order.  The order depends on (nparts, entry, E) only.
// E = 8 at k <= 64 (512 short jobs at k = 64: one memory round trip each).  E = 32 at k > 64: with 8 entries a wave's load
// touches 8 x 32 B of eight 64-KB-apart partials -- a quarter of every 128-B line; the 2048 jobs of k = 128 took 28 us as
// a launch of their own (profiles/r03_c5_kernel_stats.csv), most of it over-fetch.
// ---------------------------------------------------------------------------
constexpr int GRAM_MAX_PARTS = 1024;
__host__ __device__ inline int gram_job_entries(int k) { return k > 64 ? 32 : 8; }
__host__ __device__ inline int gram_reduce_jobs(int k) { return (k * k + gram_job_entries(k) - 1) / gram_job_entries(k); }

// `job` past the last one: the same barriers, no loads, no store (the idle half of a 512-thread workgroup)
template <int E>
__device__ __forceinline__ void gram_reduce_job_e(const float* __restrict__ partial, int nparts, int kk2,
                                                  double* __restrict__ G, double* __restrict__ G_keep, int job,
                                                  double* red /* [256] shared */, int tid) {
  constexpr int NSUB = 256 / E;
  const int el = tid % E, sub = tid / E;
  const int e = job * E + el;
  double s = 0.0;
#pragma unroll 1
  for (int j0 = 0; j0 * NSUB < nparts; j0 += 32) {
    float v[32];
    // BRANCH-FREE: indices are clamped and the values masked afterwards.  Written as `cond ? partial[..] : 0` every load
    // became its own exec-masked block with a wait for it behind it -- the "32 loads in flight" were 13 dependent memory
    // round trips per job at C3 (391 parts), 8-10 us in front of every sweep (profiles/r03_gram_partial_layout.txt).
    const int ec = e < kk2 ? e : kk2 - 1;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int b = sub + NSUB * (j0 + j);
      v[j] = partial[(int64_t)(b < nparts ? b : nparts - 1) * kk2 + ec];
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) s += (e < kk2 && sub + NSUB * (j0 + j) < nparts) ? (double)v[j] : 0.0;
  }
  red[sub * E + el] = s;
  __syncthreads();
  if (sub == 0 && e < kk2) {
    double t = red[el];
#pragma unroll
    for (int j = 1; j < NSUB; ++j) t += red[j * E + el];
    G[e] = t;
    if (G_keep) G_keep[e] = t;
  }
  __syncthreads();
}

// WIDE_K: -1 = decide at run time (the stand-alone kernels); 0 / 1 = the caller knows k <= 64 / k > 64 at compile time
// (the sweep kernels: only one of the two forms is then compiled into them -- both together cost every sweep kernel its
// register headroom: 256 VGPRs and 12 spilled where the k = 64 sweep had 250 and none)
template <int WIDE_K = -1>
__device__ __forceinline__ void gram_reduce_job(const float* __restrict__ partial, int nparts, int k,
                                                double* __restrict__ G, double* __restrict__ G_keep, int job,
                                                double* red /* [256] shared */, int tid = (int)threadIdx.x) {
  if constexpr (WIDE_K == 1) gram_reduce_job_e<32>(partial, nparts, k * k, G, G_keep, job, red, tid);
  else if constexpr (WIDE_K == 0) gram_reduce_job_e<8>(partial, nparts, k * k, G, G_keep, job, red, tid);
  else if (k > 64) gram_reduce_job_e<32>(partial, nparts, k * k, G, G_keep, job, red, tid);
  else gram_reduce_job_e<8>(partial, nparts, k * k, G, G_keep, job, red, tid);
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
// per-wave float64 partials [nwaves][k + 1], thread-strided then a fixed tree.  c > k: barriers only.
__device__ __forceinline__ void colcross_job(const double* __restrict__ vpartial, int nwaves, int k, int c,
                                             double* __restrict__ colnorm2, double* __restrict__ cross, double* red, int tid) {
  double s = 0.0;
  if (c <= k) {
    // 8 rows per thread in flight (clamped, masked; same order of additions as one row at a time)
    for (int b0 = tid; b0 < nwaves; b0 += 8 * 256) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = b0 + 256 * j;
        v[j] = vpartial[(int64_t)(b < nwaves ? b : nwaves - 1) * (k + 1) + c];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (b0 + 256 * j < nwaves) ? v[j] : 0.0;
    }
  }
  s = block_sum_256(s, red, tid);
  if (tid == 0 && c <= k) {
    if (c < k) colnorm2[c] = s;
    else *cross = s;
  }
}

// The same two sums in the column-sharded multi-rank mode, from the partials every rank's own-column V update appended
// to its block of the all-gather: column c belongs to rank c / kc; <Q, V_new> is summed per rank first, then the ranks
// in order (the order of the per-iteration form).
__device__ __forceinline__ void colcross_gathered_job(const SweepPrologue& p, int c, double* red, int tid) {
  if (c != p.k) {
    double s = 0.0;
    if (c < p.k) {
      const double* part = reinterpret_cast<const double*>(p.gath + (int64_t)(c / p.gath_kc) * p.gath_stride + p.gath_off);
      for (int b = tid; b < p.gath_nblk; b += 256) s += part[(int64_t)b * (p.gath_kc + 1) + (c % p.gath_kc)];
    }
    s = block_sum_256(s, red, tid);
    if (tid == 0 && c < p.k) p.colnorm2[c] = s;
    // the other group of a 512-thread workgroup may be summing <Q, V_new>: keep the barrier counts equal
    for (int r = 1; r < p.gath_nranks; ++r) (void)block_sum_256(0.0, red, tid);
    return;
  }
  double tot = 0.0;
  for (int r = 0; r < p.gath_nranks; ++r) {
    const double* part = reinterpret_cast<const double*>(p.gath + (int64_t)r * p.gath_stride + p.gath_off);
    double s = 0.0;
    for (int b = tid; b < p.gath_nblk; b += 256) s += part[(int64_t)b * (p.gath_kc + 1) + p.gath_kc];
    tot += block_sum_256(s, red, tid);
  }
  if (tid == 0) *p.cross = tot;
}

// The side work a sweep launch runs before its main loop (see the header comment), or a launch of its own.  The workgroup
// is one or two groups of 256 threads (the 8-wave sweep shape); a group does one job at a time, both groups in step.
// `lds`: 2 KiB of scratch per group.
template <int WIDE_K = -1>
__device__ __forceinline__ void run_sweep_prologue(const SweepPrologue& p, char* lds, int lds_bytes) {
  (void)lds_bytes;
  const int groups = (int)(blockDim.x >> 8), grp = (int)(threadIdx.x >> 8), tid = (int)(threadIdx.x & 255);
  double* red = reinterpret_cast<double*>(lds) + 256 * grp;
  const int nj = gram_reduce_jobs(p.k);
  for (int j0 = (int)blockIdx.x * groups; j0 < nj; j0 += (int)gridDim.x * groups)
    gram_reduce_job<WIDE_K>(p.gram_partial, p.nparts, p.k, p.G, p.G_keep, j0 + grp < nj ? j0 + grp : 0x3fffffff / 32, red, tid);
  if (p.kind == PROLOGUE_POST) {
    // workgroups from the far end of the grid first: at small k they are the ones without a Gram job
    if (p.vpartial || p.gath)
      for (int c0 = (int)(gridDim.x - 1 - blockIdx.x) * groups; c0 <= p.k; c0 += (int)gridDim.x * groups) {
        const int c = c0 + grp <= p.k ? c0 + grp : p.k + 1;
        if (p.vpartial) colcross_job(p.vpartial, p.nwaves, p.k, c, p.colnorm2, p.cross, red, tid);
        else colcross_gathered_job(p, c, red, tid);
      }
    if (p.gu_keep)
      for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < p.k * p.k; e += (int)(gridDim.x * blockDim.x)) p.gu_keep[e] = p.gu_src[e];
  }
}

}  // namespace prmf
