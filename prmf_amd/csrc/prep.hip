// Per-gene quantile normalisation of X on the GPU -- the `quantile_transform(X)` call of the reference's CLI
// (script/prmf_runner.py:1019-1020: sklearn.preprocessing.quantile_transform, n_quantiles = 1000, uniform output,
// axis 0).  On the host it is an np.apply_along_axis loop over the genes (one partition + two np.interp per
// column): minutes at 50000 x 20000.  Here: column blocks are gathered into contiguous segments, sorted with a
// segmented radix sort (rocprim::segmented_radix_sort_keys, a library primitive: one-time preprocessing, not the hot path), the n_quantiles order statistics are interpolated exactly as
// numpy's `linear` percentile does, and every element is mapped with the mean of an ascending and a descending
// np.interp -- all in float64, operation for operation, so that the result equals scikit-learn's to the last bits:
//
//   references = linspace(0, 1, nq)                          nq = min(n_quantiles, rows used for the fit)
//   quantiles[:, j] = nanpercentile(X[:, j], references * 100)  (virtual index (n - 1) q, numpy's two-sided lerp)
//   quantiles = maximum.accumulate(quantiles)
//   y = 0.5 * (interp(x, quantiles, references) - interp(-x, -quantiles[::-1], -references[::-1]))
//   y = 1 where x == quantiles[-1];  y = 0 where x == quantiles[0]        (in that order)
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "kernels.h"

// numpy evaluates a*b + c as two rounded operations: no fused multiply-add in this file
#pragma clang fp contract(off)

namespace prmf {

namespace {

constexpr int TT = 32;   // transpose tile

// cols[cc][r] = X[rows[r] (or r)][c0 + cc]   for cc < ncols, r < nfit; NaNs are made positive (they then sort last)
__global__ void __launch_bounds__(256)
gather_cols_kernel(const double* __restrict__ X, int64_t ld, const int64_t* __restrict__ rows, int64_t nfit, int64_t c0,
                   int ncols, double* __restrict__ cols) {
  __shared__ double tile[TT][TT + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int64_t r0 = (int64_t)blockIdx.x * TT;
  const int cb = blockIdx.y * TT;
#pragma unroll
  for (int i = 0; i < TT; i += 8) {
    const int64_t r = r0 + ty + i;
    const int cc = cb + tx;
    double v = 0.0;
    if (r < nfit && cc < ncols) {
      const int64_t row = rows ? rows[r] : r;
      v = X[row * ld + c0 + cc];
      if (v != v) v = fabs(v);
    }
    tile[ty + i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TT; i += 8) {
    const int cc = cb + ty + i;
    const int64_t r = r0 + tx;
    if (r < nfit && cc < ncols) cols[(int64_t)cc * nfit + r] = tile[tx][ty + i];
  }
}

// numpy's percentile, method "linear", on one sorted column of `cnt` valid values
__device__ __forceinline__ double np_percentile_sorted(const double* __restrict__ a, int64_t cnt, double q) {
  const double n = (double)cnt;
  const double vi = (n - 1.0) * q;                                   // method 'linear': (n - 1) * quantiles
  if (vi >= n - 1.0) return a[cnt - 1];                               // above bounds: the maximum
  if (vi < 0.0) return a[0];
  const double prev = floor(vi);
  const double t = vi - prev;                                         // gamma
  const int64_t ip = (int64_t)prev;
  const double lo = a[ip], hi = a[ip + 1];
  const double d = hi - lo;                                           // _lerp
  double r = lo + d * t;
  if (t >= 0.5) r = hi - d * (1.0 - t);
  return r;
}

// Q[cc][qi] for the block's columns; a column of only NaNs gives NaN quantiles
__global__ void __launch_bounds__(256)
quantiles_kernel(const double* __restrict__ sorted, int64_t nfit, int ncols, const double* __restrict__ qs, int nq,
                 double* __restrict__ Q) {
  const int cc = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (cc >= ncols || qi >= nq) return;
  const double* a = sorted + (int64_t)cc * nfit;
  // valid values = everything before the first NaN (NaNs sort last): binary search
  int64_t lo = 0, hi = nfit;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] != a[mid]) hi = mid; else lo = mid + 1;
  }
  const int64_t cnt = lo;
  Q[(int64_t)cc * nq + qi] = cnt > 0 ? np_percentile_sorted(a, cnt, qs[qi]) : __longlong_as_double(0x7ff8000000000000LL);
}

// np.maximum.accumulate along the quantile index (NaN propagates as in numpy's maximum)
__global__ void __launch_bounds__(64) accumulate_max_kernel(double* __restrict__ Q, int ncols, int nq) {
  const int cc = blockIdx.x * 64 + threadIdx.x;
  if (cc >= ncols) return;
  double* q = Q + (int64_t)cc * nq;
  double run = q[0];
  for (int i = 1; i < nq; ++i) {
    const double v = q[i];
    run = (run != run || v != v) ? (run != run ? run : v) : (v > run ? v : run);
    q[i] = run;
  }
}

// np.interp(x, xp, fp) with xp[i] = sx * XP[rev ? n-1-i : i], fp likewise from FP (sx = -1 with rev for the descending pass)
template <bool REV>
__device__ __forceinline__ double np_interp(double x, const double* __restrict__ XP, const double* __restrict__ FP, int n) {
  auto xp = [&](int i) { return REV ? -XP[n - 1 - i] : XP[i]; };
  auto fp = [&](int i) { return REV ? -FP[n - 1 - i] : FP[i]; };
  int lo = 0, hi = n;                                   // upper bound: first index with xp > x
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (x >= xp(mid)) lo = mid + 1; else hi = mid;
  }
  const int j = lo - 1;
  if (j < 0) return fp(0);
  if (j >= n - 1) return fp(n - 1);
  const double xj = xp(j);
  if (xj == x) return fp(j);
  const double slope = (fp(j + 1) - fp(j)) / (xp(j + 1) - xj);
  double r = slope * (x - xj) + fp(j);
  if (r != r) {
    r = slope * (x - xp(j + 1)) + fp(j + 1);
    if (r != r && fp(j) == fp(j + 1)) r = fp(j);
  }
  return r;
}

// in place on X[:, c0 : c0 + ncols]; thread (row, column): consecutive threads = consecutive columns
__global__ void __launch_bounds__(256)
transform_kernel(double* __restrict__ X, int64_t ld, int64_t m, int64_t c0, int ncols, const double* __restrict__ Q,
                 const double* __restrict__ refs, int nq) {
  const int cc = blockIdx.x * 64 + (threadIdx.x & 63);
  if (cc >= ncols) return;
  const double* q = Q + (int64_t)cc * nq;
  const double qlo = q[0], qhi = q[nq - 1];
  for (int64_t r = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6); r < m; r += (int64_t)gridDim.y * 4) {
    double* px = X + r * ld + c0 + cc;
    const double x = *px;
    if (x != x) continue;                               // NaN stays NaN
    double y = 0.5 * (np_interp<false>(x, q, refs, nq) - np_interp<true>(-x, q, refs, nq));
    if (x == qhi) y = 1.0;
    if (x == qlo) y = 0.0;
    *px = y;
  }
}

}  // namespace

// dX: m x n float64 row-major (leading dimension ld), transformed in place.  fit_rows_dev (nullable): the rows the
// quantiles are fitted on (scikit-learn's subsample when m > 1e5), n_fit of them; refs / qs: device arrays of nq doubles
// (references and references*100/100).  Scratch is allocated per call.
hipError_t quantile_transform_device(double* dX, int64_t m, int64_t n, int64_t ld, const int64_t* fit_rows_dev, int64_t n_fit,
                                     const double* refs, const double* qs, int nq, hipStream_t s) {
  if (m <= 0 || n <= 0) return hipSuccess;
  // column blocks of <= 512 MiB per buffer, at most 2^31-1 keys per sort call
  int64_t cb = (int64_t)(512u << 20) / (n_fit * 8);
  if (cb > n) cb = n;
  if (cb * n_fit > 0x7fffff00LL) cb = 0x7fffff00LL / n_fit;
  if (cb < 1) cb = 1;
  double *cols = nullptr, *sorted = nullptr, *Q = nullptr;
  int* offsets = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0;
  hipError_t e = hipSuccess;
  auto cleanup = [&]() {
    (void)hipFree(cols); (void)hipFree(sorted); (void)hipFree(Q); (void)hipFree(offsets); (void)hipFree(temp);
  };
#define PQ(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
  PQ(hipMalloc((void**)&cols, (size_t)cb * n_fit * 8));
  PQ(hipMalloc((void**)&sorted, (size_t)cb * n_fit * 8));
  PQ(hipMalloc((void**)&Q, (size_t)cb * nq * 8));
  PQ(hipMalloc((void**)&offsets, (size_t)(cb + 1) * 4));
  {
    std::vector<int> off((size_t)cb + 1);
    for (int64_t i = 0; i <= cb; ++i) off[(size_t)i] = (int)(i * n_fit);
    PQ(hipMemcpyAsync(offsets, off.data(), off.size() * 4, hipMemcpyHostToDevice, s));
    PQ(hipStreamSynchronize(s));
  }
  PQ(rocprim::segmented_radix_sort_keys(nullptr, temp_bytes, cols, sorted, (unsigned)(cb * n_fit), (unsigned)cb, offsets, offsets + 1,
                                        0, 64, s));
  PQ(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
  for (int64_t c0 = 0; c0 < n; c0 += cb) {
    const int nc = (int)(n - c0 < cb ? n - c0 : cb);
    gather_cols_kernel<<<dim3((unsigned)((n_fit + TT - 1) / TT), (unsigned)((nc + TT - 1) / TT)), 256, 0, s>>>(
        dX, ld, fit_rows_dev, n_fit, c0, nc, cols);
    PQ(hipGetLastError());
    PQ(rocprim::segmented_radix_sort_keys(temp, temp_bytes, cols, sorted, (unsigned)((int64_t)nc * n_fit), (unsigned)nc, offsets, offsets + 1,
                                          0, 64, s));
    quantiles_kernel<<<dim3((unsigned)((nq + 255) / 256), (unsigned)nc), 256, 0, s>>>(sorted, n_fit, nc, qs, nq, Q);
    accumulate_max_kernel<<<(nc + 63) / 64, 64, 0, s>>>(Q, nc, nq);
    int64_t rb = (m + 3) / 4;
    if (rb > 4096) rb = 4096;
    transform_kernel<<<dim3((unsigned)((nc + 63) / 64), (unsigned)rb), 256, 0, s>>>(dX, ld, m, c0, nc, Q, refs, nq);
    PQ(hipGetLastError());
  }
  PQ(hipStreamSynchronize(s));
#undef PQ
  cleanup();
  return hipSuccess;
}

}  // namespace prmf
