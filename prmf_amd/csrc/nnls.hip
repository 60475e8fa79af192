// Batched non-negative least squares for the held-out reconstruction error
// (measure_cv_performance, prmf/__init__.py:768-798):
//
//     err_s = min_{u >= 0} || x_s - u V^T ||_2 / || x_s ||_2       for every held-out sample s
//
// The reference calls scipy.optimize.nnls (Lawson-Hanson active set) once per sample
// on the n x k matrix V.  V is shared by all samples, so here the problem is moved to
// the normal equations once:  G = V^T V (k x k), h_s = V^T x_s (both accumulated in float64:
// the products of f32/bf16 X and f32 V are exact in double), and each sample runs the same
// active-set method on (G, h_s):
//
//     P = {} ; c = 0
//     repeat: w = h - G c ; t = argmax_{j not in P} w_j ; stop when w_t <= tol
//             P += t ; solve G_PP s = h_P (Cholesky, grown by one row per added index)
//             while min s <= 0: step c towards s until the first coordinate hits 0,
//                               drop the zeros from P, refactor, solve again
//             c_P = s
//     err^2 = ||x||^2 - 2 c.h + c^T G c
//
// The minimiser of a strictly convex NNLS problem is unique, so any exact method gives
// scipy's answer up to rounding.  One wavefront per sample; the Cholesky factor lives
// in LDS (k (k+1) doubles), vectors are distributed over the lanes (index j -> lane
// j % 64, two per lane at k = 128).
#include "kernels.h"

namespace prmf {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(uint16_t bf16_bits) { return __uint_as_float((unsigned)bf16_bits << 16); }

template <typename T>
__global__ void __launch_bounds__(256) row_normsq_kernel(const T* __restrict__ X, int64_t ld, int64_t m, int64_t n,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const T* x = X + row * ld;
  double acc = 0.0;
  for (int64_t j = lane; j < n; j += 64) {
    const double v = (double)to_f32(x[j]);
    acc += v * v;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc;
}

// H = X V in float64 (products of f32/bf16 X and f32 V are exact in double): 16 samples x all k
// columns per workgroup, 64-gene chunks of X and V staged in LDS.  thread t -> sample t/16, columns
// t%16 + 16 u.  Not a hot loop (once per run, after the solve) -- the error is obtained from the
// normal equations, where a float32 h would be amplified by ||x||^2 / err^2.
template <typename T>
__global__ void __launch_bounds__(256)
xv_f64_kernel(const T* __restrict__ X, int64_t ld, int64_t m, int64_t n, const float* __restrict__ V, int k,
              double* __restrict__ H) {
  __shared__ float xs[16][65];
  __shared__ float vs[64][MAX_KP];
  const int tid = threadIdx.x, r = tid >> 4, cg = tid & 15;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  double acc[MAX_KP / 16];
#pragma unroll
  for (int u = 0; u < MAX_KP / 16; ++u) acc[u] = 0.0;
  for (int64_t i0 = 0; i0 < n; i0 += 64) {
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += 256) {
      const int rr = e >> 6, ii = e & 63;
      const int64_t row = row0 + rr, col = i0 + ii;
      xs[rr][ii] = (row < m && col < n) ? to_f32(X[row * ld + col]) : 0.f;
    }
    for (int e = tid; e < 64 * k; e += 256) {
      const int ii = e / k, cc = e - ii * k;
      vs[ii][cc] = (i0 + ii < n) ? V[(i0 + ii) * k + cc] : 0.f;
    }
    __syncthreads();
    for (int ii = 0; ii < 64; ++ii) {
      const double x = (double)xs[r][ii];
#pragma unroll
      for (int u = 0; u < MAX_KP / 16; ++u)
        if (cg + 16 * u < k) acc[u] += x * (double)vs[ii][cg + 16 * u];
    }
  }
  const int64_t row = row0 + r;
  if (row < m)
#pragma unroll
    for (int u = 0; u < MAX_KP / 16; ++u)
      if (cg + 16 * u < k) H[row * k + cg + 16 * u] = acc[u];
}

// partial[b][a*k + c] = sum over the block's 256 genes of V[i][a] V[i][c]   (float64)
__global__ void __launch_bounds__(256)
gram_f64_partial_kernel(const float* __restrict__ V, int64_t n, int k, double* __restrict__ partial) {
  __shared__ float vs[64][MAX_KP + 1];
  const int tid = threadIdx.x;
  const int kk2 = k * k;
  const int64_t i_begin = (int64_t)blockIdx.x * 256;
  // thread t owns entries e = t, t + 256, ... of the k x k matrix (<= 64 of them)
  double acc[MAX_KP * MAX_KP / 256];
#pragma unroll
  for (int u = 0; u < MAX_KP * MAX_KP / 256; ++u) acc[u] = 0.0;
  for (int64_t i0 = i_begin; i0 < i_begin + 256 && i0 < n; i0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * k; e += 256) {
      const int ii = e / k, cc = e - ii * k;
      vs[ii][cc] = (i0 + ii < n) ? V[(i0 + ii) * k + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAX_KP * MAX_KP / 256; ++u) {
      const int e = tid + 256 * u;
      if (e < kk2) {
        const int a = e / k, c = e - a * k;
        double t = 0.0;
        for (int ii = 0; ii < 64; ++ii) t += (double)vs[ii][a] * (double)vs[ii][c];
        acc[u] += t;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MAX_KP * MAX_KP / 256; ++u) {
    const int e = tid + 256 * u;
    if (e < kk2) partial[(int64_t)blockIdx.x * kk2 + e] = acc[u];
  }
}

__global__ void __launch_bounds__(256)
gram_f64_reduce_kernel(const double* __restrict__ partial, int nparts, int kk2, double* __restrict__ G) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kk2) return;
  double t = 0.0;
  for (int b = 0; b < nparts; ++b) t += partial[(int64_t)b * kk2 + e];
  G[e] = t;
}

struct NnlsShared {
  double* L;     // [k][ld]  Cholesky rows in passive-list order
  double* h;     // [k]
  double* c;     // [k]      current iterate (by variable index)
  double* t;     // [k]      scratch / solution in passive-list order
  int* idx;      // [k]      passive list
  int* flag;     // [k]      0 free, 1 passive, 2 rejected in this round, 3 dropped (numerically dependent)
};

// forward substitution L y = b in place on sh.t[0..q)
__device__ __forceinline__ void fwd_solve(const NnlsShared& sh, int ld, int q, int lane) {
  for (int j = 0; j < q; ++j) {
    __syncthreads();
    const double yj = sh.t[j] / sh.L[j * ld + j];
    __syncthreads();
    if (lane == (j & 63)) sh.t[j] = yj;
    for (int i = lane; i < q; i += 64)
      if (i > j) sh.t[i] -= sh.L[i * ld + j] * yj;
  }
  __syncthreads();
}

// back substitution L^T s = y in place on sh.t[0..q)
__device__ __forceinline__ void bwd_solve(const NnlsShared& sh, int ld, int q, int lane) {
  for (int j = q - 1; j >= 0; --j) {
    __syncthreads();
    const double sj = sh.t[j] / sh.L[j * ld + j];
    __syncthreads();
    if (lane == (j & 63)) sh.t[j] = sj;
    for (int i = lane; i < j; i += 64) sh.t[i] -= sh.L[j * ld + i] * sj;
  }
  __syncthreads();
}

// row q of the Cholesky factor for variable v appended after idx[0..q): false if G_PP would
// become numerically singular
__device__ __forceinline__ bool chol_append(const NnlsShared& sh, const double* __restrict__ G, int k, int ld, int q, int v,
                                            int lane) {
  __syncthreads();
  for (int j = lane; j < q; j += 64) sh.t[j] = G[(int64_t)sh.idx[j] * k + v];
  fwd_solve(sh, ld, q, lane);
  double ss = 0.0;
  for (int j = lane; j < q; j += 64) {
    const double r = sh.t[j];
    sh.L[q * ld + j] = r;
    ss += r * r;
  }
  ss = wave_sum(ss);
  const double gvv = G[(int64_t)v * k + v];
  const double d2 = gvv - ss;
  if (!(d2 > 1e-13 * gvv)) return false;
  if (lane == 0) sh.L[q * ld + q] = sqrt(d2);
  __syncthreads();
  return true;
}

__global__ void __launch_bounds__(64)
nnls_kernel(const double* __restrict__ G, const double* __restrict__ H, const double* __restrict__ xx, int64_t m, int k,
            double tol_rel, int max_iter, double* __restrict__ err, int* __restrict__ n_not_converged) {
  extern __shared__ double smem[];
  const int ld = k + 1;
  NnlsShared sh;
  sh.L = smem;
  sh.h = sh.L + (size_t)k * ld;
  sh.c = sh.h + k;
  sh.t = sh.c + k;
  sh.idx = reinterpret_cast<int*>(sh.t + k);
  sh.flag = sh.idx + k;
  const int lane = threadIdx.x;

  for (int64_t s = blockIdx.x; s < m; s += gridDim.x) {
    __syncthreads();
    double hmax = 0.0;
    for (int j = lane; j < k; j += 64) {
      const double hj = H[s * k + j];
      sh.h[j] = hj;
      sh.c[j] = 0.0;
      sh.flag[j] = 0;
      hmax = fmax(hmax, fabs(hj));
    }
    hmax = wave_max(hmax);
    const double tol = tol_rel * hmax;
    int p = 0, iter = 0;
    bool converged = false;
    __syncthreads();

    while (iter < max_iter) {
      // dual vector on the free variables and its arg max (lowest index wins ties)
      double wbest = -1.0;
      int jbest = 0x7fffffff;
      for (int j = lane; j < k; j += 64) {
        if (sh.flag[j] != 0) continue;
        double w = sh.h[j];
        for (int q = 0; q < p; ++q) {
          const int v = sh.idx[q];
          w -= G[(int64_t)j * k + v] * sh.c[v];
        }
        if (w > wbest) { wbest = w; jbest = j; }
      }
      const double wmax = wave_max(wbest);
      if (!(wmax > tol)) { converged = true; break; }
      const int t = wave_min_int(wbest == wmax ? jbest : 0x7fffffff);

      if (!chol_append(sh, G, k, ld, p, t, lane)) {
        if (lane == 0) sh.flag[t] = 3;
        __syncthreads();
        continue;
      }
      if (lane == 0) { sh.idx[p] = t; sh.flag[t] = 1; }
      ++p;
      __syncthreads();

      bool first = true, accepted = false;
      while (p > 0) {
        ++iter;
        for (int q = lane; q < p; q += 64) sh.t[q] = sh.h[sh.idx[q]];
        fwd_solve(sh, ld, p, lane);
        bwd_solve(sh, ld, p, lane);
        if (first && !(sh.t[p - 1] > 0.0)) {
          // the incoming variable would go negative: leave it out of this round (Lawson-Hanson step 6')
          --p;
          __syncthreads();
          if (lane == 0) sh.flag[t] = 2;
          __syncthreads();
          break;
        }
        first = false;
        double smin = 1e300;
        for (int q = lane; q < p; q += 64) smin = fmin(smin, sh.t[q]);
        smin = wave_min(smin);
        if (smin > 0.0) {
          for (int q = lane; q < p; q += 64) sh.c[sh.idx[q]] = sh.t[q];
          accepted = true;
          __syncthreads();
          break;
        }
        // step towards s until the first passive coordinate reaches zero
        double a = 2.0;
        int qa = 0x7fffffff;
        for (int q = lane; q < p; q += 64) {
          const double sq = sh.t[q];
          if (sq <= 0.0) {
            const double cq = sh.c[sh.idx[q]];
            const double r = cq / (cq - sq);
            if (r < a) { a = r; qa = q; }
          }
        }
        const double alpha = wave_min(a);
        const int qstar = wave_min_int(a == alpha ? qa : 0x7fffffff);
        __syncthreads();
        for (int q = lane; q < p; q += 64) {
          const int v = sh.idx[q];
          double cn = sh.c[v] + alpha * (sh.t[q] - sh.c[v]);
          if (q == qstar || !(cn > 0.0)) {
            cn = 0.0;
            sh.flag[v] = 0;
          }
          sh.c[v] = cn;
        }
        __syncthreads();
        // compact the passive list (order kept) and refactor
        if (lane == 0) {
          int np = 0;
          for (int q = 0; q < p; ++q) {
            const int v = sh.idx[q];
            if (sh.flag[v] == 1) sh.idx[np++] = v;
          }
          sh.t[0] = (double)np;
        }
        __syncthreads();
        const int np = (int)sh.t[0];
        __syncthreads();
        p = 0;
        for (int q = 0; q < np; ++q) {
          const int v = sh.idx[q];
          if (chol_append(sh, G, k, ld, p, v, lane)) {
            if (lane == 0) sh.idx[p] = v;
            ++p;
          } else {
            if (lane == 0) { sh.flag[v] = 3; sh.c[v] = 0.0; }
          }
          __syncthreads();
        }
        accepted = true;
      }
      if (accepted) {
        for (int j = lane; j < k; j += 64)
          if (sh.flag[j] == 2) sh.flag[j] = 0;
      }
      __syncthreads();
    }

    // residual through the normal equations, float64
    double acc = 0.0;
    for (int j = lane; j < k; j += 64) {
      const double cj = sh.c[j];
      if (cj != 0.0) {
        double gc = 0.0;
        for (int q = 0; q < p; ++q) {
          const int v = sh.idx[q];
          gc += G[(int64_t)j * k + v] * sh.c[v];
        }
        acc += cj * (gc - 2.0 * sh.h[j]);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const double x2 = xx[s];
      const double r2 = fmax(x2 + acc, 0.0);
      err[s] = sqrt(r2) / sqrt(x2);          // 0/0 -> nan, as numpy gives for an all-zero sample
      if (!converged) atomicAdd(n_not_converged, 1);
    }
  }
}

}  // namespace

hipError_t launch_row_normsq(int x_dtype, const void* X, int64_t ld, int64_t m, int64_t n, double* out, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((m + 3) / 4);
  if (x_dtype == 1)
    row_normsq_kernel<uint16_t><<<grid, 256, 0, s>>>((const uint16_t*)X, ld, m, n, out);
  else
    row_normsq_kernel<float><<<grid, 256, 0, s>>>((const float*)X, ld, m, n, out);
  return hipGetLastError();
}

hipError_t launch_xv_f64(int x_dtype, const void* X, int64_t ld, int64_t m, int64_t n, const float* V, int k, double* H,
                         hipStream_t s) {
  if (m <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((m + 15) / 16);
  if (x_dtype == 1)
    xv_f64_kernel<uint16_t><<<grid, 256, 0, s>>>((const uint16_t*)X, ld, m, n, V, k, H);
  else
    xv_f64_kernel<float><<<grid, 256, 0, s>>>((const float*)X, ld, m, n, V, k, H);
  return hipGetLastError();
}

int gram_f64_blocks(int64_t n) { return (int)((n + 255) / 256); }

hipError_t launch_gram_f64(const float* V, int64_t n, int k, double* partial, double* G, hipStream_t s) {
  const int nb = gram_f64_blocks(n);
  gram_f64_partial_kernel<<<nb, 256, 0, s>>>(V, n, k, partial);
  gram_f64_reduce_kernel<<<(k * k + 255) / 256, 256, 0, s>>>(partial, nb, k * k, G);
  return hipGetLastError();
}

hipError_t launch_nnls(const double* G, const double* H, const double* xx, int64_t m, int k, int64_t n, double* err,
                       int* n_not_converged, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  const size_t lds = ((size_t)k * (k + 1) + 3 * (size_t)k) * sizeof(double) + 2 * (size_t)k * sizeof(int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nnls_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // scipy.optimize.nnls: tol = 10 * max(A.shape) * eps, here relative to max |h|
  const double tol_rel = 10.0 * (double)(n > k ? n : k) * 2.220446049250313e-16;
  const int64_t grid = m < 4096 ? m : 4096;
  nnls_kernel<<<dim3((unsigned)grid), 64, lds, s>>>(G, H, xx, m, k, tol_rel, 3 * k, err, n_not_converged);
  return hipGetLastError();
}

}  // namespace prmf
