// Non-negative least squares against the data matrix itself -- the two solves per latent factor of `--manifolds-init`
// (script/prmf_runner.py:308-334):
//
//     u = nnls(X^T, v)     min_{u >= 0} || X^T u - v ||     m unknowns (samples), n equations (genes)
//     v = nnls(X,   u)     min_{v >= 0} || X v - u ||       n unknowns (genes),   m equations (samples)
//
// scipy.optimize.nnls is the Lawson-Hanson active-set method on the host, one Householder update of the whole matrix per
// step: 2.4 s per latent factor at 500 x 20000 (14.4 s of a 15 s command-line run whose solve takes 0.2 s,
// profiles/r05_cli_end_to_end.txt) and minutes from a few thousand samples on.  The method is kept -- its answer is the
// reference's: both problems have a unique minimiser for data in general position (the second is underdetermined, but a
// residual that is not zero pins the optimal face; u has zeros, the columns of a normalised X none) -- and its steps are
// shared out:
//
//   device (this file)  w = A^T r for all unknowns (the gradient), the masked argmax that picks the next unknown, the inner
//                       products of the passive columns with a new column / with the residual, the residual r = b - A_P s.
//                       A is never copied or transposed: column j of A is a strided column of X or a contiguous row of it.
//   host (prmf_nnls)    the passive list, the Cholesky factor of A_P^T A_P grown by one row per accepted unknown, the
//                       feasibility steps, and ONE step of iterative refinement per solve with the residual computed from
//                       A itself (corrected semi-normal equations: the accuracy of an orthogonal factorisation without one).
//
// All sums are float64 in a fixed order (no atomics): a problem gives the same bits on every run.
#include "kernels.h"

namespace prmf {

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// element t of column j of A:  X[t * st_t + j * st_j]   (A = X: st_t = ld, st_j = 1;  A = X^T: st_t = 1, st_j = ld)

// w[j] = sum_t A(t, j) r[t], A = X (row-major): one lane per column j, rows in order
__global__ void __launch_bounds__(256)
nnlsm_grad_cols_kernel(const double* __restrict__ X, int64_t ld, int64_t rows, int64_t cols, const double* __restrict__ r,
                       double* __restrict__ w, double* __restrict__ wabs) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  double acc = 0.0, mag = 0.0;
  for (int64_t t = 0; t < rows; ++t) {
    const double p = X[t * ld + j] * r[t];
    acc += p;
    mag += fabs(p);
  }
  w[j] = acc;
  wabs[j] = mag;
}

// w[i] = sum_t X[i][t] r[t], A = X^T: one wave per row i of X
__global__ void __launch_bounds__(256)
nnlsm_grad_rows_kernel(const double* __restrict__ X, int64_t ld, int64_t rows, int64_t cols, const double* __restrict__ r,
                       double* __restrict__ w, double* __restrict__ wabs) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const double* row = X + i * ld;
  double acc = 0.0, mag = 0.0;
  for (int64_t t = lane; t < cols; t += 64) {
    const double p = row[t] * r[t];
    acc += p;
    mag += fabs(p);
  }
  acc = wave_sum_f64(acc);
  mag = wave_sum_f64(mag);
  if (lane == 0) {
    w[i] = acc;
    wabs[i] = mag;
  }
}

// The free unknown (state 0) with the largest gradient among those whose gradient stands out of the rounding noise of its own
// inner product: w[j] > c1 * sum_t |a_tj r_t|  (the forward error of the sum)  +  c2 * sum_t |a_tj b_t|  (what the residual itself
// cannot resolve: r = b - A_P s carries eps |b_t| per component).  Per column, so free of row and column scales.
// out[0] = its index (-1: none), out[1] = its w (as bits); ties -> lowest index
__global__ void __launch_bounds__(1024)
nnlsm_argmax_kernel(const double* __restrict__ w, const unsigned char* __restrict__ state, const double* __restrict__ wabs,
                    const double* __restrict__ babs, double c1, double c2, int64_t n, long long* __restrict__ out) {
  __shared__ double bv[1024];
  __shared__ long long bi[1024];
  double best = 0.0;
  long long at = -1;
  for (int64_t j = threadIdx.x; j < n; j += 1024)
    if (state[j] == 0 && w[j] > c1 * wabs[j] + c2 * babs[j] && (at < 0 || w[j] > best)) {
      best = w[j];
      at = j;
    }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = at;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const long long oi = bi[threadIdx.x + s];
      const double ov = bv[threadIdx.x + s];
      const long long mi = bi[threadIdx.x];
      if (oi >= 0 && (mi < 0 || ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < mi))) {
        bv[threadIdx.x] = ov;
        bi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = bi[0];
    out[1] = __double_as_longlong(bv[0]);
  }
}

// q[t] = A(t, j): one column of A as a contiguous vector
__global__ void __launch_bounds__(256)
nnlsm_column_kernel(const double* __restrict__ X, int64_t st_t, int64_t st_j, int64_t len, int64_t j, double* __restrict__ q) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < len) q[t] = X[t * st_t + j * st_j];
}

// out[a] = sum_t A(t, P[a]) q[t]: one wave per passive column
__global__ void __launch_bounds__(256)
nnlsm_dots_kernel(const double* __restrict__ X, int64_t st_t, int64_t st_j, int64_t len, const long long* __restrict__ P, int np,
                  const double* __restrict__ q, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int a = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= np) return;
  const double* col = X + P[a] * st_j;
  double acc = 0.0;
  for (int64_t t = lane; t < len; t += 64) acc += col[t * st_t] * q[t];
  acc = wave_sum_f64(acc);
  if (lane == 0) out[a] = acc;
}

// r[t] = b[t] - sum_a s[a] A(t, P[a]), passive columns in list order
__global__ void __launch_bounds__(256)
nnlsm_residual_kernel(const double* __restrict__ X, int64_t st_t, int64_t st_j, int64_t len, const long long* __restrict__ P, int np,
                      const double* __restrict__ s, const double* __restrict__ b, double* __restrict__ r) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= len) return;
  double acc = b[t];
  const double* base = X + t * st_t;
  for (int a = 0; a < np; ++a) acc -= s[a] * base[P[a] * st_j];
  r[t] = acc;
}

}  // namespace

hipError_t nnlsm_gradient(const double* X, int64_t ld, int64_t rows, int64_t cols, bool transposed, const double* r, double* w,
                          double* wabs, hipStream_t s) {
  if (!transposed) nnlsm_grad_cols_kernel<<<dim3((unsigned)((cols + 255) / 256)), 256, 0, s>>>(X, ld, rows, cols, r, w, wabs);
  else nnlsm_grad_rows_kernel<<<dim3((unsigned)((rows + 3) / 4)), 256, 0, s>>>(X, ld, rows, cols, r, w, wabs);
  return hipGetLastError();
}

hipError_t nnlsm_argmax(const double* w, const unsigned char* state, const double* wabs, const double* babs, double c1, double c2,
                        int64_t n, long long* out, hipStream_t s) {
  nnlsm_argmax_kernel<<<1, 1024, 0, s>>>(w, state, wabs, babs, c1, c2, n, out);
  return hipGetLastError();
}

hipError_t nnlsm_column(const double* X, int64_t st_t, int64_t st_j, int64_t len, int64_t j, double* q, hipStream_t s) {
  nnlsm_column_kernel<<<dim3((unsigned)((len + 255) / 256)), 256, 0, s>>>(X, st_t, st_j, len, j, q);
  return hipGetLastError();
}

hipError_t nnlsm_dots(const double* X, int64_t st_t, int64_t st_j, int64_t len, const long long* P, int np, const double* q,
                      double* out, hipStream_t s) {
  if (np <= 0) return hipSuccess;
  nnlsm_dots_kernel<<<dim3((unsigned)((np + 3) / 4)), 256, 0, s>>>(X, st_t, st_j, len, P, np, q, out);
  return hipGetLastError();
}

hipError_t nnlsm_residual(const double* X, int64_t st_t, int64_t st_j, int64_t len, const long long* P, int np, const double* sv,
                          const double* b, double* r, hipStream_t s) {
  nnlsm_residual_kernel<<<dim3((unsigned)((len + 255) / 256)), 256, 0, s>>>(X, st_t, st_j, len, P, np, sv, b, r);
  return hipGetLastError();
}

}  // namespace prmf
