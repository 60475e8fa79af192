// How X enters the library (prmf_set_x / prmf_set_x_device / prmf_set_x_host, capi.hip): ONE pass over the source that
//   * narrows every value to the storage type of the context (f64 -> f32, f32 -> bf16, f64 -> f32 -> bf16: round to nearest
//     even at every step -- the conversions `Tensor.to(dtype)` performs),
//   * writes the zero-padded row-major copy X and the transposed copy X^T the two sweeps stream,
//   * accumulates ||X||_F^2 of the STORED values in float64 (np.linalg.norm(X) at script/prmf_runner.py:640, before the root).
// Until round 5 these were three passes (a copy, a transpose and prmf_sum_squares: 12.5 ms at 50000 x 20000) after a
// conversion by torch; a host matrix now streams through them chunk by chunk under its own PCIe copy.
//
// The sum of squares has ONE summation tree, whatever the chunking: the matrix is cut into 64 x 64 tiles anchored at row 0 /
// column 0; a tile's 4096 squares are added per thread in row order (16 values), then in a fixed 256-leaf tree; a tile row's
// sums are added by one workgroup (strided, then the same tree); the tile rows' sums likewise.  Chunks start on multiples of
// 64 rows, so the bits do not depend on how many chunks the matrix arrived in; prmf_sum_squares uses the same kernels.
#include "kernels.h"

namespace prmf {

namespace {

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)0x7fc0;        // NaN (c10::BFloat16's rule)
  return (uint16_t)((u + (((u >> 16) & 1u) + 0x7fffu)) >> 16);
}

template <typename S>
__device__ __forceinline__ float src_to_f32(S v);
template <>
__device__ __forceinline__ float src_to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float src_to_f32<double>(double v) { return (float)v; }
template <>
__device__ __forceinline__ float src_to_f32<uint16_t>(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

template <typename D>
__device__ __forceinline__ D f32_to_dst(float v);
template <>
__device__ __forceinline__ float f32_to_dst<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint16_t f32_to_dst<uint16_t>(float v) { return f32_to_bf16_rne(v); }

template <typename D>
__device__ __forceinline__ double dst_to_f64(D v);
template <>
__device__ __forceinline__ double dst_to_f64<float>(float v) { return (double)v; }
template <>
__device__ __forceinline__ double dst_to_f64<uint16_t>(uint16_t v) { return (double)__uint_as_float((uint32_t)v << 16); }

__device__ __forceinline__ double tree_256(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}

// One 64 x 64 tile per workgroup.  src: `rows` x `cols` of this launch (its first row is a multiple of 64 rows into the
// matrix); X / XT (WRITE only) already point at this launch's first row resp. first column; partial[tile row][tile column].
template <typename S, typename D, bool WRITE>
__global__ void __launch_bounds__(256) setx_tile_kernel(const S* __restrict__ src, int64_t ld_src, int64_t rows, int64_t cols,
                                                        D* __restrict__ X, int64_t ldx, D* __restrict__ XT, int64_t ldxt,
                                                        double* __restrict__ partial) {
  __shared__ D tile[64][66];
  __shared__ double red[256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t rr = r0 + ty + 4 * i, cc = c0 + tx;
    D d = D(0);
    if (rr < rows && cc < cols) {
      d = f32_to_dst<D>(src_to_f32<S>(src[rr * ld_src + cc]));
      if (WRITE) X[rr * ldx + cc] = d;
      const double v = dst_to_f64<D>(d);
      acc += v * v;
    }
    if (WRITE) tile[ty + 4 * i][tx] = d;
  }
  const double total = tree_256(acc, red);       // (its barriers also publish the tile)
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
  if (WRITE) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t cc = c0 + ty + 4 * i, rr = r0 + tx;
      if (cc < cols && rr < rows) XT[cc * ldxt + rr] = tile[tx][ty + 4 * i];
    }
  }
}

// rowsum[tile row] = sum of that tile row's `tiles_n` tile sums
__global__ void __launch_bounds__(256) setx_rowsum_kernel(const double* __restrict__ partial, int tiles_n, double* __restrict__ rowsum) {
  __shared__ double red[256];
  const double* p = partial + (int64_t)blockIdx.x * tiles_n;
  double acc = 0.0;
  for (int j = threadIdx.x; j < tiles_n; j += 256) acc += p[j];
  const double total = tree_256(acc, red);
  if (threadIdx.x == 0) rowsum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) setx_total_kernel(const double* __restrict__ rowsum, int64_t tile_rows, double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t j = threadIdx.x; j < tile_rows; j += 256) acc += rowsum[j];
  const double total = tree_256(acc, red);
  if (threadIdx.x == 0) out[0] = total;
}

template <typename S, typename D>
hipError_t launch_tiles(const S* src, int64_t ld_src, int64_t rows, int64_t cols, D* X, int64_t ldx, D* XT, int64_t ldxt,
                        double* partial, double* rowsum, hipStream_t s) {
  const int tiles_n = (int)((cols + 63) / 64);
  int64_t done = 0;                                  // gridDim.y <= 65535: fold the tile rows
  while (done < rows) {
    const int64_t nr = rows - done < 65535LL * 64 ? rows - done : 65535LL * 64;
    const int tr = (int)((nr + 63) / 64);
    const dim3 grid((unsigned)tiles_n, (unsigned)tr);
    if (X)
      setx_tile_kernel<S, D, true><<<grid, 256, 0, s>>>(src + done * ld_src, ld_src, nr, cols, X + done * ldx, ldx, XT + done, ldxt, partial);
    else
      setx_tile_kernel<S, D, false><<<grid, 256, 0, s>>>(src + done * ld_src, ld_src, nr, cols, nullptr, 0, nullptr, 0, partial);
    setx_rowsum_kernel<<<tr, 256, 0, s>>>(partial, tiles_n, rowsum + done / 64);
    done += nr;
  }
  return hipGetLastError();
}

}  // namespace

int64_t setx_partial_doubles(int64_t rows, int64_t cols) {
  const int64_t fold = rows < 65535LL * 64 ? rows : 65535LL * 64;
  return ((fold + 63) / 64) * ((cols + 63) / 64);
}

hipError_t launch_setx_rows(int src_dtype, int x_dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols,
                            void* X, int64_t ldx, void* XT, int64_t ldxt, double* partial, double* rowsum, hipStream_t s) {
  if (rows == 0 || cols == 0) return hipSuccess;
  typedef uint16_t bf;
#define PRMF_SETX(S, D) return launch_tiles<S, D>((const S*)src, ld_src, rows, cols, (D*)X, ldx, (D*)XT, ldxt, partial, rowsum, s)
  if (x_dtype == 0) {                                // float32 storage
    if (src_dtype == 0) PRMF_SETX(float, float);
    if (src_dtype == 1) PRMF_SETX(bf, float);
    if (src_dtype == 2) PRMF_SETX(double, float);
  } else {                                           // bfloat16 storage
    if (src_dtype == 0) PRMF_SETX(float, bf);
    if (src_dtype == 1) PRMF_SETX(bf, bf);
    if (src_dtype == 2) PRMF_SETX(double, bf);
  }
#undef PRMF_SETX
  return hipErrorInvalidValue;
}

hipError_t launch_setx_total(const double* rowsum, int64_t tile_rows, double* out, hipStream_t s) {
  setx_total_kernel<<<1, 256, 0, s>>>(rowsum, tile_rows, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host side of prmf_set_x_host: rows [r0, r1) of a host matrix narrowed to the storage type into a compact buffer
// (leading dimension = cols) -- the same roundings as the kernel above (tests/test_gpu_setx.py compares them bit for bit)
// ---------------------------------------------------------------------------
static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)0x7fc0;
  return (uint16_t)((u + (((u >> 16) & 1u) + 0x7fffu)) >> 16);
}

void host_narrow_rows(int src_dtype, int x_dtype, const void* src, int64_t ld, int64_t r0, int64_t r1, int64_t cols, void* dst) {
  for (int64_t r = r0; r < r1; ++r) {
    if (src_dtype == 0 && x_dtype == 1) {
      const float* a = (const float*)src + r * ld;
      uint16_t* d = (uint16_t*)dst + (r - r0) * cols;
      for (int64_t c = 0; c < cols; ++c) d[c] = host_f32_to_bf16(a[c]);
    } else if (src_dtype == 2 && x_dtype == 0) {
      const double* a = (const double*)src + r * ld;
      float* d = (float*)dst + (r - r0) * cols;
      for (int64_t c = 0; c < cols; ++c) d[c] = (float)a[c];
    } else if (src_dtype == 2 && x_dtype == 1) {
      const double* a = (const double*)src + r * ld;
      uint16_t* d = (uint16_t*)dst + (r - r0) * cols;
      for (int64_t c = 0; c < cols; ++c) d[c] = host_f32_to_bf16((float)a[c]);
    }
  }
}

}  // namespace prmf
