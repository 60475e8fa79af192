// The dominant kernel of the PRMF iteration: the streaming sweep
//     C[M x KP] = A[M x K] * B[K x KP]
// used twice per iteration -- P = X V (A = X, prmf_runner.py:420) and
// Q = X^T U (A = the resident X^T copy, :424).  A is read exactly once from HBM
// (2 B or 4 B per element), B is a small factor re-read from L2 by every
// workgroup, so the kernel is HBM-bound; MFMA does the arithmetic.
//
// Data layout
//   A   row-major, zero padded to [round_up(M,256)] x [round_up(K,64)], so the
//       kernel has no bounds checks on loads.  A reduction "chunk" is 128 B of
//       each row (64 bf16 / 32 f32).
//   B   packed on the device by pack_*_kernel into MFMA B-fragment order so one
//       wave-wide 16 B/lane read is one fragment:
//         bf16: block (kb, t, hl) of 1 KiB, kb = k/16, t = 32-column tile,
//               hl = 0 hi / 1 lo;  lane (c = lane&31, h = lane>>5), element e:
//               B[16 kb + 8 h + e][32 t + c]   split as hi = bf16(v), lo = bf16(v - hi)
//         f32 : block (g, t) of 1 KiB, g = k/8; lane (c,h), element q (4 floats):
//               B[8 g + 4 h + q][32 t + c]
//       The packed chunk (all blocks of 64 resp. 32 reduction rows) is staged
//       once per workgroup into LDS and shared by its 4 waves.
//   C   per-split partial slabs [split][M][KP] f32, summed by the consumer kernel
//       in a fixed order (deterministic; no atomics).
//
// Wave tiling (wave64): each wave owns MT 32-row tiles x NT 32-column tiles of
// accumulators (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).  The A
// fragment of lane (r = lane&31, h = lane>>5) is 16 contiguous bytes of row r, so
// A goes straight from global memory to registers (streamed once, never shared
// between waves: no LDS round trip), PF chunks ahead of the MFMAs.
#include "kernels.h"

namespace prmf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short f2bf_bits(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float bf_bits2f(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ float factor_at(const FactorView& f, int64_t row, int col) {
  if (row >= f.rows || col >= f.cols) return 0.f;
  return f.src[(int64_t)(col / f.kc) * f.blk_stride + row * f.kc + (col % f.kc)];
}

// ---------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_bf16_kernel(FactorView f, int NT, int64_t nkb, uint4* __restrict__ out) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lane = (int)(gid & 63);
  int64_t rest = gid >> 6;
  int t = (int)(rest % NT);
  int64_t kb = rest / NT;
  if (kb >= nkb) return;
  int c = lane & 31, h = lane >> 5;
  int col = 32 * t + c;
  int64_t row0 = kb * 16 + 8 * h;
  unsigned hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    float v0 = factor_at(f, row0 + e, col), v1 = factor_at(f, row0 + e + 1, col);
    unsigned short h0 = f2bf_bits(v0), h1 = f2bf_bits(v1);
    unsigned short l0 = f2bf_bits(v0 - bf_bits2f(h0)), l1 = f2bf_bits(v1 - bf_bits2f(h1));
    hi[e >> 1] = (unsigned)h0 | ((unsigned)h1 << 16);
    lo[e >> 1] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
  int64_t blk = (kb * NT + t) * 2;
  out[blk * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  out[(blk + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ void __launch_bounds__(256) pack_f32_kernel(FactorView f, int NT, int64_t ng, float4* __restrict__ out) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lane = (int)(gid & 63);
  int64_t rest = gid >> 6;
  int t = (int)(rest % NT);
  int64_t g = rest / NT;
  if (g >= ng) return;
  int c = lane & 31, h = lane >> 5;
  int col = 32 * t + c;
  int64_t row0 = g * 8 + 4 * h;
  float4 v;
  v.x = factor_at(f, row0 + 0, col);
  v.y = factor_at(f, row0 + 1, col);
  v.z = factor_at(f, row0 + 2, col);
  v.w = factor_at(f, row0 + 3, col);
  out[(g * NT + t) * 64 + lane] = v;
}

hipError_t launch_pack_factor(const FactorView& f, int x_dtype, int NT, int64_t Kpad, void* packed, hipStream_t s) {
  if (x_dtype == 1) {
    int64_t nkb = Kpad / 16;
    int64_t threads = nkb * NT * 64;
    pack_bf16_kernel<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(f, NT, nkb, (uint4*)packed);
  } else {
    int64_t ng = Kpad / 8;
    int64_t threads = ng * NT * 64;
    pack_f32_kernel<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(f, NT, ng, (float4*)packed);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------
// BF16 = true : A bf16, B hi/lo bf16, 2 MFMA (32x32x16) per fragment pair
// BF16 = false: A f32,  B f32,        MFMA 32x32x2 f32 (exact f32 products)
template <bool BF16, int MT, int NT, int PF>
__global__ void __launch_bounds__(256, 2)
sweep_kernel(const uint4* __restrict__ A, int64_t lda16 /* row stride in 16-B units */, int64_t M_store,
             const uint4* __restrict__ Bp, int nchunks, int chunks_per_split,
             float* __restrict__ slabs, int64_t slab_stride) {
  constexpr int KP = 32 * NT;
  constexpr int BCH = BF16 ? 512 * NT : 256 * NT;   // uint4 per packed chunk
  constexpr int BPT = BCH / 256;                    // uint4 staged per thread per chunk
  constexpr int NSET = PF + 1;
  __shared__ uint4 sB[2][BCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.y;
  const int c0 = split * chunks_per_split;
  const int nch = min(nchunks - c0, chunks_per_split);
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * MT);

  // lane (r,h) reads 16 B at byte 32*s + 16*h of its row's 128-B chunk piece, s = 0..3
  const uint4* ap[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ap[mt] = A + (row0 + 32 * mt + r) * lda16 + (int64_t)c0 * 8 + h;
  const uint4* bp = Bp + (int64_t)c0 * BCH + tid;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;

  uint4 a[NSET][MT][4];
  uint4 bs[BPT];

  auto load_a = [&](uint4 (&dst)[MT][4], int chunk) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) dst[mt][s4] = ap[mt][(int64_t)chunk * 8 + 2 * s4];
  };
  auto load_b = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) bs[i] = bp[(int64_t)chunk * BCH + 256 * i];
  };
  auto store_b = [&](uint4* dst) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) dst[tid + 256 * i] = bs[i];
  };
  auto compute = [&](const uint4 (&cur)[MT][4], const uint4* sb) {
    if constexpr (BF16) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uint4 bh = sb[((s4 * NT + t) * 2 + 0) * 64 + lane];
          const uint4 bl = sb[((s4 * NT + t) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, cur[mt][s4]), __builtin_bit_cast(bf16x8, bh), acc[mt][t], 0, 0, 0);
            acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, cur[mt][s4]), __builtin_bit_cast(bf16x8, bl), acc[mt][t], 0, 0, 0);
          }
        }
      }
    } else {
      // f32: piece s4 of the chunk = floats [8 s4 + 4 h, +4) of the row = group g = s4
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uint4 bv = sb[(s4 * NT + t) * 64 + lane];
          const float b4[4] = {__uint_as_float(bv.x), __uint_as_float(bv.y), __uint_as_float(bv.z), __uint_as_float(bv.w)};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const uint4 av = cur[mt][s4];
            const float a4[4] = {__uint_as_float(av.x), __uint_as_float(av.y), __uint_as_float(av.z), __uint_as_float(av.w)};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[q], acc[mt][t], 0, 0, 0);
          }
        }
      }
    }
  };

  // prologue: B chunk 0 -> LDS, A chunks 0..PF-1 -> registers
  load_b(0);
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < nch) load_a(a[p], p);
  store_b(sB[0]);
  __syncthreads();

  int cur_b = 0;
  for (int c = 0; c < nch; c += NSET) {
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int cc = c + u;
      if (cc < nch) {
        const bool has_next = cc + 1 < nch;
        // B for chunk cc+1 first (it is waited for at the end of this step, and
        // vmcnt retires in order), then the A prefetch PF chunks ahead.
        if (has_next) load_b(cc + 1);
        if (cc + PF < nch) load_a(a[(u + PF) % NSET], cc + PF);
        compute(a[u], sB[cur_b]);
        if (has_next) store_b(sB[cur_b ^ 1]);
        __syncthreads();
        cur_b ^= 1;
      }
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (i&3) + 8*(i>>2) + 4*h
  float* out = slabs + (int64_t)split * slab_stride;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = row0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row < M_store) {
#pragma unroll
        for (int t = 0; t < NT; ++t) out[row * KP + 32 * t + r] = acc[mt][t][i];
      }
    }
}

GemmPlan plan_gemm(int x_dtype, int64_t M, int64_t Kpad, int NT, int force_splits) {
  GemmPlan p;
  p.MT = 1;
  const int rows_wg = 4 * 32 * p.MT;
  p.panels = (int)(round_up(M, rows_wg) / rows_wg);
  p.nchunks = (int)(Kpad / chunk_elems(x_dtype));
  // enough workgroups to keep 256 CUs x (2..4 resident workgroups) streaming; at
  // least 8 chunks per split so the prologue/epilogue stay small.
  int target_wgs = 256 * 4;
  int splits = force_splits > 0 ? force_splits : (target_wgs + p.panels - 1) / p.panels;
  int max_splits = p.nchunks / 8 > 0 ? p.nchunks / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  p.chunks_per_split = (p.nchunks + splits - 1) / splits;
  p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
  (void)NT;
  return p;
}

template <bool BF16, int MT, int PF>
static hipError_t launch_sweep_nt(const GemmPlan& p, const void* A, int64_t lda16, int64_t M_store, const void* Bp,
                                  int NT, float* slabs, int64_t slab_stride, hipStream_t s) {
  dim3 grid(p.panels, p.splits), block(256);
#define PRMF_LAUNCH(NTV)                                                                                   \
  sweep_kernel<BF16, MT, NTV, PF><<<grid, block, 0, s>>>((const uint4*)A, lda16, M_store, (const uint4*)Bp, \
                                                         p.nchunks, p.chunks_per_split, slabs, slab_stride)
  switch (NT) {
    case 1: PRMF_LAUNCH(1); break;
    case 2: PRMF_LAUNCH(2); break;
    case 3: PRMF_LAUNCH(3); break;
    case 4: PRMF_LAUNCH(4); break;
    default: return hipErrorInvalidValue;
  }
#undef PRMF_LAUNCH
  return hipGetLastError();
}

hipError_t launch_gemm(int x_dtype, const GemmPlan& p, const void* A, int64_t lda, int64_t M_store,
                       const void* packedB, int NT, float* slabs, int64_t slab_stride, hipStream_t s) {
  const int64_t lda16 = lda * elem_bytes(x_dtype) / 16;
  if (x_dtype == 1) return launch_sweep_nt<true, 1, 2>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s);
  return launch_sweep_nt<false, 1, 2>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s);
}

// ---------------------------------------------------------------------------
// one-time layout kernels (prmf_set_x): padded copy and padded transpose
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) copy_pad_kernel(const T* __restrict__ src, int64_t ld_src, int64_t rows, int64_t cols,
                                                       T* __restrict__ dst, int64_t ld_dst) {
  int64_t row = blockIdx.y;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cols; c += (int64_t)gridDim.x * 256)
    dst[row * ld_dst + c] = src[row * ld_src + c];
  (void)rows;
}

template <typename T>
__global__ void __launch_bounds__(256) transpose_pad_kernel(const T* __restrict__ src, int64_t ld_src, int64_t rows, int64_t cols,
                                                            T* __restrict__ dst, int64_t ld_dst) {
  __shared__ T tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int64_t rr = r0 + ty + 4 * i, cc = c0 + tx;
    tile[ty + 4 * i][tx] = (rr < rows && cc < cols) ? src[rr * ld_src + cc] : T(0);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int64_t cc = c0 + ty + 4 * i, rr = r0 + tx;
    if (cc < cols && rr < rows) dst[cc * ld_dst + rr] = tile[tx][ty + 4 * i];
  }
}

hipError_t launch_copy_pad(int x_dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst,
                           int64_t ld_dst, hipStream_t s) {
  if (rows == 0 || cols == 0) return hipSuccess;
  if (rows > 0x7fffffffLL) return hipErrorInvalidValue;
  // gridDim.y is limited to 65535: fold rows
  int64_t done = 0;
  while (done < rows) {
    int64_t nr = rows - done < 65535 ? rows - done : 65535;
    dim3 grid((unsigned)((cols + 255) / 256 > 64 ? 64 : (cols + 255) / 256), (unsigned)nr);
    if (x_dtype == 1)
      copy_pad_kernel<unsigned short><<<grid, 256, 0, s>>>((const unsigned short*)src + done * ld_src, ld_src, nr, cols,
                                                           (unsigned short*)dst + done * ld_dst, ld_dst);
    else
      copy_pad_kernel<float><<<grid, 256, 0, s>>>((const float*)src + done * ld_src, ld_src, nr, cols,
                                                  (float*)dst + done * ld_dst, ld_dst);
    done += nr;
  }
  return hipGetLastError();
}

hipError_t launch_transpose_pad(int x_dtype, const void* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst,
                                int64_t ld_dst, hipStream_t s) {
  if (rows == 0 || cols == 0) return hipSuccess;
  int64_t done = 0;
  const int64_t max_rows = 65535LL * 64;
  while (done < rows) {
    int64_t nr = rows - done < max_rows ? rows - done : max_rows;
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((nr + 63) / 64));
    if (x_dtype == 1)
      transpose_pad_kernel<unsigned short><<<grid, 256, 0, s>>>((const unsigned short*)src + done * ld_src, ld_src, nr,
                                                                cols, (unsigned short*)dst + done, ld_dst);
    else
      transpose_pad_kernel<float><<<grid, 256, 0, s>>>((const float*)src + done * ld_src, ld_src, nr, cols,
                                                       (float*)dst + done, ld_dst);
    done += nr;
  }
  return hipGetLastError();
}

}  // namespace prmf
