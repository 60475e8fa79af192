// Everything of the PRMF iteration that is not the two big sweeps: the
// multiplicative updates with their small k x k products, the Gram matrices, the
// sparse pathway terms (CSR SpMM over the active supports, staged through LDS)
// and the objective.  All of it is O((m+n) k^2) work on O((m+n) k) bytes -- a few
// percent of the sweep traffic -- so these kernels favour exact reference
// semantics and run-to-run determinism (fixed-order reductions, no atomics).
//
// Reference lines (script/prmf_runner.py):
//   U update :420-422   V update :424-444   objective :336-372
//   normalize_laplacian :56-65   restrict scoring :115-127
#include "kernels.h"
#include "prologue.h"
#include "bf16split.h"

#ifndef PRMF_EPI_CS_MIN_NT
#define PRMF_EPI_CS_MIN_NT 3   // column-split epilogue (one wave per 32-column tile) from this many column tiles (k > 64)
#endif
#ifndef PRMF_EPI_NB2
#define PRMF_EPI_NB2 2     // slab segments per load batch at k = 64 (4 measured identical: profiles/r02_ab_epilogue_slab_batch.txt)
#endif

namespace prmf {

unsigned long long* g_epi_stamps = nullptr;   // debug: per-workgroup phase stamps of the next fused-epilogue launches

// float32 machine epsilon used as the clamp value (prmf_runner.py:23)
#define PRMF_EPS 1.1920928955078125e-07f

// ---------------------------------------------------------------------------
// Gram matrix F^T F on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32 products).
// A wave loads a 32-row tile in the MFMA C/D layout (lane (c,h), register i holds
// F[row(i,h)][32t + c], row(i,h) = (i&3) + 8(i>>2) + 4h): with that value as BOTH
// operands -- A[i=c][k=h] from column tile t1, B[k=h][j=c] from column tile t2 --
// one MFMA per register adds the two rows {row(i,0), row(i,1)} to the 32x32 block
// (t1,t2) of the Gram matrix; 16 MFMAs cover the 32 rows, no data movement at all.
// Per-wave float partials (a few hundred rows each), reduced in float64 in a fixed
// order (deterministic).
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GRAM_MAX_BLOCKS = 256;   // one 32-row tile per wave up to 32768 rows (nparts <= GRAM_MAX_PARTS = 512)
int gram_blocks(int64_t rows) {            // number of per-workgroup partials the launch writes
  int64_t tiles = (rows + 31) / 32;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > GRAM_MAX_BLOCKS) blocks = GRAM_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <int NT>
__global__ void __launch_bounds__(256) gram_mfma_kernel(FactorView f, float* __restrict__ partial) {
  constexpr int NP = NT * (NT + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int k = f.cols;
  f32x16 g[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) g[p][i] = 0.f;
  const int64_t ntiles = ((int64_t)f.rows + 31) / 32;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += wstride) {
    float v[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 32 * t + c;
      const int64_t cb = (int64_t)(col / f.kc) * f.blk_stride + (col % f.kc);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        v[t][i] = (row < f.rows && col < k) ? f.src[cb + row * f.kc] : 0.f;
      }
    }
    int p = 0;
#pragma unroll
    for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
      for (int t2 = t1; t2 < NT; ++t2, ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) g[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t1][i], v[t2][i], g[p], 0, 0, 0);
  }
  // sum the block's 4 waves through LDS (one 32x32 block at a time), write one partial per workgroup
  __shared__ float red[4][16][64];
  float* out = partial + (int64_t)blockIdx.x * k * k;
  int p = 0;
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
    for (int t2 = t1; t2 < NT; ++t2, ++p) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave][i][lane] = g[p][i];
      __syncthreads();
      // thread (wave, lane) finishes registers i = 4*wave .. 4*wave+3
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * wave + ii;
        const float v = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
        const int a = 32 * t1 + (i & 3) + 8 * (i >> 2) + 4 * h, b = 32 * t2 + c;
        if (a < k && b < k) {
          out[a * k + b] = v;
          if (t1 != t2) out[b * k + a] = v;
        }
      }
    }
}

// reduce per-workgroup Gram partials into G (float64, fixed order: prologue.h)
__global__ void __launch_bounds__(256) gram_reduce_kernel(const float* __restrict__ partial, int nparts, int k,
                                                          double* __restrict__ G) {
  __shared__ double red[256];
  gram_reduce_job(partial, nparts, k, G, nullptr, blockIdx.x, red);
}

hipError_t launch_gram_partials(const FactorView& f, float* pf, hipStream_t s) {
  const int nparts = gram_blocks(f.rows);
  const int NT = (f.cols + 31) / 32;
  switch (NT) {
    case 1: gram_mfma_kernel<1><<<nparts, 256, 0, s>>>(f, pf); break;
    case 2: gram_mfma_kernel<2><<<nparts, 256, 0, s>>>(f, pf); break;
    case 3: gram_mfma_kernel<3><<<nparts, 256, 0, s>>>(f, pf); break;
    case 4: gram_mfma_kernel<4><<<nparts, 256, 0, s>>>(f, pf); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gram(const FactorView& f, double* partial, double* G, hipStream_t s) {
  const int nparts = gram_blocks(f.rows);
  float* pf = reinterpret_cast<float*>(partial);
  hipError_t e = launch_gram_partials(f, pf, s);
  if (e != hipSuccess) return e;
  gram_reduce_kernel<<<gram_reduce_jobs(f.cols), 256, 0, s>>>(pf, nparts, f.cols, G);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Fused epilogues (any k): one pass per factor after its sweep.
//
//   den = F * G  on the f32 MFMA.  A operand = F in "A layout": lane (r,h) holds
//   F[row0 + r][(KP/2) h + j], j = 0..KP/2-1 (contiguous float4 loads); MFMA step s
//   pairs column s with column s + KP/2 (the k order of an MFMA is free as long as
//   B uses the same one): B[k=h][j=c] = G[s + (KP/2) h][32 t + c] from LDS.
//   The element-wise update then runs in the MFMA C/D layout (lane (c,h), register
//   i <-> row (i&3)+8(i>>2)+4h, column 32t+c), where the sweep's slabs, F itself and
//   the result are all read/written as 128-B row segments.
//
//   Packing for the next sweep happens from the same registers: the f32 fragment
//   order needs no data movement at all; the bf16 order (8 consecutive rows per
//   lane) needs one v_permlane32_swap per register pair.
// ---------------------------------------------------------------------------

// un[t][16]: a 32-row tile in C/D layout -> packed fragments of rows [row0, row0+32)
template <int NT>
__device__ __forceinline__ void pack_tile_from_cd(const float (&un)[NT][16], int64_t row0, int lane, void* packed) {
  {
    uint4* out = reinterpret_cast<uint4*>(packed);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float e[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // lanes 32-63 of x swap with lanes 0-31 of y: afterwards lane (c,0) holds rows 0-3 | 4-7,
          // lane (c,1) rows 8-11 | 12-15 of this 16-row block
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(un[t][8 * kb + i]), __float_as_uint(un[t][8 * kb + 4 + i]), false, false);
          e[i] = __uint_as_float(sw[0]);
          e[4 + i] = __uint_as_float(sw[1]);
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi[j] = pack_bf16_pair(e[2 * j], e[2 * j + 1]);
          lo[j] = pack_bf16_pair(e[2 * j] - bf16_trunc_back(e[2 * j]), e[2 * j + 1] - bf16_trunc_back(e[2 * j + 1]));
        }
        const int64_t blk = ((row0 / 16 + kb) * NT + t) * 2;
        out[blk * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        out[(blk + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
  }
}

// The same for the fragment order of the 16x16x32 MFMA (bf16 X at k > 64, gemm.hip "W16"): block (g = row0 / 32, t16, hl),
// lane (c16, kg) holds rows 32 g + 8 kg .. + 8 of column 16 t16 + c16.  After the permlane32 swap lane (c, h) of the
// C/D layout holds the 8-row groups kg = 2 kb + h (kb = 0, 1) of its column c of the 32-column tile t; lane (c16, kg) of
// block t16 = 2 t + b takes them from lane (16 b + c16, h = kg & 1), register set kb = kg >> 1: 32 ds_bpermute per tile.
__device__ __forceinline__ void pack_cd_w16(const float (&v)[16], int64_t row0, int t, int NT, int lane, void* packed) {
  uint4* out = reinterpret_cast<uint4*>(packed);
  unsigned P[2][2][4];                 // [kb][hl][word]
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    float e[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * kb + i]), __float_as_uint(v[8 * kb + 4 + i]), false, false);
      e[i] = __uint_as_float(sw[0]);
      e[4 + i] = __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P[kb][0][j] = pack_bf16_pair(e[2 * j], e[2 * j + 1]);
      P[kb][1][j] = pack_bf16_pair(e[2 * j] - bf16_trunc_back(e[2 * j]), e[2 * j + 1] - bf16_trunc_back(e[2 * j + 1]));
    }
  }
  const int c16 = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int src = 16 * b + c16 + 32 * (kg & 1);
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      unsigned w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned v0 = (unsigned)__shfl((int)P[0][hl][j], src, 64), v1 = (unsigned)__shfl((int)P[1][hl][j], src, 64);
        w[j] = (kg >> 1) ? v1 : v0;
      }
      const int64_t blk = ((row0 / 32) * (2 * NT) + 2 * t + b) * 2 + hl;
      out[blk * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

template <int NT, bool W16>
__device__ __forceinline__ void pack_tile(const float (&un)[NT][16], int64_t row0, int lane, void* packed) {
  if constexpr (W16) {
#pragma unroll
    for (int t = 0; t < NT; ++t) pack_cd_w16(un[t], row0, t, NT, lane, packed);
  } else {
    pack_tile_from_cd<NT>(un, row0, lane, packed);
  }
}

// A-layout load of one 32-row tile of F (lane (r,h): F[row0+r][(KP/2)h .. +KP/2), zero outside).
// VEC4: k % 4 == 0, the rows of F are 16-B aligned -> one float4 per 4 columns; else (any k -- the CLI's default is 6)
// element loads with clamped addresses.  A template parameter, not a branch: both forms in one kernel cost the
// k <= 64 epilogues their last free registers (256 + 200 B of scratch against 247).
template <int NT, bool VEC4>
__device__ __forceinline__ void load_tile_a_layout(const float* __restrict__ F, int64_t rows, int k, int64_t row0, int lane,
                                                   float (&ua)[16 * NT]) {
  constexpr int HALF = 16 * NT;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row = row0 + r;
  if constexpr (VEC4) {
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const int col = HALF * h + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows && col < k) v = *reinterpret_cast<const float4*>(F + row * k + col);
      ua[4 * q] = v.x; ua[4 * q + 1] = v.y; ua[4 * q + 2] = v.z; ua[4 * q + 3] = v.w;
    }
  } else {
    const float* Fr = F + (row < rows ? row : rows - 1) * k;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
      const int col = HALF * h + j;
      const float v = Fr[col < k ? col : k - 1];
      ua[j] = (row < rows && col < k) ? v : 0.f;
    }
  }
}

// den[t] = (F tile) * G ; Gs = G in LDS, [KP][KP] floats (rows/cols >= k are zero)
template <int NT>
__device__ __forceinline__ void tile_times_gram(const float (&ua)[16 * NT], int lane, const float* __restrict__ Gs,
                                                f32x16 (&den)[NT]) {
  constexpr int KP = 32 * NT, HALF = KP / 2;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) den[t][i] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < HALF; ++s2)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      den[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[s2], Gs[(s2 + HALF * h) * KP + 32 * t + r], den[t], 0, 0, 0);
}

// G (k x k float64, row-major) -> Gs [KP][KP] floats in LDS, zero outside k.  The loads are issued in batches of 16
// per thread BEFORE the LDS stores of the batch: written as `Gs[i] = (float)G[...]` in one loop the compiler kept every
// load behind the previous store, and the staging was a chain of KP*KP/256 dependent L2 round trips -- 24 us of a
// 53 us epilogue workgroup at k = 128 (tools/epi_phases.py), ~5 us at k = 64.
template <int NT, int BMAX = 16>
__device__ __forceinline__ void stage_gram(const double* __restrict__ G, int k, int col0, int ncols, float* Gs) {
  constexpr int KP = 32 * NT, PER = KP * KP / 256;           // 4, 16, 36, 64 values per thread
  constexpr int B = BMAX < 16 ? (PER < BMAX ? PER : BMAX) : (BMAX >= 32 && PER % 32 == 0 ? 32 : (PER % 16 == 0 ? 16 : (PER % 12 == 0 ? 12 : PER)));
  static_assert(PER % B == 0 && B <= 32, "whole batches");
#pragma unroll 1
  for (int b0 = 0; b0 < PER; b0 += B) {
    double v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int i = (int)threadIdx.x + 256 * (b0 + j);
      const int q = i / KP, c = i - q * KP;
      const bool in = q < k && c < ncols;
      v[j] = G[in ? q * k + col0 + c : 0];
      if (!in) v[j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < B; ++j) Gs[(int)threadIdx.x + 256 * (b0 + j)] = (float)v[j];
  }
}

// sum of a tile's slab segments (fragment order: [t][q4][lane] float4, 1 KiB per load instruction), in slab-index order.
// The loads of two segments are issued together (a data-dependent one-segment-per-iteration
// loop costs a full memory round trip per segment: up to 8 per tile); missing slots of the last
// batch re-read the first slab and are masked out of the sum.
template <int NT, int NB = 2>
__device__ __forceinline__ void load_slab_tile(const float* __restrict__ slabs, int64_t slab_stride, segmask_t mask, int64_t tile,
                                               int lane, float (&p)[NT][16]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) p[t][i] = 0.f;
  const int64_t base = tile * NT * 256 + lane;               // float4 index of (tile, t = 0, q4 = 0, lane)
  while (mask) {
    int sg[NB];
    float on[NB];
    const int first = __ffsll((long long)mask) - 1;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      sg[j] = mask ? __ffsll((long long)mask) - 1 : first;
      on[j] = mask ? 1.f : 0.f;
      mask &= mask - 1;
    }
    float4 v[NB][NT][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float4* sl = reinterpret_cast<const float4*>(slabs + (int64_t)sg[j] * slab_stride) + base;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[j][t][q4] = sl[(t * 4 + q4) * 64];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          p[t][4 * q4] += on[j] * v[j][t][q4].x;
          p[t][4 * q4 + 1] += on[j] * v[j][t][q4].y;
          p[t][4 * q4 + 2] += on[j] * v[j][t][q4].z;
          p[t][4 * q4 + 3] += on[j] * v[j][t][q4].w;
        }
  }
}

// The same sum in two halves, for callers that put other work between the issue of a batch's loads and their use (the fused
// epilogues: the MFMA chain of the tile runs under the first batch's memory round trip).  issue: the loads of the next NB
// segments of `mask` (which it consumes); add: their sum into p, in the order load_slab_tile adds them -- bit-identical.
template <int NT, int NB>
struct SlabBatch {
  float4 v[NB][NT][4];
  float on[NB];
};
template <int NT, int NB>
__device__ __forceinline__ void slab_batch_issue(const float* __restrict__ slabs, int64_t slab_stride, segmask_t& mask, int64_t tile, int lane,
                                                 SlabBatch<NT, NB>& b) {
  const int64_t base = tile * NT * 256 + lane;
  int sg[NB];
  const int first = __ffsll((long long)mask) - 1;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    sg[j] = mask ? __ffsll((long long)mask) - 1 : first;
    b.on[j] = mask ? 1.f : 0.f;
    mask &= mask - 1;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float4* sl = reinterpret_cast<const float4*>(slabs + (int64_t)sg[j] * slab_stride) + base;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) b.v[j][t][q4] = sl[(t * 4 + q4) * 64];
  }
}
template <int NT, int NB>
__device__ __forceinline__ void slab_batch_add(const SlabBatch<NT, NB>& b, float (&p)[NT][16]) {
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        p[t][4 * q4] += b.on[j] * b.v[j][t][q4].x;
        p[t][4 * q4 + 1] += b.on[j] * b.v[j][t][q4].y;
        p[t][4 * q4 + 2] += b.on[j] * b.v[j][t][q4].z;
        p[t][4 * q4 + 3] += b.on[j] * b.v[j][t][q4].w;
      }
}

// Cooperative slab sum.  With few tiles and many segments per tile (small matrices: a 5000-row X V has 157 tiles of
// 26 segments each) one wave per tile walks a chain of dependent load batches on a handful of CUs.  `wpt` = 2 or 4
// waves then share a tile: wave `sub` sums the segments whose ordinal (in slab-index order) is congruent to sub,
// the sub-sums meet in LDS and the tile's leader wave (sub 0) adds them in sub order -- a fixed order again.
__device__ __forceinline__ segmask_t my_segments(segmask_t mask, int wpt, int sub) {
  if (wpt == 1) return mask;
  segmask_t mine = 0;
  int ord = 0;
  for (segmask_t mk = mask; mk; mk &= mk - 1, ++ord)
    if (ord % wpt == sub) mine |= mk & (segmask_t(0) - mk);
  return mine;
}

// comb: [4 waves][NT*16][64] floats.  Every wave of the workgroup calls this (two barriers).
template <int NT>
__device__ __forceinline__ void combine_tile_sums(float (&p)[NT][16], int wpt, int sub, int wave, int lane, float* comb) {
  if (wpt == 1) return;
  float* mine = comb + (size_t)wave * (NT * 16 * 64);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[(t * 16 + i) * 64 + lane] = p[t][i];
  __syncthreads();
  if (sub == 0) {
    for (int j = 1; j < wpt; ++j) {
      const float* o = comb + (size_t)(wave + j) * (NT * 16 * 64);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) p[t][i] += o[(t * 16 + i) * 64 + lane];
    }
  }
  __syncthreads();
}

// Gram partial of the tile in C/D layout (see gram_mfma_kernel): g[p] += tile^T tile
template <int NT>
__device__ __forceinline__ void gram_accumulate(const float (&un)[NT][16], f32x16 (&g)[NT * (NT + 1) / 2]) {
  int p = 0;
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
    for (int t2 = t1; t2 < NT; ++t2, ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) g[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(un[t1][i], un[t2][i], g[p], 0, 0, 0);
}

// sum the block's 4 waves through LDS (one 32x32 block at a time) and write the workgroup's partial
template <int NT>
__device__ __forceinline__ void gram_block_store(const f32x16 (&g)[NT * (NT + 1) / 2], int k, float* __restrict__ out,
                                                 float (*red)[16][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  int p = 0;
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
    for (int t2 = t1; t2 < NT; ++t2, ++p) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave][i][lane] = g[p][i];
      __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * wave + ii;
        const float v = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
        const int a = 32 * t1 + (i & 3) + 8 * (i >> 2) + 4 * h, b = 32 * t2 + c;
        if (a < k && b < k) {
          out[a * k + b] = v;
          if (t1 != t2) out[b * k + a] = v;
        }
      }
    }
}

// Per-round form of the workgroup Gram partial.  Carrying the NP x 16 accumulator registers of gram_accumulate
// across the tile loop costs 48 registers at k = 64 for the whole kernel -- with them the fused epilogues need
// 300 registers and one workgroup per CU (or spill).  Here the tile's Gram is formed in fresh accumulators (live only
// after the update), summed over the 4 waves through LDS in wave order, and added to a running total of 4 floats per
// 32x32 block per thread (thread (wave, lane) keeps registers i = 4 wave .. 4 wave + 3 of every block).
// Every wave of the workgroup calls this once per loop round (2 barriers).
template <int NT>
__device__ __forceinline__ void gram_round(const f32x16 (&gt)[NT * (NT + 1) / 2], float (&gtot)[NT * (NT + 1) / 2][4],
                                           float (*red)[16][64] /* [NP][4 waves] blocks of [16][64] */, int wave, int lane) {
  // one exchange for all NP blocks of the tile (2 barriers per round; it was 2 per block)
  constexpr int NP = NT * (NT + 1) / 2;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[4 * p + wave][i][lane] = gt[p][i];
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = 4 * wave + ii;
      gtot[p][ii] += ((red[4 * p][i][lane] + red[4 * p + 1][i][lane]) + red[4 * p + 2][i][lane]) + red[4 * p + 3][i][lane];
    }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void gram_total_store(const float (&gtot)[NT * (NT + 1) / 2][4], int k, float* __restrict__ out,
                                                 int wave, int lane) {
  const int c = lane & 31, h = lane >> 5;
  int p = 0;
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
    for (int t2 = t1; t2 < NT; ++t2, ++p)
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * wave + ii;
        const int a = 32 * t1 + (i & 3) + 8 * (i >> 2) + 4 * h, b = 32 * t2 + c;
        if (a < k && b < k) {
          out[a * k + b] = gtot[p][ii];
          if (t1 != t2) out[b * k + a] = gtot[p][ii];
        }
      }
}

// GRAM: also accumulate F_new^T F_new (per-workgroup partial [k*k] floats at gram_partial[blockIdx])
template <int NT, bool VEC4, bool GRAM, bool W16>
__global__ void __launch_bounds__(256, NT <= 2 ? 2 : 1)
u_epilogue_kernel(float* __restrict__ U, int64_t m, int k, const float* __restrict__ slabs, const segmask_t* __restrict__ segmask,
                  int panel_rows, int64_t slab_stride, const double* __restrict__ Gv, void* __restrict__ packed,
                  int64_t ntiles_pack, float* __restrict__ gram_partial, int wpt /* waves per tile: 1, 2 or 4 */,
                  int nblocks_main, const GraphJob gj, unsigned long long* __restrict__ stamps /* debug, may be null */) {
  constexpr int KP = 32 * NT, NP = NT * (NT + 1) / 2;
  constexpr int COMB = NT <= 2 ? 4 * NT * 16 * 64 : 1;              // cooperative slab sums exist for k <= 64
  __shared__ float Gs[KP * KP];
  __shared__ float red_store[(GRAM ? NP * 4 * 16 * 64 : 1) > COMB ? (GRAM ? NP * 4 * 16 * 64 : 1) : COMB];   // Gram exchange (all NP blocks of a tile at once) | slab-sum exchange
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(red_store);
  bool first_round = true;
#define EPI_STAMP(j) do { if (stamps && threadIdx.x == 0 && first_round) stamps[8 * blockIdx.x + (j)] = wall_clock64(); } while (0)
#define EPI_DEP(arr) do { if (stamps) { float t_ = 0.f; for (int t2_ = 0; t2_ < NT; ++t2_) for (int i_ = 0; i_ < 16; ++i_) t_ += arr[t2_][i_]; if (t_ == 1.2345e-33f) un[0][0] = t_; } } while (0)
  if ((int)blockIdx.x >= nblocks_main) {      // extra workgroups: the pathway terms of one latent column each (GraphJob)
    const int e = (int)blockIdx.x - nblocks_main, c = e / gj.nsplit, part = e % gj.nsplit;
    if ((size_t)gj.lib.max_support * sizeof(float) <= sizeof(Gs))
      graph_terms_body<true>(gj.lib, gj.active, gj.col0, c, gj.V, k, gj.hyper, gj.gnum, gj.gden, Gs, part, gj.nsplit);
    else
      graph_terms_body<false>(gj.lib, gj.active, gj.col0, c, gj.V, k, gj.hyper, gj.gnum, gj.gden, Gs, part, gj.nsplit);
    return;
  }
  EPI_STAMP(0);
  // the wave index is made provably uniform: tile bases then live in scalar registers
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const int tpw = 4 / wpt, sub = wave % wpt, tslot = wave / wpt;
  constexpr int NB = NT == 1 ? 4 : PRMF_EPI_NB2;
  float gtot[GRAM ? NP : 1][4];
#pragma unroll
  for (int p = 0; p < (GRAM ? NP : 1); ++p)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) gtot[p][ii] = 0.f;
  // A round is ONE memory round trip deep where it was four (G staged | the panel's segment mask | the slab batches | the tile
  // of U: 8 of the 15 us of a workgroup at 6250 x 20000, 5.5 of 9 at 5000 x 10000, tools/epi_phases.py): everything that
  // depends on nothing -- the mask, the tile of U in both layouts -- is requested first (in the first round: before G is
  // staged), the slab loads follow as soon as the mask is there, and the MFMA chain U G runs under their round trip.
  // uniform trip count over the workgroup (the cooperative sum and the Gram round have barriers)
  const int64_t base0 = (int64_t)blockIdx.x * tpw;
  for (int64_t base = base0; base < ntiles_pack; base += (int64_t)nblocks_main * tpw) {
    const int64_t tile = base + tslot;
    const int64_t row0 = tile * 32;
    const bool live = tile < ntiles_pack && row0 < m;
    const bool lead = tile < ntiles_pack && sub == 0;            // this wave finishes the tile
    const bool mine = lead && live;
    segmask_t mask = 0;
    if (live && (wpt > 1 || lead)) mask = segmask[row0 / panel_rows];
    float ua[16 * NT], uu[NT][16];
    float* const Ut = U + row0 * k;
    const int last = (int)((m - row0 < 32 ? m - row0 : 32) - 1);
    auto cd_off = [&](int t, int i) -> unsigned {
      const int rr = (i & 3) + 8 * (i >> 2) + 4 * h, col = 32 * t + c;
      return (unsigned)((rr < last ? rr : last) * k + (col < k ? col : k - 1));
    };
    if (mine) {
      load_tile_a_layout<NT, VEC4>(U, m, k, row0, lane, ua);
      if constexpr (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) uu[0][i] = Ut[cd_off(0, i)];
      }
    }
    if (base == base0) {
      stage_gram<NT, 8>(Gv, k, 0, k, Gs);        // (two batches at k > 32: with 16 doubles next to the tile of U the kernel spills)
      __syncthreads();
      EPI_STAMP(1);
    }
    float p[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) p[t][i] = 0.f;
    mask = my_segments(mask, wpt, sub);
    SlabBatch<NT, NB> sb;
    const bool any = mask != 0;
    f32x16 den[NT];
    // (k > 32: the chain first -- with the batch's 64 registers live under it the kernel spills)
    if constexpr (NT > 1) {
      if (mine) {
        tile_times_gram<NT>(ua, lane, Gs, den);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) uu[t][i] = Ut[cd_off(t, i)];      // C/D layout: travels with the slab batch
      }
    }
    if (any) slab_batch_issue<NT, NB>(slabs, slab_stride, mask, tile, lane, sb);
    if constexpr (NT == 1) {
      if (mine) tile_times_gram<NT>(ua, lane, Gs, den);
    }
    if (any) slab_batch_add<NT, NB>(sb, p);
    while (mask) {
      slab_batch_issue<NT, NB>(slabs, slab_stride, mask, tile, lane, sb);
      slab_batch_add<NT, NB>(sb, p);
    }
    if constexpr (NT <= 2) combine_tile_sums<NT>(p, wpt, sub, wave, lane, red_store);
    float un[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) un[t][i] = 0.f;
    if (mine) {
      EPI_DEP(p);
      EPI_STAMP(2);
      EPI_DEP(den);
      EPI_STAMP(3);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 32 * t + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool valid = row < m && col < k;
          const float d = den[t][i] + uu[t][i];                      // U (V^T V) + U      (prmf_runner.py:421)
          float r2 = uu[t][i] * ((d != 0.f) ? p[t][i] / d : 1.f);    // 0/0 := 1           (:422)
          if (!valid) r2 = 0.f;
          if (valid) Ut[cd_off(t, i)] = r2;
          un[t][i] = r2;
        }
      }
    }
    EPI_DEP(un);
    EPI_STAMP(4);
    if (lead) pack_tile<NT, W16>(un, row0, lane, packed);   // tiles past m are packed as zeros
    EPI_STAMP(5);
    if constexpr (GRAM) {
      f32x16 gt[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) gt[q][i] = 0.f;
      if (mine) gram_accumulate<NT>(un, gt);
      gram_round<NT>(gt, gtot, red, wave, lane);
    }
    EPI_STAMP(6);
    EPI_STAMP(7);
    first_round = false;
  }
#undef EPI_STAMP
#undef EPI_DEP
  if constexpr (GRAM) gram_total_store<NT>(gtot, k, gram_partial + (int64_t)blockIdx.x * k * k, wave, lane);
}

// waves that share one 32-row tile in the fused epilogues (see my_segments): as many as keep the launch within the
// chip's 2048 resident waves and leave every wave at least one slab segment
int epilogue_wpt(int64_t rows, int max_seg, int NT) {
  if (NT >= PRMF_EPI_CS_MIN_NT) return NT;      // column-split form: the NT waves of a tile own one column tile each
  if (NT > 2) return 1;
  const int64_t tiles = (rows + 31) / 32;
  int w = 1;
  while (w < 4 && tiles * (2 * w) <= 2048 && 2 * w <= max_seg) w *= 2;
  return w;
}

int epilogue_blocks(int64_t rows, int wpt) {
  const int64_t tiles = (rows + 31) / 32;
  const int tpw = 4 / (wpt < 1 ? 1 : wpt);
  int64_t blocks = (tiles + tpw - 1) / tpw;
  // at most 512 workgroups (the resident slots at two per CU; one Gram partial each), and every workgroup the same
  // number of rounds: 782 tiles become 391 workgroups x 2 rounds, not 512 of which 270 run twice
  if (blocks > 512) {
    const int64_t rounds = (blocks + 511) / 512;
    blocks = (blocks + rounds - 1) / rounds;
  }
  return (int)(blocks < 1 ? 1 : blocks);
}

// ---------------------------------------------------------------------------
// Column-split fused epilogue (k > 64): one wave per (32-row tile, 32-column tile).
//
// With one wave per whole tile (the kernels above) a k = 128 tile is 256 f32 MFMAs in a row, 64-register arrays in
// every phase, no room for the Gram tiles (a separate gram_mfma launch per side) and one workgroup per CU: 63 us (U) and
// 98 us (V) per iteration at 25000 x 25000 plus 2 x 22 us of Gram launches.  Here the NT waves of a tile each own one
// column tile t: they sum only their quarter of every slab segment, run 64 MFMAs of F (A layout, all k columns) against
// their 32 columns of G, update / store / pack their 32 columns from 16-register arrays, and then exchange the updated
// tile through LDS so that the NT (NT + 1) / 2 Gram blocks of the tile are shared out among them (f32 MFMA, operands
// straight from the exchange buffer).  Same arithmetic per element; the summation order of the slabs is the slab-index
// order as before.
// ---------------------------------------------------------------------------
struct EpiArgs {
  float* F;                     // U (m x k) or V (n x k), updated in place
  int64_t rows;
  int k;
  const float* slabs;           // sweep slabs, or (DENSE) Q [n][k]
  const segmask_t* segmask;
  int panel_rows;
  int64_t slab_stride;
  const double* G;              // V^T V (U side) or U^T U (V side)
  void* packed;
  int64_t ntiles_pack;
  float* gram_partial;          // [blocks][k*k]
  int nblocks_main;
  // V side only
  const int32_t* pos;
  const float* gnum;
  const float* gden;
  int max_support;
  double* partial;              // [blocks * 4][k + 1]
  float* vsup;
  unsigned long long* stamps;   // debug (tools/epi_phases.py): 8 wall-clock stamps per workgroup, first round of wave 0
};

template <int NT, bool IS_V, bool VEC4, bool DENSE, bool W16>
__global__ void __launch_bounds__(256, 2)
epilogue_cs_kernel(const EpiArgs a, const GraphJob gj) {
  constexpr int KP = 32 * NT, HALF = KP / 2, NP = NT * (NT + 1) / 2;
  constexpr int TPW = NT == 3 ? 1 : 4 / NT;                 // tiles per workgroup round
  constexpr int GMAX = (NP + NT - 1) / NT;                  // Gram blocks per wave: block p belongs to wave p % NT, slot p / NT
  __shared__ float Gs[KP * KP];
  __shared__ float exch[4][16][64];                         // the updated tile, one column tile per wave (C/D layout)
  const int k = a.k;
#define EPI_STAMP(j) do { if (a.stamps && threadIdx.x == 0 && first_round) a.stamps[8 * blockIdx.x + (j)] = wall_clock64(); } while (0)
  bool first_round = true;
  if ((int)blockIdx.x >= a.nblocks_main) {                  // extra workgroups: pathway terms of one latent column each
    const int e = (int)blockIdx.x - a.nblocks_main, c = e / gj.nsplit, part = e % gj.nsplit;
    if ((size_t)gj.lib.max_support * sizeof(float) <= sizeof(Gs))
      graph_terms_body<true>(gj.lib, gj.active, gj.col0, c, gj.V, k, gj.hyper, gj.gnum, gj.gden, Gs, part, gj.nsplit);
    else
      graph_terms_body<false>(gj.lib, gj.active, gj.col0, c, gj.V, k, gj.hyper, gj.gnum, gj.gden, Gs, part, gj.nsplit);
    return;
  }
  EPI_STAMP(0);
  // (k = 128: 64 values per thread in two batches of 32 -- at the top of the kernel the registers are free; four batches of 16
  //  were four dependent L2 round trips, 5 us of a 24 us round)
  stage_gram<NT, 32>(a.G, k, 0, k, Gs);
  __syncthreads();
  EPI_STAMP(1);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5, r = lane & 31;
  const int tslot = NT == 3 ? 0 : wave / NT, t = NT == 3 ? wave : wave % NT;
  const bool wave_on = t < NT;                              // NT = 3: the fourth wave only keeps the barriers company
  const int col = 32 * t + c, colc = col < k ? col : k - 1;
  f32x16 gacc[GMAX];
#pragma unroll
  for (int p = 0; p < GMAX; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) gacc[p][i] = 0.f;
  double colsq = 0.0, cross = 0.0;
  // A round's loads in dependence order (round 6; it was: G | mask | slab batches | tile | chain | pathway gathers, six
  // dependent round trips, tools/epi_phases.py): the panel's segment mask, the tile in both layouts and (V) the positions
  // are requested first, the chain F G follows, the pathway gathers travel under the slab batches.
  const int64_t base0 = (int64_t)blockIdx.x * TPW;
  for (int64_t base = base0; base < a.ntiles_pack; base += (int64_t)a.nblocks_main * TPW) {
    const int64_t tile = base + tslot;
    const int64_t row0 = tile * 32;
    const bool in_range = wave_on && tile < a.ntiles_pack;
    const bool live = in_range && row0 < a.rows;
    float* const Ft = a.F + (live ? row0 : 0) * k;
    const int last = live ? (int)((a.rows - row0 < 32 ? a.rows - row0 : 32) - 1) : 0;
    auto cd_off = [&](int i) -> unsigned {
      const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
      return (unsigned)((rr < last ? rr : last) * k + colc);
    };
    float vn[16], q[16], fv[16], gn[16], gd[16];
    int ps[16];
    f32x16 den;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      vn[i] = 0.f;
      q[i] = 0.f;
      den[i] = 0.f;
    }
    // (k % 4 != 0 -- element-wise tile loads, a kernel that spills as it is -- keeps the old order: slab sum, then the tile)
    constexpr bool EARLY = VEC4;
    segmask_t mask = 0;
    float ua[HALF];
    // den = F_tile (all k columns, A layout) x G[:, this column tile] on the f32 MFMA; the old values of this
    // column tile (C/D layout) and, V side, the position map
    auto load_tile = [&]() {
      load_tile_a_layout<NT, VEC4>(a.F, a.rows, k, row0, lane, ua);
#pragma unroll
      for (int i = 0; i < 16; ++i) fv[i] = Ft[cd_off(i)];
      if constexpr (IS_V) {
        const int32_t* const pt = a.pos + row0 * k;
#pragma unroll
        for (int i = 0; i < 16; ++i) ps[i] = pt[cd_off(i)];
      }
    };
    auto chain = [&]() {
#pragma unroll
      for (int s2 = 0; s2 < HALF; ++s2)
        den = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[s2], Gs[(s2 + HALF * h) * KP + 32 * t + r], den, 0, 0, 0);
      if constexpr (IS_V) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {        // branch-free gathers: off-support entries read element 0 and are masked
          const unsigned goff = ps[i] >= 0 ? (unsigned)colc * (unsigned)a.max_support + (unsigned)ps[i] : 0u;
          gn[i] = a.gnum[goff];
          gd[i] = a.gden[goff];
        }
      }
    };
    if (live) {
      if constexpr (!DENSE) mask = a.segmask[row0 / a.panel_rows];
      if constexpr (EARLY) load_tile();
      if constexpr (DENSE) {
        const float* const Qt = a.slabs + row0 * k;
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i] = Qt[cd_off(i)];
      }
    }
    if (live) {
      // this wave's column tile of the slab segments, in slab-index order, 4 segments per round trip
      const int64_t fbase = (tile * NT + t) * 256 + lane;         // float4 index of (tile, t, q4 = 0, lane)
      float4 v[4][4];
      float on[4];
      auto issue = [&]() {
        int sg[4];
        const int first = __ffsll((long long)mask) - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sg[j] = mask ? __ffsll((long long)mask) - 1 : first;
          on[j] = mask ? 1.f : 0.f;
          mask &= mask - 1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4* sl = reinterpret_cast<const float4*>(a.slabs + (int64_t)sg[j] * a.slab_stride) + fbase;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) v[j][q4] = sl[q4 * 64];
        }
      };
      auto add = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            q[4 * q4] += on[j] * v[j][q4].x;
            q[4 * q4 + 1] += on[j] * v[j][q4].y;
            q[4 * q4 + 2] += on[j] * v[j][q4].z;
            q[4 * q4 + 3] += on[j] * v[j][q4].w;
          }
      };
      // (the first batch requested before the chain, to travel under it, spills on both sides: measured in registers only)
      if constexpr (EARLY) chain();
      if constexpr (!DENSE) {
        while (mask) {
          issue();
          add();
        }
      }
      if (a.stamps) { float t_ = 0.f; for (int i = 0; i < 16; ++i) t_ += q[i]; if (t_ == 1.2345e-33f) vn[0] = t_; EPI_STAMP(2); }
      if constexpr (!EARLY) {
        load_tile();
        chain();
      }
    }
    if (a.stamps) { float t_ = 0.f; for (int i = 0; i < 16; ++i) t_ += den[i]; if (t_ == 1.2345e-33f) vn[1] = t_; EPI_STAMP(3); }
    // The tile's waves update F in place, and every one of them has just read ALL columns of the old tile (A layout):
    // nobody stores before everybody's loads have been consumed by the MFMA chain above.
    __syncthreads();
    EPI_STAMP(4);
    if (live) {
      // (3) the element-wise update of this column tile
      if constexpr (IS_V) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool on = ps[i] >= 0, valid = row < a.rows && col < k;
          const float num = q[i] + (on ? gn[i] : 0.f);
          float d = den[i] + (on ? gd[i] : 0.f);
          if (d < PRMF_EPS) d = PRMF_EPS;                              // :442
          float r2 = fv[i] * (num / d);                                // :443
          if (r2 < PRMF_EPS) r2 = PRMF_EPS;                            // :444
          if (!valid) r2 = 0.f;
          if (valid) Ft[cd_off(i)] = r2;
          if (a.vsup && on && valid) a.vsup[(unsigned)col * (unsigned)a.max_support + (unsigned)ps[i]] = r2;
          colsq += (double)r2 * (double)r2;
          cross += (double)(valid ? q[i] : 0.f) * (double)r2;
          vn[i] = r2;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool valid = row < a.rows && col < k;
          const float d = den[i] + fv[i];                              // U (V^T V) + U      (prmf_runner.py:421)
          float r2 = fv[i] * ((d != 0.f) ? q[i] / d : 1.f);            // 0/0 := 1           (:422)
          if (!valid) r2 = 0.f;
          if (valid) Ft[cd_off(i)] = r2;
          vn[i] = r2;
        }
      }
    }
    if (a.stamps) { float t_ = 0.f; for (int i = 0; i < 16; ++i) t_ += vn[i]; if (t_ == 1.2345e-33f) vn[2] = t_; EPI_STAMP(5); }
    // (4) this column tile's fragments of the packed factor (tiles past the end are packed as zeros)
    if constexpr (W16) {
      if (in_range) pack_cd_w16(vn, row0, t, NT, lane, a.packed);
    } else if (in_range) {
      {
        uint4* out = reinterpret_cast<uint4*>(a.packed);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          float e[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(vn[8 * kb + i]), __float_as_uint(vn[8 * kb + 4 + i]), false, false);
            e[i] = __uint_as_float(sw[0]);
            e[4 + i] = __uint_as_float(sw[1]);
          }
          unsigned hi[4], lo[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hi[j] = pack_bf16_pair(e[2 * j], e[2 * j + 1]);
            lo[j] = pack_bf16_pair(e[2 * j] - bf16_trunc_back(e[2 * j]), e[2 * j + 1] - bf16_trunc_back(e[2 * j + 1]));
          }
          const int64_t blk = ((row0 / 16 + kb) * NT + t) * 2;
          out[blk * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          out[(blk + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
    }
    // (5) Gram blocks of the tile: exchange the column tiles through LDS; block p = (t1, t2 >= t1) belongs to wave p % NT
#pragma unroll
    for (int i = 0; i < 16; ++i) exch[wave][i][lane] = vn[i];
    __syncthreads();
    EPI_STAMP(6);
    if (wave_on) {
      const int w0 = NT == 3 ? 0 : tslot * NT;
      int p = 0;
#pragma unroll
      for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
        for (int t2 = t1; t2 < NT; ++t2, ++p)
          if (p % NT == t) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              gacc[p / NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(exch[w0 + t1][i][lane], exch[w0 + t2][i][lane], gacc[p / NT], 0, 0, 0);
          }
    }
    __syncthreads();
    EPI_STAMP(7);
    first_round = false;
  }
#undef EPI_STAMP
  if constexpr (IS_V) {
    // per-wave partials [k + 1]: this wave's 32 columns, zeros elsewhere, <Q, V_new> last (fixed order)
    double* out = a.partial + ((int64_t)blockIdx.x * 4 + wave) * (k + 1);
    const double tot = colsq + __shfl_xor(colsq, 32);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      if (h == 0 && 32 * tt + c < k) out[32 * tt + c] = (wave_on && tt == t) ? tot : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cross += __shfl_xor(cross, off);
    if (lane == 0) out[k] = cross;
  }
  // the workgroup's Gram partial: the tile slots of a workgroup (k <= 64: 2 or 4) are summed through LDS in slot order
  float* gout = a.gram_partial + (int64_t)blockIdx.x * k * k;
  constexpr int SLOTS = NT == 3 ? 1 : 4 / NT;
  int p = 0;
#pragma unroll
  for (int t1 = 0; t1 < NT; ++t1)
#pragma unroll
    for (int t2 = t1; t2 < NT; ++t2, ++p) {
      const bool own = wave_on && (p % NT == t);
      f32x16 g = gacc[p / NT];
#pragma unroll
      for (int sl = 1; sl < SLOTS; ++sl) {
        __syncthreads();
        if (own && tslot == sl)
#pragma unroll
          for (int i = 0; i < 16; ++i) exch[0][i][lane] = g[i];
        __syncthreads();
        if (own && tslot == 0)
#pragma unroll
          for (int i = 0; i < 16; ++i) g[i] += exch[0][i][lane];
      }
      if (own && tslot == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ra = 32 * t1 + (i & 3) + 8 * (i >> 2) + 4 * h, cb = 32 * t2 + c;
          if (ra < k && cb < k) {
            gout[ra * k + cb] = g[i];
            if (t1 != t2) gout[cb * k + ra] = g[i];
          }
        }
      }
    }
}


template <bool IS_V>
static hipError_t launch_epilogue_cs(const EpiArgs& a_in, const GraphJob& gj, int NT, bool bf16, bool dense, int grid, hipStream_t s, bool w16) {
  if (!bf16) return hipErrorInvalidValue;      // the factors are always packed as hi/lo bf16 fragments
  const bool vec4 = (a_in.k & 3) == 0;         // rows of F 16-B aligned
  EpiArgs a = a_in;
  a.stamps = g_epi_stamps;
#define PRMF_CS2(NTV, V4, DN)                                                                              \
  do {                                                                                                   \
    if (w16) epilogue_cs_kernel<NTV, IS_V, V4, DN, true><<<grid, 256, 0, s>>>(a, gj);                       \
    else epilogue_cs_kernel<NTV, IS_V, V4, DN, false><<<grid, 256, 0, s>>>(a, gj);                          \
  } while (0)
#define PRMF_CS(NTV)                                                                                     \
  if (vec4) {                                                                                            \
    if (dense) PRMF_CS2(NTV, true, IS_V); else PRMF_CS2(NTV, true, false);                                 \
  } else {                                                                                               \
    if (dense) PRMF_CS2(NTV, false, IS_V); else PRMF_CS2(NTV, false, false);                               \
  }
  switch (NT) {
#if PRMF_EPI_CS_MIN_NT <= 1
    case 1: PRMF_CS(1) break;
#endif
#if PRMF_EPI_CS_MIN_NT <= 2
    case 2: PRMF_CS(2) break;
#endif
    case 3: PRMF_CS(3) break;
    case 4: PRMF_CS(4) break;
    default: return hipErrorInvalidValue;
  }
#undef PRMF_CS
#undef PRMF_CS2
  return hipGetLastError();
}

hipError_t launch_u_epilogue(float* U, int64_t m, int k, const float* slabs, const segmask_t* segmask, int panel_rows,
                             int64_t slab_stride, int KP, const double* Gv, int x_dtype, void* packed, int64_t Kpad,
                             float* gram_partial, bool* gram_done, int wpt, hipStream_t s, const GraphJob* graph, bool w16) {
  const int NT = KP / 32;
  if (w16 && NT < 2) return hipErrorInvalidValue;      // the 16x16x32 fragment order exists for k > 32 only
  const int64_t ntiles = Kpad / 32;          // every tile of the packed buffer is (re)written, zero beyond m
  const GraphJob gj = graph ? *graph : GraphJob{};
  if (x_dtype != 1) return hipErrorInvalidValue;   // fragment layout of the packed factor: hi/lo bf16 only
  if (NT >= PRMF_EPI_CS_MIN_NT) {
    EpiArgs a{};
    a.F = U; a.rows = m; a.k = k; a.slabs = slabs; a.segmask = segmask; a.panel_rows = panel_rows; a.slab_stride = slab_stride;
    a.G = Gv; a.packed = packed; a.ntiles_pack = ntiles; a.gram_partial = gram_partial;
    a.nblocks_main = epilogue_blocks(Kpad, NT);
    *gram_done = true;
    return launch_epilogue_cs<false>(a, gj, NT, x_dtype == 1, false, a.nblocks_main + (gj.ncols > 0 ? gj.ncols * gj.nsplit : 0), s, w16);
  }
  if (NT > 2) wpt = 1;
  const int nmain = epilogue_blocks(Kpad, wpt);
  const int grid = nmain + (gj.ncols > 0 ? gj.ncols * gj.nsplit : 0);
  *gram_done = NT <= 2;
#define PRMF_UEP2(NTV, GR, WV)                                                                                     \
  if ((k & 3) == 0) u_epilogue_kernel<NTV, true, GR, WV><<<grid, 256, 0, s>>>(U, m, k, slabs, segmask, panel_rows, slab_stride, Gv, packed, ntiles, gram_partial, wpt, nmain, gj, g_epi_stamps); \
  else u_epilogue_kernel<NTV, false, GR, WV><<<grid, 256, 0, s>>>(U, m, k, slabs, segmask, panel_rows, slab_stride, Gv, packed, ntiles, gram_partial, wpt, nmain, gj, g_epi_stamps)
#define PRMF_UEP(NTV, GR) PRMF_UEP2(NTV, GR, false)
  switch (NT) {
    case 1: PRMF_UEP(1, true); break;
    case 2: if (w16) { PRMF_UEP2(2, true, true); } else { PRMF_UEP(2, true); } break;
#if PRMF_EPI_CS_MIN_NT > 3      // k > 64 takes the column-split kernel above
    case 3: PRMF_UEP(3, false); break;
    case 4: PRMF_UEP(4, false); break;
#endif
    default: return hipErrorInvalidValue;
  }
#undef PRMF_UEP
#undef PRMF_UEP2
  return hipGetLastError();
}

// Single-rank V epilogue: V <- V * (Q + g_num) / max(V (U^T U) + g_den, eps), clamp
// (prmf_runner.py:424-444), in place, + packing + per-wave partial column sums of
// squares and <Q, V_new> (float64) + (GRAM) V_new^T V_new partials.
template <int NT, bool VEC4, bool GRAM, bool DENSE, bool W16>
__global__ void __launch_bounds__(256, NT <= 2 ? 2 : 1)
v_epilogue_kernel(float* __restrict__ V, int64_t n, int k, const float* __restrict__ slabs /* DENSE: Q [n][k] */, const segmask_t* __restrict__ segmask,
                  int panel_rows, int64_t slab_stride, const double* __restrict__ Gu, const int32_t* __restrict__ pos,
                  const float* __restrict__ gnum, const float* __restrict__ gden, int max_support,
                  void* __restrict__ packed, int64_t ntiles_pack, double* __restrict__ partial /*[waves][k+1]*/,
                  float* __restrict__ gram_partial, float* __restrict__ vsup /* null or [k][max_support]: V_new on the supports */,
                  int wpt /* waves per tile (cooperative slab sum, see my_segments): 1, 2 or 4; 1 with DENSE */,
                  unsigned long long* __restrict__ stamps /* debug (tools/epi_phases.py), may be null */) {
  constexpr int KP = 32 * NT, NP = NT * (NT + 1) / 2;
  constexpr int COMB = (NT <= 2 && !DENSE) ? 4 * NT * 16 * 64 : 1;
  // phase stamps as in u_epilogue_kernel, but compiled in only with -DPRMF_EPI_V_STAMPS (a variant build for tools/epi_phases.py):
  // the k = 33..64 forms have no register to spare for them
#ifdef PRMF_EPI_V_STAMPS
  bool first_round = true;
#define EPI_STAMP(j) do { if (stamps && threadIdx.x == 0 && first_round) stamps[8 * blockIdx.x + (j)] = wall_clock64(); } while (0)
#define EPI_DEP(arr) do { if (stamps) { float t_ = 0.f; for (int t2_ = 0; t2_ < NT; ++t2_) for (int i_ = 0; i_ < 16; ++i_) t_ += arr[t2_][i_]; if (t_ == 1.2345e-33f) vn[0][0] = t_; } } while (0)
#define EPI_ROUND_DONE() first_round = false
#else
  (void)stamps;
#define EPI_STAMP(j) do { } while (0)
#define EPI_DEP(arr) do { } while (0)
#define EPI_ROUND_DONE() do { } while (0)
#endif
  __shared__ float Gs[KP * KP];
  __shared__ float red_store[(GRAM ? NP * 4 * 16 * 64 : 1) > COMB ? (GRAM ? NP * 4 * 16 * 64 : 1) : COMB];   // Gram exchange (all NP blocks of a tile at once) | slab-sum exchange
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(red_store);
  EPI_STAMP(0);
  stage_gram<NT>(Gu, k, 0, k, Gs);
  __syncthreads();
  EPI_STAMP(1);
  // the wave index is made provably uniform: tile bases then live in scalar registers and every per-lane
  // address is one 32-bit offset (the kernel is register-bound: 2 workgroups per CU need <= 256 VGPRs)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const int tpw = 4 / wpt, sub = wave % wpt, tslot = wave / wpt;
  double colsq[NT], cross = 0.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) colsq[t] = 0.0;
  float gtot[GRAM ? NP : 1][4];
#pragma unroll
  for (int p = 0; p < (GRAM ? NP : 1); ++p)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) gtot[p][ii] = 0.f;
  // uniform trip count over the workgroup (the cooperative sum and the Gram round have barriers)
  for (int64_t base = (int64_t)blockIdx.x * tpw; base < ntiles_pack; base += (int64_t)gridDim.x * tpw) {
    const int64_t tile = base + tslot;
    const int64_t row0 = tile * 32;
    const bool live = tile < ntiles_pack && row0 < n;
    const bool lead = tile < ntiles_pack && sub == 0;            // this wave finishes the tile
    float q[NT][16];
    if constexpr (NT <= 2 && !DENSE) {
      if (wpt > 1) {
        if (live) load_slab_tile<NT, NT == 1 ? 4 : PRMF_EPI_NB2>(slabs, slab_stride, my_segments(segmask[row0 / panel_rows], wpt, sub), tile, lane, q);
        else load_slab_tile<NT>(slabs, slab_stride, segmask_t(0), tile, lane, q);
        combine_tile_sums<NT>(q, wpt, sub, wave, lane, red_store);
      }
    }
    float vn[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) vn[t][i] = 0.f;
    if constexpr (NT <= 2 && !DENSE) {
      if (wpt > 1) { EPI_DEP(q); }
    }
    EPI_STAMP(2);
    if (lead && live) {
      // C/D-layout accesses of this tile: uniform tile base + 32-bit element offset (rows past n re-read the last row)
      float* const Vt = V + row0 * k;
      const int32_t* const pt = pos + row0 * k;
      const int last = (int)((n - row0 < 32 ? n - row0 : 32) - 1);
      auto cd_off = [&](int t, int i) -> unsigned {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h, col = 32 * t + c;
        return (unsigned)((rr < last ? rr : last) * k + (col < k ? col : k - 1));
      };
      if constexpr (DENSE) {
        // Q already summed over segments and ranks (replicated multi-rank V update): row-major [n][k]
        const float* const Qt = slabs + row0 * k;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) q[t][i] = Qt[cd_off(t, i)];
      } else {
        if (wpt == 1) load_slab_tile<NT, NT == 1 ? 4 : PRMF_EPI_NB2>(slabs, slab_stride, segmask[row0 / panel_rows], tile, lane, q);
      }
      float ua[16 * NT];
      load_tile_a_layout<NT, VEC4>(V, n, k, row0, lane, ua);
      f32x16 den[NT];
      tile_times_gram<NT>(ua, lane, Gs, den);
      EPI_DEP(den);
      EPI_STAMP(3);
      // The element-wise update, one 32-column half at a time (register budget: q, den and the new tile stay live;
      // V in C/D layout, the position map and the gathered pathway terms are per half).
      // The pathway terms are gathered BRANCH-FREE: entries off the support (98% of them) read element 0 and are
      // masked out.  As `if (ps >= 0) num += gnum[...]` every (register, column-tile) pair with one supported lane
      // became its own exec-masked block with a load and an immediate use: a chain of dependent memory round trips.
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 32 * t + c;
        if constexpr (NT > 1) __builtin_amdgcn_sched_barrier(0);
        float vv[16], gn[16], gd[16];
        int ps[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          vv[i] = Vt[cd_off(t, i)];
          ps[i] = pt[cd_off(t, i)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          // columns past k read the position map of column k - 1 (cd_off clamps): clamp the column here too
          const unsigned goff = ps[i] >= 0 ? (unsigned)(col < k ? col : k - 1) * (unsigned)max_support + (unsigned)ps[i] : 0u;
          gn[i] = gnum[goff];
          gd[i] = gden[goff];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          const bool on = ps[i] >= 0;
          const bool valid = row < n && col < k;
          const float num = q[t][i] + (on ? gn[i] : 0.f);
          float d = den[t][i] + (on ? gd[i] : 0.f);
          if (d < PRMF_EPS) d = PRMF_EPS;
          float r2 = vv[i] * (num / d);
          if (r2 < PRMF_EPS) r2 = PRMF_EPS;
          if (!valid) r2 = 0.f;
          if (valid) Vt[cd_off(t, i)] = r2;
          if (vsup && on && valid) vsup[(unsigned)col * (unsigned)max_support + (unsigned)ps[i]] = r2;   // for the deferred objective
          colsq[t] += (double)r2 * (double)r2;
          cross += (double)(valid ? q[t][i] : 0.f) * (double)r2;
          vn[t][i] = r2;
        }
      }
    }
    EPI_DEP(vn);
    EPI_STAMP(4);
    if (lead) pack_tile<NT, W16>(vn, row0, lane, packed);   // tiles past n are packed as zeros
    EPI_STAMP(5);
    if constexpr (GRAM) {
      f32x16 gt[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) gt[p][i] = 0.f;
      if (lead && live) gram_accumulate<NT>(vn, gt);
      gram_round<NT>(gt, gtot, red, wave, lane);
    }
    EPI_STAMP(6);
    EPI_STAMP(7);
    EPI_ROUND_DONE();
  }
#undef EPI_STAMP
#undef EPI_DEP
#undef EPI_ROUND_DONE
  // per-wave partials: column sums from the two half-waves, cross over all lanes (fixed order)
  double* out = partial + ((int64_t)blockIdx.x * 4 + wave) * (k + 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const double tot = colsq[t] + __shfl_xor(colsq[t], 32);
    if (h == 0 && 32 * t + c < k) out[32 * t + c] = tot;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cross += __shfl_xor(cross, off);
  if (lane == 0) out[k] = cross;
  if constexpr (GRAM) gram_total_store<NT>(gtot, k, gram_partial + (int64_t)blockIdx.x * k * k, wave, lane);
}

int v_epilogue_waves(int64_t n, int wpt) { return epilogue_blocks(round_up(n, K_PAD), wpt) * 4; }

hipError_t launch_v_epilogue(float* V, int64_t n, int k, const float* slabs, const segmask_t* segmask, int panel_rows,
                             int64_t slab_stride, int KP, const double* Gu, const int32_t* pos, const float* gnum,
                             const float* gden, int max_support, int x_dtype, void* packed, int64_t Kpad,
                             double* partial, float* gram_partial, bool* gram_done, const float* qdense, float* vsup, int wpt,
                             hipStream_t s, bool w16) {
  const int NT = KP / 32;
  if (w16 && NT < 2) return hipErrorInvalidValue;
  const int64_t ntiles = Kpad / 32;
  if (x_dtype != 1) return hipErrorInvalidValue;
  if (NT >= PRMF_EPI_CS_MIN_NT) {
    EpiArgs a{};
    a.F = V; a.rows = n; a.k = k; a.slabs = qdense ? qdense : slabs; a.segmask = segmask; a.panel_rows = panel_rows;
    a.slab_stride = slab_stride; a.G = Gu; a.packed = packed; a.ntiles_pack = ntiles; a.gram_partial = gram_partial;
    a.nblocks_main = epilogue_blocks(Kpad, NT);
    a.pos = pos; a.gnum = gnum; a.gden = gden; a.max_support = max_support; a.partial = partial; a.vsup = vsup;
    *gram_done = true;
    return launch_epilogue_cs<true>(a, GraphJob{}, NT, x_dtype == 1, qdense != nullptr, a.nblocks_main, s, w16);
  }
  // `wpt` must be the value the caller sized `partial` with (v_epilogue_waves): every partial row is written by
  // exactly one wave; the dense-Q form has no slabs to share and runs one wave per tile on the same grid
  if (NT > 2) wpt = 1;
  const int grid = epilogue_blocks(Kpad, wpt);
  const int kwpt = qdense ? 1 : wpt;
  *gram_done = NT <= 2;
  const float* qsrc = qdense ? qdense : slabs;
#define PRMF_VEP3(NTV, GR, V4, DN, WV)                                                                              \
  v_epilogue_kernel<NTV, V4, GR, DN, WV><<<grid, 256, 0, s>>>(V, n, k, qsrc, segmask, panel_rows, slab_stride, Gu, pos, gnum, gden, max_support, packed, ntiles, partial, gram_partial, vsup, kwpt, g_epi_stamps)
#define PRMF_VEP2(NTV, GR, V4, DN)                                                                                  \
  do { if (NTV == 2 && w16) { PRMF_VEP3(2, GR, V4, DN, true); } else { PRMF_VEP3(NTV, GR, V4, DN, false); } } while (0)
#define PRMF_VEP(NTV, GR)                                                                                           \
  if ((k & 3) == 0) { if (qdense) PRMF_VEP2(NTV, GR, true, true); else PRMF_VEP2(NTV, GR, true, false); }           \
  else { if (qdense) PRMF_VEP2(NTV, GR, false, true); else PRMF_VEP2(NTV, GR, false, false); }
  switch (NT) {
    case 1: PRMF_VEP(1, true); break;
    case 2: PRMF_VEP(2, true); break;
#if PRMF_EPI_CS_MIN_NT > 3      // k > 64 takes the column-split kernel above
    case 3: PRMF_VEP(3, false); break;
    case 4: PRMF_VEP(4, false); break;
#endif
    default: return hipErrorInvalidValue;
  }
#undef PRMF_VEP
#undef PRMF_VEP2
#undef PRMF_VEP3
  return hipGetLastError();
}

// Multi-rank phase_post in one pass: the gathered per-rank column blocks -> row-major V (caller's buffer),
// packed fragments for the next sweep, V^T V partials (GRAM).  Same tile machinery as the epilogues.
// STAGED (kc % 4 == 0 and the rank blocks together no wider than the tile): a tile's share of every rank block is one contiguous run of 32 kc floats; the wave copies the
// runs with 16-B loads into a private LDS tile and reads its C/D-layout values from there.  Read straight from the
// blocks in the C/D layout, one load instruction touched 2 rows x 32 / kc blocks with kc * 4 bytes each (32 B at eight
// ranks of k = 64): 25.5 us per iteration at 20000 genes against 18 us for the whole fused V epilogue.
template <int NT, bool GRAM, bool STAGED, bool W16>
__global__ void __launch_bounds__(256)
v_post_kernel(float* __restrict__ V, int64_t n, int k, const float* __restrict__ v_recv, int64_t blk_stride, int kc,
              void* __restrict__ packed, int64_t ntiles_pack, float* __restrict__ gram_partial,
              const int32_t* __restrict__ pos_all /* null or [n][k] */, float* __restrict__ vsup /* [k][max_support] */, int max_support) {
  constexpr int NP = NT * (NT + 1) / 2, KP = 32 * NT, PITCH = KP + 4;
  __shared__ float red[GRAM ? 4 : 1][16][64];
  __shared__ float stage[STAGED ? 4 : 1][STAGED ? 32 * PITCH : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  const int nblocks_src = (k + kc - 1) / kc;
  f32x16 g[GRAM ? NP : 1];
  if constexpr (GRAM) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) g[p][i] = 0.f;
  }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles_pack; tile += wstride) {
    const int64_t row0 = tile * 32;
    float vn[NT][16];
    // Everything below is written BRANCH-FREE around its loads (clamped addresses, masked uses): behind a per-element
    // `if` the compiler gives every load a block of its own with a full wait after it -- this kernel was a chain of up
    // to 35 (k = 64) / 128 (k = 128) dependent memory round trips per tile.
    if constexpr (STAGED) {
      float* T = stage[wave];
      const int rows_here = row0 < n ? (int)(n - row0 < 32 ? n - row0 : 32) : 0;
      const int run4 = rows_here * kc / 4;                  // float4 per rank block
      const int total4 = nblocks_src * run4;                // <= 8 KP = 64 lanes x 4 NT (launch_v_post: nblocks_src kc <= KP)
      float4 x[4 * NT];
      int dst[4 * NT];
#pragma unroll
      for (int j = 0; j < 4 * NT; ++j) {
        const int q = lane + 64 * j;
        const bool on = q < total4;
        const int qq = on ? q : 0;
        const int b = run4 > 0 ? qq / run4 : 0, f = qq - b * run4;
        const int e = 4 * f, rr = e / kc, cc = e - rr * kc;
        dst[j] = on ? rr * PITCH + b * kc + cc : -1;
        x[j] = reinterpret_cast<const float4*>(v_recv + (int64_t)b * blk_stride + (rows_here > 0 ? row0 : 0) * kc)[f];     // tiles past n read row 0 (unused)
      }
#pragma unroll
      for (int j = 0; j < 4 * NT; ++j)
        if (dst[j] >= 0) {
          float* d = T + dst[j];
          d[0] = x[j].x; d[1] = x[j].y; d[2] = x[j].z; d[3] = x[j].w;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 32 * t + c;
      const int colc = col < k ? col : k - 1;
      const int64_t cb = (int64_t)(colc / kc) * blk_stride + (colc % kc);
      int ps[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int64_t row = row0 + rr, rowc = row < n ? row : n - 1;
        if constexpr (STAGED) vn[t][i] = stage[wave][rr * PITCH + col];
        else vn[t][i] = v_recv[cb + rowc * kc];
        ps[i] = -1;
      }
      if (pos_all) {               // uniform: ONE block with the tile column's 16 position loads
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          ps[i] = pos_all[(row < n ? row : n - 1) * k + colc];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool valid = row < n && col < k;
        const float v = valid ? vn[t][i] : 0.f;
        vn[t][i] = v;
        if (valid) V[row * k + col] = v;
        // V_new on the active supports, for the deferred objective (prmf_end)
        if (valid && ps[i] >= 0) vsup[(unsigned)col * (unsigned)max_support + (unsigned)ps[i]] = v;
      }
    }
    if constexpr (GRAM) {
      if (row0 < n) gram_accumulate<NT>(vn, g);
    }
    if constexpr (W16) {
#pragma unroll
      for (int t = 0; t < NT; ++t) pack_cd_w16(vn[t], row0, t, NT, lane, packed);
    } else {
      pack_tile_from_cd<NT>(vn, row0, lane, packed);
    }
  }
  if constexpr (GRAM) gram_block_store<NT>(g, k, gram_partial + (int64_t)blockIdx.x * k * k, red);
}

hipError_t launch_v_post(float* V, int64_t n, int k, const float* v_recv, int64_t blk_stride, int kc, int KP, int x_dtype,
                         void* packed, int64_t Kpad, float* gram_partial, bool* gram_done, const int32_t* pos_all, float* vsup,
                         int max_support, hipStream_t s, bool w16) {
  if (x_dtype != 1 || (pos_all && !vsup) || (w16 && KP < 64)) return hipErrorInvalidValue;
  const int NT = KP / 32;
  const int64_t ntiles = Kpad / 32;
  const int grid = epilogue_blocks(Kpad);
  *gram_done = NT <= 2;
  // the staged copy places rank block b at columns [b kc, (b + 1) kc) of a KP-wide tile: when the blocks together are wider
  // than the tile (k = 96 on five ranks: 5 x 20 = 100 > 96) the copy neither fits the tile nor its 4 NT load slots per lane
  // -- those shapes read the blocks directly
  const int nblocks_src = kc > 0 ? (k + kc - 1) / kc : 0;
  const bool staged = (kc & 3) == 0 && (blk_stride & 3) == 0 && nblocks_src * kc <= KP;
#define PRMF_VPOST(NTV, GR)                                                                                      \
  do {                                                                                                           \
    if (staged) v_post_kernel<NTV, GR, true, false><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support); \
    else v_post_kernel<NTV, GR, false, false><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support); \
  } while (0)
#define PRMF_VPOST_W(NTV)                                                                                        \
  do {                                                                                                           \
    if (!w16) { PRMF_VPOST(NTV, false); }                                                                        \
    else if (staged) v_post_kernel<NTV, false, true, true><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support); \
    else v_post_kernel<NTV, false, false, true><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support); \
  } while (0)
  switch (NT) {
    case 1: PRMF_VPOST(1, true); break;
    case 2:
      if (!w16) { PRMF_VPOST(2, true); }
      else if (staged) v_post_kernel<2, true, true, true><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support);
      else v_post_kernel<2, true, false, true><<<grid, 256, 0, s>>>(V, n, k, v_recv, blk_stride, kc, packed, ntiles, gram_partial, pos_all, vsup, max_support);
      break;
    case 3: PRMF_VPOST_W(3); break;
    case 4: PRMF_VPOST_W(4); break;
    default: return hipErrorInvalidValue;
  }
#undef PRMF_VPOST_W
#undef PRMF_VPOST
  return hipGetLastError();
}

// reduce `nparts` per-workgroup Gram partials (written by an epilogue) into G
hipError_t launch_gram_reduce(const float* partial, int nparts, int k, double* G, hipStream_t s);

hipError_t launch_gram_reduce(const float* partial, int nparts, int k, double* G, hipStream_t s) {
  gram_reduce_kernel<<<gram_reduce_jobs(k), 256, 0, s>>>(partial, nparts, k, G);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Q slabs -> per-owner column blocks  q_send[r][j][c],  column kk = r*kc + c
// ---------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256)
slab_sum_blocks_kernel(const float* __restrict__ slabs, const segmask_t* __restrict__ segmask, int panel_rows,
                       int64_t slab_stride, int64_t n, int k, int kc, float* __restrict__ q_send, int wpt) {
  // `wpt` waves per 32-row tile (cooperative slab sum, see my_segments): fragment-order 16-B loads, summed over the
  // tile's segments, written by owner block
  constexpr int COMB = NT <= 2 ? 4 * NT * 16 * 64 : 1;
  __shared__ float comb[COMB];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5;
  const int tpw = 4 / wpt, sub = wave % wpt, tslot = wave / wpt;
  const int64_t ntiles = (n + 31) / 32;
  for (int64_t base = (int64_t)blockIdx.x * tpw; base < ntiles; base += (int64_t)gridDim.x * tpw) {
    const int64_t tile = base + tslot;
    const int64_t row0 = tile * 32;
    const bool live = tile < ntiles;
    float q[NT][16];
    if (live) load_slab_tile<NT, NT == 1 ? 4 : PRMF_EPI_NB2>(slabs, slab_stride, my_segments(segmask[row0 / panel_rows], wpt, sub), tile, lane, q);
    else load_slab_tile<NT>(slabs, slab_stride, segmask_t(0), tile, lane, q);
    if constexpr (NT <= 2) combine_tile_sums<NT>(q, wpt, sub, wave, lane, comb);
    if (!live || sub != 0) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 32 * t + c;
      if (col < k) {
        const int r = col / kc, cc = col - r * kc;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (row < n) q_send[((int64_t)r * n + row) * kc + cc] = q[t][i];
        }
      }
    }
  }
}

hipError_t launch_slab_sum_blocks(const float* slabs, const segmask_t* segmask, int panel_rows, int64_t slab_stride, int KP,
                                  int64_t n, int k, int kc, int nranks, float* q_send, int max_seg, hipStream_t s) {
  (void)nranks;
  const int NT = KP / 32;
  const int wpt = epilogue_wpt(n, max_seg, NT);
  const int64_t tiles = (n + 31) / 32;
  const int tpw = 4 / wpt;
  const int grid = (int)((tiles + tpw - 1) / tpw < 2048 ? (tiles + tpw - 1) / tpw : 2048);
  switch (NT) {
    case 1: slab_sum_blocks_kernel<1><<<grid, 256, 0, s>>>(slabs, segmask, panel_rows, slab_stride, n, k, kc, q_send, wpt); break;
    case 2: slab_sum_blocks_kernel<2><<<grid, 256, 0, s>>>(slabs, segmask, panel_rows, slab_stride, n, k, kc, q_send, wpt); break;
    case 3: slab_sum_blocks_kernel<3><<<grid, 256, 0, s>>>(slabs, segmask, panel_rows, slab_stride, n, k, kc, q_send, 1); break;
    case 4: slab_sum_blocks_kernel<4><<<grid, 256, 0, s>>>(slabs, segmask, panel_rows, slab_stride, n, k, kc, q_send, 1); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host scalars -> device state without a staging buffer (kernel arguments): no
// pinned-memory reuse hazard, so prmf_begin / prmf_set_active need no stream sync
// ---------------------------------------------------------------------------
__global__ void set_hyper_kernel(double* __restrict__ hyper, double gamma, double delta, double tradeoff) {
  hyper[0] = gamma;
  hyper[1] = delta;
  hyper[2] = tradeoff;
  hyper[3] = 0.0;
}
hipError_t launch_set_hyper(double* hyper, double gamma, double delta, double tradeoff, hipStream_t s) {
  set_hyper_kernel<<<1, 1, 0, s>>>(hyper, gamma, delta, tradeoff);
  return hipGetLastError();
}
struct ActiveArgs { int32_t v[MAX_KP]; };
__global__ void set_active_kernel(int32_t* __restrict__ active, ActiveArgs a, int k) {
  if ((int)threadIdx.x < k) active[threadIdx.x] = a.v[threadIdx.x];
}
hipError_t launch_set_active(int32_t* active_dev, const int32_t* host_vals, int k, hipStream_t s) {
  ActiveArgs a;
  for (int i = 0; i < MAX_KP; ++i) a.v[i] = i < k ? host_vals[i] : 0;
  set_active_kernel<<<1, MAX_KP, 0, s>>>(active_dev, a, k);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// active pathway -> dense position map for the own columns
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
build_pos_kernel(PathwayLibDev lib, const int32_t* __restrict__ active, int col0, int kc, int32_t* __restrict__ pos) {
  const int c = blockIdx.x;
  const int p = active[col0 + c];
  const int64_t s0 = lib.sup_ptr[p];
  const int ns = (int)(lib.sup_ptr[p + 1] - s0);
  for (int s = threadIdx.x; s < ns; s += 256) pos[(int64_t)lib.sup_idx[s0 + s] * kc + c] = s;
}

hipError_t launch_build_pos(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols, int kc, int64_t n,
                            int32_t* pos, hipStream_t s) {
  hipError_t e = hipMemsetAsync(pos, 0xFF, (size_t)n * kc * sizeof(int32_t), s);
  if (e != hipSuccess) return e;
  if (ncols > 0) build_pos_kernel<<<ncols, 256, 0, s>>>(lib, active_dev, col0, kc, pos);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sparse pathway terms of the V update for the own columns   (:431-438)
//   gnum = gamma * (W_k v_k) + delta * (v_k + 1)^-2   on the support S_k
//   gden = gamma * deg_k * v_k
// One workgroup per latent column: the column's support values are gathered
// once into LDS, the CSR rows (support-local column indices) then read LDS only.
// ---------------------------------------------------------------------------
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
graph_terms_kernel(PathwayLibDev lib, const int32_t* __restrict__ active, int col0, const float* __restrict__ V, int k,
                   const double* __restrict__ hyper, float* __restrict__ gnum, float* __restrict__ gden) {
  extern __shared__ __attribute__((aligned(16))) float vs[];
  graph_terms_body<USE_LDS>(lib, active, col0, blockIdx.x, V, k, hyper, gnum, gden, vs);
}

// One launch for the two small jobs between the U epilogue and the second sweep (single-rank path):
// blocks [0, nred): reduce the U^T U partials; blocks [nred, nred + k): pathway terms of column b - nred.
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
mid_kernel(const float* __restrict__ gram_partial, int nparts, int k, double* __restrict__ Gu, int nred, PathwayLibDev lib,
           const int32_t* __restrict__ active, int col0, const float* __restrict__ V, const double* __restrict__ hyper,
           float* __restrict__ gnum, float* __restrict__ gden) {
  extern __shared__ __attribute__((aligned(16))) float vs[];
  __shared__ double red[256];
  if ((int)blockIdx.x < nred) gram_reduce_job(gram_partial, nparts, k, Gu, nullptr, blockIdx.x, red);
  else graph_terms_body<USE_LDS>(lib, active, col0, blockIdx.x - nred, V, k, hyper, gnum, gden, vs);
}

// (col0, ncols): the latent columns this rank owns (all k with one rank)
hipError_t launch_mid(const float* gram_partial, int nparts, int k, double* Gu, const PathwayLibDev& lib,
                      const int32_t* active_dev, int col0, int ncols, const float* V, const double* hyper, float* gnum,
                      float* gden, hipStream_t s) {
  const int nred = gram_reduce_jobs(k);
  const size_t lds = (size_t)lib.max_support * sizeof(float);
  if (lds <= 64 * 1024) mid_kernel<true><<<nred + ncols, 256, lds, s>>>(gram_partial, nparts, k, Gu, nred, lib, active_dev, col0, V, hyper, gnum, gden);
  else mid_kernel<false><<<nred + ncols, 256, 0, s>>>(gram_partial, nparts, k, Gu, nred, lib, active_dev, col0, V, hyper, gnum, gden);
  return hipGetLastError();
}

hipError_t launch_graph_terms(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols, const float* V,
                              int k, const double* hyper, float* gnum, float* gden, hipStream_t s) {
  if (ncols <= 0) return hipSuccess;
  size_t lds = (size_t)lib.max_support * sizeof(float);
  if (lds <= 64 * 1024)
    graph_terms_kernel<true><<<ncols, 256, lds, s>>>(lib, active_dev, col0, V, k, hyper, gnum, gden);
  else
    graph_terms_kernel<false><<<ncols, 256, 0, s>>>(lib, active_dev, col0, V, k, hyper, gnum, gden);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// V update for the own columns   (:424-425, :440-444), column-sharded multi-rank mode
//   V_own <- V_own * (Q_own + gnum) / max(V (U^T U)[:, own] + gden, eps), clamp at eps
// One wave per 32-row tile, as the fused epilogues: den = V_tile (A layout, all k columns) x G[:, own columns] on the
// f32 MFMA, the element-wise update in the C/D layout (lane = own column, registers = rows), own columns written to
// this rank's all-gather block, per-workgroup float64 sums of squares / <Q, V_new> to its tail.
// (The first version -- one thread per (row, 4 columns), scalar dot products from an LDS copy of the rows, 157 workgroups
// at 20000 genes -- took 19.5 us per iteration for 8 own columns: profiles/r03_share8_kernel_stats.txt.)
// ---------------------------------------------------------------------------
constexpr int VUPD_MAX_BLOCKS = 256;   // its per-workgroup sums travel in the tail of the all-gather: keep them few
int v_update_blocks(int64_t n) {
  const int64_t blocks = ((n + 31) / 32 + 3) / 4;
  return (int)(blocks < 1 ? 1 : (blocks < VUPD_MAX_BLOCKS ? blocks : VUPD_MAX_BLOCKS));
}

// NT: 32-column tiles of all k columns; NTO: 32-column tiles of the own columns (kc <= 32 NTO)
template <int NT, int NTO, bool VEC4>
__global__ void __launch_bounds__(256)
v_update_kernel(const float* __restrict__ V, int64_t n, int k, int col0, int ncols, int kc, const float* __restrict__ q_recv,
                const double* __restrict__ Gu, const int32_t* __restrict__ pos, const float* __restrict__ gnum,
                const float* __restrict__ gden, int max_support, float* __restrict__ v_send, double* __restrict__ partial) {
  constexpr int KP = 32 * NT, HALF = KP / 2, OW = 32 * NTO;
  __shared__ float Gs[KP * OW];                  // G[:, col0 .. col0 + ncols), zero beyond
  __shared__ double red[4][OW + 1];
  {
    // loads in batches BEFORE the LDS stores of the batch, indices clamped and values masked (see stage_gram: one load and
    // one store per loop iteration is a chain of KP OW / 256 dependent L2 round trips)
    constexpr int PER = KP * OW / 256, B = PER % 16 == 0 ? 16 : (PER % 12 == 0 ? 12 : (PER % 8 == 0 ? 8 : 4));
    static_assert(PER % B == 0, "whole batches");
#pragma unroll 1
    for (int b0 = 0; b0 < PER; b0 += B) {
      double v[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int i = (int)threadIdx.x + 256 * (b0 + j);
        const int q = i / OW, oc = i - q * OW;
        const bool in = q < k && oc < ncols;
        v[j] = Gu[in ? q * k + col0 + oc : 0];
        if (!in) v[j] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < B; ++j) Gs[(int)threadIdx.x + 256 * (b0 + j)] = (float)v[j];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 31, h = lane >> 5, r = lane & 31;
  double colsq[NTO], cross = 0.0;
#pragma unroll
  for (int to = 0; to < NTO; ++to) colsq[to] = 0.0;
  const int64_t ntiles = (n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row0 = tile * 32;
    float ua[16 * NT];
    load_tile_a_layout<NT, VEC4>(V, n, k, row0, lane, ua);
    f32x16 den[NTO];
#pragma unroll
    for (int to = 0; to < NTO; ++to)
#pragma unroll
      for (int i = 0; i < 16; ++i) den[to][i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < HALF; ++s2)
#pragma unroll
      for (int to = 0; to < NTO; ++to)
        den[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[s2], Gs[(s2 + HALF * h) * OW + 32 * to + r], den[to], 0, 0, 0);
    const int last = (int)((n - row0 < 32 ? n - row0 : 32) - 1);
#pragma unroll
    for (int to = 0; to < NTO; ++to) {
      const int oc = 32 * to + c, occ = oc < ncols ? oc : ncols - 1;       // own column (clamped for the loads)
      float qv[16], vv[16], gn[16], gd[16];
      int ps[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int64_t row = row0 + (rr < last ? rr : last);
        qv[i] = q_recv[row * kc + occ];
        vv[i] = V[row * k + col0 + occ];
        ps[i] = pos[row * kc + occ];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {          // branch-free gathers: off-support entries read element 0 and are masked
        const unsigned goff = ps[i] >= 0 ? (unsigned)occ * (unsigned)max_support + (unsigned)ps[i] : 0u;
        gn[i] = gnum[goff];
        gd[i] = gden[goff];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int64_t row = row0 + rr;
        const bool on = ps[i] >= 0, valid = row < n && oc < ncols;
        const float num = qv[i] + (on ? gn[i] : 0.f);
        float d = den[to][i] + (on ? gd[i] : 0.f);
        if (d < PRMF_EPS) d = PRMF_EPS;                              // :442
        float vn = vv[i] * (num / d);                                // :443
        if (vn < PRMF_EPS) vn = PRMF_EPS;                            // :444
        if (valid) {
          v_send[row * kc + oc] = vn;
          colsq[to] += (double)vn * (double)vn;
          cross += (double)qv[i] * (double)vn;
        }
      }
    }
  }
  // per-workgroup sums (fixed order: half-waves, lanes, then the 4 waves): own-column sums of squares, <Q, V_new> last
#pragma unroll
  for (int to = 0; to < NTO; ++to) {
    const double tot = colsq[to] + __shfl_xor(colsq[to], 32);
    if (h == 0) red[wave][32 * to + c] = tot;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cross += __shfl_xor(cross, off);
  if (lane == 0) red[wave][OW] = cross;
  __syncthreads();
  double* out = partial + (int64_t)blockIdx.x * (kc + 1);
  for (int j = threadIdx.x; j <= kc; j += 256) {
    const int src = j < kc ? j : OW;
    out[j] = (j < kc && j >= ncols) ? 0.0 : ((red[0][src] + red[1][src]) + red[2][src]) + red[3][src];
  }
}

hipError_t launch_v_update(const float* V, int64_t n, int k, int col0, int ncols, int kc, const float* q_recv,
                           const double* Gu, const int32_t* pos, const float* gnum, const float* gden, int max_support,
                           float* v_send, double* partial, hipStream_t s) {
  const int nblk = v_update_blocks(n);           // every workgroup writes its row of `partial` (zeros if it had no tile)
  if (ncols <= 0) return hipMemsetAsync(partial, 0, (size_t)nblk * (kc + 1) * sizeof(double), s);
  const int NT = (k + 31) / 32, NTO = (kc + 31) / 32;
  const bool vec4 = (k & 3) == 0;
#define PRMF_VUPD(NTV, NTOV)                                                                                             \
  do {                                                                                                                    \
    if (vec4) v_update_kernel<NTV, NTOV, true><<<nblk, 256, 0, s>>>(V, n, k, col0, ncols, kc, q_recv, Gu, pos, gnum, gden, max_support, v_send, partial); \
    else v_update_kernel<NTV, NTOV, false><<<nblk, 256, 0, s>>>(V, n, k, col0, ncols, kc, q_recv, Gu, pos, gnum, gden, max_support, v_send, partial); \
  } while (0)
  if (NTO == 1) {
    switch (NT) {
      case 1: PRMF_VUPD(1, 1); break;
      case 2: PRMF_VUPD(2, 1); break;
      case 3: PRMF_VUPD(3, 1); break;
      case 4: PRMF_VUPD(4, 1); break;
      default: return hipErrorInvalidValue;
    }
  } else if (NTO == 2) {                         // kc in (32, 64]: two ranks at k > 64
    switch (NT) {
      case 3: PRMF_VUPD(3, 2); break;
      case 4: PRMF_VUPD(4, 2); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    return hipErrorInvalidValue;                 // kc > 64 needs nranks == 1, which takes the fused epilogue
  }
#undef PRMF_VUPD
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// objective pieces of the own columns   (:344-352 with normalize_laplacian :56-65)
//   vhat = v / ||v|| ; y = D^-1/2 vhat on the support
//   manifold_k = y^T (D y - W y) ; ignore_k = sum_{i in S} 1 / (vhat_i + 1)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void objective_col_body(const PathwayLibDev& lib, const int32_t* __restrict__ active, int col0, int kc, int c,
                                                   const float* __restrict__ v_send, const double* __restrict__ partial, int nblocks,
                                                   double* __restrict__ colres, double* __restrict__ scratch, double* red) {
  const int tid = threadIdx.x;
  double sq = 0.0;
  for (int b = tid; b < nblocks; b += 256) sq += partial[(int64_t)b * (kc + 1) + c];
  sq = block_sum_256(sq, red);
  const double inv = 1.0 / sqrt(sq);
  const int p = active[col0 + c];
  const int64_t s0 = lib.sup_ptr[p];
  const int ns = (int)(lib.sup_ptr[p + 1] - s0);
  double* y = scratch + (int64_t)c * lib.max_support;
  double ign = 0.0;
  for (int s = tid; s < ns; s += 256) {
    const double vh = (double)v_send[(int64_t)lib.sup_idx[s0 + s] * kc + c] * inv;
    y[s] = vh * lib.sup_dis[s0 + s];
    ign += 1.0 / (vh + 1.0);
  }
  __threadfence_block();
  __syncthreads();
  double man = 0.0;
  for (int s = tid; s < ns; s += 256) {
    const double ys = y[s];
    double wy = 0.0;
    for (int64_t e = lib.row_ptr[s0 + s]; e < lib.row_ptr[s0 + s + 1]; ++e) wy += lib.w[e] * y[lib.col_local[e]];
    man += ys * (lib.sup_deg[s0 + s] * ys - wy);
  }
  man = block_sum_256(man, red);
  ign = block_sum_256(ign, red);
  if (tid == 0) {
    colres[2 * c] = man;
    colres[2 * c + 1] = ign;
  }
}

__global__ void __launch_bounds__(256)
objective_col_kernel(PathwayLibDev lib, const int32_t* __restrict__ active, int col0, int kc,
                     const float* __restrict__ v_send, const double* __restrict__ partial, int nblocks,
                     double* __restrict__ colres, double* __restrict__ scratch) {
  __shared__ double red[256];
  objective_col_body(lib, active, col0, kc, blockIdx.x, v_send, partial, nblocks, colres, scratch, red);
}

// One launch after the V epilogue (single-rank path): blocks [0, nred): reduce the V^T V partials;
// blocks [nred, nred + k): objective pieces of column b - nred.
__global__ void __launch_bounds__(256)
post_kernel(const float* __restrict__ gram_partial, int nparts, int k, double* __restrict__ Gv, int nred, PathwayLibDev lib,
            const int32_t* __restrict__ active, const float* __restrict__ V, const double* __restrict__ partial, int nblocks,
            double* __restrict__ colres, double* __restrict__ scratch) {
  __shared__ double red[256];
  if ((int)blockIdx.x < nred) gram_reduce_job(gram_partial, nparts, k, Gv, nullptr, blockIdx.x, red);
  else objective_col_body(lib, active, 0, k, blockIdx.x - nred, V, partial, nblocks, colres, scratch, red);
}

hipError_t launch_post(const float* gram_partial, int nparts, int k, double* Gv, const PathwayLibDev& lib,
                       const int32_t* active_dev, const float* V, const double* partial, int nblocks, double* colres,
                       double* scratch, hipStream_t s) {
  const int nred = gram_reduce_jobs(k);
  post_kernel<<<nred + k, 256, 0, s>>>(gram_partial, nparts, k, Gv, nred, lib, active_dev, V, partial, nblocks, colres, scratch);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256)
rank_extras_kernel(const double* __restrict__ partial, int nblocks, int kc, int ncols, const double* __restrict__ colres,
                   double* __restrict__ extras) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double cross = 0.0;
  for (int b = tid; b < nblocks; b += 256) cross += partial[(int64_t)b * (kc + 1) + kc];
  cross = block_sum_256(cross, red);
  double man = 0.0, ign = 0.0;
  for (int c = tid; c < ncols; c += 256) {
    man += colres[2 * c];
    ign += colres[2 * c + 1];
  }
  man = block_sum_256(man, red);
  ign = block_sum_256(ign, red);
  if (tid == 0) {
    extras[0] = cross;
    extras[1] = man;
    extras[2] = ign;
    extras[3] = 0.0;
  }
}

hipError_t launch_objective_terms(const PathwayLibDev& lib, const int32_t* active_dev, int col0, int ncols, int kc,
                                  int64_t n, const float* v_send, const double* partial, int nblocks, double* colres,
                                  double* scratch, double* extras_out, hipStream_t s) {
  (void)n;
  if (ncols > 0)
    objective_col_kernel<<<ncols, 256, 0, s>>>(lib, active_dev, col0, kc, v_send, partial, nblocks, colres, scratch);
  rank_extras_kernel<<<1, 256, 0, s>>>(partial, nblocks, kc, ncols, colres, extras_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// replicated V (row-major) from the gathered per-rank column blocks
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// objective record (:362-371) + tradeoff feedback (:542-548)
//   recon^2 = ||X||^2 - 2 <X^T U, V> + <U^T U, V^T V>
// ---------------------------------------------------------------------------
// `loc_partial` != nullptr (single rank): cross / manifold / ignore are summed here from the V epilogue's
// per-wave partials and the per-column results instead of being read from the gathered rank extras.
__global__ void __launch_bounds__(256)
finalize_kernel(const double* __restrict__ Gu, const double* __restrict__ Gv, int k, const float* __restrict__ v_recv,
                int nranks, int64_t blk_stride, int64_t extras_off, double norm_x_sq, double* __restrict__ hyper,
                double* __restrict__ record, const double* __restrict__ loc_partial, int loc_nblocks,
                const double* __restrict__ loc_colres,
                int* __restrict__ need /* nullable: direct-residual request (residual.hip) */, int force, double direct_below) {
  __shared__ double red[5][256];
  const int tid = threadIdx.x;
  // five sums, one tree reduction (fixed order: deterministic)
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // <Gu,Gv>, trace Gu, cross, manifold, ignore
  for (int i = tid; i < k * k; i += 256) acc[0] += Gu[i] * Gv[i];
  for (int a = tid; a < k; a += 256) acc[1] += Gu[a * k + a];
  if (loc_partial) {
    for (int b = tid; b < loc_nblocks; b += 256) acc[2] += loc_partial[(int64_t)b * (k + 1) + k];
    for (int c = tid; c < k; c += 256) {
      acc[3] += loc_colres[2 * c];
      acc[4] += loc_colres[2 * c + 1];
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) red[j][tid] = acc[j];
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int j = 0; j < 5; ++j) red[j][tid] += red[j][tid + s];
    }
    __syncthreads();
  }
  const double gg = red[0][0], fro = red[1][0], lcross = red[2][0], lman = red[3][0], lign = red[4][0];
  if (tid == 0) {
    double cross = lcross, man = lman, ign = lign;
    if (!loc_partial)
      for (int r = 0; r < nranks; ++r) {
        const double* ex = reinterpret_cast<const double*>(v_recv + (int64_t)r * blk_stride + extras_off);
        cross += ex[0];
        man += ex[1];
        ign += ex[2];
      }
    const double gamma = hyper[0], delta = hyper[1], t = hyper[2];
    const double recon_sq = norm_x_sq - 2.0 * cross + gg;
    const double recon = sqrt(recon_sq > 0.0 ? recon_sq : 0.0);
    record[0] = recon;
    record[1] = man;
    record[2] = ign;
    record[3] = fro;
    record[4] = gamma;
    record[5] = delta;
    record[6] = recon + gamma * man + delta * ign + fro;
    record[7] = recon_sq;
    // The identity's recon^2 is exact to ~1e-8 (k <= 64) / ~1e-7 (k > 64) ||X||^2: below `direct_below` the record is marked, and
    // the last record of a batch (every record in the per-iteration form) asks for the direct residual, which then
    // replaces recon / obj and drives the tradeoff feedback.  A negative value is never clamped silently.
    const bool low = recon_sq < direct_below;
    const bool ask = need && (low || force);
    record[REC_FLAGS] = (double)((low ? REC_FLAG_IMPRECISE : 0) | (recon_sq < 0.0 ? REC_FLAG_CLAMPED : 0));
    record[REC_RECON_SQ_IDENTITY] = recon_sq;
    if (need) *need = ask ? 1 : 0;
    if (t >= 0.0 && !ask) {
      const double denom = t * man;
      const double g2 = (denom == 0.0) ? 1.0 : ((1.0 - t) * recon) / denom;
      hyper[0] = g2;
      hyper[1] = g2;
    }
  }
}

hipError_t launch_finalize(const double* Gu, const double* Gv, int k, const float* v_recv, int nranks,
                           int64_t blk_stride, int64_t extras_off, double norm_x_sq, double* hyper, double* record,
                           const double* loc_partial, int loc_nblocks, const double* loc_colres, int* need, bool force,
                           double direct_below, hipStream_t s) {
  finalize_kernel<<<1, 256, 0, s>>>(Gu, Gv, k, v_recv, nranks, blk_stride, extras_off, norm_x_sq, hyper, record,
                                    loc_partial, loc_nblocks, loc_colres, need, force ? 1 : 0, direct_below);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Deferred objective (single rank, tradeoff off).  Nothing of the objective record (:336-372) feeds the next
// update unless the tradeoff feedback is on, so the iterations only KEEP what the record needs -- U^T U and
// V^T V (reduced anyway), <Q, V_new>, the column norms and V_new on the active supports (a few hundred KB per
// iteration) -- and the whole batch is evaluated here in two launches at prmf_end.  Same arithmetic, same
// reduction orders as objective_col_body / finalize_kernel: the records are bit-identical to the per-iteration form.
// ---------------------------------------------------------------------------
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
objective_all_kernel(PathwayLibDev lib, const int32_t* __restrict__ active, int k, const double* __restrict__ colnorm2,
                     const float* __restrict__ vsup, double* __restrict__ colres, double* __restrict__ scratch, int it0) {
  extern __shared__ double y_lds[];
  __shared__ double red[256];
  const int c = blockIdx.x, it = it0 + blockIdx.y, tid = threadIdx.x;
  const double inv = 1.0 / sqrt(colnorm2[(int64_t)it * k + c]);
  const int p = active[c];
  const int64_t s0 = lib.sup_ptr[p];
  const int ns = (int)(lib.sup_ptr[p + 1] - s0);
  double* y = USE_LDS ? y_lds : scratch + (int64_t)c * lib.max_support;
  const float* vs = vsup + ((int64_t)it * k + c) * lib.max_support;
  double ign = 0.0;
  for (int s = tid; s < ns; s += 256) {
    const double vh = (double)vs[s] * inv;
    y[s] = vh * lib.sup_dis[s0 + s];
    ign += 1.0 / (vh + 1.0);
  }
  __threadfence_block();
  __syncthreads();
  double man = 0.0;
  for (int s = tid; s < ns; s += 256) {
    const double ys = y[s];
    double wy = 0.0;
    for (int64_t e = lib.row_ptr[s0 + s]; e < lib.row_ptr[s0 + s + 1]; ++e) wy += lib.w[e] * y[lib.col_local[e]];
    man += ys * (lib.sup_deg[s0 + s] * ys - wy);
  }
  man = block_sum_256(man, red);
  ign = block_sum_256(ign, red);
  if (tid == 0) {
    colres[((int64_t)it * k + c) * 2] = man;
    colres[((int64_t)it * k + c) * 2 + 1] = ign;
  }
}

hipError_t launch_objective_all(const PathwayLibDev& lib, const int32_t* active_dev, int k, int iters, const double* colnorm2,
                                const float* vsup, double* colres, double* scratch, hipStream_t s) {
  if (iters <= 0) return hipSuccess;
  const size_t lds = (size_t)lib.max_support * sizeof(double);
  if (lds <= 48 * 1024) {
    objective_all_kernel<true><<<dim3(k, iters), 256, lds, s>>>(lib, active_dev, k, colnorm2, vsup, colres, scratch, 0);
  } else {
    // supports too large for LDS: one iteration per launch through the [k][max_support] global scratch
    for (int it = 0; it < iters; ++it)
      objective_all_kernel<false><<<dim3(k, 1), 256, 0, s>>>(lib, active_dev, k, colnorm2, vsup, colres, scratch, it);
  }
  return hipGetLastError();
}

__global__ void __launch_bounds__(256)
records_all_kernel(const double* __restrict__ Gu_keep, const double* __restrict__ Gv_keep, int k, const double* __restrict__ cross,
                   const double* __restrict__ colres, double norm_x_sq, const double* __restrict__ hyper,
                   double* __restrict__ records, int* __restrict__ need, int force, double direct_below) {
  __shared__ double red[4][256];
  const int tid = threadIdx.x, it = blockIdx.x;
  const double* Gu = Gu_keep + (int64_t)it * k * k;
  const double* Gv = Gv_keep + (int64_t)it * k * k;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // <Gu,Gv>, trace Gu, manifold, ignore
  for (int i = tid; i < k * k; i += 256) acc[0] += Gu[i] * Gv[i];
  for (int a = tid; a < k; a += 256) acc[1] += Gu[a * k + a];
  for (int c = tid; c < k; c += 256) {
    acc[2] += colres[((int64_t)it * k + c) * 2];
    acc[3] += colres[((int64_t)it * k + c) * 2 + 1];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[j][tid] = acc[j];
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[j][tid] += red[j][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double gg = red[0][0], fro = red[1][0], man = red[2][0], ign = red[3][0];
    const double gamma = hyper[0], delta = hyper[1];
    const double recon_sq = norm_x_sq - 2.0 * cross[it] + gg;
    const double recon = sqrt(recon_sq > 0.0 ? recon_sq : 0.0);
    double* record = records + (int64_t)it * REC_STRIDE;
    record[0] = recon;
    record[1] = man;
    record[2] = ign;
    record[3] = fro;
    record[4] = gamma;
    record[5] = delta;
    record[6] = recon + gamma * man + delta * ign + fro;
    record[7] = recon_sq;
    const bool low = recon_sq < direct_below;                                  // see finalize_kernel
    record[REC_FLAGS] = (double)((low ? REC_FLAG_IMPRECISE : 0) | (recon_sq < 0.0 ? REC_FLAG_CLAMPED : 0));
    record[REC_RECON_SQ_IDENTITY] = recon_sq;
    if (need && it == (int)gridDim.x - 1) *need = (low || force) ? 1 : 0;     // the last record of the batch
  }
}

hipError_t launch_records_all(const double* Gu_keep, const double* Gv_keep, int k, int iters, const double* cross,
                              const double* colres, double norm_x_sq, const double* hyper, double* records, int* need, bool force,
                              double direct_below, hipStream_t s) {
  if (iters <= 0) return hipSuccess;
  records_all_kernel<<<iters, 256, 0, s>>>(Gu_keep, Gv_keep, k, cross, colres, norm_x_sq, hyper, records, need, force ? 1 : 0, direct_below);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// candidate scoring for restrict (:115-127): one wave per (pathway, latent)
//   score = ||vhat[S_p]||_2 + 1 - vhat^T Lhat_p vhat
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) colnorm_kernel(const float* __restrict__ V, int64_t n, int k, double* __restrict__ out) {
  __shared__ double red[256];
  const int kk = blockIdx.x;
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    double v = (double)V[j * k + kk];
    s += v * v;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[kk] = sqrt(s);
}

// One workgroup per (pathway p, block of 32 latent columns).  Everything the inner loop touches is staged in LDS
// first: the |S_p| x 32 block of V as 128-B row segments (coalesced: 32 consecutive latents of one gene) and the
// pathway's CSR (row pointers, support-local column indices, weights) with D^-1/2 and the degrees.  Lane j of every
// wave then owns latent kk0 + j and walks the CSR rows of its eighth of the support from LDS only (bank = latent:
// conflict-free; CSR entries are wave-uniform broadcasts).  The first version (one wave per (p, latent), single-float
// gathers of V and CSR loads from global memory per non-zero) took 0.15 / 0.98 ms of kernel time at (k=64, P=300) /
// (k=128, P=1000); staging V alone left the dependent global CSR loads: 0.078 / 0.47 ms.
struct ScoreLds {
  float* ys;       // [ns][33]
  double* dis;     // [ns]
  double* deg;     // [ns]
  double* w;       // [nnz]
  int* rowp;       // [ns + 1]  (relative to the pathway's first entry)
  int* col;        // [nnz]
};
__host__ __device__ inline size_t score_lds_bytes(int ns, int nnz) {
  return (size_t)ns * 33 * 4 + (size_t)ns * 16 + (size_t)nnz * 8 + (size_t)(ns + 1) * 4 + (size_t)nnz * 4 + 16;
}

// RAW (prmf_match_terms: the weights of force_distinct_lapls, prmf_runner.py:227-246): the same walk on the column as it
// is -- no column normalisation, D^-1/2 = 1 -- gives v^T L_p v = sum_s v_s (deg_s v_s - sum_e w_e v_c(e)) into `scores`
// and sum_s 1 / (v_s + 1) over the support into `scores2`.
template <bool USE_LDS, bool RAW>
__global__ void __launch_bounds__(256)
score_kernel(PathwayLibDev lib, const float* __restrict__ V, int k, const double* __restrict__ colnorm,
             double* __restrict__ scores, double* __restrict__ scores2) {
  extern __shared__ double score_smem[];
  __shared__ double red[2][4][32];
  const int p = blockIdx.x, kk0 = blockIdx.y * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, half = lane >> 5;    // latent within the block; the two half-waves split the rows
  const int kk = kk0 + j;
  const int64_t s0 = lib.sup_ptr[p];
  const int ns = (int)(lib.sup_ptr[p + 1] - s0);
  const int64_t e0 = lib.row_ptr[s0];
  const int nnz = (int)(lib.row_ptr[s0 + ns] - e0);
  const double inv = RAW ? 1.0 : (kk < k ? 1.0 / colnorm[kk] : 0.0);
  ScoreLds L;
  if constexpr (USE_LDS) {
    // doubles first (8-byte aligned), then the 4-byte arrays
    L.dis = score_smem;
    L.deg = L.dis + ns;
    L.w = L.deg + ns;
    L.ys = reinterpret_cast<float*>(L.w + nnz);
    L.rowp = reinterpret_cast<int*>(L.ys + (size_t)ns * 33);
    L.col = L.rowp + (ns + 1);
    for (int e = tid; e < ns * 32; e += 256) {
      const int s = e >> 5, jj = e & 31;
      const int col = kk0 + jj;
      L.ys[s * 33 + jj] = col < k ? V[(int64_t)lib.sup_idx[s0 + s] * k + col] : 0.f;
    }
    for (int s = tid; s < ns; s += 256) {
      L.dis[s] = RAW ? 1.0 : lib.sup_dis[s0 + s];
      L.deg[s] = lib.sup_deg[s0 + s];
    }
    for (int s = tid; s <= ns; s += 256) L.rowp[s] = (int)(lib.row_ptr[s0 + s] - e0);
    for (int e = tid; e < nnz; e += 256) {
      L.col[e] = lib.col_local[e0 + e];
      L.w[e] = lib.w[e0 + e];
    }
    __syncthreads();
  }
  double mass = 0.0, man = 0.0;
  if (kk < k) {
    for (int s = wave * 2 + half; s < ns; s += 8) {
      if constexpr (USE_LDS) {
        const double vh = (double)L.ys[s * 33 + j] * inv;
        const double ysv = vh * L.dis[s];
        double wy = 0.0;
        for (int e = L.rowp[s]; e < L.rowp[s + 1]; ++e) {
          const int cl = L.col[e];
          wy += L.w[e] * ((double)L.ys[cl * 33 + j] * inv * L.dis[cl]);
        }
        mass += RAW ? 1.0 / (vh + 1.0) : vh * vh;
        man += ysv * (L.deg[s] * ysv - wy);
      } else {
        const double vh = (double)V[(int64_t)lib.sup_idx[s0 + s] * k + kk] * inv;
        const double ysv = RAW ? vh : vh * lib.sup_dis[s0 + s];
        double wy = 0.0;
        for (int64_t e = lib.row_ptr[s0 + s]; e < lib.row_ptr[s0 + s + 1]; ++e) {
          const int cl = lib.col_local[e];
          wy += lib.w[e] * ((double)V[(int64_t)lib.sup_idx[s0 + cl] * k + kk] * inv * (RAW ? 1.0 : lib.sup_dis[s0 + cl]));
        }
        mass += RAW ? 1.0 / (vh + 1.0) : vh * vh;
        man += ysv * (lib.sup_deg[s0 + s] * ysv - wy);
      }
    }
  }
  // fixed-order reduction: half-waves, then the 4 waves
  mass += __shfl_xor(mass, 32);
  man += __shfl_xor(man, 32);
  if (half == 0) {
    red[0][wave][j] = mass;
    red[1][wave][j] = man;
  }
  __syncthreads();
  if (tid < 32 && kk0 + tid < k) {
    const double ms = ((red[0][0][tid] + red[0][1][tid]) + red[0][2][tid]) + red[0][3][tid];
    const double mn = ((red[1][0][tid] + red[1][1][tid]) + red[1][2][tid]) + red[1][3][tid];
    if (RAW) {
      scores[(int64_t)(kk0 + tid) * lib.P + p] = mn;
      scores2[(int64_t)(kk0 + tid) * lib.P + p] = ms;
    } else {
      scores[(int64_t)(kk0 + tid) * lib.P + p] = sqrt(ms) + (1.0 - mn);
    }
  }
}

template <bool RAW>
static hipError_t launch_score_form(const PathwayLibDev& lib, const float* V, int k, const double* colnorm, double* scores,
                                    double* scores2, hipStream_t s) {
  const dim3 grid(lib.P, (k + 31) / 32);
  const size_t lds = score_lds_bytes(lib.max_support, lib.max_nnz);
  if (lds <= 150 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_kernel<true, RAW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    score_kernel<true, RAW><<<grid, 256, lds, s>>>(lib, V, k, colnorm, scores, scores2);
  } else {
    score_kernel<false, RAW><<<grid, 256, 0, s>>>(lib, V, k, colnorm, scores, scores2);   // pathway too large for LDS: global gathers
  }
  return hipGetLastError();
}

hipError_t launch_score_candidates(const PathwayLibDev& lib, const float* V, int64_t n, int k, double* colnorm_scratch,
                                   double* scores, hipStream_t s) {
  colnorm_kernel<<<k, 256, 0, s>>>(V, n, k, colnorm_scratch);
  return launch_score_form<false>(lib, V, k, colnorm_scratch, scores, nullptr, s);
}

hipError_t launch_match_terms(const PathwayLibDev& lib, const float* V, int k, double* quad, double* ignore, hipStream_t s) {
  return launch_score_form<true>(lib, V, k, nullptr, quad, ignore, s);
}

}  // namespace prmf
