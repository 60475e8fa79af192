// Direct reconstruction error  sum_ij (X - U V^T)_ij^2  -- what the reference evaluates at
// script/prmf_runner.py:337 (`np.linalg.norm(X - U.dot(V.transpose()))`) -- as one more sweep over the resident X.
//
// Why it exists.  The objective record normally takes recon from the Gram identity
//     ||X - U V^T||^2 = ||X||^2 - 2 <X^T U, V> + <U^T U, V^T V>
// whose three terms are each ~||X||^2 and carry ~1e-7 of relative error (float32 MFMA accumulation of X^T U and of the
// Gram tiles).  The difference is exact to ~1e-7 ||X||^2, i.e. recon to 1e-7 (||X|| / recon)^2 / 2: fine at the 6% fit
// of the reference's test problem (1e-5), useless at a 1% fit (1e-3, the size of the outer loop's `tol`).  The reference
// never cancels.  So when a record's identity value falls below RECON_DIRECT_THRESHOLD * ||X||^2 the library
// evaluates the residual directly (capi.hip: the last record of a batch; every record with the tradeoff feedback on or
// PRMF_FLAG_DIRECT_RESIDUAL) and says so in the record's flags.
//
// The kernel.  Same streaming structure as the sweep (gemm.hip): a workgroup is 4 wave64s x (32 MT) rows, X is loaded
// line-complete one chunk ahead and bounced through the wave-private XOR-swizzled LDS tile.  Per chunk (64 bf16 / 32 f32
// columns of X = GT 32-gene tiles) a wave forms, per (row tile, gene tile), ONE accumulator chain on the bf16 MFMA
//     D = sum_kb (-U)[32 x 16] (V^T)[16 x 32]     both factors split EXACTLY into three bf16 pieces h + m + l (truncation,
//                                                 8 + 8 + 8 mantissa bits), 8 MFMAs per 16 latent columns: every
//                                                 product down to 2^-24 (only l l, 2^-32, is left out)
//       + X_tile[32 x 16 s] I[16 s x 32]          X through the identity: bf16 X is exact in the product, f32 X goes
//                                                 in as its three exact bf16 pieces
// so D = X - U V^T arrives in the accumulators without any transposition of X, and sum D^2 is added up in float64.
// U fragments (A layout: lane (r, h) holds U[row r][16 kb + 8 h .. + 8)) are split from the float32 factor once per
// segment and stay in registers; the chunk's V fragments (B layout of V^T = the same 8 consecutive latents of gene c)
// are split by the workgroup into LDS.
// Why exact pieces and not the sweeps' hi/lo pairs (16 bits): a rounding error dU of the factor is NOT independent
// across the residual's elements -- it enters as the rank-k matrix dU V^T, and <R, dU V^T> = <R V, dU> averages over
// the m k error sources only, not over m n elements.  With 16-bit factors that is 4e-6 (||X|| / recon) / sqrt(m) on
// recon -- 1e-4 at a 1e-3 fit (measured with the first version of this kernel).  With exact pieces the only error left
// is the float32 rounding of the accumulator chain: independent per element, ~1e-7 |x|, invisible in the sum.
// MFMA work is ~3x a sweep's; it runs once per batch at most.
#include "bf16split.h"
#include "kernels.h"
#include "prologue.h"

namespace prmf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Raw8 { float v[8]; };

// F[row][col0 .. col0 + 8), zero outside rows x k.  VEC4: k % 4 == 0 (16-B aligned rows)
template <bool VEC4>
__device__ __forceinline__ Raw8 load8(const float* __restrict__ F, int64_t rows, int k, int64_t row, int col0) {
  Raw8 r;
  if constexpr (VEC4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int col = col0 + 4 * q;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows && col < k) x = *reinterpret_cast<const float4*>(F + row * k + col);
      r.v[4 * q] = x.x; r.v[4 * q + 1] = x.y; r.v[4 * q + 2] = x.z; r.v[4 * q + 3] = x.w;
    }
  } else {
    const float* Fr = F + (row < rows ? row : rows - 1) * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = col0 + e;
      const float x = Fr[col < k ? col : k - 1];
      r.v[e] = (row < rows && col < k) ? x : 0.f;
    }
  }
  return r;
}

// f32 -> three bf16 pieces by truncation, x = h + m + l exactly (as gemm.hip's split_fragment)
__device__ __forceinline__ void split3(const u32x4& f0, const u32x4& f1, u32x4& fh, u32x4& fm, u32x4& fl) {
  const unsigned x[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
  constexpr unsigned UPPER_HALVES = 0x07060302u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned r1[2], r2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned xb = x[2 * j + e];
      r1[e] = __float_as_uint(__uint_as_float(xb) - __uint_as_float(xb & 0xFFFF0000u));
      r2[e] = __float_as_uint(__uint_as_float(r1[e]) - __uint_as_float(r1[e] & 0xFFFF0000u));
    }
    fh[j] = __builtin_amdgcn_perm(x[2 * j + 1], x[2 * j], UPPER_HALVES);
    fm[j] = __builtin_amdgcn_perm(r1[1], r1[0], UPPER_HALVES);
    fl[j] = __builtin_amdgcn_perm(r2[1], r2[0], UPPER_HALVES);
  }
}

__device__ __forceinline__ void split3_raw(const Raw8& r, u32x4& fh, u32x4& fm, u32x4& fl) {
  const u32x4 f0 = {__float_as_uint(r.v[0]), __float_as_uint(r.v[1]), __float_as_uint(r.v[2]), __float_as_uint(r.v[3])};
  const u32x4 f1 = {__float_as_uint(r.v[4]), __float_as_uint(r.v[5]), __float_as_uint(r.v[6]), __float_as_uint(r.v[7])};
  split3(f0, f1, fh, fm, fl);
}

__device__ __forceinline__ void mm(f32x16& d, const u32x4& a, const u32x4& b) {
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), d, 0, 0, 0);
}

}  // namespace

// MODE 1: X bf16 (chunk = 64 columns = 2 gene tiles) | MODE 2: X f32 (chunk = 32 columns = 1 gene tile)
// NKB = ceil(k / 16) rounded to whole 32-column tiles (KP / 16); BDB: V chunk double-buffered in LDS (k <= 32)
template <int MODE, int MT, int NKB, bool BDB, bool VEC4>
__global__ void __launch_bounds__(256, 2)
residual_kernel(const uint4* __restrict__ A, int64_t lda16, const float* __restrict__ U, int64_t m, const float* __restrict__ V,
                int64_t n, int k, int npanels, int nchunks, double* __restrict__ partial /* [grid * 4] */,
                const int* __restrict__ need /* null: always; else run only if *need != 0 */) {
  if (need && *need == 0) return;
  constexpr int GT = MODE == 1 ? 2 : 1;
  constexpr int BCH = GT * NKB * 3 * 64;                 // uint4 of h/m/l V fragments per chunk
  constexpr int ITEMS = GT * NKB * 64;                   // (gene tile, kb, lane) 8-value loads per chunk
  constexpr int IPT = (ITEMS + 255) / 256;
  __shared__ u32x4 sB[BDB ? 2 : 1][BCH];
  __shared__ u32x4 sA[4][MT * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;

  // identity fragments: element e of lane (c, h) is 1 iff 16 par + 8 h + e == c (c = lane & 31), par = parity of the
  // 16-column slice within its 32-column gene tile
  u32x4 idf[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int es = r - 16 * par - 8 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) idf[par][j] = (es == 2 * j ? 0x3F80u : 0u) | (es == 2 * j + 1 ? 0x3F800000u : 0u);
  }

  unsigned avoff[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + (lane >> 3);
      const int piece = (lane & 7) ^ ((lr >> 1) & 7);               // source-side swizzle (see gemm.hip)
      avoff[mt][i] = (unsigned)(((int64_t)(wave * (32 * MT) + 32 * mt + lr) * lda16 + piece) * 16);
    }
  u32x4* const myA = sA[wave];

  const int64_t T = (int64_t)npanels * nchunks;
  int64_t unit = T * blockIdx.x / gridDim.x;
  const int64_t unit_end = T * (blockIdx.x + 1) / gridDim.x;
  double sumsq = 0.0;

  while (unit < unit_end) {
    const int panel = (int)(unit / nchunks);
    const int c0 = (int)(unit - (int64_t)panel * nchunks);
    const int nch = (int)min((int64_t)(nchunks - c0), unit_end - unit);
    unit += nch;
    const int64_t prow0 = (int64_t)panel * (128 * MT);
    const char* abase = reinterpret_cast<const char*>(A + prow0 * lda16 + (int64_t)c0 * 8);

    // -U fragments of this wave's rows: the three exact pieces, all latent columns
    u32x4 uf[MT][NKB][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        split3_raw(load8<VEC4>(U, m, k, prow0 + wave * (32 * MT) + 32 * mt + r, 16 * kb + 8 * h), uf[mt][kb][0], uf[mt][kb][1], uf[mt][kb][2]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int j = 0; j < 4; ++j) uf[mt][kb][pc][j] ^= 0x80008000u;
      }

    auto load_v = [&](Raw8 (&raw)[IPT], int chunk) {
#pragma unroll
      for (int i = 0; i < IPT; ++i) {
        const int it = tid + 256 * i;
        if (ITEMS % 256 == 0 || it < ITEMS) {
          const int L = it & 63, rest = it >> 6, kb = rest % NKB, gt = rest / NKB;
          raw[i] = load8<VEC4>(V, n, k, ((int64_t)chunk * GT + gt) * 32 + (L & 31), 16 * kb + 8 * (L >> 5));
        }
      }
    };
    auto store_v = [&](u32x4* sb, const Raw8 (&raw)[IPT]) {
#pragma unroll
      for (int i = 0; i < IPT; ++i) {
        const int it = tid + 256 * i;
        if (ITEMS % 256 == 0 || it < ITEMS) {
          u32x4 fh, fm, fl;
          split3_raw(raw[i], fh, fm, fl);
          const int L = it & 63, rest = it >> 6;          // rest = gt * NKB + kb
          sb[(rest * 3 + 0) * 64 + L] = fh;
          sb[(rest * 3 + 1) * 64 + L] = fm;
          sb[(rest * 3 + 2) * 64 + L] = fl;
        }
      }
    };
    auto load_a = [&](u32x4 (&a)[MT][4], int c) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a[mt][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(abase + (int64_t)c * 128 + avoff[mt][i]));
    };
    auto store_a = [&](const u32x4 (&a)[MT][4]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) myA[mt * 256 + i * 64 + lane] = a[mt][i];
    };

    Raw8 raw[IPT];
    u32x4 a[MT][4];
    load_v(raw, c0);
    load_a(a, 0);
    store_v(sB[0], raw);
    store_a(a);
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
      const bool next = c + 1 < nch;
      if (next) {                       // the next chunk's loads are in flight during the MFMA chains
        load_v(raw, c0 + c + 1);
        load_a(a, c + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4* sb = sB[BDB ? (c & 1) : 0];
#pragma unroll
      for (int gt = 0; gt < GT; ++gt) {
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const u32x4 bh = sb[((gt * NKB + kb) * 3 + 0) * 64 + lane];
          const u32x4 bm = sb[((gt * NKB + kb) * 3 + 1) * 64 + lane];
          const u32x4 bl = sb[((gt * NKB + kb) * 3 + 2) * 64 + lane];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            mm(acc[mt], uf[mt][kb][1], bl);      // smallest terms first: m l, l m (2^-24), h l, l h, m m (2^-16), ...
            mm(acc[mt], uf[mt][kb][2], bm);
            mm(acc[mt], uf[mt][kb][0], bl);
            mm(acc[mt], uf[mt][kb][2], bh);
            mm(acc[mt], uf[mt][kb][1], bm);
            mm(acc[mt], uf[mt][kb][0], bm);
            mm(acc[mt], uf[mt][kb][1], bh);
            mm(acc[mt], uf[mt][kb][0], bh);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (MODE == 1) {
#pragma unroll
            for (int par = 0; par < 2; ++par) mm(acc[mt], myA[mt * 256 + r * 8 + ((2 * (2 * gt + par) + h) ^ swz)], idf[par]);
          } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              u32x4 ah, am, al;
              split3(myA[mt * 256 + r * 8 + ((4 * kb + 2 * h) ^ swz)], myA[mt * 256 + r * 8 + ((4 * kb + 2 * h + 1) ^ swz)], ah, am, al);
              mm(acc[mt], al, idf[kb]);
              mm(acc[mt], am, idf[kb]);
              mm(acc[mt], ah, idf[kb]);
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const double d = (double)acc[mt][i];
            sumsq += d * d;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!BDB) __syncthreads();          // everyone has finished reading this chunk's V fragments
      if (next) {
        store_v(sB[BDB ? ((c + 1) & 1) : 0], raw);
        store_a(a);
      }
      __syncthreads();
    }
  }
  // per-wave partial, lanes added in a fixed order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sumsq += __shfl_xor(sumsq, off);
  if (lane == 0) partial[(int64_t)blockIdx.x * 4 + wave] = sumsq;
}

hipError_t launch_residual(int x_dtype, const void* X, int64_t ldx, int64_t m, int64_t n, int k, const float* U, const float* V,
                           int n_cus, double* partial, int* nparts_out, const int* need, hipStream_t s) {
  const int KP = (int)round_up(k, 32), NKB = KP / 16;
  const int MT = (NKB <= 2 || (NKB == 4 && x_dtype != 1)) ? 2 : 1;     // as instantiated below
  const int panel_rows = 128 * MT;
  const int npanels = (int)(round_up(m > 0 ? m : 1, panel_rows) / panel_rows);
  const int nchunks = (int)(round_up(n, K_PAD) / chunk_elems(x_dtype));
  const int64_t T = (int64_t)npanels * nchunks;
  int64_t grid = (int64_t)n_cus * 2;
  if (grid > T) grid = T;
  if (grid > RESIDUAL_MAX_BLOCKS) grid = RESIDUAL_MAX_BLOCKS;
  if (grid < 1) grid = 1;
  *nparts_out = (int)grid * 4;
  const int64_t lda16 = ldx * elem_bytes(x_dtype) / 16;
  const bool vec4 = (k & 3) == 0;
#define PRMF_RES(MODE, MTV, NKBV, BDBV)                                                                                   \
  do {                                                                                                                    \
    if (vec4) residual_kernel<MODE, MTV, NKBV, BDBV, true><<<(unsigned)grid, 256, 0, s>>>((const uint4*)X, lda16, U, m, V, n, k, npanels, nchunks, partial, need); \
    else residual_kernel<MODE, MTV, NKBV, BDBV, false><<<(unsigned)grid, 256, 0, s>>>((const uint4*)X, lda16, U, m, V, n, k, npanels, nchunks, partial, need); \
  } while (0)
  if (x_dtype == 1) {
    switch (NKB) {
      case 2: PRMF_RES(1, 2, 2, true); break;
      case 4: PRMF_RES(1, 1, 4, false); break;      // two row tiles per wave spill 32 registers here
      case 6: PRMF_RES(1, 1, 6, false); break;
      case 8: PRMF_RES(1, 1, 8, false); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (NKB) {
      case 2: PRMF_RES(2, 2, 2, true); break;
      case 4: PRMF_RES(2, 2, 4, false); break;
      case 6: PRMF_RES(2, 1, 6, false); break;
      case 8: PRMF_RES(2, 1, 8, false); break;
      default: return hipErrorInvalidValue;
    }
  }
#undef PRMF_RES
  return hipGetLastError();
}

// The record takes the direct value: recon, obj, recon_sq; flags: direct set, imprecise cleared.  With the tradeoff
// feedback on, gamma = delta = (1 - t) recon / (t manifold) (prmf_runner.py:542-548) is computed from it -- the kernel
// that wrote the identity-based record left the hyper-parameters alone when it raised `need`.
__device__ __forceinline__ void record_take_direct(double rs, double* __restrict__ record, double* __restrict__ hyper, int feedback) {
  const double recon = sqrt(rs);
  record[REC_RECON_SQ_IDENTITY] = record[7];
  record[0] = recon;
  record[6] = recon + record[4] * record[1] + record[5] * record[2] + record[3];
  record[7] = rs;
  record[REC_FLAGS] = (double)(((int)record[REC_FLAGS] | REC_FLAG_DIRECT) & ~REC_FLAG_IMPRECISE);
  if (feedback && hyper[2] >= 0.0) {
    const double t = hyper[2], denom = t * record[1];
    const double g2 = (denom == 0.0) ? 1.0 : ((1.0 - t) * recon) / denom;
    hyper[0] = g2;
    hyper[1] = g2;
  }
}

// sum of the per-wave partials (fixed order) -> out[0] (0 when the residual kernel did not run); single rank: the
// record is rewritten in the same launch
__global__ void __launch_bounds__(256) residual_sum_kernel(const double* __restrict__ partial, int nparts, const int* __restrict__ need,
                                                           double* __restrict__ out, double* __restrict__ record,
                                                           double* __restrict__ hyper, int feedback) {
  __shared__ double red[256];
  if (need && *need == 0) {
    if (threadIdx.x == 0) out[0] = 0.0;
    return;
  }
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    out[0] = s;
    if (record) record_take_direct(s, record, hyper, feedback);
  }
}

hipError_t launch_residual_sum(const double* partial, int nparts, const int* need, double* out, double* record, double* hyper,
                               bool feedback, hipStream_t s) {
  residual_sum_kernel<<<1, 256, 0, s>>>(partial, nparts, need, out, record, hyper, feedback ? 1 : 0);
  return hipGetLastError();
}

__global__ void record_fix_kernel(const int* __restrict__ need, const double* __restrict__ resid_sq, double* __restrict__ record,
                                  double* __restrict__ hyper, int feedback) {
  if (*need == 0) return;
  record_take_direct(resid_sq[0], record, hyper, feedback);
}

hipError_t launch_record_fix(const int* need, const double* resid_sq, double* record, double* hyper, bool feedback, hipStream_t s) {
  record_fix_kernel<<<1, 1, 0, s>>>(need, resid_sq, record, hyper, feedback ? 1 : 0);
  return hipGetLastError();
}

}  // namespace prmf
