// The dominant kernel of the PRMF iteration: the streaming sweep
//     C[M x KP] = A[M x K] * B[K x KP]
// used twice per iteration -- P = X V (A = X, prmf_runner.py:420) and
// Q = X^T U (A = the resident X^T copy, :424).  A is read exactly once from HBM
// (2 B or 4 B per element), B is a small factor re-read from L2 by every
// workgroup, so the kernel is HBM-bound; MFMA does the arithmetic.
//
// Data layout
//   A   row-major, zero padded to [round_up(M, 512)] x [round_up(K, 128)], so the
//       kernel has no bounds checks on loads.  A reduction "chunk" is 128 B of
//       each row (64 bf16 / 32 f32).
//   B   the factor, packed on the device (pack_bf16_kernel, or the epilogue that produced it) into MFMA B-fragment
//       order so that one wave-wide 16 B/lane read is one fragment -- for bf16 X and f32 X alike:
//         block (kb, t, hl) of 1 KiB, kb = k/16, t = 32-column tile, hl = 0 hi / 1 lo;
//         lane (c = lane&31, h = lane>>5), element e:  B[16 kb + 8 h + e][32 t + c],
//         split as hi = bf16(v), lo = bf16(v - hi)  (16 mantissa bits)
//       bf16 X at k > 64 and f32 X at k > 32 ("W16", round 3) multiply with v_mfma_f32_16x16x32_bf16 and pack accordingly:
//         block (g, t16, hl) of 1 KiB, g = k/32, t16 = 16-column tile; lane (c16 = lane&15, kg = lane>>4), element e:
//         B[32 g + 8 kg + e][16 t16 + c16]
//       (same bytes per chunk, so the staging is the same).
//       The packed chunk (all blocks of the chunk's 64 resp. 32 reduction rows) is staged
//       once per workgroup into LDS and shared by its 4 waves.
//   C   per-segment partial slabs in MFMA fragment order, [slab][32-row tile][t][q4][lane][4 floats] (1 KiB per
//       store / load instruction); which slabs hold a segment of a row panel is a per-panel bit mask, and the consumer
//       kernel sums them in slab-index order (deterministic; no atomics).
//
// Wave tiling (wave64): a workgroup is 4 waves x 64 rows, two per CU -- or ONE workgroup of 8 waves x 64 rows per CU
// (512-row panels, factor chunk double-buffered, the next chunk's loads issued between the MFMA steps, no reduction
// slices below 32 MB) when the packed factor is >= 8 MB and k > 32: see plan_gemm.  Each wave owns MT = 2 32-row tiles
// (1 for f32 X at k <= 32 and k > 64) x NT 32-column tiles of accumulators (v_mfma_f32_32x32x16_bf16 for bf16 X at
// k <= 64 and f32 X at k <= 32, the 16x16x32 shape otherwise; f32 X is split exactly into three bf16 pieces in
// registers), so every B fragment read from LDS feeds two MFMAs.
// A is loaded line-complete (8 rows x 128 B per wave instruction, non-temporal), one chunk ahead of the MFMAs, and
// bounced through a wave-private XOR-swizzled LDS tile from which the 16-B A fragments are read (see "A path" below:
// fragment-shaped global loads were 22% slower).  B is staged once per workgroup per chunk through registers into LDS.
#include "kernels.h"
#include <type_traits>
#include "prologue.h"


#ifndef PRMF_SWEEP_PRIO
#define PRMF_SWEEP_PRIO 1   // see the chunk loop
#endif
#ifndef PRMF_PHASE_PROF
#define PRMF_PHASE_PROF 0
#endif
#ifndef PRMF_BF16_ACC_LO
#define PRMF_BF16_ACC_LO 1    // bf16 X, k <= 64: the products with the factor's low piece in accumulators of their own (see sweep_compute)
#endif
#ifndef PRMF_SWEEP_MINW
#define PRMF_SWEEP_MINW 2     // 1 (the compiler then takes 268 registers at k = 128: one workgroup per CU) is 12% slower there
#endif

namespace prmf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native 128-bit value (asm "v" operand)

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
__global__ void __launch_bounds__(256) pack_bf16_kernel(FactorView f, int NT, int64_t nkb, uint4* __restrict__ out, int w16) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int lane = (int)(gid & 63);
  int64_t rest = gid >> 6;
  int t = (int)(rest % NT);
  int64_t kb = rest / NT;
  if (kb >= nkb) return;
  int c = lane & 31, h = lane >> 5;
  int col = 32 * t + c;
  int64_t row0 = kb * 16 + 8 * h;
  if (w16) {        // the same thread count, dealt as (g = kb / 2, t16 = 2 t + (kb & 1), lane (c16, kg))
    const int t16 = 2 * t + (int)(kb & 1);
    col = 16 * t16 + (lane & 15);
    row0 = (kb >> 1) * 32 + 8 * (lane >> 4);
  }
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
  if (w16) blk = ((kb >> 1) * (2 * NT) + 2 * t + (kb & 1)) * 2;
  out[blk * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  out[(blk + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

hipError_t launch_pack_factor(const FactorView& f, int x_dtype, int NT, int64_t Kpad, void* packed, hipStream_t s, bool w16) {
  if (x_dtype != 1) return hipErrorInvalidValue;   // every sweep multiplies with the hi/lo bf16 fragment layout
  int64_t nkb = Kpad / 16;                         // Kpad is a multiple of 128: whole 32-row groups
  int64_t threads = nkb * NT * 64;
  pack_bf16_kernel<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(f, NT, nkb, (uint4*)packed, w16 ? 1 : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------
// MODE 1: A bf16, B hi/lo bf16, 2 MFMA (32x32x16) per fragment pair
// MODE 2: A f32 split exactly into three bf16 pieces in registers, B hi/lo bf16, 5 MFMAs per fragment pair
// (the f32 MFMA form of round 1 -- 1/16 of the bf16 rate -- is gone: at k = 128 it was 40% slower than MODE 2)
//
// A path.  The MFMA A fragment of lane (r = lane&31, h = lane>>5) is 16 B of row
// r, i.e. a wave-wide fragment load touches 32 different 128-B lines with 32 B
// each -- measured 22% slower than line-complete loads (texture-address bound).
// So A is loaded line-complete (one instruction = 8 rows x 128 B, lane l -> row
// l>>3, 16-B piece l&7), PF chunks ahead into registers, then bounced through a
// wave-private 4 KiB LDS tile from which the fragments are read with ds_read_b128.
// The tile is XOR-swizzled on the SOURCE side (LDS image linear per instruction):
//   LDS slot (16-B units) of (row r, piece q) = 8 r + (q ^ ((r >> 1) & 7))
// which makes every 16-lane group of the fragment read hit 16 distinct 16-B slots
// of the 256-B bank row (rows of equal parity in a group differ in (r>>1)&7).
// No barrier is needed for A: a wave only reads what it wrote, and the LDS
// executes one wave's instructions in order.
// f32 -> three bf16 pieces by truncation: x = h + m + l exactly (8 + 8 + 8 mantissa bits).  h, m, l are the upper halves
// of x, r1 = x - h and r2 = r1 - m, so the bf16 pairs are packed straight from those with one v_perm_b32 each; only the
// two subtrahends need a mask.
// 8 floats (two 16-B pieces) -> bf16x8 fragments of the high, middle and low pieces
__device__ __forceinline__ void split_fragment(const u32x4& f0, const u32x4& f1, u32x4& fh, u32x4& fm, u32x4& fl) {
  const unsigned x[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
  constexpr unsigned UPPER_HALVES = 0x07060302u;    // v_perm_b32 (S0 = odd element, S1 = even): [S0.hi16 : S1.hi16]
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

// MODE 1: A bf16, B hi/lo bf16 | MODE 2: A f32 split into three bf16 pieces in registers, B hi/lo bf16 (5 bf16 MFMAs
// per fragment instead of 8 f32 MFMAs at 1/16 of the rate)
// acc_lo (MODE 2 only): the cross terms (<= 2^-8 of the main products) are summed in their own accumulators -- added
// into the main ones they fall below half an ulp of a long running sum and are always rounded DOWN (a -4.8e-6 bias of
// X V at 50000 x 20000: exactly the mean of the lowest piece of X)
// Does the kernel keep a second accumulator set?  f32 X: always (cross terms).  bf16 X at k <= 64 (PRMF_BF16_ACC_LO):
// the products with the factor's low piece (2^-9 of the main ones) are then not absorbed one MFMA at a time into a
// running sum 10^4 times their size.
template <int MODE, int NT>
struct SweepLo { static constexpr bool value = MODE == 2 || (PRMF_BF16_ACC_LO != 0 && NT <= 2); };
// Which sweeps multiply with v_mfma_f32_16x16x32_bf16 and the W16 fragment order: bf16 X at k > 64, f32 X at k > 32 (the
// shapes whose matrix-instruction load makes the chip lower its clock: -5..10% per sweep, nothing at bf16 k <= 64 or f32
// k <= 32 -- profiles/r03_ab_mfma_16x16x32.txt).  capi.hip's `w16` flag and the packing kernels follow the same rule.
template <int MODE, int NT>
struct SweepW16 { static constexpr bool value = (MODE == 1 && NT > 2) || (MODE == 2 && NT >= 2); };

struct NoStepHook { __device__ __forceinline__ void operator()(int) const {} };
// number of hook calls sweep_compute makes per chunk (0: the form has no step structure)
template <int MODE, int NT> struct SweepSteps { static constexpr int value = MODE == 1 ? 4 * NT : ((MODE == 2 && NT >= 2) ? 2 * NT : 0); };

template <int MODE, int MT, int NT, typename Hook = NoStepHook>
__device__ __forceinline__ void sweep_compute(f32x16 (&acc)[MT][NT], f32x16 (&acc_lo)[(SweepLo<MODE, NT>::value ? MT : 1)][(SweepLo<MODE, NT>::value ? NT : 1)],
                                              const u32x4* __restrict__ sa, const u32x4* __restrict__ sb, int lane, const Hook& hook = Hook()) {
  const int r = lane & 31, h = lane >> 5;
  const int swz = (r >> 1) & 7;
  if constexpr (MODE == 2 && SweepW16<MODE, NT>::value) {
    // f32 X, W16: the chunk (32 columns) is ONE 32-element step of the 16x16x32 MFMA.  The three exact pieces of every
    // 16-row group's A fragment (lane (r16, kg): 8 consecutive floats = 16-B pieces 2 kg, 2 kg + 1 of its row) are split
    // once and kept; the factor fragments stream by, one 16-column tile at a time.  Accumulator storage as for bf16 X.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    constexpr int NG = 2 * MT, T16 = 2 * NT;
    const int r16 = lane & 15, kg = lane >> 4;
    u32x4 ah[NG], am[NG], al[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int rr = 16 * (g & 1) + r16, sw = (rr >> 1) & 7;
      split_fragment(sa[(g >> 1) * 256 + rr * 8 + ((2 * kg) ^ sw)], sa[(g >> 1) * 256 + rr * 8 + ((2 * kg + 1) ^ sw)], ah[g], am[g], al[g]);
    }
    auto mm16 = [&](f32x4_& d, const u32x4& a, const u32x4& b) {
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), d, 0, 0, 0);
    };
#pragma unroll
    for (int t16 = 0; t16 < T16; ++t16) {
      const u32x4 bh = sb[(t16 * 2 + 0) * 64 + lane];
      const u32x4 bl = sb[(t16 * 2 + 1) * 64 + lane];
      if constexpr (!std::is_same<Hook, NoStepHook>::value) {
        hook(t16);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int mt = g >> 1, t = t16 >> 1, o = 4 * (2 * (g & 1) + (t16 & 1));
        f32x4_ lo4 = {acc_lo[mt][t][o], acc_lo[mt][t][o + 1], acc_lo[mt][t][o + 2], acc_lo[mt][t][o + 3]};
        f32x4_ hi4 = {acc[mt][t][o], acc[mt][t][o + 1], acc[mt][t][o + 2], acc[mt][t][o + 3]};
        mm16(lo4, al[g], bh);      // smallest terms first
        mm16(lo4, am[g], bl);
        mm16(lo4, ah[g], bl);
        mm16(lo4, am[g], bh);
        mm16(hi4, ah[g], bh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc_lo[mt][t][o + i] = lo4[i];
          acc[mt][t][o + i] = hi4[i];
        }
      }
    }
    return;
  }
  if constexpr (MODE == 2) {
    // 64-row wave tiles (k <= 64): not unrolled over the two k-blocks -- the three-piece split is register-hungry (222
    // registers at k = 64 next to the 2 x 64 accumulator registers; two waves per SIMD allow 256) and unrolled it is no
    // faster at k = 16 or k = 64.  32-row wave tiles (k > 64): unrolled, so that the split of the second k-block runs
    // under the MFMAs of the first (-2.5%, 229 registers at k = 128).
#pragma unroll MT == 1 ? 2 : 1
    for (int kb = 0; kb < 2; ++kb) {
      u32x4 bh[NT], bl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bh[t] = sb[((kb * NT + t) * 2 + 0) * 64 + lane];
        bl[t] = sb[((kb * NT + t) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        u32x4 ah, am, al;
        {
          const u32x4 f0 = sa[mt * 256 + r * 8 + ((4 * kb + 2 * h) ^ swz)];
          const u32x4 f1 = sa[mt * 256 + r * 8 + ((4 * kb + 2 * h + 1) ^ swz)];
          split_fragment(f0, f1, ah, am, al);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          auto mm = [&](f32x16& d, const u32x4& a, const u32x4& b) {
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), d, 0, 0, 0);
          };
          mm(acc_lo[mt][t], al, bh[t]);      // smallest terms first
          mm(acc_lo[mt][t], am, bl[t]);
          mm(acc_lo[mt][t], ah, bl[t]);
          mm(acc_lo[mt][t], am, bh[t]);
          mm(acc[mt][t], ah, bh[t]);
        }
      }
    }
    return;
  }
  static_assert(MODE == 1 || MODE == 2, "MODE 1: bf16 X, MODE 2: f32 X");
  if constexpr (MODE == 1 && NT > 2) {
    // k > 64 ("W16"): v_mfma_f32_16x16x32_bf16 -- under load the chip holds a higher clock with this shape than with
    // 32x32x16 at the same cycles per flop (the guide's DVFS note; same-session ablation with realistic operands:
    // X V of 25000 x 25000, k = 128 0.307 -> 0.290 ms, of 200000 x 25000 2.37 -> 2.23 ms; nothing at k = 64:
    // profiles/r03_ab_mfma_16x16x32.txt).  A wave's 64 rows are four 16-row groups (A operand: lane (r16, kg) holds
    // row r16, 8 consecutive reduction elements 8 kg ..), the chunk is two 32-element steps, a column tile is 16 wide.
    // The accumulators keep their [MT][NT] x 16 storage: 16-row group rg16 = 2 mt + a and 16-column tile t16 = 2 t + b
    // own floats 4 (2 a + b) .. + 4 of acc[mt][t]  (C layout: lane (c16, rg) holds rows 4 rg + i, i = 0..3).
    // The fragment reads of step j + 1 are issued before the MFMAs of step j (scheduling barriers keep them there).
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    static_assert(MT == 2 && SweepW16<MODE, NT>::value, "four 16-row groups per wave");
    const int r16 = lane & 15, kg = lane >> 4;
    u32x4 av[4], bb[2][2];     // (A fragments double-buffered too: 16 more registers, 45 spilled instead of 26 at k = 128)
    auto a_frag = [&](int step, int rg16) -> u32x4 {
      const int rr = 16 * (rg16 & 1) + r16;
      return sa[(rg16 >> 1) * 256 + rr * 8 + ((4 * step + kg) ^ ((rr >> 1) & 7))];
    };
    bb[0][0] = sb[0 * 64 + lane];
    bb[0][1] = sb[1 * 64 + lane];
    constexpr int T16 = 2 * NT;
#pragma unroll
    for (int j = 0; j < 2 * T16; ++j) {
      const int step = j / T16, t16 = j % T16, jn = j + 1;
      if (t16 == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) av[g] = a_frag(step, g);
      }
      if (jn < 2 * T16) {
        bb[jn & 1][0] = sb[(jn * 2 + 0) * 64 + lane];
        bb[jn & 1][1] = sb[(jn * 2 + 1) * 64 + lane];
      }
      hook(j);      // the caller's share of this step (the next chunk's global loads, one or two per step)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mt = g >> 1, t = t16 >> 1, o = 4 * (2 * (g & 1) + (t16 & 1));
        f32x4_ c4 = {acc[mt][t][o], acc[mt][t][o + 1], acc[mt][t][o + 2], acc[mt][t][o + 3]};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[g]), __builtin_bit_cast(bf16x8, bb[j & 1][0]), c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[g]), __builtin_bit_cast(bf16x8, bb[j & 1][1]), c4, 0, 0, 0);
        acc[mt][t][o] = c4[0];
        acc[mt][t][o + 1] = c4[1];
        acc[mt][t][o + 2] = c4[2];
        acc[mt][t][o + 3] = c4[3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    u32x4 av[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) av[mt] = sa[mt * 256 + r * 8 + ((2 * s4 + h) ^ swz)];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4 bh = sb[((s4 * NT + t) * 2 + 0) * 64 + lane];
      const u32x4 bl = sb[((s4 * NT + t) * 2 + 1) * 64 + lane];
      if constexpr (!std::is_same<Hook, NoStepHook>::value) {
        hook(s4 * NT + t);      // the caller's share of this step (the next chunk's global loads), kept in place
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, av[mt]), __builtin_bit_cast(bf16x8, bh), acc[mt][t], 0, 0, 0);
        if constexpr (SweepLo<MODE, NT>::value)
          acc_lo[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, av[mt]), __builtin_bit_cast(bf16x8, bl), acc_lo[mt][t], 0, 0, 0);
        else
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, av[mt]), __builtin_bit_cast(bf16x8, bl), acc[mt][t], 0, 0, 0);
      }
    }
  }
}

// ---- loads ----------------------------------------------------------------------
// The next chunk's A lines and B slice are loaded (plain loads into registers) BEFORE the MFMAs of the current
// chunk, so they are in flight while the wave computes from LDS, and are written to LDS afterwards (one full
// wait per chunk).  With 64-row wave tiles this order is worth +5% at k = 128 (where a chunk is 64 MFMAs per
// wave) and +8% on the sweeps of the f32 k = 16 case; at k = 64 it is neutral (it was 2% slower with the earlier
// 8 x 32-row shape, whose waves had half the MFMA work per chunk to hide the loads behind).
// History (kept because it cost a day): a deeper pipeline with hand-issued asm loads and counted
// `s_waitcnt vmcnt(N)` (newest loads left in flight across iterations) was (a) not faster and (b) UNSAFE: at
// 50000x20000 about 1e-5 of the wave-steps saw an OLDER load still outstanding with only the N youngest
// pending -> 2-4 wrong 32-row tiles per launch, different every run.  The design therefore never carries a
// load across a chunk boundary: every chunk ends with all of its loads landed and stored.
// (Round 3 re-tried two chunks in flight for the 32-row-tile sweeps with plain C++ loads into a second register set --
// compiler-counted waits, correct -- and it is not faster either: 5000 x 10000 f32 k = 16 0.0447 -> 0.0459 ms,
// profiles/r03_ab_two_chunks_in_flight.txt.)

#ifndef PRMF_A_NT
#define PRMF_A_NT 1
#endif
constexpr bool NT_LOADS = PRMF_A_NT != 0;   // A is read exactly once: non-temporal loads keep it from displacing the factor in L2
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef PRMF_LOADS_FIRST
#define PRMF_LOADS_FIRST 1
#endif
constexpr bool LOADS_FIRST = PRMF_LOADS_FIRST != 0;
#ifndef PRMF_SKIP_PAD_TILES
#define PRMF_SKIP_PAD_TILES 1
#endif
#ifndef PRMF_SPRINKLE_LPS
#define PRMF_SPRINKLE_LPS 2     // loads per MFMA step (0: as few as cover the chunk).  2: the 12 loads of a k = 128 chunk leave in the first 6 of
                                // its 16 steps, so the rest of the chunk covers their latency: 1-2 % ahead of 1 and of 3 per step
#endif
#ifndef PRMF_SPRINKLE
#define PRMF_SPRINKLE 8     // workgroup size (in waves) whose k > 64 sweeps issue the next chunk's loads one per MFMA step; 0: none
#endif
// Sprinkled loads (see the chunk loop).  Measured in one session at 200000 x 25000, k = 128 (profiles/r03_ab_sprinkled_loads.txt):
// the 8-wave form (one workgroup per CU: its two waves per SIMD run the same phase at the same time, so nothing else
// overlaps a wave's load burst with MFMAs) 2.214 -> 2.059 ms; the 4-wave form (two workgroups per CU that alternate
// phases by themselves) 2.132 -> 2.319 ms, 25000 x 25000 equal -- hence the 8-wave form only.
template <int NW> struct Sprinkle { static constexpr bool value = PRMF_SPRINKLE != 0 && (PRMF_SPRINKLE == NW || PRMF_SPRINKLE == 1); };
#ifndef PRMF_PP
#define PRMF_PP 1             // 1: the 8-wave bf16 sweeps of PRMF_PP_MIN_NT..PRMF_PP_MAX_NT column tiles run the ping-pong chunk loop (see the
                              // kernel); 2: the same with the loads issued at the end of STAGE (k = 128: spills, 2.4x slower); 0: off.
                              // Same-session A/B (profiles/r04_ab_pingpong_k64.txt): C3 X V -2.5 %, X^T U -5.5 %, iteration -3.3 %; its
                              // two-rank share -5.9 %; k > 64 equal (profiles/r04_ab_k128_pingpong.txt), f32 X +2.8 % -- hence bf16, k = 33..64 only
#endif
#ifndef PRMF_PP_MIN_NT
#define PRMF_PP_MIN_NT 2      // column tiles from / up to which the 8-wave sweeps take the ping-pong loop
#endif
#ifndef PRMF_PP_MAX_NT
#define PRMF_PP_MAX_NT 2
#endif
#ifndef PRMF_PP_F32
#define PRMF_PP_F32 0         // 1: f32 X (three-piece split) too
#endif
#ifndef PRMF_PP_WIDE_MB
#define PRMF_PP_WIDE_MB 4.0   // a ping-pong plan takes the 8-wave form from this factor size (instead of PRMF_WIDE_SMALL_MB) ...
#endif
#ifndef PRMF_PP_WIDE_MIN_ROWS
#define PRMF_PP_WIDE_MIN_ROWS 16384   // ... when the matrix has at least this many rows (enough 512-row panels)
#endif
#ifndef PRMF_TAIL_PROBE
#define PRMF_TAIL_PROBE 0     // 1: the hand-off probe behind the segments loop is compiled in (it costs the kernels 2 registers)
#endif
#ifndef PRMF_ABL_B_HOT
#define PRMF_ABL_B_HOT 0      // ABLATION (wrong arithmetic): every workgroup reads its factor chunks from one window of this many
                              // chunks, which then stays in every L2 -- prices the factor's trips through the fabric (tools/ab_sweep.py)
#endif
#ifndef PRMF_UNIT_CHUNKS
#define PRMF_UNIT_CHUNKS 2
#endif
// chunks per work unit of the balanced decomposition.  2: every segment starts on a 256-B boundary of its rows.  1 (finer
// balance: shares differ by one 2.8 us chunk instead of two) was measured 2% SLOWER at 50000 x 20000 and no faster at
// 5000 x 10000 or 6250 x 20000 (profiles/r02_ab_unit_chunks.txt) -- the imbalance of one unit is not what small sweeps lose.
constexpr int UNIT = PRMF_UNIT_CHUNKS;

template <int MT>
__device__ __forceinline__ void load_a_chunk(u32x4 (&dst)[MT][4], const unsigned (&voff)[MT][4], const uint4* base) {
  const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4* src = reinterpret_cast<const u32x4*>(b + voff[mt][i]);
      dst[mt][i] = NT_LOADS ? __builtin_nontemporal_load(src) : *src;
    }
}

// thread tid takes uint4 #tid of every NTHREADS*16-byte slice of the packed chunk (the last slice may be partial)
template <int NB, int NTHREADS, int BCH>
__device__ __forceinline__ void load_b_chunk(u32x4 (&dst)[NB], const uint4* base, int tid) {
  const u32x4* b = reinterpret_cast<const u32x4*>(base);
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if ((i + 1) * NTHREADS <= BCH || tid + NTHREADS * i < BCH) dst[i] = b[tid + NTHREADS * i];
}

template <int NB, int NTHREADS, int BCH>
__device__ __forceinline__ void store_b_chunk(u32x4* sb, const u32x4 (&src)[NB], int tid) {
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if ((i + 1) * NTHREADS <= BCH || tid + NTHREADS * i < BCH) sb[tid + NTHREADS * i] = src[i];
}

// One workgroup = NW waves x MT 32-row tiles (a panel of 32*MT*NW rows) and one
// contiguous range of reduction chunks (a "split").  Per chunk c:
//     compute chunk c: A fragments from the wave's LDS tile, B from LDS
//     load    B(c+1) -> bs,  A(c+1) -> a      (clamped to the segment's last chunk)
//     write   bs -> LDS (other buffer if BDB, else after a barrier) ; a -> the wave's A tile
//     barrier
// BDB = true : B double-buffered in LDS, one barrier per chunk
// BDB = false: B single-buffered, two barriers per chunk, 16*NT KiB less LDS per
//              workgroup -> more resident workgroups (bytes in flight) per CU
// The chunk loop is unrolled x2 with compile-time LDS buffer offsets; a segment may have any number of chunks.
template <int MODE, int MT, int NT, int NW, bool BDB, bool PP = false>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : PRMF_SWEEP_MINW)   // two workgroups of 4 waves per CU, or one of 8: at most 256 registers (arch + accumulator) per wave
sweep_kernel(const uint4* __restrict__ A, int64_t lda16 /* row stride in 16-B units */, int64_t M_store,
             const uint4* __restrict__ Bp, int npanels, int g_begin, int ngroups /* work units (UNIT chunks each) [g_begin, g_begin+ngroups) of every row */,
             int nx, int pmax, int seg_base, float* __restrict__ slabs, int64_t slab_stride, int slab_nt,
             unsigned long long* __restrict__ stamps /* debug, may be null */, const SweepPrologue pro,
             int* __restrict__ tail /* debug (tools/tail_probe.py), may be null */, int tail_launch) {
  if (stamps && blockIdx.x >= STAMP_MAX_WGS) stamps = nullptr;      // (no launch has that many workgroups: one per resident slot)
  if (stamps && threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = wall_clock64();
    // where this workgroup runs (tools/wg_where.py): HW_ID (wave / SIMD / CU / SH / SE ids) and XCC_ID, for the question
    // what the systematic finish-time spread of equal shares goes with
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    stamps[STAMP_WHERE + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
  }
  constexpr int NTHR = 64 * NW;
  constexpr int BCH = MODE == 1 ? 512 * NT : 256 * NT;   // uint4 per packed chunk (64 bf16 or 32 f32 elements of the reduction)
  constexpr int BPT = (BCH + NTHR - 1) / NTHR;      // uint4 staged per thread per chunk (last one may be partial)
  // LDS: the factor chunk (double-buffered if BDB) and one A tile per wave (MT tiles of 32 rows x 128 B).  The 8-wave shape
  // with a double-buffered k = 128 chunk needs 128 KiB: dynamic shared memory there, static arrays everywhere else.
  constexpr bool DYN_LDS = NW == 8;
  constexpr bool SKIP_PAD_TILES = MODE == 1 && NW == 8 && PRMF_SKIP_PAD_TILES;   // where the registers allow it (232 / 244 at k = 64 / 128)
  constexpr int LDS_B = (BDB ? 2 : 1) * BCH, LDS_A = NW * MT * 256;
  extern __shared__ u32x4 sweep_dyn_lds[];
  __shared__ u32x4 sweep_stat_lds[DYN_LDS ? 1 : LDS_B + LDS_A];
  u32x4* const lds_base = DYN_LDS ? sweep_dyn_lds : sweep_stat_lds;
  u32x4 (*const sB)[BCH] = reinterpret_cast<u32x4 (*)[BCH]>(lds_base);
  u32x4 (*const sA)[MT * 256] = reinterpret_cast<u32x4 (*)[MT * 256]>(lds_base + LDS_B);

  // the small reductions between the epilogue before this sweep and the one after it (prologue.h); the LDS
  // scratch is the A tile area (>= 16 KiB with 4 waves x 64 rows), untouched until the main loop's own prologue
  if constexpr (NW == 4 || NW == 8) {
    if (pro.kind != PROLOGUE_NONE) {
      run_sweep_prologue<(NT > 2) ? 1 : 0>(pro, reinterpret_cast<char*>(&sA[0][0]), LDS_A * 16);
      __syncthreads();
    }
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if PRMF_PHASE_PROF
  unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_sum[6] = {0, 0, 0, 0, 0, 0};
#define PH_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_sum[i] += t_ - ph_t; ph_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PH_MARK(i) do {} while (0)
#endif

  // Balanced, XCD-aware persistent decomposition.  The launch has one workgroup per
  // resident slot.  The reduction range (ngroups units of UNIT chunks) is cut into `nx` slices
  // (nx = 8: one per XCD; workgroups b, b+8, ... share an XCD and its 4 MiB L2, so
  // the nx-th of the packed factor they all stream stays L2-resident instead of
  // being re-fetched through the fabric by every workgroup -- placement is used for
  // speed only, never for correctness).  Within slice x the npanels x L_x units (a
  // unit = UNIT chunks of one row panel) are cut into gridDim.x/nx equal
  // contiguous ranges in panel-major order.  A range that crosses a panel boundary is
  // processed as separate segments; the segment of panel p in slice x that is the
  // `part`-th to touch it goes to slab x*pmax + part, and the consumer sums the slabs
  // named by the panel's bit mask in index order.
  // block b runs on XCD b % 8; slice x = (b % 8) % nx, so XCDs x, x + nx, ... share slice x
  const int xcd = (int)(blockIdx.x & 7), x = xcd % nx;
  const int64_t Gx = gridDim.x / nx;                             // workgroups per slice
  const int wj = (int)(blockIdx.x >> 3) * (8 / nx) + xcd / nx;   // index within the slice
  const int gs0 = g_begin + (int)((int64_t)ngroups * x / nx);
  const int L = g_begin + (int)((int64_t)ngroups * (x + 1) / nx) - gs0;      // units in this slice
  const int64_t T = (int64_t)npanels * L;
  int64_t unit = T * wj / Gx;
  const int64_t unit_end = L > 0 ? T * (wj + 1) / Gx : 0;

  unsigned avoff0[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = 8 * i + (lane >> 3);                           // row within the 32-row tile
      const int piece = (lane & 7) ^ ((lr >> 1) & 7);               // source-side swizzle
      avoff0[mt][i] = (unsigned)(((int64_t)(wave * (32 * MT) + 32 * mt + lr) * lda16 + piece) * 16);
    }
  u32x4* const myA = sA[wave];

  while (unit < unit_end) {
  const int panel = (int)(unit / L);
  const int g0 = (int)(unit - (int64_t)panel * L);
  const int g1 = (int)min((int64_t)L, g0 + (unit_end - unit));
  const int part = (int)(wj - (((int64_t)panel * L + 1) * Gx - 1) / T);   // wj minus the first workgroup of this (panel, slice)
  const int seg = seg_base + x * pmax + part;
  unit += g1 - g0;
  const int c0 = UNIT * (gs0 + g0);
  const int nch = UNIT * (g1 - g0);
  const int64_t prow0 = (int64_t)panel * (32 * MT * NW);            // first row of this panel
  const int64_t row0 = prow0 + wave * (32 * MT);                    // first row of this wave
  // 512-row panels (bf16 X): a 32-row tile that lies entirely in the zero padding behind the matrix (20000 genes are
  // 39 panels + 32 rows: 15 of the last panel's 16 tiles) does not stream its rows from memory -- all of its lanes read
  // the panel's last row instead, which is padding too: 128 B per chunk out of the cache.  Same zeros, same arithmetic.
  unsigned avoff[MT][4];
  {
    const unsigned zoff = (unsigned)(((int64_t)(32 * MT * NW - 1) * lda16 + (lane & 7)) * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool dead = SKIP_PAD_TILES && row0 + 32 * mt >= M_store;
#pragma unroll
      for (int i = 0; i < 4; ++i) avoff[mt][i] = dead ? zoff : avoff0[mt][i];
    }
  }

  // uniform bases (SGPR) + per-lane byte offsets (VGPR, < panel rows * row bytes)
  const uint4* abase = A + prow0 * lda16 + (int64_t)c0 * 8;         // chunk = 8 x 16 B of each row
  const uint4* bbase = Bp + (int64_t)c0 * BCH;
#if PRMF_ABL_B_HOT
#define B_CHUNK_AT(ci) (Bp + (int64_t)((c0 + (ci)) % PRMF_ABL_B_HOT) * BCH)
#else
#define B_CHUNK_AT(ci) (bbase + (int64_t)(ci) * BCH)
#endif

  f32x16 acc[MT][NT];
  constexpr bool HAS_LO = SweepLo<MODE, NT>::value;
  f32x16 acc_lo[HAS_LO ? MT : 1][HAS_LO ? NT : 1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[mt][t][i] = 0.f;
        if constexpr (HAS_LO) acc_lo[mt][t][i] = 0.f;
      }

  u32x4 a[MT][4];
  u32x4 bs[BPT];

  // prologue: B(0), A(0) -> LDS
  load_b_chunk<BPT, NTHR, BCH>(bs, B_CHUNK_AT(0), tid);
  load_a_chunk<MT>(a, avoff, abase);
  store_b_chunk<BPT, NTHR, BCH>(sB[0], bs, tid);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) myA[mt * 256 + i * 64 + lane] = a[mt][i];
  if constexpr (PP) {
    // ---- PING-PONG chunk loop (8 waves, one workgroup per CU; round 4) ------------------------------------------------
    // The two waves of a SIMD belong to the same workgroup and used to run the same phase at the same time: while both
    // waited for their loads, wrote LDS and met at the barrier the matrix pipe idled (k = 128: 65 % MFMA-busy).  Here the
    // workgroup is two halves -- waves 0-3 (H0) and 4-7 (H1), one wave of each per SIMD -- half a chunk apart:
    //     slot 2c    : H0 COMPUTE(c) [its loads for chunk c + 1 go out between the MFMA steps]   | H1 STAGE
    //     slot 2c + 1: H0 STAGE: wait, A(c + 1) -> its tile, its half of B(c + 1) -> buffer (c+1)&1 | H1 COMPUTE(c)
    // so that exactly one wave per SIMD is in its MFMA phase at any time.  Every slot ends at one workgroup barrier
    // (H1 takes one extra barrier in front of its loop, H0 one behind: equal counts).  The factor chunk stays
    // double-buffered: H0 stages its slice of B(c + 1) in slot 2c + 1, H1 its slice of B(c + 2) in slot 2c + 2 (H1 loads
    // the factor two chunks ahead, its slice of B(1) in the segment prologue); buffer (c+1)&1 was last read in slot
    // 2c - 1 (H1's COMPUTE(c - 1)), buffer c&1 in slot 2c + 1 (H1's COMPUTE(c)): nobody overwrites what is being read.
    constexpr int STEPS = SweepSteps<MODE, NT>::value;
    static_assert(NW == 8 && BDB && STEPS > 0 && UNIT == 2, "ping-pong: the 8-wave double-buffered form, an even number of chunks per segment");
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);
    if (half == 1) {      // H1's slice of B(1)
      load_b_chunk<BPT, NTHR, BCH>(bs, B_CHUNK_AT(nch > 1 ? 1 : 0), tid);
      store_b_chunk<BPT, NTHR, BCH>(sB[1], bs, tid);
    }
    // PRMF_PP == 2: the loads do not go out between the MFMA steps of COMPUTE but at the END of STAGE, straight after the
    // registers they land in have been written to LDS -- one chunk further ahead (A(c + 2); the factor slice c + 2 / c + 3),
    // in flight through the barrier, the whole COMPUTE slot and the next STAGE's wait: a wave's 12 KiB are on their way
    // practically all the time instead of from the middle of its COMPUTE to its STAGE, and the wave that queues them is the
    // one that has no MFMAs to issue.
    constexpr bool LOADS_IN_STAGE = PRMF_PP == 2;
    auto issue_loads = [&](int ca, int cb) {
      load_b_chunk<BPT, NTHR, BCH>(bs, B_CHUNK_AT(cb), tid);
      load_a_chunk<MT>(a, avoff, abase + (int64_t)ca * 8);
    };
    if constexpr (LOADS_IN_STAGE) issue_loads(nch > 1 ? 1 : 0, 1 + half < nch ? 1 + half : nch - 1);
    __syncthreads();
    if (half == 1) __syncthreads();      // H1 starts one slot later
    constexpr int NLD = BPT + 4 * MT, LPS = PRMF_SPRINKLE_LPS * STEPS >= NLD ? PRMF_SPRINKLE_LPS : (NLD + STEPS - 1) / STEPS;
    for (int c = 0; c < nch; c += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cc = c + u;                                           // < nch: a segment is whole units of two chunks
        const int ca = cc + 1 < nch ? cc + 1 : cc;                      // the last chunk re-loads itself (no branch around loads)
        const int cb = cc + 1 + half < nch ? cc + 1 + half : nch - 1;
        const u32x4* const bnext = reinterpret_cast<const u32x4*>(B_CHUNK_AT(cb));
        const char* const anext = reinterpret_cast<const char*>(abase + (int64_t)ca * 8);
        auto hook = [&](int j) {
          if constexpr (LOADS_IN_STAGE) return;
#pragma unroll
          for (int q = 0; q < LPS; ++q) {
            const int li = j * LPS + q;
            if (li < BPT) {
              if ((li + 1) * NTHR <= BCH || tid + NTHR * li < BCH) bs[li] = bnext[tid + NTHR * li];
            } else if (li < NLD) {
              const u32x4* src = reinterpret_cast<const u32x4*>(anext + avoff[(li - BPT) / 4][(li - BPT) % 4]);
              a[(li - BPT) / 4][(li - BPT) % 4] = NT_LOADS ? __builtin_nontemporal_load(src) : *src;
            }
          }
        };
#if PRMF_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        sweep_compute<MODE, MT, NT>(acc, acc_lo, myA, sB[u], lane, hook);
#if PRMF_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                // end of this half's COMPUTE slot
        store_b_chunk<BPT, NTHR, BCH>(half == 0 ? sB[u ^ 1] : sB[u], bs, tid);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) myA[mt * 256 + i * 64 + lane] = a[mt][i];
        if constexpr (LOADS_IN_STAGE)       // the next-but-one chunk (clamped to the segment: the last ones re-load the last chunk)
          issue_loads(cc + 2 < nch ? cc + 2 : nch - 1, cc + 2 + half < nch ? cc + 2 + half : nch - 1);
        __syncthreads();                                                // end of this half's STAGE slot
      }
    }
    if (half == 0) __syncthreads();      // H0 waits for H1's last slot: the halves leave the segment together
  } else {
  __syncthreads();

  // The chunk loop is unrolled x2 so that the LDS buffer offsets are compile-time constants; it handles any number of
  // chunks per segment (odd only with UNIT = 1), and the last chunk of a segment loads nothing (it used to re-load
  // itself: one wasted chunk per segment).  `have` / `next` are workgroup-uniform.
  // f32 X on the bf16 MFMA (MODE 2): loads first at k <= 32 (-6% on the 5000 x 10000, k = 16 sweeps) and at k > 64 (32-row
  // wave tiles: -3%); at k = 64 the chunk in flight costs 22 more registers (244, no spills) and the sweeps are 0.5%
  // slower (profiles/r02_ab_f32_loads_first.txt, r02_ab_f32_wide.txt)
  constexpr bool LF_MODE_OK = MODE != 2 || NT == 1 || MT == 1;
  for (int c = 0; c < nch; c += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool have = c + u < nch, next = c + u + 1 < nch;
      if (have) {
        const int cn = c + u + 1;
        PH_MARK(0);
        constexpr int STEPS = Sprinkle<NW>::value ? SweepSteps<MODE, NT>::value : 0;
        if constexpr (LOADS_FIRST && LF_MODE_OK && STEPS == 0) {
          // the next chunk's loads are in flight while this chunk is computed from LDS
          if (next) {
            load_b_chunk<BPT, NTHR, BCH>(bs, B_CHUNK_AT(cn), tid);
            load_a_chunk<MT>(a, avoff, abase + (int64_t)cn * 8);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        PH_MARK(1);
        // The compute segment runs at raised wave priority: the wave that shares this SIMD (it belongs to the CU's other
        // workgroup) is then the one that yields issue slots while it queues loads or LDS writes.  Same-session A/B:
        // 1-2.5% on both sweeps at 50000 x 20000, 4.5% at k = 128, neutral at 5000 x 10000; the inverse (loaders high)
        // and a priority kept through the LDS writes are 1-4% slower (profiles/r02_ab_priority_and_pipelined_reads.txt).
        // The CU's second workgroup (upper half of the grid) runs 1-5% longer than the first -- the arbiter serves the
        // older wave of a SIMD first (tools/wg_spread.py) -- but raising ITS priority on every other chunk does not
        // shorten the launch (k = 128: +29%): profiles/r03_ab_priority_balance.txt.
#if PRMF_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        if constexpr (STEPS > 0) {
          // k > 64: the next chunk's BPT + 4 MT loads go out one (or two) per MFMA step instead of as a burst in front of
          // the chunk -- a wave that sits in a full memory queue issues no MFMAs
          constexpr int NLD = BPT + 4 * MT, LPS = PRMF_SPRINKLE_LPS * STEPS >= NLD ? PRMF_SPRINKLE_LPS : (NLD + STEPS - 1) / STEPS;
          // no branch around the loads (behind one the compiler waits vmcnt(0) before every load): the last chunk of a
          // segment re-loads itself, one wasted chunk per segment
          const int cl = next ? cn : c + u;
          const u32x4* const bnext = reinterpret_cast<const u32x4*>(B_CHUNK_AT(cl));
          const char* const anext = reinterpret_cast<const char*>(abase + (int64_t)cl * 8);
          auto hook = [&](int j) {
#pragma unroll
            for (int q = 0; q < LPS; ++q) {
              const int li = j * LPS + q;
              if (li < BPT) {
                if ((li + 1) * NTHR <= BCH || tid + NTHR * li < BCH) bs[li] = bnext[tid + NTHR * li];
              } else if (li < NLD) {
                const u32x4* src = reinterpret_cast<const u32x4*>(anext + avoff[(li - BPT) / 4][(li - BPT) % 4]);
                a[(li - BPT) / 4][(li - BPT) % 4] = NT_LOADS ? __builtin_nontemporal_load(src) : *src;
              }
            }
          };
          sweep_compute<MODE, MT, NT>(acc, acc_lo, myA, sB[BDB ? u : 0], lane, hook);
        } else {
          sweep_compute<MODE, MT, NT>(acc, acc_lo, myA, sB[BDB ? u : 0], lane);
        }
#if PRMF_SWEEP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        PH_MARK(2);
        if constexpr (!(LOADS_FIRST && LF_MODE_OK) && STEPS == 0) {
          if (next) {
            load_b_chunk<BPT, NTHR, BCH>(bs, B_CHUNK_AT(cn), tid);
            load_a_chunk<MT>(a, avoff, abase + (int64_t)cn * 8);
          }
        } else {
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!BDB) __syncthreads();          // everyone has finished reading B(c)
        PH_MARK(3);
        if (next || STEPS > 0) {      // STEPS > 0: unconditional like the loads (or the compiler sinks the loads into this block)
          store_b_chunk<BPT, NTHR, BCH>(sB[BDB ? (u ^ 1) : 0], bs, tid);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) myA[mt * 256 + i * 64 + lane] = a[mt][i];
        }
        PH_MARK(4);
        __syncthreads();      // also orders this chunk's LDS reads before the next segment's prologue writes
        PH_MARK(5);
      }
    }
  }
  }   // !PP

  // epilogue: the accumulators go out in FRAGMENT order -- slab[seg][tile32][t][q4][lane][4 floats]: every
  // store instruction writes 1 KiB contiguous (whole 128-B lines), non-temporal (the slab is read next by
  // another kernel on other XCDs: it has to reach memory anyway, and as full-line streaming writes it does
  // so during the sweep instead of as an L2 write-back at the end: -16 us per launch at 50000x20000,
  // tools/stream_ladder2.hip).  Element (row, col) of a slab:
  //   tile = row/32, rr = row%32, lane = col%32 + 32*((rr>>2)&1), q4 = rr>>3, j = rr&3, t = col/32
  f32x4* out = reinterpret_cast<f32x4*>(slabs + (int64_t)seg * slab_stride);
  if constexpr (SweepW16<MODE, NT>::value) {
    // W16 accumulators (see sweep_compute) into the SAME slab order: floats 4 (2 a + b) .. + 4 of acc[mt][t] are rows
    // 16 a + 4 rg + i, column 16 b + c16 of the 32 x 32 tile (mt, t) -> q4 = 2 a + (rg >> 1), lane' = 16 b + c16 + 32 (rg & 1):
    // every store instruction writes four 256-B runs
    const int c16 = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t tile = row0 / 32 + mt;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) {
            const int o = 4 * (2 * a2 + b2);
            f32x4 v = {acc[mt][t][o], acc[mt][t][o + 1], acc[mt][t][o + 2], acc[mt][t][o + 3]};
            if constexpr (HAS_LO) {
              const f32x4 w = {acc_lo[mt][t][o], acc_lo[mt][t][o + 1], acc_lo[mt][t][o + 2], acc_lo[mt][t][o + 3]};
              v += w;
            }
            f32x4* dst = out + ((tile * NT + t) * 4 + (2 * a2 + (rg >> 1))) * 64 + (16 * b2 + c16 + 32 * (rg & 1));
            if (slab_nt) __builtin_nontemporal_store(v, dst);     // (uniform)
            else *dst = v;
          }
    }
  } else {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t tile = row0 / 32 + mt;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 v = {acc[mt][t][4 * q4], acc[mt][t][4 * q4 + 1], acc[mt][t][4 * q4 + 2], acc[mt][t][4 * q4 + 3]};
        if constexpr (HAS_LO) {
          const f32x4 w = {acc_lo[mt][t][4 * q4], acc_lo[mt][t][4 * q4 + 1], acc_lo[mt][t][4 * q4 + 2], acc_lo[mt][t][4 * q4 + 3]};
          v += w;
        }
        f32x4* dst = out + ((tile * NT + t) * 4 + q4) * 64 + lane;
        if (slab_nt) __builtin_nontemporal_store(v, dst);     // (uniform)
        else *dst = v;
      }
  }
  }
#undef B_CHUNK_AT
  }   // segments
#if PRMF_TAIL_PROBE
  // ---- PROBE (a -DPRMF_TAIL_PROBE=1 variant build; tools/tail_probe.py): what would a hand-off to an epilogue in the sweep's tail cost?  Every
  // workgroup publishes its slabs (release fence + one agent-scope atomic per panel it touched), then claims groups of
  // output tiles from a ticket counter, waits until the group's panel has all its segments (acquire) and reads the
  // group's tiles of slab 0.  No arithmetic depends on it.  Counters only grow: launch L works with tickets
  // [L (ngroups + grid), ..) and panel targets (L + 1) expected[p].  Layout: tail[0] ticket counter, [1] ngroups,
  // [2] tiles per group, [3] error flag, [4] checksum sink, [64 + p] arrivals of panel p, [64 + 4096 + p] expected[p].
  // PRMF_TAIL_PROBE = 2: the same without cache-wide operations -- no release fence (the wave only waits for its own
  // stores, s_waitcnt vmcnt(0)) and relaxed agent-scope atomics instead of acquire / release: what the hand-off costs if
  // the slabs themselves travel with system-scope (sc0 sc1) stores and loads, which need no L2 write-back / invalidate
  constexpr int TAIL_REL = PRMF_TAIL_PROBE == 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE;
  constexpr int TAIL_ACQ = PRMF_TAIL_PROBE == 2 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE;
  if (tail) {
    if constexpr (PRMF_TAIL_PROBE == 2) __builtin_amdgcn_s_waitcnt(0);
    else __threadfence();
    __syncthreads();
    {
      const int64_t u0 = L > 0 ? T * wj / Gx : 0;
      if (threadIdx.x == 0 && unit_end > u0)
        for (int pp = (int)(u0 / L); pp <= (int)((unit_end - 1) / L); ++pp)
          __hip_atomic_fetch_add(&tail[64 + pp], 1, TAIL_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int ngroups = tail[1], gtiles = tail[2], per_launch = ngroups + (int)gridDim.x;
    __shared__ int s_ticket;
    float sink = 0.f;
    for (;;) {
      if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&tail[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tail_launch * per_launch;
      __syncthreads();
      const int g = s_ticket;
      __syncthreads();
      if (g >= ngroups) break;
      const int pp = (int)(((int64_t)g * gtiles * 32) / (32 * MT * NW));
      const int target = (tail_launch + 1) * tail[64 + 4096 + pp];
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(&tail[64 + pp], TAIL_ACQ, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 20) || __hip_atomic_load(&tail[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // never hang: give up, flagged,
            __hip_atomic_store(&tail[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                 // and nobody waits after that
            break;
          }
        }
      }
      __syncthreads();
      if (lane == 0) (void)__hip_atomic_load(&tail[64 + pp], TAIL_ACQ, __HIP_MEMORY_SCOPE_AGENT);      // every wave acquires
      const f32x4* sl = reinterpret_cast<const f32x4*>(slabs + (int64_t)seg_base * slab_stride);
      for (int tt = wave; tt < gtiles; tt += NW) {
        const int64_t tile = (int64_t)g * gtiles + tt;
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) {
          const f32x4 v = sl[(tile * NT * 4 + q) * 64 + lane];
          sink += v[0] + v[3];
        }
      }
    }
    if (sink == 1.2345e-33f) tail[4] = 1;
  }
#endif
#if PRMF_PHASE_PROF
  if (stamps && lane == 0)
    for (int i = 0; i < 6; ++i) stamps[STAMP_PHASE + (blockIdx.x * 8 + wave) * 8 + i] = ph_sum[i];
#endif
  if (stamps && threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}

// The one workgroup shape (alternatives measured and retired: 8 waves x 32 rows -- twice the LDS fragment traffic per
// byte of X, 3% slower at k = 64 and 5% at k = 128; 32-row wave tiles with 3 workgroups per CU -- 11% slower at k = 128;
// 4 waves x 128 rows, one workgroup per CU -- 10% slower at k = 64; the factor chunk staged by LDS-DMA
// (global_load_lds_dwordx4) instead of through registers -- 8% slower at k = 64, 14% at k = 16:
// profiles/r02_ab_factor_lds_dma.txt; one or three workgroups per CU -- 35% / 12% slower at 5000 x 10000; two 4-wave
// K-groups per workgroup that add their accumulators through LDS (half the slabs) -- 5% / 15% slower on the two C3
// sweeps; an 8-wave workgroup on a 512-row panel -- equal at C3, 14% slower at C2: profiles/r02_ab_kgroups_and_8wave.txt):
//   4 waves x 64 rows (MT = 2), two workgroups per CU;
//   k <= 64: factor chunk double-buffered in LDS (32 KiB of A tiles + 2 x 8 NT KiB), one barrier per chunk;
//   k  > 64: factor chunk single-buffered (k = 128: 32 + 32 KiB, 116 VGPRs + 128 accumulator registers), two barriers.
//   bf16 X, k = 33..64, packed factor >= 4 MB and >= 16384 rows (round 4): the 8-wave form with the PING-PONG chunk loop (its two
//   waves per SIMD half a chunk apart; see the kernel) -- both sweeps of 50000 x 20000 run it.
//   bf16 X, k > 64, packed factor >= 32 MB (round 3): ONE workgroup of 8 waves x 64 rows per CU with the factor chunk
//   double-buffered (64 KiB of A tiles + 2 x 32 KiB at k = 128, dynamic LDS), one barrier per chunk.  Same-session A/B
//   (profiles/r03_ab_k128_8wave_double_buffer.txt): X^T U of 200000 x 25000 (102 MB factor, re-read from the Infinity
//   Cache by every segment) 2.62 -> 2.21 ms; with a 12.8 MB factor -2% at 25000 rows and +7% at 200000 rows -- hence the
//   threshold.
constexpr int SWEEP_NW = 4, SWEEP_MT = 2, SWEEP_WGS_PER_CU = 2, SWEEP_NW_WIDE = 8;
constexpr int EXACT_MAX_K = 512, EXACT_PANEL_ROWS = 256, EXACT_WIDE_ROWS = 4096;      // the exact short-reduction product (exact_product_kernel below): padded depth x column tiles <= 512; depth <= 1024 at one column tile
#ifndef PRMF_WIDE_MB
#define PRMF_WIDE_MB 32.0
#endif
constexpr double SWEEP_WIDE_FACTOR_BYTES = PRMF_WIDE_MB * 1024 * 1024;
#ifndef PRMF_WIDE_SMALL_MB
#define PRMF_WIDE_SMALL_MB 8.0
#endif
constexpr double SWEEP_WIDE_SMALL_FACTOR_BYTES = PRMF_WIDE_SMALL_MB * 1024 * 1024;      // the 8-wave form from this factor size up (unsliced below the 32 MB)

// 32-row tiles per wave.  bf16 X: 2 at every k.  f32 X (three-piece split, two accumulator sets): 2 at k = 33..64; 1 at
// k > 64, where two do not fit 256 registers, and at k <= 32, where the 128-row panels measure 4% faster on the
// 5000 x 10000, k = 16 sweeps (half as many segments per panel); at k = 64 one tile per wave is 1-2% slower
// (profiles/r02_ab_f32_wide.txt).
static int f32_wave_tiles(int x_dtype, int NT) { return (x_dtype == 0 && NT != 2) ? 1 : SWEEP_MT; }

GemmPlan plan_gemm(int x_dtype, int64_t M, int64_t Kpad, int NT, int n_cus, const SweepTuning& tune) {
  GemmPlan p;
  if ((Kpad * NT <= EXACT_MAX_K || (NT == 1 && Kpad <= 2 * EXACT_MAX_K && M <= EXACT_WIDE_ROWS)) && !tune.no_exact) {
    // a short reduction: the exact float32 product, one slab per panel (exact_product_kernel)
    p.exact = true;
    p.nw = 4;
    p.mt = 1;
    p.bdb = false;
    p.panel_rows = EXACT_PANEL_ROWS;
    p.panels = (int)(round_up(M, p.panel_rows) / p.panel_rows);
    p.nchunks = (int)(Kpad / 32);
    p.ngroups = p.nchunks;
    // reduction slices (= slabs per panel): enough waves for eight per CU, at least two chunks per slice, at most 8
    const int64_t pairs = (int64_t)p.panels * p.panel_rows / 32 * NT;
    int64_t sl = (8 * (int64_t)n_cus + pairs - 1) / pairs;
    if (sl > p.nchunks / 2) sl = p.nchunks / 2;
    if (sl > 8) sl = 8;
    if (sl < 1) sl = 1;
    p.nx = 1;
    p.pmax = (int)sl;
    p.max_seg = (int)sl;
    p.grid = (int)((pairs * sl + 3) / 4);
    p.slab_nt = false;
    p.segmask.assign(p.panels, (segmask_t(1) << sl) - 1);
    return p;
  }
  p.nw = SWEEP_NW;
  p.bdb = NT <= 2;
  p.mt = f32_wave_tiles(x_dtype, NT);
  // bf16 X, k > 64, a factor far beyond the caches (the X^T U sweep of 200000 samples: 102 MB): one 8-wave workgroup per
  // CU on 512-row panels -- every 32 KiB factor chunk then serves twice the rows -- with the chunk double-buffered
  // Since its loads are sprinkled through the MFMA steps (chunk loop) the shape also wins at smaller factors -- unsliced
  // (see below): from 8 MB up at every row count (profiles/r03_ab_sprinkled_loads.txt).
  const double factor_bytes = (double)Kpad * 32.0 * NT * 4.0;
  // k = 33..64 as well (8 steps of 4 MFMAs per chunk, low-piece accumulators, 32x32x16): X^T U of 50000 samples
  // (12.8 MB factor, 4 slices in the 4-wave form) 0.3845 -> 0.3795 ms unsliced in the 8-wave form, iteration 0.8420 ->
  // 0.8325 ms, 7 of 7 interleaved rounds; with 2 slices it is slower than the 4-wave form.
  // f32 X too (three-piece split; hook per 16-column tile): X^T U at the C3 shape 0.777 -> 0.745 ms, the N = 8 share of C5
  // 0.725 / 0.715 -> 0.711 / 0.705 ms -- unsliced; with the 4 slices of the 4-wave form it had measured equal.
  const bool pp_ok = PRMF_PP != 0 && (x_dtype == 1 || (PRMF_PP_F32 != 0 && NT == 2)) && NT >= PRMF_PP_MIN_NT && NT <= PRMF_PP_MAX_NT;
#ifndef PRMF_WIDE_F32_MB
#define PRMF_WIDE_F32_MB 4.0      // f32 X, k = 33..64, >= PRMF_PP_WIDE_MIN_ROWS rows: the 8-wave form from this factor size (8 = the general rule)
#endif
  double wide_from = (pp_ok && M >= PRMF_PP_WIDE_MIN_ROWS) ? PRMF_PP_WIDE_MB * 1024 * 1024 : SWEEP_WIDE_SMALL_FACTOR_BYTES;
  // f32 X at k = 33..64 too (round 5): X V of the 50000 x 20000 shape with the CLI's default storage (5.1 MB factor) 0.786 -> 0.756 ms in the
  // 8-wave unsliced form, iteration -1.7 %, 5 of 5 interleaved rounds (profiles/r05_ab_f32_xv_8wave.txt); k > 64 not measured, not changed
  if (x_dtype == 0 && NT == 2 && M >= PRMF_PP_WIDE_MIN_ROWS && PRMF_WIDE_F32_MB * 1024 * 1024 < wide_from) wide_from = PRMF_WIDE_F32_MB * 1024 * 1024;
  if (NT >= 2 && factor_bytes >= wide_from) {
    p.nw = SWEEP_NW_WIDE;
    p.bdb = true;
    p.pp = pp_ok;
  }
  p.panel_rows = p.nw * 32 * p.mt;
  p.panels = (int)(round_up(M, p.panel_rows) / p.panel_rows);
  p.nchunks = (int)(Kpad / chunk_elems(x_dtype));          // even (K_PAD)
  p.ngroups = p.nchunks / UNIT;                            // work units per row panel
  // one workgroup per resident slot (a multiple of 8: one share per XCD); fewer for a matrix too small to give every slot work:
  int64_t G = (int64_t)n_cus * (p.nw == SWEEP_NW_WIDE ? 1 : SWEEP_WGS_PER_CU);
  const int64_t total_units = (int64_t)p.panels * p.ngroups;
  // A matrix too small to give every slot work: down to 4 chunks per workgroup while that still fills one workgroup per CU,
  // down to 2 below that (but then at most one workgroup per CU).  Round 5 (profiles/r05_small_matrix_grid.txt; the rule was
  // "at least 8 chunks each"): 500 x 20000 f32 k = 6 -- the reference's typical size -- iteration 78.6 -> 73.7 us (X V 23.3 ->
  // 18.9 us on 256 instead of 156 workgroups); 100 x 2000 k = 30 / 128: 92 -> 53 / 197 -> 91 us; no shape of the scan slower.
  // A short sweep is bounded by loads in flight (workgroups x 32 KiB), not by bytes.
  {
    const int64_t chunks = total_units * UNIT;
    const int64_t cap = chunks / 4 >= n_cus ? chunks / 4 : (chunks / 2 < n_cus ? chunks / 2 : (int64_t)n_cus);
    if (G > cap) G = cap;
  }
  G = G / 8 * 8;
  p.nx = 1;
  if (G >= 8 && p.ngroups * UNIT >= 16) {
    // as many slices as keep >= ~1.5 (panel, slice) pieces per workgroup: shorter segments cost
    // more in prologues and slab traffic than the L2 residency of the factor slice wins back
    const double units_per_wg = (double)total_units / (double)G;
    // Slice only when the packed factor cannot stay in the 4 MiB XCD L2 (with the A stream loaded
    // non-temporally a slice that fits is served from L2 instead of the fabric).  Measured at
    // 50000x20000, k=64 (interleaved runs): X^T*U (12.8 MB factor) 0.425 -> 0.36-0.39 ms at 4 slices;
    // X*V (5.1 MB factor) is fastest unsliced (0.35-0.375 vs 0.40-0.42: the extra segments cost more).
    const double factor_mb = (double)Kpad * 32.0 * NT * 4.0 / (1024.0 * 1024.0);
    if (factor_mb > 7.0) {
      const double slice_mb = 3.5;      // 8 slices (2 MB each) were bimodal: 0.355 / 0.413 ms
      p.nx = 2;
      while (p.nx < 8 && factor_mb / p.nx > slice_mb) p.nx *= 2;
      while (p.nx > 1 && units_per_wg < 0.5 * ((double)p.ngroups / p.nx)) p.nx /= 2;
    } else if (factor_mb > 4.0 && units_per_wg <= 0.5 * ((double)p.ngroups / 2.0)) {
      // A factor that barely exceeds the L2 (5.1 MB: V at 20000 genes, k = 64; 4.3 / 6.4 MB: U of a 16667 / 25000-row
      // share) is worth two slices when a workgroup's share is at most half a slice row, i.e. for the row shares of a
      // multi-GPU run: X V 69.7 -> 61.8 us at 6250 rows, 112 -> 103 at 12500, 136 -> 127 at 16667, 192 -> 188 at 25000;
      // X^T U 207 -> 181 us at 25000 samples.  With longer shares (50000 rows: 0.77 of a slice row) the shares
      // straddle slice boundaries and two slices are 0-6% slower (profiles/r02_slices_small_shares.txt).
      p.nx = 2;
    }
    // The 8-wave form reads half the factor bytes per byte of X, and unsliced it writes a third of the slab segments:
    // below the 32 MB that its big-factor rule starts at it is fastest with no slices at all -- X V of 25000 / 50000 /
    // 100000 / 200000 rows x 25000 genes, k = 128 (12.8 MB factor): 0.3215 -> 0.3162, 0.5798 -> 0.5701, 1.127 -> 1.050,
    // 2.166 -> 2.083 ms (and 2.108 in the 4-wave form with 4 slices); from 51 MB up the slices stay (X^T U of 100000
    // samples 1.066 against 1.085 ms unsliced; 200000 samples equal).  profiles/r03_ab_sprinkled_loads.txt.
#ifndef PRMF_WIDE_KEEP_SLICES
#define PRMF_WIDE_KEEP_SLICES 0     // experiment: 1 = an 8-wave plan keeps the two slices a 4-7 MB factor gets on short shares
#endif
    if (p.nw == SWEEP_NW_WIDE && factor_bytes < SWEEP_WIDE_FACTOR_BYTES && !(PRMF_WIDE_KEEP_SLICES && p.nx == 2 && factor_bytes < 7.0 * 1048576)) p.nx = 1;
    if (tune.force_slices == 1 || tune.force_slices == 2 || tune.force_slices == 4 || tune.force_slices == 8) p.nx = tune.force_slices;
  }
  if (G < 1) G = 1;
  // which slabs does every panel receive?  simulate the kernel's decomposition for one launch
  auto simulate = [&](int ngroups_sub, int nx, int64_t Gl, int seg_base, int* pmax_out, std::vector<segmask_t>* mask) -> bool {
    const int64_t Gx = Gl / nx;
    std::vector<int> parts((size_t)p.panels * nx, 0);
    int pmax = 1;
    for (int x = 0; x < nx; ++x) {
      const int gs0 = (int)((int64_t)ngroups_sub * x / nx);
      const int L = (int)((int64_t)ngroups_sub * (x + 1) / nx) - gs0;
      if (L <= 0) continue;
      const int64_t T = (int64_t)p.panels * L;
      for (int q = 0; q < p.panels; ++q) {
        const int64_t w0 = (((int64_t)q * L + 1) * Gx - 1) / T;
        const int64_t w1 = (((int64_t)(q + 1) * L) * Gx - 1) / T;
        const int n = (int)(w1 - w0 + 1);
        parts[(size_t)q * nx + x] = n;
        if (n > pmax) pmax = n;
      }
    }
    *pmax_out = pmax;
    if (seg_base + nx * pmax > MAX_SLABS) return false;
    if (mask)
      for (int q = 0; q < p.panels; ++q)
        for (int x = 0; x < nx; ++x)
          for (int j = 0; j < parts[(size_t)q * nx + x]; ++j) (*mask)[q] |= segmask_t(1) << (seg_base + x * pmax + j);
    return true;
  };
  for (;;) {
    p.grid = (int)G;
    p.segmask.assign(p.panels, segmask_t(0));
    if (simulate(p.ngroups, p.nx, G, 0, &p.pmax, &p.segmask)) {
      p.max_seg = p.nx * p.pmax;
      break;
    }
    // too many segments per panel for the 64-bit mask: drop the slices, then workgroups -- eight at a time (round 5; it was
    // "halve": a two-panel X V got 64 workgroups where 120 fit, profiles/r05_small_matrix_grid.txt)
    if (p.nx > 1) p.nx = 1;
    else G = G > 16 ? G - 8 : 1;
  }
  // Slabs leave as non-temporal full-line stores (see the sweep's epilogue) -- except when a launch writes only a few
  // MB of them (5000 x 10000, k = 16: 9 MB): plain stores are then 2% faster on the iteration; from 34 MB up (k = 64 at
  // any row count: (workgroups + panels) x 64 KiB) non-temporal is equal or up to 2% faster
  // (profiles/r02_ab_slab_stores.txt).
  p.slab_nt = (int64_t)(p.grid + p.panels) * p.panel_rows * 32 * NT * 4 > (int64_t)16 << 20;
  return p;
}

unsigned long long* g_sweep_stamps = nullptr;   // debug: per-workgroup start/end stamps (tools/wg_spread.py)
int* g_tail_probe = nullptr;                    // debug: hand-off probe in the sweep's tail (tools/tail_probe.py)
int g_tail_launch = 0;
const GemmPlan* g_tail_plan = nullptr;          // the one plan whose launches carry the probe

template <int MODE, int NW, bool BDB, int MTV>
static hipError_t launch_sweep_nt(const GemmPlan& p, const void* A, int64_t lda16, int64_t M_store, const void* Bp,
                                  int NT, float* slabs, int64_t slab_stride, hipStream_t s, const SweepPrologue& pro, bool attr_only) {
  // dynamic LDS only for the 8-wave shape (see the kernel).  More than 64 KiB of it needs the function attribute raised
  // on the device once: prmf_create does that for the context's two plans (attr_only) -- no state is kept here.
  const size_t dyn = NW == 8 ? (size_t)((BDB ? 2 : 1) * (MODE == 1 ? 512 : 256) * NT + NW * MTV * 256) * 16 : 0;
#define PRMF_LAUNCH_PP(NTV, PPV)                                                                               \
  if (attr_only) {                                                                                             \
    if (dyn <= 64 * 1024) return hipSuccess;                                                                   \
    return hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<MODE, MTV, NTV, NW, BDB, PPV>),      \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                          \
  }                                                                                                            \
  sweep_kernel<MODE, MTV, NTV, NW, BDB, PPV><<<dim3(p.grid), dim3(64 * NW), dyn, s>>>(                           \
      (const uint4*)A, lda16, M_store, (const uint4*)Bp, p.panels, 0, p.ngroups, p.nx, p.pmax, 0, slabs, slab_stride, \
      p.slab_nt ? 1 : 0, g_sweep_stamps, pro, (&p == g_tail_plan) ? g_tail_probe : nullptr, (&p == g_tail_plan) ? g_tail_launch++ : 0)
#define PRMF_LAUNCH(NTV) PRMF_LAUNCH_PP(NTV, false)
  // the ping-pong chunk loop exists for bf16 X in the 8-wave form (plan_gemm decides: p.pp)
#define PRMF_LAUNCH_8W(NTV)                                                    \
  do {                                                                         \
    if constexpr (PRMF_PP != 0 && (MODE == 1 || (PRMF_PP_F32 != 0 && NTV == 2)) && NTV >= PRMF_PP_MIN_NT && NTV <= PRMF_PP_MAX_NT) { \
      if (p.pp) { PRMF_LAUNCH_PP(NTV, true); } else { PRMF_LAUNCH_PP(NTV, false); } \
    } else {                                                                   \
      if (p.pp) return hipErrorInvalidValue;                                   \
      PRMF_LAUNCH_PP(NTV, false);                                              \
    }                                                                          \
  } while (0)
  if constexpr (NW == 8) {
    // bf16 X: k > 32; f32 X: k = 33..64 with 64-row wave tiles, k > 64 with 32-row ones (f32_wave_tiles)
    constexpr bool K64 = MODE == 1 || MTV == SWEEP_MT, KWIDE = MODE == 1 || MTV == 1;
    if (NT == 2) {
      if constexpr (K64) { PRMF_LAUNCH_8W(2); } else return hipErrorInvalidValue;
    } else if (NT == 3) {
      if constexpr (KWIDE) { PRMF_LAUNCH_8W(3); } else return hipErrorInvalidValue;
    } else if (NT == 4) {
      if constexpr (KWIDE) { PRMF_LAUNCH_8W(4); } else return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
  } else if constexpr (BDB) {
    switch (NT) {
      case 1: PRMF_LAUNCH(1); break;
      case 2: PRMF_LAUNCH(2); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (NT) {
      case 3: PRMF_LAUNCH(3); break;
      case 4: PRMF_LAUNCH(4); break;
      default: return hipErrorInvalidValue;
    }
  }
#undef PRMF_LAUNCH
#undef PRMF_LAUNCH_8W
#undef PRMF_LAUNCH_PP
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Short reductions (round 6): the exact product
// ---------------------------------------------------------------------------
// The streaming sweep multiplies with the factor packed to hi + lo bf16: 16-17 mantissa bits.  Over a long reduction that
// rounding averages out (1.1e-6 / sqrt(K) per latent column: 5e-9 at 50000 samples); over a SHORT one it does not -- every
// output row of a column gets the same relative shift, because X is positive and every row sums the same K rounded factor
// entries: 1.5e-7 at 301 samples, 3e-7 at 100 (the reference's own test problems), which the normalised objective terms and
// the tradeoff feedback amplify (profiles/r06_first_step_errors.txt: on the reference's 100 x 1000 case the packing IS the
// library's error -- trace 1.7e-6 with it, 1.2e-8 without, tools/packing_model_small.py).  A reduction of at most
// EXACT_MAX_K elements (after padding) therefore runs on v_mfma_f32_32x32x2_f32 -- exact float32 products, a fused
// multiply-add chain bit for bit (the guide) -- against the float32 factor itself, unpacked.  1/16 of the bf16 rate on a
// product that is microseconds long (tools/ab_exact_product.py).  Longer reductions gain less (the shift falls like
// 1 / sqrt(K)) and pay more (the float32 pipe, in proportion to depth x column tiles): the limit is
// (padded depth) x (32-column tiles) <= 512 -- 512 samples at k <= 32, 256 at k <= 64, 128 at k <= 128 -- and a depth of up
// to 1024 at k <= 32 when the product has at most 4096 rows (X V of the reference's 100-sample, 1000-gene test problems: few
// tiles, so the float32 pipe costs nothing there; with 20000 rows the same depth measured +13 % on the iteration).  Measured against the
// streaming form in one session: 500 x 20000, k = 6 (the reference's typical use) 57.8 -> 60.2 us per iteration, 100 x 1000
// 34.4 -> 35.2 us; at k = 64 / 128 with 500 samples it would have been +12 % (hence the product rule).
//
// One wave per (32-row tile, 32-column tile, reduction slice); the A tile of a 32-element chunk travels line-complete into a wave-private
// LDS tile (row stride 36 floats), the 16 B-fragment values of the chunk come straight from the factor (L2), both one
// chunk ahead of the 16 MFMAs that consume them.  A slice's result leaves as slab `slice` in the sweep's fragment order:
// every consumer (fused epilogues, dense-Q sum, prmf_debug_slab_sum) reads a panel with that many segments.

template <bool A_BF16>
__global__ void __launch_bounds__(256) exact_product_kernel(const void* __restrict__ Av, int64_t lda /* elements */, int nchunks,
                                                            const float* __restrict__ B, int64_t k_valid, int k, int NT,
                                                            int nslices, int64_t npairs, float* __restrict__ slabs, int64_t slab_stride,
                                                            const SweepPrologue pro) {
  __shared__ __attribute__((aligned(16))) float tileA[4][2][32 * 36];
  if (pro.kind != PROLOGUE_NONE) {                 // the reductions that ride in a sweep ride here too (uniform branch)
    run_sweep_prologue(pro, reinterpret_cast<char*>(&tileA[0][0][0]), (int)sizeof(tileA));
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= npairs * nslices) return;            // (no workgroup barrier below: the tile is wave-private)
  // a (tile, column tile) pair is shared by `nslices` waves, each a contiguous range of the reduction chunks and a slab of
  // its own (summed in slab order by the consumer): a short product on few rows would otherwise be a handful of waves
  // walking their chunks one memory round trip at a time
  const int slice = (int)(item / npairs);
  const int64_t pair = item % npairs;
  const int c_lo = (int)((int64_t)nchunks * slice / nslices), c_hi = (int)((int64_t)nchunks * (slice + 1) / nslices);
  float* const slab = slabs + (int64_t)slice * slab_stride;
  const int64_t tile = pair / NT;
  const int t = (int)(pair % NT);
  const int r = lane & 31, h = lane >> 5;
  const int col = 32 * t + r;
  const float colmask = col < k ? 1.f : 0.f;
  const int colc = col < k ? col : k - 1;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  u32x4 a_next[4];
  float b_next[16];
  auto issue = [&](int c) {
    if (A_BF16) {        // 32 rows x 64 B: lane -> row 16 j + (lane >> 2), 16-B piece lane & 3 (8 values)
      const unsigned short* A = (const unsigned short*)Av + (tile * 32) * lda + (int64_t)c * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        a_next[j] = *reinterpret_cast<const u32x4*>(A + (int64_t)(16 * j + (lane >> 2)) * lda + 8 * (lane & 3));
    } else {             // 32 rows x 128 B: lane -> row 8 j + (lane >> 3), 16-B piece lane & 7 (4 values)
      const float* A = (const float*)Av + (tile * 32) * lda + (int64_t)c * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a_next[j] = *reinterpret_cast<const u32x4*>(A + (int64_t)(8 * j + (lane >> 3)) * lda + 4 * (lane & 7));
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      int64_t kr = (int64_t)c * 32 + 2 * s2 + h;          // rows past the factor's end meet zero columns of A: clamped, finite
      kr = kr < k_valid ? kr : k_valid - 1;
      b_next[s2] = B[kr * k + colc];
    }
  };
  if (c_lo < c_hi) issue(c_lo);
  for (int c = c_lo; c < c_hi; ++c) {
    float* T = tileA[wave][c & 1];
    float b_cur[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) b_cur[s2] = b_next[s2] * colmask;
    if (A_BF16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = 16 * j + (lane >> 2), c0 = 8 * (lane & 3);
        f32x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          lo4[2 * e] = __uint_as_float(a_next[j][e] << 16);
          lo4[2 * e + 1] = __uint_as_float(a_next[j][e] & 0xFFFF0000u);
          hi4[2 * e] = __uint_as_float(a_next[j][2 + e] << 16);
          hi4[2 * e + 1] = __uint_as_float(a_next[j][2 + e] & 0xFFFF0000u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          T[row * 36 + c0 + e] = lo4[e];
          T[row * 36 + c0 + 4 + e] = hi4[e];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(8 * j + (lane >> 3)) * 36 + 4 * (lane & 7) + e] = __uint_as_float(a_next[j][e]);
    }
    if (c + 1 < c_hi) issue(c + 1);                  // in flight under this chunk's MFMAs
    // the chunk's 32 products in a chain of their own, then ONE addition into the running sum: a float32 chain's rounding
    // grows like the root of its length -- sqrt(16 + K / 32) instead of sqrt(K / 2) steps (3 x less at K = 1024)
    f32x16 part;
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      part = __builtin_amdgcn_mfma_f32_32x32x2f32(T[r * 36 + 2 * s2 + h], b_cur[s2], part, 0, 0, 0);
    acc += part;
  }
  f32x4* out = reinterpret_cast<f32x4*>(slab);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const f32x4 v = {acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
    out[((tile * NT + t) * 4 + q4) * 64 + lane] = v;
  }
}

static hipError_t launch_exact_product(int x_dtype, const GemmPlan& p, const void* A, int64_t lda, const float* B, int64_t k_valid,
                                       int k, int NT, float* slabs, int64_t slab_stride, hipStream_t s, const SweepPrologue& pro) {
  if (!B || k_valid < 1 || k < 1) return hipErrorInvalidValue;
  const int64_t tiles = (int64_t)p.panels * p.panel_rows / 32, npairs = tiles * NT;
  const int nchunks = p.nchunks;                    // 32 reduction elements each (both storages)
  const unsigned grid = (unsigned)((npairs * p.max_seg + 3) / 4);
  if (x_dtype == 1) exact_product_kernel<true><<<grid, 256, 0, s>>>(A, lda, nchunks, B, k_valid, k, NT, p.max_seg, npairs, slabs, slab_stride, pro);
  else exact_product_kernel<false><<<grid, 256, 0, s>>>(A, lda, nchunks, B, k_valid, k, NT, p.max_seg, npairs, slabs, slab_stride, pro);
  return hipGetLastError();
}

hipError_t launch_gemm(int x_dtype, const GemmPlan& p, const void* A, int64_t lda, int64_t M_store,
                       const void* packedB, int NT, float* slabs, int64_t slab_stride, hipStream_t s, bool split,
                       const SweepPrologue* prologue, bool attr_only, const float* factor, int64_t k_valid, int k) {
  const int64_t lda16 = lda * elem_bytes(x_dtype) / 16;
  SweepPrologue pro = prologue ? *prologue : SweepPrologue{};
  if (pro.kind != PROLOGUE_NONE && pro.nparts > GRAM_MAX_PARTS) return hipErrorInvalidValue;
  if (p.exact) {                                    // a short reduction: the exact float32 product (see above)
    if (attr_only) return hipSuccess;
    if (pro.kind != PROLOGUE_NONE) {
      if (pro.ncols != 0) return hipErrorInvalidValue;
      const int jobs = gram_reduce_jobs(pro.k) + (pro.kind == PROLOGUE_POST ? pro.k + 1 : 0);
      if (jobs > 4 * p.grid) {                      // too few workgroups to share the jobs: a launch of their own (as for the sweep)
        const hipError_t e = launch_prologue_alone(pro, s);
        if (e != hipSuccess) return e;
        pro = SweepPrologue{};
      }
    }
    return launch_exact_product(x_dtype, p, A, lda, factor, k_valid, k, NT, slabs, slab_stride, s, pro);
  }
  if (pro.kind != PROLOGUE_NONE && !attr_only) {
    // The prologue's jobs (k^2 / 8 Gram entries + k + 1 column sums) are dealt over the sweep's workgroups.  A small
    // matrix launches few of them (one for 100 x 2000): at k = 64 its 512 jobs then ran one after the other -- 0.79 ms
    // in front of a 10 us sweep (tools/shape_scan.py).  More than 4 jobs per workgroup: a launch of their own.
    if (pro.ncols != 0) return hipErrorInvalidValue;      // the pathway terms ride in the U epilogue launch (GraphJob)
    const int jobs = gram_reduce_jobs(pro.k) + (pro.kind == PROLOGUE_POST ? pro.k + 1 : 0);
    if (jobs > 4 * p.grid * (p.nw / 4)) {          // a 512-thread workgroup runs two jobs side by side
      const hipError_t e = launch_prologue_alone(pro, s);
      if (e != hipSuccess) return e;
      pro = SweepPrologue{};
    }
  }
  // arithmetic: 1 = bf16 X x hi/lo bf16 factor; 2 = f32 X split into three bf16 pieces (every k)
  const int mode = x_dtype == 1 ? 1 : 2;
  if (x_dtype != 1 && !split) return hipErrorInvalidValue;
  if (p.nw == SWEEP_NW_WIDE) {
    if (NT < 2 || !p.bdb) return hipErrorInvalidValue;
    if (mode == 2) {
      if (p.mt == 1) return launch_sweep_nt<2, SWEEP_NW_WIDE, true, 1>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
      return launch_sweep_nt<2, SWEEP_NW_WIDE, true, SWEEP_MT>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
    }
    if (p.mt != SWEEP_MT) return hipErrorInvalidValue;
    return launch_sweep_nt<1, SWEEP_NW_WIDE, true, SWEEP_MT>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
  }
  if (p.nw != SWEEP_NW || p.mt != f32_wave_tiles(x_dtype, NT) || p.bdb != (NT <= 2)) return hipErrorInvalidValue;
  if (NT <= 2) {
    if (mode == 1) return launch_sweep_nt<1, SWEEP_NW, true, SWEEP_MT>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
    if (p.mt == 1) return launch_sweep_nt<2, SWEEP_NW, true, 1>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
    return launch_sweep_nt<2, SWEEP_NW, true, SWEEP_MT>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
  }
  if (mode == 1) return launch_sweep_nt<1, SWEEP_NW, false, SWEEP_MT>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
  return launch_sweep_nt<2, SWEEP_NW, false, 1>(p, A, lda16, M_store, packedB, NT, slabs, slab_stride, s, pro, attr_only);
}

// the prologue work without a sweep behind it (end of a batch)
__global__ void __launch_bounds__(256) prologue_alone_kernel(const SweepPrologue pro) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 1024];
  run_sweep_prologue(pro, lds, (int)sizeof(lds));
}

hipError_t launch_prologue_alone(const SweepPrologue& p, hipStream_t s) {
  if (p.kind == PROLOGUE_NONE) return hipSuccess;
  if (p.nparts > GRAM_MAX_PARTS) return hipErrorInvalidValue;
  if (p.ncols != 0) return hipErrorInvalidValue;
  int jobs = gram_reduce_jobs(p.k);
  const int extra = p.kind == PROLOGUE_POST ? p.k + 1 : 0;
  if (jobs < extra) jobs = extra;
  prologue_alone_kernel<<<jobs, 256, 0, s>>>(p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// calibration: a pure stream of the resident matrix (prmf_debug_stream; bench.py's `roofline.stream_gbs`)
// ---------------------------------------------------------------------------
// What this box delivers for the sweep's bytes with nothing else going on: one workgroup per resident slot of the
// sweep (2 per CU x 256 threads), each a contiguous share, every wave instruction 1 KiB contiguous (16 B per lane),
// 8 non-temporal loads in flight per lane -- the loads of the sweep without its LDS bounce, factor and MFMAs.  The sweep's
// rate divided by this one separates the kernel from the box (boxes of the pool differ by up to 8%).
__global__ void __launch_bounds__(256) stream_read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
  unsigned acc = 0;
  for (size_t base = lo; base < hi; base += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t i = base + (size_t)j * 256 + threadIdx.x;
      v[j] = __builtin_nontemporal_load(src + (i < hi ? i : lo));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x12345678u && lo > hi) sink[blockIdx.x] = acc;   // never true: keeps the loads alive
}

hipError_t launch_stream_read(const void* p, size_t bytes, int n_cus, void* sink, hipStream_t s) {
  if (!p || bytes < 16) return hipSuccess;
  stream_read_kernel<<<dim3(2 * n_cus), 256, 0, s>>>((const u32x4*)p, bytes / 16, (unsigned*)sink);
  return hipGetLastError();
}

}  // namespace prmf
