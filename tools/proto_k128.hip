// Prototype of the k = 128 sweep shape DESIGN.md section 10 (3) describes -- NOT part of the library.
//   C (M x 128, f32) = X (M x K, bf16, row-major) * (B_hi + B_lo) (K x 128 as packed bf16 fragments)
// One workgroup of 4 waves per CU, ONE wave per SIMD; a wave owns 128 rows x 128 columns = 8 x 8 accumulator tiles of
// v_mfma_f32_16x16x32_bf16 (256 accumulator registers).  A fragments are loaded from global memory straight in the
// operand order (lane = (row & 15, 8 consecutive k): no LDS for X), one chunk (64 k) ahead; the 32 KiB factor chunk is
// double-buffered in LDS, one barrier per chunk; every B fragment read from LDS feeds 8 row tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto_k128.hip -o /tmp/proto_k128 && /tmp/proto_k128 [panels_per_wg] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NCOL = 128, CT = NCOL / 16, RT = 8, WROWS = 16 * RT, PANEL = 4 * WROWS;   // 512-row panels
constexpr int CHUNK_K = 64, KB = CHUNK_K / 32;
constexpr int BCHUNK_U4 = KB * CT * 2 * 64;                      // uint4 per factor chunk: 2048 = 32 KiB

#ifndef NABUF
#define NABUF 3       // A register sets: chunk in use + NABUF - 1 chunks in flight
#endif
#ifndef LOADS_PER_STEP
#define LOADS_PER_STEP 1       // A loads issued per (k-block, column tile) step of 16 MFMAs (16 loads per chunk)
#endif

#ifndef ABL
#define ABL 0        // timing ablations (wrong results): 1 no A loads in the loop, 2 no B staging (loads + LDS writes), 4 no barrier, 8 no B fragment reads, 16 no MFMAs
#endif
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// The accumulators are pinned to AccVGPRs: left to the compiler, the 64 accumulator tiles were shuffled between VGPRs and
// AccVGPRs all the time (1300 v_accvgpr_read / write per 512 MFMAs).
__device__ __forceinline__ void mfma16(f32x4& acc, u32x4 a, u32x4 b) {
#if ABL & 16
  asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[0]) : "v"(a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0]));      // consume the operands, no matrix work
#else
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
proto_kernel(const u32x4* __restrict__ X, long ldx16 /* row pitch in 16-B units */, const u32x4* __restrict__ B, int nchunks,
             int npanels, float* __restrict__ C) {
  __shared__ u32x4 sB[2][BCHUNK_U4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, akg = lane >> 4;
  for (int panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const long row0 = (long)panel * PANEL + wave * WROWS;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane's A pointer: row (row0 + 16 rt + arow), 16-B unit (chunk * 8 + kb * 4 + akg)
    const unsigned voff = (unsigned)(arow * ldx16 + akg);      // per-lane part of every A address; the rest is uniform
    u32x4 a[NABUF][RT][KB];
    // prologue: A of chunks 0 .. NABUF-2, B chunk 0 into LDS buffer 0
#pragma unroll
    for (int d = 0; d < NABUF - 1; ++d)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a[d][r][kb] = __builtin_nontemporal_load((X + (row0 + 16 * r) * ldx16 + d * 8 + kb * 4) + voff);
#pragma unroll
    for (int j = 0; j < BCHUNK_U4 / 256; ++j) sB[0][j * 256 + tid] = B[j * 256 + tid];
    __syncthreads();
    // nchunks is a multiple of 2 NABUF; X rows and B carry NABUF chunks of padding, so no prefetch needs a guard
    for (int c = 0; c < nchunks; c += 2 * NABUF) {
      auto body = [&](auto cura, auto curb, int cc) {
        constexpr int CA = decltype(cura)::value, NA = (CA + NABUF - 1) % NABUF;      // A set of chunk cc | of chunk cc + NABUF - 1
        constexpr int CUR = decltype(curb)::value, NXT = CUR ^ 1;
        // B chunk cc + 1: global -> LDS buffer NXT directly (LDS-DMA: no staging registers); that buffer was last read in
        // chunk cc - 1, which every wave left through the barrier
        if (!(ABL & 2)) {
#pragma unroll
          for (int j = 0; j < BCHUNK_U4 / 256; ++j)
            __builtin_amdgcn_global_load_lds((const void*)(B + (long)(cc + 1) * BCHUNK_U4 + j * 256 + tid),
                                             (__attribute__((address_space(3))) void*)(&sB[NXT][j * 256 + wave * 64]), 16, 0, 0);
        }
        // B fragments one step ahead of their MFMAs (the MFMAs are volatile asm: the compiler does not move the reads itself)
        u32x4 bh_n = sB[CUR][0 * 64 + lane], bl_n = sB[CUR][1 * 64 + lane];
#pragma unroll
        for (int step = 0; step < KB * CT; ++step) {
          const int kb = step / CT, ct = step % CT;
          const u32x4 bh = bh_n, bl = bl_n;
          if (step + 1 < KB * CT && !(ABL & 8)) {
            bh_n = sB[CUR][((step + 1) * 2 + 0) * 64 + lane];
            bl_n = sB[CUR][((step + 1) * 2 + 1) * 64 + lane];
          }
          // sprinkle the A loads of chunk cc + NABUF - 1 through the MFMA stream (the two halves of a row's line in
          // consecutive steps: 8 steps apart costs 25 %)
#pragma unroll
          for (int q = 0; q < LOADS_PER_STEP; ++q) {
            const int li = step * LOADS_PER_STEP + q;          // 0 .. 15
            if (li < RT * KB && !(ABL & 1)) {
              const int r = li / KB, k2 = li % KB;
              a[NA][r][k2] = __builtin_nontemporal_load((X + (row0 + 16 * r) * ldx16 + (long)(cc + NABUF - 1) * 8 + k2 * 4) + voff);
            }
          }
#pragma unroll
          for (int r = 0; r < RT; ++r) mfma16(acc[r][ct], a[CA][r][kb], bh);
#pragma unroll
          for (int r = 0; r < RT; ++r) mfma16(acc[r][ct], a[CA][r][kb], bl);
        }
        // the LDS-DMA of this chunk is older than its RT KB A loads: wait until only those are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RT * KB) : "memory");
        if (!(ABL & 4)) __syncthreads();
      };
      auto six = [&](auto i0) {};
      (void)six;
#define BODY(i) body(std::integral_constant<int, (i) % NABUF>{}, std::integral_constant<int, (i) & 1>{}, c + (i))
      BODY(0); BODY(1);
      if constexpr (NABUF >= 2) { BODY(2); BODY(3); }
      if constexpr (NABUF >= 3) { BODY(4); BODY(5); }
#undef BODY
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> v_accvgpr_read: the hazard the compiler cannot see through inline asm
    // C/D layout: lane (col = lane & 15, rows 4 (lane >> 4) + i)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          C[(row0 + 16 * r + 4 * (lane >> 4) + i) * NCOL + 16 * ct + (lane & 15)] = acc[r][ct][i];
  }
}

__global__ void fill_kernel(unsigned short* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // bf16 in [0.5, 1): sign 0, exponent 126, 7 random mantissa bits
    p[i] = (unsigned short)((126u << 7) | (h & 127u));
  }
}

static float bf16_to_f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int ppw = argc > 1 ? atoi(argv[1]) : 2;            // panels per workgroup
  const int K = argc > 2 ? atoi(argv[2]) : 24960;          // a multiple of 64 * 2 * NABUF
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int npanels = ncu * ppw; const long M = (long)npanels * PANEL;
  const int nchunks = K / CHUNK_K;
  const long ldx = K + 64 * NABUF;                         // NABUF chunks of padding per row
  if ((K / 64) % (2 * NABUF)) { printf("K / 64 must be a multiple of %d\n", 2 * NABUF); return 1; }
  const long xelems = M * ldx;
  unsigned short* X; u32x4* B; float* C;
  CHK(hipMalloc(&X, xelems * 2)); CHK(hipMalloc(&B, (size_t)(nchunks + 2) * BCHUNK_U4 * 16)); CHK(hipMalloc(&C, M * NCOL * 4));
  fill_kernel<<<4096, 256>>>(X, xelems, 1u);
  fill_kernel<<<512, 256>>>((unsigned short*)B, (long)(nchunks + 2) * BCHUNK_U4 * 8, 7u);
  CHK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CHK(hipEventRecord(e0));
    proto_kernel<<<ncu, 256>>>((const u32x4*)X, ldx / 8, B, nchunks, npanels, C);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  CHK(hipGetLastError());
  const double bytes = (double)M * K * 2;
  printf("M=%ld K=%d (X %.2f GB) NABUF=%d LOADS_PER_STEP=%d: %.4f ms  %.0f GB/s of X  (%.1f TFLOP/s hi+lo)\n", M, K, bytes / 1e9, NABUF, LOADS_PER_STEP, best,
         bytes / best / 1e6, 2.0 * M * K * NCOL * 2 / best / 1e9);
  // check 3 x 16 rows against the host
  std::vector<unsigned short> hB((size_t)nchunks * BCHUNK_U4 * 8); CHK(hipMemcpy(hB.data(), B, hB.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  for (long rbase : {0L, M / 2 + 48, M - 16}) {
    std::vector<unsigned short> hX((size_t)16 * ldx); CHK(hipMemcpy(hX.data(), X + rbase * ldx, hX.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> hC((size_t)16 * NCOL); CHK(hipMemcpy(hC.data(), C + rbase * NCOL, hC.size() * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < 16; ++r)
      for (int col = 0; col < NCOL; ++col) {
        double s = 0;
        for (int k = 0; k < K; ++k) {
          const int c = k / 64, kb = (k % 64) / 32, kg = (k % 32) / 8, j = k % 8, ct = col / 16, ln = (col % 16) + 16 * kg;
          const size_t base = ((size_t)c * BCHUNK_U4 + ((kb * CT + ct) * 2) * 64 + ln) * 8 + j;
          s += (double)bf16_to_f(hX[(size_t)r * ldx + k]) * ((double)bf16_to_f(hB[base]) + (double)bf16_to_f(hB[base + 64 * 8]));
        }
        const double e = fabs(hC[r * NCOL + col] - s) / fabs(s);
        if (e > worst) worst = e;
      }
  }
  printf("max relative error on 48 checked rows: %.2e\n", worst);
  return 0;
}
