// What does each structural element of the sweep cost on top of pure streaming?
// 512-thread workgroups, 8 waves x 32 rows, 128 B per row per step, line-complete loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 stream, 1 +barrier/step, 2 +LDS bounce, 3 +B stage (16 KiB/step via LDS, double-buffered)
__global__ void __launch_bounds__(512) ladder(const u32x4* __restrict__ A, long lda16, const u32x4* __restrict__ B,
                                              int nch_total, int nsplits, unsigned* out) {
  extern __shared__ u32x4 lds[];
  u32x4* sA = lds;                 // 8 waves x 256
  u32x4* sB = lds + 8 * 256;       // 2 x 1024
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = ((long)blockIdx.x * 8 + wave) * 32;
  const int per = nch_total / nsplits, c0 = blockIdx.y * per;
  const u32x4* p = A + (row0 + (lane >> 3)) * lda16 + (long)c0 * 8 + (lane & 7);
  const u32x4* bp = B + (long)c0 * 1024 + tid;
  unsigned acc = 0;
  for (int c = 0; c < per; ++c) {
    u32x4 v[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = p[(long)i * 8 * lda16];
    if (MODE >= 3) { b[0] = bp[0]; b[1] = bp[512]; }
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sA[wave * 256 + i * 64 + lane] = v[i];
      if (MODE >= 3) { sB[(c & 1) * 1024 + tid] = b[0]; sB[(c & 1) * 1024 + 512 + tid] = b[1]; }
    }
    if (MODE >= 1) __syncthreads();
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { u32x4 t = sA[wave * 256 + ((lane * 4 + i) & 255)]; acc += t[0] ^ t[3]; }
      if (MODE >= 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { u32x4 t = sB[(c & 1) * 1024 + ((i * 64 + lane) & 1023)]; acc += t[1]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    }
    p += 8; bp += 1024;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const long m = 50176, n = 20096;
  const long lda16 = n * 2 / 16;
  u32x4 *A, *B; unsigned* out;
  hipMalloc(&A, m * n * 2); hipMalloc(&B, 314L * 16384); hipMalloc(&out, 4);
  hipMemset(A, 1, m * n * 2); hipMemset(B, 1, 314L * 16384);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int panels = 196, nch = 314;
  auto run = [&](int mode, int nsplits, int lds_kb) {
    dim3 grid(panels, nsplits);
    size_t lds = (size_t)lds_kb * 1024;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) ladder<0><<<grid, 512, lds>>>(A, lda16, B, nch - nch % nsplits, nsplits, out);
      if (mode == 1) ladder<1><<<grid, 512, lds>>>(A, lda16, B, nch - nch % nsplits, nsplits, out);
      if (mode == 2) ladder<2><<<grid, 512, lds>>>(A, lda16, B, nch - nch % nsplits, nsplits, out);
      if (mode == 3) ladder<3><<<grid, 512, lds>>>(A, lda16, B, nch - nch % nsplits, nsplits, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double bytes = (double)panels * 256 * (nch - nch % nsplits) * 128;
    printf("mode %d splits %2d lds %3d KB (=%d WG/CU): %.3f ms  %.2f TB/s\n", mode, nsplits, lds_kb, 160 / lds_kb > 4 ? 4 : 160 / lds_kb, best, bytes / best / 1e9);
  };
  for (int a = 0; a < 4; ++a) { hipFuncSetAttribute((const void*)(a == 0 ? (void*)ladder<0> : a == 1 ? (void*)ladder<1> : a == 2 ? (void*)ladder<2> : (void*)ladder<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  for (int mode = 0; mode < 4; ++mode)
    for (int lds : {40, 64, 80})
      for (int sp : {2, 6, 13}) run(mode, sp, lds);
  return 0;
}
