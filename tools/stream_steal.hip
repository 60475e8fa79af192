// Long-running balanced workgroups vs the same with a dynamically assigned tail pool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void run_range(const u32x4* __restrict__ A, long lda16, const u32x4* __restrict__ B, long row_panel, int c0, int nch,
                          u32x4* sA, u32x4* sB, unsigned& acc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row0 = row_panel * 256 + wave * 32;
  const u32x4* p = A + (row0 + (lane >> 3)) * lda16 + (long)c0 * 8 + (lane & 7);
  const u32x4* bp = B + (long)c0 * 1024 + tid;
  for (int c = 0; c < nch; ++c) {
    u32x4 v[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = p[(long)i * 8 * lda16];
    b[0] = bp[0]; b[1] = bp[512];
#pragma unroll
    for (int i = 0; i < 4; ++i) sA[wave * 256 + i * 64 + lane] = v[i];
    sB[(c & 1) * 1024 + tid] = b[0]; sB[(c & 1) * 1024 + 512 + tid] = b[1];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { u32x4 t = sA[wave * 256 + ((lane * 4 + i) & 255)]; acc += t[0] ^ t[3]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { u32x4 t = sB[(c & 1) * 1024 + ((i * 64 + lane) & 1023)]; acc += t[1]; }
    p += 8; bp += 1024;
  }
}

// units = panels x nch chunks; static part: chunks [0, nstat) of every panel, cut into G equal contiguous ranges;
// dynamic part: chunks [nstat, nch) of every panel in items of `item` chunks, handed out by an atomic counter.
__global__ void __launch_bounds__(512) steal(const u32x4* __restrict__ A, long lda16, const u32x4* __restrict__ B, int panels, int nch, int nstat,
                                             int item, unsigned* counter, unsigned* out) {
  extern __shared__ u32x4 lds[];
  u32x4* sA = lds; u32x4* sB = lds + 8 * 256;
  __shared__ unsigned s_item;
  unsigned acc = 0;
  const long T = (long)panels * nstat, G = gridDim.x, w = blockIdx.x;
  long u = w * T / G; const long ue = (w + 1) * T / G;
  while (u < ue) {
    const int panel = (int)(u / nstat); const int c0 = (int)(u - (long)panel * nstat);
    const int c1 = (int)min((long)nstat, c0 + (ue - u));
    run_range(A, lda16, B, panel, c0, c1 - c0, sA, sB, acc);
    u += c1 - c0;
  }
  const int ipp = (nch - nstat + item - 1) / (item > 0 ? item : 1);   // items per panel
  const unsigned nitems = item > 0 ? (unsigned)panels * ipp : 0;
  while (nitems) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const unsigned it = s_item;
    if (it >= nitems) break;
    const int panel = it / ipp, j = it % ipp;
    const int c0 = nstat + j * item; const int c1 = min(nch, c0 + item);
    run_range(A, lda16, B, panel, c0, c1 - c0, sA, sB, acc);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  // shapes: 50000 x 20000 bf16 (196 panels x 314 chunks of 128 B), 5000 x 10000 f32 (20 x 316), 6250 x 20000 bf16 (25 x 314)
  const int shapes[3][2] = {{196, 314}, {20, 316}, {25, 314}};
  const long m = 50176, n = 20096; const long lda16 = n * 2 / 16;
  u32x4 *A, *B; unsigned *out, *counter;
  hipMalloc(&A, m * n * 2); hipMalloc(&B, 316L * 16384); hipMalloc(&out, 4); hipMalloc(&counter, 4);
  hipMemset(A, 1, m * n * 2); hipMemset(B, 1, 316L * 16384);
  hipFuncSetAttribute((const void*)steal, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int sh = 0; sh < 3; ++sh) {
    const int panels = shapes[sh][0], nch = shapes[sh][1];
    printf("--- %d panels x %d chunks (%.0f MB)\n", panels, nch, (double)panels * 256 * nch * 128 / 1e6);
    auto run = [&](int G, int tail_pct, int item) {
      int nstat = item > 0 ? nch - nch * tail_pct / 100 : nch;
      float best = 1e9, sum = 0;
      const int reps = 9;
      for (int rep = 0; rep < reps; ++rep) {
        hipMemset(counter, 0, 4);
        hipEventRecord(e0);
        steal<<<G, 512, 64 * 1024>>>(A, lda16, B, panels, nch, nstat, item, counter, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        if (rep > 0) sum += ms;
      }
      double bytes = (double)panels * 256 * nch * 128;
      printf("G=%4d tail=%3d%% item=%2d chunks: best %.4f ms  mean %.4f ms  %.2f TB/s (best)\n", G, tail_pct, item, best, sum / (reps - 1), bytes / best / 1e9);
    };
    run(512, 0, 0);
    for (int tail : {10, 20, 30, 50}) for (int item : {2, 4, 8}) run(512, tail, item);
    run(512, 100, 2); run(512, 100, 4); run(512, 100, 8);
  }
  return 0;
}
