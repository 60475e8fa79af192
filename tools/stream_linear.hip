// How fast can 2 GB be read when every workgroup streams ONE contiguous region (blocked layout)
// compared with the sweep's row-strided pattern (256 rows x 128 B per step per workgroup)?
//   mode 0: strided  -- workgroup = 256-row panel, step = 128 B of each row (row stride = ld bytes)
//   mode 1: linear   -- same bytes per step (32 KB), but contiguous: workgroup w, step s -> offset (w*S + s) * 32 KB
// Both: 8 waves per workgroup, one 16-B load per lane x 4 per step (4 KB per wave per step), persistent grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ void __launch_bounds__(512) k(const u32x4* __restrict__ A, long ld16, int panels, int chunks, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long T = (long)panels * chunks;
  long u = T * blockIdx.x / gridDim.x;
  const long u1 = T * (blockIdx.x + 1) / gridDim.x;
  unsigned acc = 0;
  for (; u < u1; ++u) {
    const int p = (int)(u / chunks), c = (int)(u - (long)p * chunks);
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + 8 * i + (lane >> 3);
      const u32x4* src = MODE == 0 ? A + ((long)p * 256 + r) * ld16 + (long)c * 8 + (lane & 7)
                                   : A + (u * 256 + r) * 8 + (lane & 7);
      v[i] = NT ? __builtin_nontemporal_load(src) : *src;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 50176, rowbytes = argc > 2 ? atol(argv[2]) : 40192;   // X of C3 (padded): 196 panels x 314 chunks
  const int panels = rows / 256, chunks = rowbytes / 128;
  u32x4* A; unsigned* out;
  hipMalloc(&A, rows * rowbytes); hipMalloc(&out, 4);
  hipMemset(A, 1, rows * rowbytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int mode, bool nt, int grid) {
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (mode == 0 && nt) k<0, true><<<grid, 512>>>(A, rowbytes / 16, panels, chunks, out);
      if (mode == 0 && !nt) k<0, false><<<grid, 512>>>(A, rowbytes / 16, panels, chunks, out);
      if (mode == 1 && nt) k<1, true><<<grid, 512>>>(A, rowbytes / 16, panels, chunks, out);
      if (mode == 1 && !nt) k<1, false><<<grid, 512>>>(A, rowbytes / 16, panels, chunks, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)rows * rowbytes;
    printf("%s nt=%d grid=%4d  best %.3f ms %.2f TB/s   mean %.3f ms %.2f TB/s\n", mode ? "linear " : "strided", (int)nt, grid,
           best, bytes / best / 1e9, sum / 5, bytes / (sum / 5) / 1e9);
  };
  printf("rows %ld x %ld bytes = %.1f MB\n", rows, rowbytes, rows * rowbytes / 1e6);
  for (int grid : {512, 1024})
    for (int mode : {0, 1})
      for (int nt : {1, 0}) run(mode, nt, grid);
  return 0;
}
