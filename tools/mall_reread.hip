// Decision measurement for the one-pass-over-X design:
// how fast is a re-read of a row group of X that was just streamed from HBM, i.e. what can the 256 MiB
// Infinity Cache (MALL) give the second product (Q += X_g^T U_g) of a fused iteration?
//
//   for S in sizes:   flush (stream 1.5 GB of another buffer)
//                     cold = stream S bytes of X            (HBM)
//                     warm = stream the same S bytes again  (MALL if the group stayed resident), x4
//   grouped:          for g in groups of S bytes over 2 GB: read group g, read group g again  (= the fused order)
//                     against 2 x one plain pass over the 2 GB  (= today's two sweeps)
//
// Access pattern = the sweep's: 512 persistent workgroups x 256 threads, every wave instruction reads 1 KiB
// contiguous (16 B per lane), 8 loads in flight per lane, first pass with plain or non-temporal loads.
// Build/run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mall tools/mall_reread.hip && /tmp/mall
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  // workgroup b owns the contiguous range [b, b+1) * n16 / grid, processed in steps of 256 threads x 8 loads
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
  unsigned acc = 0;
  for (size_t base = lo; base < hi; base += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t i = base + (size_t)j * 256 + threadIdx.x;
      const u32x4* p = src + (i < hi ? i : lo);
      v[j] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;   // never true in practice: keeps the loads alive
}

static float timed(hipStream_t s, hipEvent_t a, hipEvent_t b, bool nt, const void* p, size_t bytes, unsigned* sink, int reps = 1) {
  CHECK(hipEventRecord(a, s));
  for (int r = 0; r < reps; ++r) {
    if (nt) stream_kernel<true><<<512, 256, 0, s>>>((const u32x4*)p, bytes / 16, sink);
    else stream_kernel<false><<<512, 256, 0, s>>>((const u32x4*)p, bytes / 16, sink);
  }
  CHECK(hipEventRecord(b, s));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const size_t MB = 1000 * 1000;
  const size_t xbytes = 2000 * MB, fbytes = 1500 * MB;
  char *X = nullptr, *F = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc(&X, xbytes));
  CHECK(hipMalloc(&F, fbytes));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(X, 1, xbytes));
  CHECK(hipMemset(F, 2, fbytes));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) timed(s, a, b, true, X, xbytes, sink);

  printf("# full pass over 2 GB\n");
  for (int nt = 0; nt < 2; ++nt) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      const float ms = timed(s, a, b, nt, X, xbytes, sink);
      best = ms < best ? ms : best;
    }
    printf("full_pass  loads=%s  %.3f ms  %.0f GB/s\n", nt ? "nt" : "plain", best, xbytes / best / 1e6);
  }

  printf("# cold read of S bytes, then 4 re-reads (first-pass load flavour x re-read flavour)\n");
  const size_t sizes[] = {32 * MB, 64 * MB, 96 * MB, 128 * MB, 160 * MB, 192 * MB, 224 * MB, 256 * MB, 320 * MB, 512 * MB};
  for (size_t S : sizes)
    for (int nt1 = 0; nt1 < 2; ++nt1)
      for (int nt2 = 0; nt2 < 2; ++nt2) {
        float cold = 0, warm = 0;
        const int R = 3;
        for (int r = 0; r < R; ++r) {
          timed(s, a, b, true, F, fbytes, sink);                 // flush
          cold += timed(s, a, b, nt1, X, S, sink);
          warm += timed(s, a, b, nt2, X, S, sink, 4);
        }
        cold /= R;
        warm /= R;
        printf("S=%4zu MB  first=%-5s reread=%-5s  cold %.4f ms %6.0f GB/s   reread %.4f ms %6.0f GB/s  (x%.2f)\n", S / MB,
               nt1 ? "nt" : "plain", nt2 ? "nt" : "plain", cold, S / cold / 1e6, warm, S / warm / 1e6, cold / warm);
      }

  printf("# fused order over 2 GB: per group read, read again (2 launches per group) vs two full passes\n");
  for (size_t S : {50 * MB, 100 * MB, 200 * MB, 400 * MB}) {
    for (int nt1 = 0; nt1 < 2; ++nt1) {
      float best = 1e9f;
      for (int r = 0; r < 3; ++r) {
        timed(s, a, b, true, F, fbytes, sink);
        CHECK(hipEventRecord(a, s));
        for (size_t off = 0; off < xbytes; off += S) {
          const size_t len = off + S <= xbytes ? S : xbytes - off;
          if (nt1) stream_kernel<true><<<512, 256, 0, s>>>((const u32x4*)(X + off), len / 16, sink);
          else stream_kernel<false><<<512, 256, 0, s>>>((const u32x4*)(X + off), len / 16, sink);
          stream_kernel<true><<<512, 256, 0, s>>>((const u32x4*)(X + off), len / 16, sink);
        }
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      printf("grouped  S=%3zu MB first=%-5s  total %.3f ms for 2 x 2 GB  (%.0f GB/s effective)\n", S / MB, nt1 ? "nt" : "plain",
             best, 2.0 * xbytes / best / 1e6);
    }
  }
  {
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      CHECK(hipEventRecord(a, s));
      stream_kernel<true><<<512, 256, 0, s>>>((const u32x4*)X, xbytes / 16, sink);
      stream_kernel<true><<<512, 256, 0, s>>>((const u32x4*)X, xbytes / 16, sink);
      CHECK(hipEventRecord(b, s));
      CHECK(hipEventSynchronize(b));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, a, b));
      best = ms < best ? ms : best;
    }
    printf("two_full_passes (nt)  total %.3f ms for 2 x 2 GB  (%.0f GB/s effective)\n", best, 2.0 * xbytes / best / 1e6);
  }
  return 0;
}
