// Prices each structural element of the sweep on top of pure non-temporal streaming, with the sweep's own
// persistent balanced decomposition (512 workgroups x 8 waves, unit = one 128-B chunk of a 256-row panel).
//   0 nt stream | 1 + barrier per chunk | 2 + A through LDS | 3 + packed factor (16 KiB/chunk) staged through LDS
//   4 + 16 bf16 MFMAs per wave per chunk | 5 + 64 KiB slab store per segment
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) ladder(const u32x4* __restrict__ A, long ld16, const u32x4* __restrict__ B, int panels,
                                              int chunks, float* __restrict__ slabs, unsigned* out) {
  __shared__ u32x4 sA[8][256];
  __shared__ u32x4 sB[2][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long T = (long)panels * chunks;
  long u = T * blockIdx.x / gridDim.x;
  const long u1 = T * (blockIdx.x + 1) / gridDim.x;
  unsigned acc = 0;
  f32x16 c0 = {0}, c1 = {0};
  int seg = 0;
  while (u < u1) {
    const int p = (int)(u / chunks), cb = (int)(u - (long)p * chunks);
    const int ce = (int)((u1 - u) < (chunks - cb) ? cb + (u1 - u) : chunks);
    for (int c = cb; c < ce; ++c) {
      u32x4 v[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + 8 * i + (lane >> 3);
        v[i] = __builtin_nontemporal_load(A + ((long)p * 256 + r) * ld16 + (long)c * 8 + (lane & 7));
      }
      if (MODE >= 3 && MODE != 12) { const long cbx = MODE == 11 ? (c & 3) : c; b[0] = B[cbx * 1024 + tid]; b[1] = B[cbx * 1024 + 512 + tid]; }
      if (MODE == 12) {
        // every wave reads its 16 B fragments of the chunk straight from global memory (L1/L2): no LDS staging, no barrier
#pragma unroll
        for (int i = 0; i < 4; ++i) sA[wave][i * 64 + lane] = v[i];
        u32x4 bf[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) bf[f] = B[(long)c * 1024 + f * 64 + lane];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const u32x4 a = sA[wave][(lane & 31) * 8 + ((2 * s4 + (lane >> 5)) ^ ((lane >> 1) & 7))];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x16& cc = t ? c1 : c0;
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bf[(s4 * 2 + t) * 2]), cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bf[(s4 * 2 + t) * 2 + 1]), cc, 0, 0, 0);
          }
        }
        continue;
      }
      if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sA[wave][i * 64 + lane] = v[i];
        if (MODE >= 3) { sB[c & 1][tid] = b[0]; sB[c & 1][512 + tid] = b[1]; }
      }
      if (MODE >= 1) __syncthreads();
      if (MODE >= 4) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const u32x4 a = sA[wave][(lane & 31) * 8 + ((2 * s4 + (lane >> 5)) ^ ((lane >> 1) & 7))];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const u32x4 bh = sB[c & 1][((s4 * 2 + t) * 2 + 0) * 64 + lane], bl = sB[c & 1][((s4 * 2 + t) * 2 + 1) * 64 + lane];
            f32x16& cc = t ? c1 : c0;
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bh), cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bl), cc, 0, 0, 0);
          }
        }
      } else if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { u32x4 t = sA[wave][(lane * 4 + i) & 255]; acc += t[0] ^ t[3]; }
        if (MODE >= 3) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { u32x4 t = sB[c & 1][(i * 64 + lane) & 1023]; acc += t[1]; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
      }
    }
    if (((MODE >= 5 && MODE <= 10) || MODE >= 13) &&
        !(MODE == 16 && u + (ce - cb) >= u1) && !(MODE == 17 && u + (ce - cb) < u1)) {      // 16: only the segments that are not the workgroup's last store; 17: only the last one
      // 5: distinct slab per segment | 6: same, non-temporal stores | 7: one slab per workgroup, overwritten
      // 8: every workgroup writes the same 64 KiB (no write traffic to speak of)
      // 9: as 5 but every store instruction writes 1 KiB contiguous ([q][lane] instead of [lane][q]) | 10: same, non-temporal
      const long slab = MODE == 8 ? 0 : (MODE == 7 ? (long)blockIdx.x * 2 : (long)blockIdx.x * 2 + seg);
      if (MODE >= 9) {      // (13: sc1, 14: sc0 sc1, 15: sc1 nt -- round 5)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* o = reinterpret_cast<f4*>(slabs + slab * 16384) + (long)wave * 2 * 256 + lane;
        unsigned long long t0 = 0;      // 19: how long until this wave's store burst is acknowledged (wave 0 of every workgroup)
        if (MODE == 19) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t0 = wall_clock64(); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f4 x = {c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]}, y = {c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
          if (MODE == 10 || MODE >= 16) { __builtin_nontemporal_store(x, o + q * 64); __builtin_nontemporal_store(y, o + 256 + q * 64); }
          else if (MODE == 13) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o + q * 64), "v"(x) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o + 256 + q * 64), "v"(y) : "memory"); }
          else if (MODE == 14) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(o + q * 64), "v"(x) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(o + 256 + q * 64), "v"(y) : "memory"); }
          else if (MODE == 15) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(o + q * 64), "v"(x) : "memory"); asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(o + 256 + q * 64), "v"(y) : "memory"); }
          else { o[q * 64] = x; o[256 + q * 64] = y; }
        }
        if (MODE == 19) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long t1 = wall_clock64();
          if (lane == 0 && wave == 0) { const bool last = u + (ce - cb) >= u1; out[16 + blockIdx.x * 4 + (last ? 1 : 0)] = (unsigned)(t1 - t0); out[16 + blockIdx.x * 4 + 2 + (last ? 1 : 0)] = 1; }
        }
        seg ^= 1;
        u += ce - cb;
        continue;
      }
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4* o = reinterpret_cast<f4*>(slabs + slab * 16384) + ((long)wave * 2 * 64 + lane) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 x = {c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]}, y = {c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        if (MODE == 6) { __builtin_nontemporal_store(x, o + q); __builtin_nontemporal_store(y, o + 256 + q); }
        else { o[q] = x; o[256 + q] = y; }
      }
      seg ^= 1;
    }
    u += ce - cb;
  }
  if (MODE >= 4) acc += (unsigned)(c0[0] + c1[3]);

  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const long rows = 50176, rowbytes = 40192;
  const int panels = rows / 256, chunks = rowbytes / 128;
  u32x4 *A, *B; unsigned* out; float* slabs;
  (void)hipMalloc(&A, rows * rowbytes); (void)hipMalloc(&B, (long)chunks * 16384); (void)hipMalloc(&out, 65536); (void)hipMemset(out, 0, 65536);
  (void)hipMalloc(&slabs, 1024L * 16384 * 4);
  (void)hipMemset(A, 0, rows * rowbytes); (void)hipMemset(B, 0, (long)chunks * 16384);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 512;
  auto run = [&](int mode) {
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
      (void)hipEventRecord(e0);
      switch (mode) {
        case 0: ladder<0><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 1: ladder<1><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 2: ladder<2><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 3: ladder<3><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 4: ladder<4><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 5: ladder<5><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 6: ladder<6><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 7: ladder<7><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 8: ladder<8><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 9: ladder<9><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 10: ladder<10><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 11: ladder<11><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 12: ladder<12><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 13: ladder<13><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 14: ladder<14><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 15: ladder<15><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 16: ladder<16><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 17: ladder<17><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
        case 19: ladder<19><<<grid, 512>>>(A, rowbytes / 16, B, panels, chunks, slabs, out); break;
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)rows * rowbytes;
    printf("mode %d: best %.3f ms %.2f TB/s   mean %.3f ms %.2f TB/s\n", mode, best, bytes / best / 1e9, sum / 6, bytes / (sum / 6) / 1e9);
  };
  for (int rep = 0; rep < 3; ++rep)
    for (int mode : {4, 10, 19}) run(mode);
  {
    static unsigned h[16384];
    (void)hipMemcpy(h, out, 65536, hipMemcpyDeviceToHost);
    double sm = 0, sl = 0, mxm = 0, mxl = 0; int nm = 0, nl = 0;
    for (int b = 0; b < grid; ++b) {
      if (h[16 + b * 4 + 2]) { const double t = h[16 + b * 4] / 100.0; sm += t; ++nm; if (t > mxm) mxm = t; }
      if (h[16 + b * 4 + 3]) { const double t = h[16 + b * 4 + 1] / 100.0; sl += t; ++nl; if (t > mxl) mxl = t; }
    }
    printf("mode 19: store burst -> vmcnt(0): mid-stream %d bursts mean %.2f us max %.2f us | final %d bursts mean %.2f us max %.2f us\n", nm, nm ? sm / nm : 0, mxm, nl, nl ? sl / nl : 0, mxl);
  }
  return 0;
}
