// Drives libprmf_hip.so through its C-ABI from plain C++ (no Python, no torch): device buffers from hipMalloc,
// inputs from a tiny LCG that tests/test_gpu_native.py repeats in numpy, three iterations, records printed as text.
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include tests/native/capi_native.cpp -o capi_native \\
//       -L prmf_amd/lib -lprmf_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/prmf_amd/lib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>

#include "prmf_hip.h"

static uint32_t lcg_state = 12345u;
static double lcg() {   // uniform in (0, 1]
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return ((lcg_state >> 8) + 1) / 16777216.0;
}
#define CK(x) do { int _rc = (x); if (_rc != PRMF_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, _rc, prmf_last_error(ctx)); return 2; } } while (0)
#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 3; } } while (0)

int main() {
  const int64_t m = 70, n = 150;
  const int k = 4, P = 3, S = 12;                       // 3 path graphs of 12 genes each
  std::vector<float> X(m * n), U(m * k), V(n * k);
  double nsq = 0.0;
  for (auto& x : X) { x = (float)lcg(); nsq += (double)x * x; }
  for (auto& u : U) u = (float)(3.0 * lcg());
  for (auto& v : V) v = (float)(3.0 * lcg());
  // pathway p: genes p*20 + 0..11 in a chain, unit weights
  std::vector<int64_t> sup_ptr(P + 1), row_ptr(P * S + 1);
  std::vector<int32_t> sup_idx(P * S), col_local;
  std::vector<double> sup_deg(P * S), sup_ldiag(P * S), w;
  for (int p = 0; p <= P; ++p) sup_ptr[p] = (int64_t)p * S;
  row_ptr[0] = 0;
  for (int p = 0; p < P; ++p)
    for (int s = 0; s < S; ++s) {
      const int g = p * S + s;
      sup_idx[g] = p * 20 + s;
      int deg = 0;
      if (s > 0) { col_local.push_back(s - 1); w.push_back(1.0); ++deg; }
      if (s + 1 < S) { col_local.push_back(s + 1); w.push_back(1.0); ++deg; }
      sup_deg[g] = deg;
      sup_ldiag[g] = deg;
      row_ptr[g + 1] = (int64_t)col_local.size();
    }
  prmf_ctx* ctx = nullptr;
  if (prmf_abi_version() != PRMF_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
  CK(prmf_create(&ctx, 0, m, n, k, PRMF_X_F32, 1, 0, 0));
  float *dX = nullptr, *dU = nullptr, *dV = nullptr;
  HK(hipMalloc(&dX, X.size() * 4)); HK(hipMalloc(&dU, U.size() * 4)); HK(hipMalloc(&dV, V.size() * 4));
  HK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  CK(prmf_set_x(ctx, dX, n, nsq, nullptr));
  CK(prmf_set_pathway_library(ctx, P, sup_ptr.data(), sup_idx.data(), sup_deg.data(), sup_ldiag.data(), row_ptr.data(),
                              col_local.data(), w.data(), nullptr));
  const int32_t active[4] = {0, 1, 2, 0};
  CK(prmf_set_active(ctx, active, nullptr));
  double obj[3 * PRMF_OBJ_STRIDE], gd[2];
  CK(prmf_step(ctx, dU, dV, 3, 2.5, 0.125, -1.0, obj, gd, nullptr));
  HK(hipMemcpy(U.data(), dU, U.size() * 4, hipMemcpyDeviceToHost));
  HK(hipMemcpy(V.data(), dV, V.size() * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < 3; ++i)
    std::printf("obj %d %.10e %.10e %.10e %.10e %.10e\n", i, obj[i * PRMF_OBJ_STRIDE + PRMF_OBJ_RECON], obj[i * PRMF_OBJ_STRIDE + PRMF_OBJ_MANIFOLD],
                obj[i * PRMF_OBJ_STRIDE + PRMF_OBJ_IGNORE], obj[i * PRMF_OBJ_STRIDE + PRMF_OBJ_FRO], obj[i * PRMF_OBJ_STRIDE + PRMF_OBJ_OBJ]);
  double su = 0, sv = 0;
  for (float u : U) su += u;
  for (float v : V) sv += v;
  std::printf("sumU %.10e sumV %.10e\n", su, sv);
  // error path: a second prmf_step with a NULL factor must fail cleanly
  if (prmf_step(ctx, nullptr, dV, 1, 1.0, 1.0, -1.0, obj, gd, nullptr) != PRMF_ERR_ARG) { std::fprintf(stderr, "NULL U accepted\n"); return 4; }
  CK(prmf_destroy(ctx));
  (void)hipFree(dX); (void)hipFree(dU); (void)hipFree(dV);
  std::printf("native OK\n");
  std::fflush(stdout);
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
  // under the host sanitizers: leave without running the HIP / HSA runtimes' exit handlers -- AMD's ASan runtime aborts in its
  // device allocator once the HSA runtime has unloaded ("dev_runtime_unloaded_" CHECK), after everything of ours has been torn down
  _exit(0);
#endif
#endif
  return 0;
}
