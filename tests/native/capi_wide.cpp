// The rest of the C-ABI from plain C++ (tests/test_gpu_native.py): several contexts of different plans
// (f32 / bf16 X, k = 6 .. 128, 4- and 8-wave sweeps, sliced reductions, ragged sizes), prmf_step with and without the
// tradeoff feedback and in the eager / direct-residual forms, the phase calls, candidate scores, held-out error, direct
// residual, timing, introspection, the quantile transform, the text reader and writer, the matrix NNLS, invalidation, error paths on a live context.  Values are only
// checked for being finite and sane -- the parity tests own the numbers; what this run is for is every entry point and
// every host-side buffer the library sizes from a plan, from a caller that is not Python.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include <unistd.h>

#include "prmf_hip.h"

static uint32_t lcg_state = 777u;
static double lcg() {
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return ((lcg_state >> 8) + 1) / 16777216.0;
}
#define CK(x) do { int _rc = (x); if (_rc != PRMF_OK) { std::fprintf(stderr, "line %d: %s -> %d: %s\n", __LINE__, #x, _rc, prmf_last_error(ctx)); return 2; } } while (0)
#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { std::fprintf(stderr, "line %d: %s: %s\n", __LINE__, #x, hipGetErrorString(_e)); return 3; } } while (0)
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "line %d: %s failed\n", __LINE__, #c); return 4; } } while (0)

static uint16_t to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct Library {
  int P;
  std::vector<int64_t> sup_ptr, row_ptr;
  std::vector<int32_t> sup_idx, col_local;
  std::vector<double> sup_deg, sup_ldiag, w;
};
// P chains of S genes each, pathway p on genes (p * stride + s) % n, weights 0.5 .. 1.5
static Library chains(int64_t n, int P, int S) {
  Library L;
  L.P = P;
  L.sup_ptr.resize(P + 1);
  L.row_ptr.push_back(0);
  const int64_t stride = n / P > S ? n / P : S;
  for (int p = 0; p <= P; ++p) L.sup_ptr[p] = (int64_t)p * S;
  for (int p = 0; p < P; ++p)
    for (int s = 0; s < S; ++s) {
      L.sup_idx.push_back((int32_t)(((int64_t)p * stride + s) % n));
      double deg = 0;
      if (s > 0) { L.col_local.push_back(s - 1); L.w.push_back(0.5 + (s % 3) * 0.5); deg += L.w.back(); }
      if (s + 1 < S) { L.col_local.push_back(s + 1); L.w.push_back(0.5 + ((s + 1) % 3) * 0.5); deg += L.w.back(); }
      L.sup_deg.push_back(deg);
      L.sup_ldiag.push_back(deg);
      L.row_ptr.push_back((int64_t)L.col_local.size());
    }
  return L;
}

static int run_case(int64_t m, int64_t n, int k, int x_dtype, unsigned flags, int P, int S, double tradeoff, int steps) {
  prmf_ctx* ctx = nullptr;
  std::vector<float> X((size_t)(m * n)), U((size_t)(m * k)), V((size_t)(n * k));
  double nsq = 0.0;
  for (auto& x : X) {
    x = (float)lcg();
    if (x_dtype == PRMF_X_BF16) { uint32_t u = (uint32_t)to_bf16(x) << 16; std::memcpy(&x, &u, 4); }
    nsq += (double)x * x;
  }
  for (auto& u : U) u = (float)(3.0 * lcg());
  for (auto& v : V) v = (float)(3.0 * lcg());
  void* dX = nullptr;
  float *dU = nullptr, *dV = nullptr;
  if (x_dtype == PRMF_X_BF16) {
    std::vector<uint16_t> Xh(X.size());
    for (size_t i = 0; i < X.size(); ++i) Xh[i] = to_bf16(X[i]);
    HK(hipMalloc(&dX, Xh.size() * 2));
    HK(hipMemcpy(dX, Xh.data(), Xh.size() * 2, hipMemcpyHostToDevice));
  } else {
    HK(hipMalloc(&dX, X.size() * 4));
    HK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  }
  HK(hipMalloc(&dU, U.size() * 4 + 16));
  HK(hipMalloc(&dV, V.size() * 4 + 16));
  HK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  CK(prmf_create(&ctx, 0, m, n, k, x_dtype, 1, 0, flags));
  CK(prmf_set_x(ctx, dX, n, nsq, nullptr));
  const Library L = chains(n, P, S);
  CK(prmf_set_pathway_library(ctx, L.P, L.sup_ptr.data(), L.sup_idx.data(), L.sup_deg.data(), L.sup_ldiag.data(), L.row_ptr.data(),
                              L.col_local.data(), L.w.data(), nullptr));
  std::vector<int32_t> active((size_t)k);
  for (int i = 0; i < k; ++i) active[i] = i % P;
  CK(prmf_set_active(ctx, active.data(), nullptr));
  prmf_sweep_info info;
  CK(prmf_get_sweep_info(ctx, &info));
  REQUIRE(info.grid_xv >= 1 && info.grid_xtu >= 1 && info.splits_xv >= 1 && info.splits_xtu >= 1);
  CK(prmf_set_timing(ctx, 1));
  std::vector<double> obj((size_t)steps * PRMF_OBJ_STRIDE);
  double gd[2] = {0, 0};
  const double gamma = std::sqrt(nsq) / k, delta = 10.0 / std::sqrt(nsq);
  CK(prmf_step(ctx, dU, dV, steps, gamma, delta, tradeoff, obj.data(), gd, nullptr));
  for (int i = 0; i < steps; ++i) REQUIRE(std::isfinite(obj[(size_t)i * PRMF_OBJ_STRIDE + PRMF_OBJ_OBJ]) && obj[(size_t)i * PRMF_OBJ_STRIDE + PRMF_OBJ_RECON] >= 0.0);
  double a = 0, b = 0, c = 0;
  CK(prmf_get_timing(ctx, &a, &b, &c));
  REQUIRE(a > 0 && b > 0);
  CK(prmf_set_timing(ctx, 0));
  // the phase calls (what a host that drives the collectives itself uses), two iterations
  CK(prmf_begin(ctx, dU, dV, gamma, delta, -1.0, nullptr));
  for (int it = 0; it < 2; ++it) {
    CK(prmf_phase_u(ctx, nullptr));
    CK(prmf_phase_q(ctx, nullptr));
    CK(prmf_phase_v(ctx, nullptr));
    CK(prmf_phase_post(ctx, nullptr));
  }
  double obj2[2 * PRMF_OBJ_STRIDE];
  CK(prmf_end(ctx, 2, obj2, gd, nullptr));
  REQUIRE(std::isfinite(obj2[PRMF_OBJ_STRIDE + PRMF_OBJ_OBJ]));
  // candidate scores, direct residual, held-out error
  std::vector<double> scores((size_t)k * P), err((size_t)m);
  CK(prmf_score_candidates(ctx, dV, scores.data(), nullptr));
  for (double s : scores) REQUIRE(std::isfinite(s));
  double rsq = -1.0;
  CK(prmf_residual_sq(ctx, dU, dV, &rsq, nullptr));
  REQUIRE(rsq >= 0.0 && std::isfinite(rsq));
  if (m <= 4000) {
    CK(prmf_cv_error(ctx, dV, err.data(), nullptr));
    for (double e : err) REQUIRE(e >= 0.0 && e <= 1.0 + 1e-9);
  }
  // the terms of the final matching (ABI 3): finite, the quadratic form of a Laplacian and a sum of positive terms
  {
    std::vector<double> quad((size_t)k * P), ign((size_t)k * P);
    CK(prmf_match_terms(ctx, dV, quad.data(), ign.data(), nullptr));
    for (size_t i = 0; i < quad.size(); ++i) REQUIRE(std::isfinite(quad[i]) && quad[i] >= -1e-9 && ign[i] > 0.0);
  }
  // the one-pass uploads (ABI 3): the same matrix from the HOST (narrowed by worker threads, and on the device) and from the
  // device in its source type leave the same ||X||^2 as the sum the caller computed, and the same next step
  {
    double l_host = -1, l_host0 = -1, l_dev = -1;
    std::vector<double> o1(PRMF_OBJ_STRIDE), o2(PRMF_OBJ_STRIDE), o3(PRMF_OBJ_STRIDE);
    float *U2 = nullptr, *V2 = nullptr;
    HK(hipMalloc(&U2, U.size() * 4 + 16));
    HK(hipMalloc(&V2, V.size() * 4 + 16));
    auto one_step = [&](std::vector<double>& o) -> int {
      HK(hipMemcpy(U2, U.data(), U.size() * 4, hipMemcpyHostToDevice));
      HK(hipMemcpy(V2, V.data(), V.size() * 4, hipMemcpyHostToDevice));
      CK(prmf_invalidate_factors(ctx));
      CK(prmf_step(ctx, U2, V2, 1, gamma, delta, -1.0, o.data(), nullptr, nullptr));
      return 0;
    };
    CK(prmf_set_x_host(ctx, X.data(), PRMF_SRC_F32, n, 3, &l_host, nullptr));
    if (one_step(o1)) return 1;
    CK(prmf_set_x_host(ctx, X.data(), PRMF_SRC_F32, n, 0, &l_host0, nullptr));
    if (one_step(o2)) return 1;
    float* dXf = nullptr;
    HK(hipMalloc(&dXf, X.size() * 4));
    HK(hipMemcpy(dXf, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(prmf_set_x_device(ctx, dXf, PRMF_SRC_F32, n, &l_dev, nullptr));
    if (one_step(o3)) return 1;
    REQUIRE(l_host == l_host0 && l_host == l_dev && std::fabs(l_host - nsq) <= 1e-10 * nsq);   // (nsq: the naive running sum above, 4e7 terms)
    REQUIRE(std::memcmp(o1.data(), o2.data(), o1.size() * 8) == 0 && std::memcmp(o1.data(), o3.data(), o1.size() * 8) == 0);
    REQUIRE(prmf_set_x_host(ctx, X.data(), 7, n, 1, &l_host, nullptr) == PRMF_ERR_ARG);
    REQUIRE(prmf_set_x_host(ctx, X.data(), PRMF_SRC_F32, n - 1, 1, &l_host, nullptr) == PRMF_ERR_ARG);
    REQUIRE(prmf_set_norm_x_sq(ctx, -1.0) == PRMF_ERR_ARG);
    CK(prmf_set_norm_x_sq(ctx, nsq));
    HK(hipFree(dXf));
    HK(hipFree(U2));
    HK(hipFree(V2));
  }
  // the caller rewrites V: invalidate, step again; then error paths on the live context
  HK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice));
  CK(prmf_invalidate_factors(ctx));
  CK(prmf_step(ctx, dU, dV, 1, gamma, delta, -1.0, obj.data(), gd, nullptr));
  REQUIRE(prmf_step(ctx, nullptr, dV, 1, 1.0, 1.0, -1.0, obj.data(), gd, nullptr) == PRMF_ERR_ARG);
  REQUIRE(prmf_step(ctx, dU, dV, -1, 1.0, 1.0, -1.0, obj.data(), gd, nullptr) == PRMF_ERR_ARG);
  active[0] = P;      // out of range
  REQUIRE(prmf_set_active(ctx, active.data(), nullptr) == PRMF_ERR_ARG);
  REQUIRE(prmf_step_dist(ctx, dU, dV, 1, 1.0, 1.0, -1.0, obj.data(), gd, nullptr) == PRMF_ERR_STATE);      // no communicator
  CK(prmf_destroy(ctx));
  (void)hipFree(dX); (void)hipFree(dU); (void)hipFree(dV);
  std::printf("case %lld x %lld k=%d %s flags %u tradeoff %.1f: ok (grid %d/%d, slabs %d/%d, sweeps %.3f/%.3f ms)\n", (long long)m, (long long)n, k,
              x_dtype == PRMF_X_BF16 ? "bf16" : "f32", flags, tradeoff, info.grid_xv, info.grid_xtu, info.splits_xv, info.splits_xtu, a, b);
  return 0;
}

int main() {
  if (prmf_abi_version() != PRMF_ABI_VERSION) return 1;
  int rc;
  if ((rc = run_case(70, 150, 6, PRMF_X_F32, 0, 3, 12, -1.0, 3))) return rc;                     // the CLI's default k: element-wise factor loads
  if ((rc = run_case(1, 37, 1, PRMF_X_BF16, 0, 1, 2, -1.0, 2))) return rc;                       // one sample, one latent column
  if ((rc = run_case(333, 2500, 16, PRMF_X_F32, 0, 5, 40, 0.5, 3))) return rc;                   // tradeoff feedback: 7 launches + direct residual
  if ((rc = run_case(900, 700, 47, PRMF_X_F32, PRMF_FLAG_EAGER_OBJECTIVE, 4, 30, -1.0, 3))) return rc;
  if ((rc = run_case(1500, 9000, 64, PRMF_X_BF16, PRMF_FLAG_DIRECT_RESIDUAL, 8, 60, -1.0, 2))) return rc;
  if ((rc = run_case(640, 70000, 64, PRMF_X_BF16, 0, 6, 50, -1.0, 2))) return rc;                // X V: 18 MB factor, 8-wave form, 2 row panels
  if ((rc = run_case(40000, 700, 64, PRMF_X_BF16, 0, 3, 50, -1.0, 2))) return rc;                // X^T U: 10 MB factor over 2 panels
  if ((rc = run_case(2100, 1300, 100, PRMF_X_F32, 0, 7, 35, -1.0, 2))) return rc;                // f32 X, k > 64: 32-row wave tiles, column-split epilogue
  if ((rc = run_case(3000, 2600, 128, PRMF_X_BF16, 0, 9, 70, 0.3, 2))) return rc;                // k = 128 with the feedback
  {
    // the quantile transform (no context): 500 x 30 float64 with a NaN and a constant column
    const int64_t m = 500, n = 30;
    std::vector<double> A((size_t)(m * n));
    for (auto& x : A) x = lcg();
    for (int64_t i = 0; i < m; ++i) A[(size_t)(i * n + 5)] = 2.0;
    A[7 * n + 3] = NAN;
    double* dA = nullptr;
    HK(hipMalloc(&dA, A.size() * 8));
    HK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    prmf_ctx* ctx = nullptr;
    CK(prmf_quantile_transform(0, dA, m, n, n, nullptr, 0, 1000, nullptr));
    std::vector<int64_t> rows;
    for (int64_t i = 0; i < m; i += 3) rows.push_back(i);
    CK(prmf_quantile_transform(0, dA, m, n, n, rows.data(), (int64_t)rows.size(), 100, nullptr));
    HK(hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < m; ++i)
      for (int64_t j = 0; j < n; ++j) {
        const double v = A[(size_t)(i * n + j)];
        if (i == 7 && j == 3) { REQUIRE(std::isnan(v)); } else { REQUIRE(v >= 0.0 && v <= 1.0); }
      }
    (void)hipFree(dA);
  }
  {
    // the text reader (no context): 40 lines of 1 name + 25 numbers, appended in pieces of 97 bytes, last line not terminated
    prmf_ctx* ctx = nullptr;
    std::string body;
    std::vector<double> want;
    char word[64];
    for (int i = 0; i < 40; ++i) {
      std::snprintf(word, sizeof word, "row %d", i);
      body += word;
      for (int j = 0; j < 25; ++j) {
        const double v = (lcg() - 0.5) * std::pow(10.0, (double)((i + j) % 9 - 4));
        std::snprintf(word, sizeof word, ";%.9g", v);
        body += word;
        want.push_back(std::strtod(word + 1, nullptr));          // <= 10 digits: pandas' conversion is the exact one there
      }
      if (i != 39) body += i % 2 ? "\r\n" : "\n";
    }
    prmf_text* t = nullptr;
    CK(prmf_text_create(&t, 0, (int64_t)body.size()));
    for (size_t off = 0; off < body.size(); off += 97) CK(prmf_text_append(t, body.data() + off, (int64_t)std::min<size_t>(97, body.size() - off)));
    int64_t rows = 0, bad = 0;
    int flags = -1;
    CK(prmf_text_index(t, ';', 26, &rows, &flags));
    REQUIRE(rows == 40 && flags == 0);
    std::vector<double> out((size_t)(40 * 25));
    std::vector<int64_t> spans(80);
    std::vector<int32_t> is_int(25);
    CK(prmf_text_parse(t, 1, out.data(), spans.data(), is_int.data(), &flags, &bad));
    REQUIRE(flags == 0 && bad == -1);
    for (size_t i = 0; i < out.size(); ++i) REQUIRE(out[i] == want[i]);
    REQUIRE(body.substr((size_t)spans[2 * 17], (size_t)(spans[2 * 17 + 1] - spans[2 * 17])) == "row 17");
    for (int j = 0; j < 25; ++j) REQUIRE(is_int[(size_t)j] == 0);
    CK(prmf_text_destroy(t));
    CK(prmf_text_create(&t, 0, 12));                              // a ragged body is reported, not parsed
    CK(prmf_text_append(t, "1;2\n3;4;5\n6\n", 12));
    CK(prmf_text_index(t, ';', 2, &rows, &flags));
    REQUIRE(flags == PRMF_TEXT_RAGGED && rows == 0);
    CK(prmf_text_destroy(t));
  }
  {
    // sum of squares, NNLS against a device matrix (both orientations; checked through its optimality conditions), CSV writer
    prmf_ctx* ctx = nullptr;
    const int64_t m = 30, n = 50;
    std::vector<double> A((size_t)(m * n));
    std::vector<float> Af(A.size());
    double want_ss = 0.0;
    for (size_t i = 0; i < A.size(); ++i) {
      Af[i] = (float)lcg();
      A[i] = (double)Af[i];
      want_ss += A[i] * A[i];
    }
    double* dA = nullptr;
    float* dAf = nullptr;
    HK(hipMalloc(&dA, A.size() * 8));
    HK(hipMalloc(&dAf, A.size() * 4));
    HK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(dAf, Af.data(), A.size() * 4, hipMemcpyHostToDevice));
    double ss = -1.0;
    CK(prmf_sum_squares(0, dAf, PRMF_X_F32, m, n, n, &ss, nullptr));
    REQUIRE(std::fabs(ss - want_ss) <= 1e-12 * want_ss);
    for (int transposed = 0; transposed < 2; ++transposed) {
      const int64_t eq = transposed ? n : m, un = transposed ? m : n;
      std::vector<double> b((size_t)eq), x((size_t)un, -1.0);
      for (auto& v : b) v = lcg();
      double rnorm = -1.0;
      int iters = -1;
      CK(prmf_nnls(0, dA, m, n, n, transposed, b.data(), x.data(), &rnorm, &iters, 0, nullptr));
      REQUIRE(iters > 0 && rnorm >= 0.0);
      std::vector<double> r(b);                      // r = b - A x, then the gradient A^T r: <= 0 where x = 0, = 0 where x > 0
      for (int64_t t = 0; t < eq; ++t)
        for (int64_t j = 0; j < un; ++j) r[(size_t)t] -= (transposed ? A[(size_t)(j * n + t)] : A[(size_t)(t * n + j)]) * x[(size_t)j];
      double rr = 0.0;
      for (double v : r) rr += v * v;
      REQUIRE(std::fabs(std::sqrt(rr) - rnorm) <= 1e-10);
      for (int64_t j = 0; j < un; ++j) {
        double g = 0.0;
        for (int64_t t = 0; t < eq; ++t) g += (transposed ? A[(size_t)(j * n + t)] : A[(size_t)(t * n + j)]) * r[(size_t)t];
        REQUIRE(x[(size_t)j] >= 0.0);
        if (x[(size_t)j] > 0.0) REQUIRE(std::fabs(g) <= 1e-9); else REQUIRE(g <= 1e-9);
      }
    }
    REQUIRE(prmf_nnls(0, dA, m, n, n - 1, 0, A.data(), A.data(), nullptr, nullptr, 0, nullptr) == PRMF_ERR_ARG);
    char path[] = "/tmp/prmf_capi_wide_XXXXXX";
    const int fd = mkstemp(path);
    REQUIRE(fd >= 0);
    close(fd);
    const char prefixes[] = "\"a\",\"b\",";
    const int64_t offs[3] = {0, 4, 8};
    const double vals[4] = {0.1, 1e16, -0.0, 5e-324};
    CK(prmf_write_csv(path, "\"\",\"LV0\",\"LV1\"\n", prefixes, offs, vals, 2, 2, 2, ',', 1));
    std::FILE* fh = std::fopen(path, "rb");
    REQUIRE(fh != nullptr);
    char text[256];
    const size_t got = std::fread(text, 1, sizeof text - 1, fh);
    std::fclose(fh);
    unlink(path);
    text[got] = 0;
    REQUIRE(std::string(text) == "\"\",\"LV0\",\"LV1\"\n\"a\",0.1,1e+16\n\"b\",-0.0,5e-324\n");
    (void)hipFree(dA);
    (void)hipFree(dAf);
  }
  REQUIRE(prmf_release_staging() == PRMF_OK);       // the staging buffers prmf_set_x_host keeps between calls
  REQUIRE(prmf_release_staging() == PRMF_OK);       // (idempotent)
  std::printf("wide OK\n");
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
