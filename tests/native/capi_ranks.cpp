// The in-library multi-rank step from plain C++: N host threads, one context each (rank r of N on the one GPU), collectives
// through the RCCL test double (PRMF_RCCL_LIB=libmock_rccl.so), both V-update modes, ragged shards, ranks without a row or without a latent column.  Checks: every rank
// returns the same records and the same V bit for bit, U / V / objective within 1e-5 of a single-rank prmf_step on the
// whole matrix.  tests/test_gpu_native.py builds and runs it.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "prmf_hip.h"

static uint32_t lcg_state = 4242u;
static double lcg() {
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return ((lcg_state >> 8) + 1) / 16777216.0;
}

struct Problem {
  int64_t m, n;
  int k, P, S, steps;
  std::vector<float> X, U0, V0;
  double nsq;
  std::vector<int64_t> sup_ptr, row_ptr;
  std::vector<int32_t> sup_idx, col_local, active;
  std::vector<double> sup_deg, sup_ldiag, w;
};

static Problem make(int64_t m, int64_t n, int k, int P, int S, int steps) {
  Problem p;
  p.m = m; p.n = n; p.k = k; p.P = P; p.S = S; p.steps = steps;
  p.X.resize((size_t)(m * n)); p.U0.resize((size_t)(m * k)); p.V0.resize((size_t)(n * k));
  p.nsq = 0;
  for (auto& x : p.X) { x = (float)lcg(); p.nsq += (double)x * x; }
  for (auto& u : p.U0) u = (float)(3.0 * lcg());
  for (auto& v : p.V0) v = (float)(3.0 * lcg());
  p.sup_ptr.resize(P + 1);
  p.row_ptr.push_back(0);
  const int64_t stride = n / P > S ? n / P : S;
  for (int q = 0; q <= P; ++q) p.sup_ptr[q] = (int64_t)q * S;
  for (int q = 0; q < P; ++q)
    for (int s = 0; s < S; ++s) {
      p.sup_idx.push_back((int32_t)(((int64_t)q * stride + s) % n));
      double deg = 0;
      if (s > 0) { p.col_local.push_back(s - 1); p.w.push_back(1.0); deg += 1; }
      if (s + 1 < S) { p.col_local.push_back(s + 1); p.w.push_back(1.0); deg += 1; }
      p.sup_deg.push_back(deg);
      p.sup_ldiag.push_back(deg);
      p.row_ptr.push_back((int64_t)p.col_local.size());
    }
  for (int i = 0; i < k; ++i) p.active.push_back(i % P);
  return p;
}

struct Result {
  int rc = -1;
  std::vector<float> U, V;
  std::vector<double> obj;
  char err[512] = {0};
};

// rows [lo, hi) of rank r of N: the library's shard rule (prmf_amd.synthetic.shard_rows): ceil-sized leading shards
static void shard(int64_t m, int N, int r, int64_t* lo, int64_t* hi) {
  const int64_t base = m / N, extra = m % N;
  *lo = r * base + (r < extra ? r : extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}

static void run_rank(const Problem& p, int N, int r, unsigned flags, const unsigned char* id, Result* out) {
  prmf_ctx* ctx = nullptr;
  auto fail = [&](const char* what, int rc) {
    std::snprintf(out->err, sizeof out->err, "rank %d: %s -> %d: %s", r, what, rc, prmf_last_error(ctx));
    out->rc = rc ? rc : -1;
  };
  if (hipSetDevice(0) != hipSuccess) return fail("hipSetDevice", -1);
  int64_t lo, hi;
  shard(p.m, N, r, &lo, &hi);
  const int64_t ml = hi - lo;
  int rc;
  if ((rc = prmf_create(&ctx, 0, ml, p.n, p.k, PRMF_X_F32, N, r, flags))) return fail("prmf_create", rc);
  float *dX = nullptr, *dU = nullptr, *dV = nullptr;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return fail("stream", -1);
  if (hipMalloc(&dX, (size_t)(ml > 0 ? ml : 1) * p.n * 4) != hipSuccess || hipMalloc(&dU, (size_t)(ml > 0 ? ml : 1) * p.k * 4) != hipSuccess ||
      hipMalloc(&dV, (size_t)p.n * p.k * 4) != hipSuccess) return fail("hipMalloc", -1);
  if (ml > 0) {
    (void)hipMemcpy(dX, p.X.data() + lo * p.n, (size_t)ml * p.n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dU, p.U0.data() + lo * p.k, (size_t)ml * p.k * 4, hipMemcpyHostToDevice);
  }
  (void)hipMemcpy(dV, p.V0.data(), (size_t)p.n * p.k * 4, hipMemcpyHostToDevice);
  if ((rc = prmf_set_x(ctx, dX, p.n, p.nsq, s))) return fail("prmf_set_x", rc);
  if ((rc = prmf_set_pathway_library(ctx, p.P, p.sup_ptr.data(), p.sup_idx.data(), p.sup_deg.data(), p.sup_ldiag.data(), p.row_ptr.data(),
                                     p.col_local.data(), p.w.data(), s))) return fail("library", rc);
  if ((rc = prmf_set_active(ctx, p.active.data(), s))) return fail("active", rc);
  if (N > 1 && (rc = prmf_comm_init(ctx, id))) return fail("prmf_comm_init", rc);
  out->obj.assign((size_t)p.steps * PRMF_OBJ_STRIDE, 0.0);
  double gd[2];
  const double gamma = std::sqrt(p.nsq) / p.k, delta = 10.0 / std::sqrt(p.nsq);
  rc = N > 1 ? prmf_step_dist(ctx, dU, dV, p.steps, gamma, delta, -1.0, out->obj.data(), gd, s)
             : prmf_step(ctx, dU, dV, p.steps, gamma, delta, -1.0, out->obj.data(), gd, s);
  if (rc) return fail("step", rc);
  out->U.resize((size_t)ml * p.k);
  out->V.resize((size_t)p.n * p.k);
  if (ml > 0) (void)hipMemcpy(out->U.data(), dU, out->U.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(out->V.data(), dV, out->V.size() * 4, hipMemcpyDeviceToHost);
  if ((rc = prmf_destroy(ctx))) { ctx = nullptr; return fail("destroy", rc); }
  (void)hipFree(dX); (void)hipFree(dU); (void)hipFree(dV);
  (void)hipStreamDestroy(s);
  out->rc = 0;
}

static double rel(const std::vector<float>& a, const std::vector<float>& b) {
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { const double d = (double)a[i] - b[i]; num += d * d; den += (double)b[i] * b[i]; }
  return std::sqrt(num / (den > 0 ? den : 1));
}

static int run_case(int64_t m, int64_t n, int k, int N, unsigned flags, int steps) {
  const Problem p = make(m, n, k, 3, 10, steps);
  Result one;
  run_rank(p, 1, 0, 0, nullptr, &one);
  if (one.rc) { std::fprintf(stderr, "single rank: %s\n", one.err); return 2; }
  unsigned char id[PRMF_COMM_ID_BYTES];
  if (prmf_comm_unique_id(id) != PRMF_OK) { std::fprintf(stderr, "prmf_comm_unique_id: %s\n", prmf_last_error(nullptr)); return 3; }
  std::vector<Result> res((size_t)N);
  std::vector<std::thread> th;
  for (int r = 0; r < N; ++r) th.emplace_back(run_rank, std::cref(p), N, r, flags, id, &res[(size_t)r]);
  for (auto& t : th) t.join();
  std::vector<float> U;
  for (int r = 0; r < N; ++r) {
    if (res[(size_t)r].rc) { std::fprintf(stderr, "%s\n", res[(size_t)r].err); return 4; }
    U.insert(U.end(), res[(size_t)r].U.begin(), res[(size_t)r].U.end());
    if (res[(size_t)r].V != res[0].V || res[(size_t)r].obj != res[0].obj) { std::fprintf(stderr, "rank %d: replicated state differs from rank 0\n", r); return 5; }
  }
  const double eu = rel(U, one.U), ev = rel(res[0].V, one.V);
  double eo = 0;
  for (int i = 0; i < steps; ++i) {
    const double a = res[0].obj[(size_t)i * PRMF_OBJ_STRIDE + PRMF_OBJ_OBJ], b = one.obj[(size_t)i * PRMF_OBJ_STRIDE + PRMF_OBJ_OBJ];
    eo = std::fmax(eo, std::fabs(a - b) / std::fabs(b));
  }
  std::printf("ranks %d (%s) %lld x %lld k=%d: U %.1e V %.1e obj %.1e\n", N, (flags & PRMF_FLAG_REPLICATED) ? "replicated" : "colshard", (long long)m,
              (long long)n, k, eu, ev, eo);
  if (!(eu < 1e-5 && ev < 1e-5 && eo < 1e-5)) { std::fprintf(stderr, "beyond 1e-5\n"); return 6; }
  return 0;
}

int main() {
  int rc;
  if ((rc = run_case(70, 150, 4, 2, 0, 3))) return rc;
  if ((rc = run_case(333, 900, 16, 3, PRMF_FLAG_REPLICATED, 2))) return rc;
  if ((rc = run_case(5, 400, 6, 8, 0, 2))) return rc;                       // ranks without a row, ranks without a column
  if ((rc = run_case(1200, 700, 65, 5, 0, 2))) return rc;                   // k > 64: column-split epilogue, own-column update on 13-column blocks
  if ((rc = run_case(801, 2100, 48, 8, PRMF_FLAG_REPLICATED, 2))) return rc;
  std::printf("ranks OK\n");
  std::fflush(stdout);
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
  _exit(0);      // see capi_native.cpp: AMD's ASan runtime aborts inside the HSA runtime's exit handlers
#endif
#endif
  return 0;
}
