// Host side of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer ("host ASAN/UBSAN on the
// C-ABI layer"; GPU sanitizers are not available on this pool).  Linked against a HOST-ONLY build of the library
// (prmf_amd._build.build_sanitized: the host code of every translation unit instrumented, device code as always); it
// runs without a GPU -- tests/test_sanitize_cpu.py builds both and runs this.  What it drives is the host arithmetic
// that decides how kernels are launched: the sweep planner (balanced decomposition, slab masks: 64-bit shifts, vectors
// sized by the plan) over a wide random range of shapes, and the argument / error paths that run before any device work.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include <unistd.h>

#include "prmf_hip.h"
#include "prmf_hip_debug.h"

static int failures = 0;
#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      std::printf("FAILED %s: ", #cond);         \
      std::printf(__VA_ARGS__);                  \
      std::printf("\n");                         \
      ++failures;                                \
    }                                            \
  } while (0)

int main(int argc, char** argv) {
  const long cases = argc > 1 ? std::atol(argv[1]) : 200000;
  std::mt19937_64 rng(20261005);
  auto uni = [&](long long lo, long long hi) { return (long long)(lo + (long long)(rng() % (unsigned long long)(hi - lo + 1))); };
  long long checked = 0;
  for (long i = 0; i < cases; ++i) {
    const int xd = (int)(rng() & 1);
    // log-uniform sizes: 1 .. 2^19 rows / depth, every k, CU counts of whole and harvested parts
    const long long rows = uni(1, 1LL << uni(0, 19)), depth = uni(1, 1LL << uni(0, 19));
    const int k = (int)uni(1, 128);
    const int cus_list[] = {1, 7, 8, 64, 104, 228, 256, 304};
    const int n_cus = cus_list[rng() % 8];
    int out[8];
    std::memset(out, 0xff, sizeof out);
    const int rc = prmf_debug_plan_sweep(xd, rows, depth, k, n_cus, out);
    CHECK(rc == PRMF_OK, "rc %d for x_dtype %d rows %lld depth %lld k %d cus %d", rc, xd, rows, depth, k, n_cus);
    if (rc != PRMF_OK) continue;
    const int waves = out[0], tiles = out[1], panel_rows = out[2], grid = out[3], slices = out[4], slabs = out[5], panels = out[7];
    CHECK((waves == 4 || waves == 8) && (tiles == 1 || tiles == 2), "waves %d tiles %d", waves, tiles);
    CHECK(panels == (int)((rows + panel_rows - 1) / panel_rows), "panels %d for %lld rows of %d", panels, rows, panel_rows);
    CHECK(slabs >= 1 && slabs <= 64, "slabs per panel %d (64-bit panel mask)", slabs);
    const int NT = (k + 31) / 32;
    const long long kpad = (depth + 127) / 128 * 128;
    const bool short_reduction = kpad * NT <= 512 || (NT == 1 && kpad <= 1024 && rows <= 4096);
    CHECK(((out[6] & 4) != 0) == short_reduction, "exact-product flag %d for depth %lld (padded %lld) x %d column tiles, %lld rows", out[6] & 4, depth, kpad, NT, rows);
    if (out[6] & 4) {
      // the exact short-reduction product (round 6): one wave per (32-row tile, 32-column tile, reduction slice), 256-row panels,
      // at most 8 slices of at least two 32-element chunks, one slab per slice
      const long long pairs = (long long)panels * 8 * NT, chunks = kpad / 32;
      CHECK(panel_rows == 256 && waves == 4 && tiles == 1 && slices == 1, "exact plan shape %d %d %d %d", panel_rows, waves, tiles, slices);
      CHECK(slabs <= 8 && (slabs == 1 || 2LL * slabs <= chunks), "exact plan: %d slices of %lld chunks", slabs, chunks);
      CHECK(grid == (int)((pairs * slabs + 3) / 4), "exact plan: grid %d for %lld pairs x %d slices", grid, pairs, slabs);
      ++checked;
      continue;
    }
    CHECK(panel_rows == waves * tiles * 32, "panel rows %d", panel_rows);
    CHECK(grid >= 1 && grid <= 2 * n_cus + 8, "grid %d for %d CUs", grid, n_cus);
    CHECK(slices == 1 || slices == 2 || slices == 4 || slices == 8, "slices %d", slices);
    ++checked;
  }
  // argument and error paths that end before any device work (or at the first HIP call when there is no GPU)
  prmf_ctx* ctx = nullptr;
  CHECK(prmf_create(nullptr, 0, 10, 10, 4, 0, 1, 0, 0) == PRMF_ERR_ARG, "null out");
  CHECK(prmf_create(&ctx, 0, 10, 0, 4, 0, 1, 0, 0) == PRMF_ERR_ARG && ctx == nullptr, "n = 0");
  CHECK(std::strstr(prmf_last_error(nullptr), "bad shape") != nullptr, "message: %s", prmf_last_error(nullptr));
  CHECK(prmf_create(&ctx, 0, -1, 10, 4, 0, 1, 0, 0) == PRMF_ERR_ARG, "m < 0");
  CHECK(prmf_create(&ctx, 0, 10, 10, 4000, 0, 1, 0, 0) == PRMF_ERR_UNSUPPORTED, "k too large");
  CHECK(prmf_create(&ctx, 0, 10, 10, 4, 7, 1, 0, 0) == PRMF_ERR_ARG, "x_dtype");
  CHECK(prmf_create(&ctx, 0, 10, 10, 4, 0, 2, 2, 0) == PRMF_ERR_ARG, "rank >= nranks");
  CHECK(prmf_create(&ctx, 0, 10, 10, 4, 0, 0, 0, 0) == PRMF_ERR_ARG, "nranks < 1");
  {
    const int rc = prmf_create(&ctx, 0, 64, 300, 6, 0, 1, 0, 0);      // a GPU box creates it; without one the first HIP call fails loudly
    CHECK(rc == PRMF_OK || (rc == PRMF_ERR_HIP && ctx == nullptr), "create rc %d", rc);
    if (rc == PRMF_OK) CHECK(prmf_destroy(ctx) == PRMF_OK, "destroy");
    ctx = nullptr;
  }
  CHECK(prmf_destroy(nullptr) == PRMF_OK, "destroy(NULL)");
  CHECK(prmf_step(nullptr, nullptr, nullptr, 1, 1.0, 1.0, -1.0, nullptr, nullptr, nullptr) == PRMF_ERR_ARG, "step(NULL)");
  CHECK(prmf_step_dist(nullptr, nullptr, nullptr, 1, 1.0, 1.0, -1.0, nullptr, nullptr, nullptr) == PRMF_ERR_ARG, "step_dist(NULL)");
  CHECK(prmf_set_comm_timeout(nullptr, 1.0) == PRMF_ERR_ARG && prmf_set_global_rows(nullptr, 1) == PRMF_ERR_ARG, "NULL ctx setters");
  CHECK(prmf_comm_init(nullptr, nullptr) == PRMF_ERR_ARG, "comm_init(NULL)");
  {
    int w = 0, ph = 0, mx = 0;
    const int n = prmf_debug_stamp_layout(&w, &ph, &mx);
    CHECK(n > ph && ph > w && w >= 2 * mx, "stamp layout %d %d %d %d", n, w, ph, mx);
    CHECK(prmf_debug_set_stamps((void*)0x1000, (size_t)n - 1) == PRMF_ERR_ARG, "short stamp buffer");
    CHECK(prmf_debug_set_stamps((void*)0x1000, (size_t)n) == 0 && prmf_debug_set_stamps(nullptr, 0) == 0, "stamp buffer");
  }
  {
    // the CSV writer is host code end to end: 2000 x 7 values of every magnitude through the threaded path and the single one,
    // row prefixes, NaN / infinities / signed zeros / subnormals -- read back and compared with strtod
    const int64_t rows = 2000, cols = 7;
    std::vector<double> v((size_t)(rows * cols));
    for (size_t i = 0; i < v.size(); ++i) {
      uint64_t bits = rng();
      double d;
      std::memcpy(&d, &bits, 8);
      if (i % 5 == 0) d = (double)(int64_t)(rng() % 1000000) / 997.0;
      v[i] = d;
    }
    v[0] = 0.0; v[1] = -0.0; v[2] = 5e-324; v[3] = 1.7976931348623157e308; v[4] = 1e16; v[5] = 1e-5; v[6] = 123456789012345680.0;
    std::string prefixes;
    std::vector<int64_t> off((size_t)rows + 1, 0);
    for (int64_t r = 0; r < rows; ++r) {
      prefixes += "\"r" + std::to_string(r) + "\",";
      off[(size_t)r + 1] = (int64_t)prefixes.size();
    }
    char path[] = "/tmp/prmf_sanitize_XXXXXX";
    const int fd = mkstemp(path);
    CHECK(fd >= 0, "mkstemp");
    if (fd >= 0) {
      close(fd);
      for (int threads : {1, 4, 0}) {
        CHECK(prmf_write_csv(path, "\"\",\"a\"\n", prefixes.data(), off.data(), v.data(), rows, cols, cols, ',', threads) == PRMF_OK, "write_csv");
        std::FILE* fh = std::fopen(path, "rb");
        CHECK(fh != nullptr, "reopen");
        if (!fh) break;
        std::string text;
        char buf[65536];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, fh)) > 0) text.append(buf, got);
        std::fclose(fh);
        size_t pos = text.find('\n') + 1;
        long bad = 0;
        for (int64_t r = 0; r < rows && !bad; ++r) {
          pos = text.find(',', pos) + 1;                       // past the row label
          for (int64_t c = 0; c < cols; ++c) {
            const size_t end = text.find_first_of(",\n", pos);
            const std::string cell = text.substr(pos, end - pos);
            const double want = v[(size_t)(r * cols + c)];
            if (want != want) bad += cell != "\"\"";
            else bad += std::strtod(cell.c_str(), nullptr) != want || std::signbit(std::strtod(cell.c_str(), nullptr)) != std::signbit(want);
            pos = end + 1;
          }
        }
        CHECK(bad == 0 && pos == text.size(), "%ld cells read back differently (threads %d)", bad, threads);
      }
      CHECK(prmf_write_csv(path, nullptr, nullptr, nullptr, v.data(), 3, 2, 2, ';', 1) == PRMF_OK, "no header, no prefixes");
      unlink(path);
    }
    CHECK(prmf_write_csv("/nonexistent-dir/x.csv", nullptr, nullptr, nullptr, v.data(), 1, 1, 1, ',', 1) == PRMF_ERR_IO, "unwritable path");
    CHECK(prmf_write_csv(path, nullptr, prefixes.data(), nullptr, v.data(), 1, 1, 1, ',', 1) == PRMF_ERR_ARG, "prefixes without offsets");
    CHECK(prmf_write_csv(nullptr, nullptr, nullptr, nullptr, v.data(), 1, 1, 1, ',', 1) == PRMF_ERR_ARG, "NULL path");
  }
  {
    // argument checks of the device-side entry points that end before any device work
    prmf_text* t = nullptr;
    int64_t rows = 0;
    int flags = 0;
    double out = 0.0, x[4] = {0, 0, 0, 0};
    CHECK(prmf_text_create(nullptr, 0, 10) == PRMF_ERR_ARG && prmf_text_create(&t, 0, -5) == PRMF_ERR_ARG && t == nullptr, "text_create");
    CHECK(prmf_text_append(nullptr, "1", 1) == PRMF_ERR_ARG && prmf_text_index(nullptr, ',', 2, &rows, &flags) == PRMF_ERR_ARG, "text NULL handle");
    CHECK(prmf_text_parse(nullptr, 0, nullptr, nullptr, nullptr, &flags, nullptr) == PRMF_ERR_ARG && prmf_text_destroy(nullptr) == PRMF_OK, "text parse / destroy NULL");
    CHECK(prmf_sum_squares(0, nullptr, PRMF_X_F32, 2, 2, 2, &out, nullptr) == PRMF_ERR_ARG && prmf_sum_squares(0, nullptr, 9, 0, 0, 0, &out, nullptr) == PRMF_ERR_ARG, "sum_squares");
    CHECK(prmf_sum_squares(0, nullptr, PRMF_X_BF16, 0, 5, 5, &out, nullptr) == PRMF_OK && out == 0.0, "sum_squares of nothing");
    CHECK(prmf_nnls(0, nullptr, 2, 2, 2, 0, x, x, nullptr, nullptr, 0, nullptr) == PRMF_ERR_ARG, "nnls NULL matrix");
    CHECK(prmf_nnls(0, x, 2, 2, 1, 0, x, x, nullptr, nullptr, 0, nullptr) == PRMF_ERR_ARG && prmf_nnls(0, x, 0, 2, 2, 0, x, x, nullptr, nullptr, 0, nullptr) == PRMF_ERR_ARG, "nnls shapes");
  }
  CHECK(prmf_abi_version() >= 2, "abi version %d", prmf_abi_version());
  std::printf("%lld plans checked, %d failures\n", checked, failures);
  return failures ? 1 : 0;
}
