// Does a captured hipGraph shorten the gap between DEPENDENT kernels of one stream?  (the iteration is 4 dependent
// launches; rocprofv3 shows 3.5-3.8 us between them: 13% of the 5000 x 10000 step)
//   chain of N kernels, each 512 x 256 threads busy for ~T us, (a) launched into a stream, (b) replayed from a graph
// hipcc --offload-arch=gfx950 -O3 -o /tmp/gg tools/graph_gap.hip && /tmp/gg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) busy(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffull) out[0] = 1;
}

int main() {
  unsigned* out; CHECK(hipMalloc(&out, 4));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int N = 40;
  for (int us : {0, 10, 30}) {
    const unsigned long long ticks = 100ull * us;   // wall_clock64: 100 MHz
    auto chain = [&]() { for (int i = 0; i < N; ++i) busy<<<512, 256, 0, s>>>(ticks, out); };
    for (int w = 0; w < 3; ++w) chain();
    CHECK(hipStreamSynchronize(s));
    float best_s = 1e9f, best_g = 1e9f, ms;
    for (int r = 0; r < 10; ++r) {
      CHECK(hipEventRecord(a, s)); chain(); CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best_s) best_s = ms;
    }
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    for (int r = 0; r < 10; ++r) {
      CHECK(hipEventRecord(a, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best_g) best_g = ms;
    }
    printf("kernel ~%2d us x %d dependent launches: stream %.1f us (%.2f us per launch beyond the kernel), graph %.1f us (%.2f)\n", us, N,
           best_s * 1e3, best_s * 1e3 / N - us, best_g * 1e3, best_g * 1e3 / N - us);
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  return 0;
}
