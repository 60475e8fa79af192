// Host build of prmf_amd/csrc/xstrtod.h (the very function the GPU parse kernel calls) for tests/test_textio_cpu.py:
// fields [off[i], off[i + 1] - 1) of `text` -> values and verdict bits (1 ok, 2 is_int, 4 wide_int).
//   g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off -I prmf_amd/csrc tests/native/xstrtod_host.cpp -o libxstrtod_host.so
#include "xstrtod.h"

extern "C" int parse_many(const char* text, const long long* off, long long n, double* out, int* verdict) {
  for (long long i = 0; i < n; ++i) {
    const prmf::ParsedField r =
        prmf::parse_decimal(reinterpret_cast<const unsigned char*>(text) + off[i], reinterpret_cast<const unsigned char*>(text) + off[i + 1] - 1);
    out[i] = r.value;
    verdict[i] = (r.ok ? 1 : 0) | (r.is_int ? 2 : 0) | (r.wide_int ? 4 : 0);
  }
  return 0;
}
