// The numeric body of a delimited text matrix, parsed on the GPU -- what `pd.read_csv(args.data, sep=delimiter, ...)` does
// for the reference's command line (script/prmf_runner.py:942).  At the reference's typical size (500 samples x 20000
// genes, 187 MB of text) the pandas parse is 1.7 s of a 2.0 s run whose solve takes 0.08 s (profiles/r05_cli_end_to_end.txt);
// it is byte work with no reuse, so it belongs where the bandwidth is:
//
//   1. count       every workgroup takes 4 KiB of text (16 B per lane, one dwordx4 load) and counts its separators
//                  (the delimiter and '\n');
//   2. scan        exclusive sum of the per-workgroup counts (hipcub::DeviceScan, a library primitive: n / 4096 values);
//   3. positions   the same 4 KiB again: an in-workgroup scan gives every separator its global number j, its byte offset is
//                  stored, and the line structure is checked on the way -- separator j ends a line exactly when
//                  (j + 1) mod fields_per_line = 0, so every line has the same number of fields or a flag is raised;
//   4. parse       one lane per field: [sep_pos[j-1] + 1, sep_pos[j]) through parse_decimal (xstrtod.h: pandas' own sequence
//                  of float64 operations), value stored at out[row][column]; the first field of a line can be skipped
//                  (row names: their byte spans go back to the host, which lets pandas itself read them).
//
// Traffic: the text twice (count, positions) + once through the cache (parse), 8 B per field of offsets written and read,
// 8 B per field of output: ~4.5 x the text size, ~1 ms per 187 MB at HBM speed; the measured cost is the host's read of the
// file and the two PCIe copies.  Nothing here is approximate and nothing is guessed: a field outside the plain decimal
// grammar, a quote outside the first column, a ragged line raise a flag and the caller gives the file to pandas.
#include <hipcub/hipcub.hpp>

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "xstrtod.h"

namespace prmf {

namespace {

constexpr int TEXT_TPB = 256;                       // lanes per workgroup
constexpr int TEXT_BPL = 16;                        // bytes per lane
constexpr int TEXT_TILE = TEXT_TPB * TEXT_BPL;      // 4 KiB per workgroup

// bit i of *sep: byte base + i is a separator; of *nl: it is '\n'
__device__ __forceinline__ void classify16(const unsigned char* __restrict__ text, int64_t base, int64_t L, unsigned char delim,
                                           unsigned* sep, unsigned* nl, unsigned* cr = nullptr) {
  unsigned s = 0, n = 0, r = 0;
  if (base + TEXT_BPL <= L) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + base);        // base is a multiple of 16, text is hipMalloc'ed
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < TEXT_BPL; ++i) {
      const unsigned char c = (unsigned char)(w[i >> 2] >> ((i & 3) * 8));
      n |= (unsigned)(c == '\n') << i;
      r |= (unsigned)(c == '\r') << i;
      s |= (unsigned)(c == '\n' || c == delim) << i;
    }
  } else {
    for (int i = 0; i < TEXT_BPL && base + i < L; ++i) {
      const unsigned char c = text[base + i];
      n |= (unsigned)(c == '\n') << i;
      r |= (unsigned)(c == '\r') << i;
      s |= (unsigned)(c == '\n' || c == delim) << i;
    }
  }
  *sep = s;
  *nl = n;
  if (cr) *cr = r;
}

__global__ void __launch_bounds__(TEXT_TPB)
text_count_kernel(const unsigned char* __restrict__ text, int64_t L, unsigned char delim, int64_t* __restrict__ counts) {
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const int64_t base = ((int64_t)blockIdx.x * TEXT_TPB + threadIdx.x) * TEXT_BPL;
  unsigned sep = 0, nl = 0;
  if (base < L) classify16(text, base, L, delim, &sep, &nl);
  int c = __popc(sep);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(TEXT_TPB)
text_positions_kernel(const unsigned char* __restrict__ text, int64_t L, unsigned char delim, const int64_t* __restrict__ block_base,
                      int64_t fields_per_line, int64_t* __restrict__ sep_pos, int* __restrict__ flags) {
  __shared__ int wave_total[TEXT_TPB / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = ((int64_t)blockIdx.x * TEXT_TPB + tid) * TEXT_BPL;
  unsigned sep = 0, nl = 0, cr = 0;
  if (base < L) classify16(text, base, L, delim, &sep, &nl, &cr);
  // a carriage return that is not followed by a line feed ends a line for pandas: such a file is not read here
  if (cr) {
    unsigned followed = nl >> 1;                       // bit i: byte i + 1 is '\n'
    if ((cr >> 15) & 1u) followed |= (unsigned)(base + TEXT_BPL < L && text[base + TEXT_BPL] == '\n') << 15;
    if (cr & ~followed) atomicOr(flags, TEXT_FLAG_BARE_CR);
  }
  const int mine = __popc(sep);
  int incl = mine;                                   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  int before = incl - mine;
  for (int w = 0; w < wave; ++w) before += wave_total[w];
  if (!mine) return;
  int64_t j = block_base[blockIdx.x] + before;       // global number of this lane's first separator
  int64_t phase = (j + 1) % fields_per_line;         // (j + 1) mod F, carried along instead of divided again
  bool ragged = false;
  while (sep) {
    const int i = __ffs(sep) - 1;
    sep &= sep - 1;
    sep_pos[j] = base + i;
    ragged |= (((nl >> i) & 1u) != 0) != (phase == 0);
    ++j;
    if (++phase == fields_per_line) phase = 0;
  }
  if (ragged) atomicOr(flags, TEXT_FLAG_RAGGED);
}

__global__ void __launch_bounds__(TEXT_TPB)
text_parse_kernel(const unsigned char* __restrict__ text, const int64_t* __restrict__ sep_pos, int64_t n_fields, int64_t fields_per_line,
                  int skip_first, double* __restrict__ out, int64_t* __restrict__ name_spans, int* __restrict__ col_not_int,
                  int* __restrict__ flags, unsigned long long* __restrict__ first_bad) {
  const int64_t j = (int64_t)blockIdx.x * TEXT_TPB + threadIdx.x;
  if (j >= n_fields) return;
  const int64_t start = j ? sep_pos[j - 1] + 1 : 0;
  int64_t end = sep_pos[j];
  const int64_t row = j / fields_per_line, col = j - row * fields_per_line;
  if (col == fields_per_line - 1 && end > start && text[end - 1] == '\r') --end;      // "\r\n" line ends
  if (skip_first && col == 0) {
    name_spans[2 * row] = start;
    name_spans[2 * row + 1] = end;
    return;
  }
  // no number is 4 KiB long: such a field (a file with the wrong delimiter is ONE field per line) is refused unread
  const ParsedField r = end - start <= 4096 ? parse_decimal(text + start, text + end) : ParsedField{0.0, false, false, false};
  if (!r.ok || r.wide_int) {
    int why = r.ok ? TEXT_FLAG_WIDE_INT : TEXT_FLAG_NOT_NUMERIC;
    if (!r.ok)                                           // (a look at its first 4 KiB: a refused field can be a whole malformed file)
      for (int64_t p = start; p < end && p < start + 4096; ++p)
        if (text[p] == '"') why = TEXT_FLAG_QUOTED;
    atomicOr(flags, why);
    atomicMin(first_bad, (unsigned long long)start);
    return;
  }
  const int64_t width = fields_per_line - skip_first, c = col - skip_first;
  if (!r.is_int && !col_not_int[c]) col_not_int[c] = 1;
  out[row * width + c] = r.value;
}

}  // namespace

int64_t text_blocks(int64_t nbytes) { return (nbytes + TEXT_TILE - 1) / TEXT_TILE; }

hipError_t text_count(const unsigned char* text, int64_t nbytes, unsigned char delim, int64_t* counts, hipStream_t s) {
  const int64_t nb = text_blocks(nbytes);
  if (nb == 0) return hipSuccess;
  text_count_kernel<<<dim3((unsigned)nb), TEXT_TPB, 0, s>>>(text, nbytes, delim, counts);
  return hipGetLastError();
}

hipError_t text_scan(const int64_t* counts, int64_t* base, int64_t n_values, void* tmp, size_t* tmp_bytes, hipStream_t s) {
  return hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, counts, base, (int)n_values, s);
}

hipError_t text_positions(const unsigned char* text, int64_t nbytes, unsigned char delim, const int64_t* block_base,
                          int64_t fields_per_line, int64_t* sep_pos, int* flags, hipStream_t s) {
  const int64_t nb = text_blocks(nbytes);
  if (nb == 0) return hipSuccess;
  text_positions_kernel<<<dim3((unsigned)nb), TEXT_TPB, 0, s>>>(text, nbytes, delim, block_base, fields_per_line, sep_pos, flags);
  return hipGetLastError();
}

hipError_t text_parse(const unsigned char* text, const int64_t* sep_pos, int64_t n_fields, int64_t fields_per_line, int skip_first,
                      double* out, int64_t* name_spans, int* col_not_int, int* flags, unsigned long long* first_bad, hipStream_t s) {
  if (n_fields == 0) return hipSuccess;
  const int64_t nb = (n_fields + TEXT_TPB - 1) / TEXT_TPB;
  text_parse_kernel<<<dim3((unsigned)nb), TEXT_TPB, 0, s>>>(text, sep_pos, n_fields, fields_per_line, skip_first, out, name_spans,
                                                             col_not_int, flags, first_bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// float64 matrix -> delimited text, the way `DataFrame.to_csv(path, sep=",", quoting=csv.QUOTE_NONNUMERIC)` writes U.csv and
// V.csv (script/prmf_runner.py:1068-1071): Python's csv module prints a float with repr() -- the shortest digit string that
// reads back as the same double (std::to_chars gives the same digits), laid out by CPython's rule: positional notation when
// the decimal point falls in (-4, 16], else d[.ddd]e[+-]XX; ".0" appended to positional integers; NaN is the quoted
// empty string (na_rep under QUOTE_NONNUMERIC).  Host work (one thread per row range), here because this file is the text
// I/O of the data matrix; 50000 x 64 + 20000 x 64: 3.0 s in pandas (profiles/r05_cli_end_to_end.txt).
// ---------------------------------------------------------------------------
namespace {

// -> characters written (at most 26)
inline int format_repr(double x, char* out) {
  if (x != x) {
    out[0] = '"';
    out[1] = '"';
    return 2;
  }
  char* o = out;
  if (std::signbit(x)) {
    *o++ = '-';
    x = -x;
  }
  if (std::isinf(x)) {
    std::memcpy(o, "inf", 3);
    return (int)(o + 3 - out);
  }
  char sci[40];                                      // d[.ddd]e[+-]XX, shortest round-trip digits
  const std::to_chars_result r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
  char digits[24];
  int nd = 0;
  const char* p = sci;
  for (; p < r.ptr && *p != 'e'; ++p)
    if (*p != '.') digits[nd++] = *p;
  ++p;                                               // 'e'
  const bool eneg = *p == '-';
  ++p;
  int e10 = 0;
  for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
  if (eneg) e10 = -e10;
  const int decpt = e10 + 1;                         // digits are 0.d1d2... x 10^decpt
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -decpt; ++i) *o++ = '0';
      std::memcpy(o, digits, (size_t)nd);
      o += nd;
    } else if (decpt >= nd) {
      std::memcpy(o, digits, (size_t)nd);
      o += nd;
      for (int i = nd; i < decpt; ++i) *o++ = '0';
      *o++ = '.';
      *o++ = '0';
    } else {
      std::memcpy(o, digits, (size_t)decpt);
      o += decpt;
      *o++ = '.';
      std::memcpy(o, digits + decpt, (size_t)(nd - decpt));
      o += nd - decpt;
    }
  } else {
    *o++ = digits[0];
    if (nd > 1) {
      *o++ = '.';
      std::memcpy(o, digits + 1, (size_t)(nd - 1));
      o += nd - 1;
    }
    *o++ = 'e';
    int e = decpt - 1;
    *o++ = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    if (e >= 100) *o++ = (char)('0' + e / 100);
    *o++ = (char)('0' + (e / 10) % 10);
    *o++ = (char)('0' + e % 10);
  }
  return (int)(o - out);
}

}  // namespace

// rows [r0, r1) of `values` as text lines appended to *out: prefix (nullable offsets), fields joined by `sep`, '\n'
void text_format_rows(const double* values, int64_t ld, int64_t cols, int64_t r0, int64_t r1, const char* prefixes,
                      const int64_t* prefix_off, char sep, std::string* out) {
  char cell[32];
  for (int64_t r = r0; r < r1; ++r) {
    if (prefixes) out->append(prefixes + prefix_off[r], (size_t)(prefix_off[r + 1] - prefix_off[r]));
    const double* row = values + r * ld;
    for (int64_t c = 0; c < cols; ++c) {
      if (c) out->push_back(sep);
      out->append(cell, (size_t)format_repr(row[c], cell));
    }
    out->push_back('\n');
  }
}

// 0 on success, else errno of the failing stdio call
int text_write_matrix(const char* path, const char* header, const char* prefixes, const int64_t* prefix_off, const double* values,
                      int64_t rows, int64_t cols, int64_t ld, char sep, int n_threads) {
  std::FILE* fh = std::fopen(path, "wb");
  if (!fh) return errno ? errno : -1;
  int err = 0;
  if (header && std::fputs(header, fh) < 0) err = errno ? errno : -1;
  if (n_threads < 1) n_threads = 1;
  const int64_t block = 4096 * (int64_t)n_threads;       // rows formatted between two writes
  std::vector<std::string> parts((size_t)n_threads);
  for (int64_t b0 = 0; b0 < rows && !err; b0 += block) {
    const int64_t b1 = b0 + block < rows ? b0 + block : rows;
    const int64_t per = (b1 - b0 + n_threads - 1) / n_threads;
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) {
      const int64_t r0 = b0 + t * per, r1 = r0 + per < b1 ? r0 + per : b1;
      parts[(size_t)t].clear();
      if (r0 >= r1) continue;
      parts[(size_t)t].reserve((size_t)(r1 - r0) * (size_t)(cols * 26 + 64));      // once: clear() keeps it for the next block
      if (n_threads == 1) {
        text_format_rows(values, ld, cols, r0, r1, prefixes, prefix_off, sep, &parts[0]);
      } else {
        pool.emplace_back(text_format_rows, values, ld, cols, r0, r1, prefixes, prefix_off, sep, &parts[(size_t)t]);
      }
    }
    for (auto& th : pool) th.join();
    for (int t = 0; t < n_threads && !err; ++t)
      if (!parts[(size_t)t].empty() && std::fwrite(parts[(size_t)t].data(), 1, parts[(size_t)t].size(), fh) != parts[(size_t)t].size())
        err = errno ? errno : -1;
  }
  if (std::fclose(fh) != 0 && !err) err = errno ? errno : -1;
  return err;
}

}  // namespace prmf
