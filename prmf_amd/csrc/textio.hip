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

#include "kernels.h"
#include "xstrtod.h"

namespace prmf {

namespace {

constexpr int TEXT_TPB = 256;                       // lanes per workgroup
constexpr int TEXT_BPL = 16;                        // bytes per lane
constexpr int TEXT_TILE = TEXT_TPB * TEXT_BPL;      // 4 KiB per workgroup

// bit i of *sep: byte base + i is a separator; of *nl: it is '\n'
__device__ __forceinline__ void classify16(const unsigned char* __restrict__ text, int64_t base, int64_t L, unsigned char delim,
                                           unsigned* sep, unsigned* nl) {
  unsigned s = 0, n = 0;
  if (base + TEXT_BPL <= L) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + base);        // base is a multiple of 16, text is hipMalloc'ed
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < TEXT_BPL; ++i) {
      const unsigned char c = (unsigned char)(w[i >> 2] >> ((i & 3) * 8));
      n |= (unsigned)(c == '\n') << i;
      s |= (unsigned)(c == '\n' || c == delim) << i;
    }
  } else {
    for (int i = 0; i < TEXT_BPL && base + i < L; ++i) {
      const unsigned char c = text[base + i];
      n |= (unsigned)(c == '\n') << i;
      s |= (unsigned)(c == '\n' || c == delim) << i;
    }
  }
  *sep = s;
  *nl = n;
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
  unsigned sep = 0, nl = 0;
  if (base < L) classify16(text, base, L, delim, &sep, &nl);
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
  const ParsedField r = parse_decimal(text + start, text + end);
  if (!r.ok || r.wide_int) {
    int why = r.ok ? TEXT_FLAG_WIDE_INT : TEXT_FLAG_NOT_NUMERIC;
    if (!r.ok)
      for (int64_t p = start; p < end; ++p)
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

}  // namespace prmf
