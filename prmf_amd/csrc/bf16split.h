// float32 -> (hi, lo) bfloat16 pairs: hi = bf16(v), lo = bf16(v - hi), round-to-nearest-even both times, so that
// hi + lo carries 16 mantissa bits of v.  Shared by the fused epilogues (packing the factors for the next sweep,
// update.hip) and the direct-residual kernel (residual.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace prmf {

__device__ __forceinline__ unsigned pack_bf16_pair(float a, float b) {
  unsigned short ha = __builtin_bit_cast(unsigned short, (__bf16)a), hb = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (unsigned)ha | ((unsigned)hb << 16);
}
__device__ __forceinline__ float bf16_trunc_back(float a) {
  unsigned short ha = __builtin_bit_cast(unsigned short, (__bf16)a);
  return __uint_as_float(((unsigned)ha) << 16);
}

}  // namespace prmf
