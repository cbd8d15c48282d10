// slot / L and row / A without a division: a 32-bit divide by a kernel argument is ~30 VALU instructions on gfx950, and the
// small-group vote kernel met four of them per slot (profiles/r05_cfg5.md).  No HIP in this header: the kernels include it
// (fpx_kernels.hpp), make_geom (fpx_api.hip) computes the constants with it, and tests/test_fastdiv.py compiles it with
// g++ and holds it against `/` for every divisor up to 2^12, the powers of two and their neighbours, and random ones.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FPX_FASTDIV_FN __host__ __device__ __forceinline__
#else
#define FPX_FASTDIV_FN inline
#endif

namespace fpx {

// (magic, shift) of a divisor d >= 2: shift = ceil(log2 d) - 1, so that 2^shift < d <= 2^(shift + 1), and
// magic = floor(2^(32 + shift) / d) + 1 -- at most 2^31 + 1 for d = 2^(shift + 1), below 2^32 otherwise.
// d < 2: magic = 0 (callers take s / 1 = s)
FPX_FASTDIV_FN void fast_div_magic(int d, uint32_t* magic, int32_t* shift) {
  *magic = 0, *shift = 0;
  if (d < 2) return;
  int k = 0;
  while ((1ll << (k + 1)) < d) ++k;
  *shift = k, *magic = (uint32_t)(((1ull << (32 + k)) / (uint64_t)d) + 1ull);
}

// the high 32 bits of a 32 x 32-bit product (v_mul_hi_u32 on the device)
FPX_FASTDIV_FN uint32_t fast_div_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// s / d for 0 <= s < 2^31.  Exact: magic = 2^(32 + shift) / d + e with 0 < e <= 1, so s * magic / 2^(32 + shift) exceeds
// s / d by s * e / 2^(32 + shift) < 2^31 / 2^(32 + shift) <= 1 / d (d <= 2^(shift + 1)) -- too little to reach the next
// integer from a quotient whose fractional part is at most (d - 1) / d
FPX_FASTDIV_FN int fast_div(int s, uint32_t magic, int shift) { return (int)(fast_div_mulhi((uint32_t)s, magic) >> shift); }

}  // namespace fpx
