// philox.hpp — Philox4x32-10 (Salmon et al., SC'11) as the wide kernels use it, compilable for the HOST as well: tests/test_philox.py
// builds it with g++ and pins it on the Random123 known-answer vectors and on oracle/philox.py (the device code and the check
// compile the same source).  One 32 x 32 -> 64 multiply per lane of a round (v_mad_u64_u32) instead of a mul_lo / mul_hi pair.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define PHILOX_FN __device__ __forceinline__
#else
#define PHILOX_FN static inline
#endif

namespace mapdn {

// Philox4x32-10 (Salmon et al. SC'11), keyed (seed) / counter (env, draw, stream, block); mapping documented in oracle/philox.py
PHILOX_FN void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53 random bits as a double in [0, 2^53)
PHILOX_FN double u53(uint32_t hi, uint32_t lo) { return (double)(((uint64_t)(hi >> 5) << 26) + (uint64_t)(lo >> 6)); }

}  // namespace mapdn
