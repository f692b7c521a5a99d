// rowtile.hpp — lane-level helpers shared by the learner's MFMA kernels (critic.hip, policy_bwd.hip): a tile of 16 rows of 64 features
// in "A layout" (lane l = (j = l & 15, g = l >> 4) holds row j, features 16 c + 4 g + q), reductions over the four lanes of a row and
// over the 16 lanes of a DPP row, LayerNorm statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mapdn {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

constexpr int HS = 68;                 // row stride (floats) of a 16 x 64 LDS staging tile: C-layout reads hit 64 distinct banks
constexpr int HP = 4416;               // floats of parameter-gradient partials per wavefront / workgroup: dW2 4096 | dgamma | dbeta | db2 | dw3 | db3 + pad

struct HeadArgs {
  const float* x;                      // [rows][64] — or base [rows / n][64] when per_n != nullptr
  const float* per_n;                  // [n][64] or nullptr
  int n;
  const float* gamma; const float* beta; float eps;
  const float* w2; const float* b2; const float* w3; const float* b3;
};

// sum over the four lanes that hold one row (j, j + 16, j + 32, j + 48), every lane receiving it: two row swaps, no LDS
__device__ __forceinline__ float sum_g(float v) {
  u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a.x) + __uint_as_float(a.y);
  a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a.x) + __uint_as_float(a.y);
}
// sum over the 16 lanes of a DPP row
__device__ __forceinline__ float sum_j(float v) {
  v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
  return v;
}
__device__ __forceinline__ float relu_nan(float o) { return o > 0.0f ? o : (o != o ? o : 0.0f); }      // torch.relu keeps NaN
__device__ __forceinline__ float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
// GRU gate non-linearities on the hardware transcendentals: v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of the library's tanhf (~40
// instructions) and a full-precision division (~10): absolute error ~2e-7, the size of one f32 rounding of the gate pre-activation —
// the gate arithmetic was 45 % of the policy kernels' VALU instructions, and those bound them (SQ counters, profiles/r06_*)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }

// LayerNorm statistics of the lane's row; xa becomes xhat = (x - mean) rstd
__device__ __forceinline__ float ln_stats(f4 (&xa)[4], float eps) {
  const float mu = sum_g((hsum(xa[0]) + hsum(xa[1])) + (hsum(xa[2]) + hsum(xa[3]))) * (1.0f / 64.0f);
  float var = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { xa[c] = xa[c] - mu; var += hsum(xa[c] * xa[c]); }
  const float rs = rsqrtf(sum_g(var) * (1.0f / 64.0f) + eps);
#pragma unroll
  for (int c = 0; c < 4; ++c) xa[c] = xa[c] * rs;
  return rs;
}

}  // namespace mapdn
