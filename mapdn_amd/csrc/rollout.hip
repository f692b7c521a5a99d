// rollout.hip — the glue of one batched rollout step around the env and the policy, gfx950 (MI355X).
//
// Reference: Model.train_process (models/model.py:197-262) per env step: select_action's exploration sample (utilities/util.py:52-76),
// translate_action (utilities/util.py:123-132), the running means of reward / info (models/model.py:243-248) and the replay insertion
// (utilities/replay_buffer.py:25-29).  In PyTorch each of these is a chain of one-line elementwise launches — ~45 per step next to
// the 4 launches that do the work (policy forward, k_nr_tree, k_advance, k_gather), a fifth of the end-to-end loop.  Here:
//   k_explore        tanh(mean + std * eps), the availability mask, and translate_action's clamp + affine map: one launch, the same
//                    f32 operations in the same order as the PyTorch chain (no contraction: bit-identical actions);
//   k_rollout_stats  sum over the live envs of the 11 info values and the reward, the live count, and alive &= ~done: one block,
//                    fixed summation order (deterministic);
//   k_copy_segments  every field of the transition (state, action, reward, next_state, done, ..., hidden states) into its ring
//                    positions (and their mirror) as ONE launch over up to 48 contiguous segments instead of ~20 copy launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mapdn.h"

namespace mapdn {

__global__ void __launch_bounds__(256)
k_explore(const float* __restrict__ mean, const float* __restrict__ eps, const float* __restrict__ avail, float stdv, int tanh_bound,
          float half_range2, float low, float* __restrict__ action, float* __restrict__ action_pol, float* __restrict__ actual, long n) {
#pragma clang fp contract(off)      // every PyTorch op of the chain rounds on its own: no mul + add -> fma here (hipcc contracts by default,
                                    // and the operators must be written out here: inlined __fmul_rn / __fadd_rn bring their own contract flag)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    // x_t = mean + std * eps ; y_t = tanh(x_t)       (util.py:57-66; Normal(mean, std).rsample())
    const float se = stdv * eps[i];
    const float x = mean[i] + se;
    const float y = tanh_bound ? tanhf(x) : x;
    action[i] = y;
    // restore_mask * actions (maddpg.py:92-93): 1 - (avail == 0)
    const float yp = avail ? (avail[i] == 0.0f ? 0.0f : 1.0f) * y : y;
    if (action_pol) action_pol[i] = yp;
    // translate_action (util.py:123-132) of the stored action: clamp to [-1, 1], 0.5 (cp + 1) (high - low) + low
    const float cp = fminf(fmaxf(y, -1.0f), 1.0f);
    const float t1 = cp + 1.0f;
    const float t2 = 0.5f * t1;
    const float t3 = t2 * half_range2;
    actual[i] = t3 + low;
  }
}

// sums[0..10] += sum_e alive[e] info[e][k]; sums[11] += sum_e alive[e] reward[e]; sums[12] += sum_e alive[e]; alive_out[e] = alive[e] & !done[e]
// (alive_out may be alive itself)
__global__ void __launch_bounds__(1024)
k_rollout_stats(const double* __restrict__ info, const double* __restrict__ reward, const uint8_t* alive, const uint8_t* __restrict__ done,
                uint8_t* alive_out, double* __restrict__ sums, int B) {
  __shared__ double s[13][33];
  double acc[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) acc[k] = 0.0;
  for (int e = threadIdx.x; e < B; e += 1024) {
    const double w = alive[e] ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) acc[k] += info[(size_t)e * 11 + k] * w;
    acc[11] += reward[e] * w; acc[12] += w;
    alive_out[e] = (alive[e] && !done[e]) ? 1 : 0;
  }
  const int lane = threadIdx.x & 31, warp = threadIdx.x >> 5;         // 32 groups of 32 threads: fixed two-level tree
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    double v = acc[k];
    for (int o = 16; o > 0; o >>= 1) v += __shfl_down(v, o, 32);
    if (lane == 0) s[k][warp] = v;
  }
  __syncthreads();
  if (threadIdx.x < 13) {
    double t = 0.0;
    for (int g = 0; g < 32; ++g) t += s[threadIdx.x][g];
    sums[threadIdx.x] += t;
  }
}

struct CopySegs { const uint4* src[48]; uint4* dst[48]; long units[48]; int n; };

__global__ void __launch_bounds__(256) k_copy_segments(CopySegs c) {
  const int sgm = blockIdx.y;
  if (sgm >= c.n) return;
  const uint4* __restrict__ s = c.src[sgm];
  uint4* __restrict__ d = c.dst[sgm];
  const long n = c.units[sgm];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = s[i];
}

}  // namespace mapdn

extern "C" int mapdn_explore_actions(const float* mean, const float* eps, const float* avail, float stdv, int32_t tanh_bound, double action_scale,
                                     double action_bias, float* action, float* action_pol, float* actual, int64_t n, void* stream) {
  using namespace mapdn;
  if (!mean || !eps || !action || !actual || n < 1) return MAPDN_E_INVALID;
  // util.py:128-129: low / high and (high - low) are Python doubles; PyTorch rounds each scalar to f32 where it meets the f32 tensor
  const double low = action_bias - action_scale, high = action_bias + action_scale;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_explore, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mean, eps, avail, stdv, (int)tanh_bound, (float)(high - low), (float)low,
                     action, action_pol, actual, (long)n);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int mapdn_rollout_stats(const double* info, const double* reward, const uint8_t* alive, const uint8_t* done, uint8_t* alive_out, double* sums,
                                   int32_t n_envs, void* stream) {
  using namespace mapdn;
  if (!info || !reward || !alive || !done || !alive_out || !sums || n_envs < 1) return MAPDN_E_INVALID;
  hipLaunchKernelGGL(k_rollout_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, info, reward, alive, done, alive_out, sums, (int)n_envs);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int mapdn_copy_segments(const void* const* src, void* const* dst, const int64_t* nbytes, int32_t n_segments, void* stream) {
  using namespace mapdn;
  if (!src || !dst || !nbytes || n_segments < 1 || n_segments > 48) return MAPDN_E_INVALID;
  CopySegs c;
  long most = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (!src[i] || !dst[i] || nbytes[i] < 0 || (nbytes[i] & 15) || ((uintptr_t)src[i] & 15) || ((uintptr_t)dst[i] & 15)) return MAPDN_E_INVALID;
    c.src[i] = (const uint4*)src[i]; c.dst[i] = (uint4*)dst[i]; c.units[i] = (long)(nbytes[i] >> 4);
    most = c.units[i] > most ? c.units[i] : most;
  }
  c.n = n_segments;
  if (most == 0) return MAPDN_OK;
  const int bx = (int)((most + 255) / 256 < 2048 ? (most + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_copy_segments, dim3(bx, n_segments), dim3(256), 0, (hipStream_t)stream, c);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}
