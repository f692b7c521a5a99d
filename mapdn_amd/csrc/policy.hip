// policy.hip — rollout-side forward of MAPDN's shared-parameter recurrent agent, gfx950 (MI355X).
//
// Reference: agents/rnn_agent.py:5-32 (fc1 -> LayerNorm -> ReLU -> GRUCell -> fc2) as called once per env step by the
// rollout (models/model.py:101-139, 204-262) on [envs x agents] rows of 58-82 observation values + a one-hot agent id.
// The stock PyTorch path is five launches whose small GEMMs / LayerNorm / GRU-cell kernels run far below any roof at this
// shape (profiles/e2e): the batched env made them the bulk of a training step.  This kernel does the whole forward in one
// launch for inference (no autograd; training-time forward passes stay in PyTorch): one wavefront per tile of 16 rows, the
// three products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, like the reference's fp32 modules), the
// whole parameter set (~130 KB) staged once per workgroup in LDS in MFMA operand order (weight-stationary), LayerNorm /
// gates / fc2 on the accumulator layout with DPP-row reductions.  Results agree with PyTorch to ~1e-6 (summation order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/mapdn.h"
#include "rowtile.hpp"

namespace mapdn {

constexpr int PH = 64;          // hidden size (args.hid_size of the reference's default.yaml)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
// sum over the 16 lanes of a DPP row (lanes that share lane >> 4)
__device__ __forceinline__ float row_sum16(float v) {
  v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
  return v;
}

// One wavefront = one tile of 16 rows at a time; every product runs on v_mfma_f32_16x16x4_f32 (exact fp32 at the vector
// rate, but 1024 MACs per 256 bytes of operand traffic — the VALU form of this kernel, one thread per row with the weights
// as LDS broadcasts, measured 1.02 ms for 311 k rows, exactly PyTorch's time: LDS-bound at 0.25 MAC per LDS byte).
// Operand maps (CDNA4 guide): A lane l = A[i = l & 15][k], B lane l = B[k][j = l & 15], C/D reg r of lane l = row 4 (l >> 4) + r,
// column l & 15.  The k index of MFMA step s of chunk c is 16 c + 4 (l >> 4) + s for BOTH operands, so a lane's four steps of
// a chunk are 16 contiguous bytes: one b128 load per operand and chunk.  Weights are staged ONCE per workgroup in LDS in
// exactly that operand order (Wop[n-tile][chunk][lane] = 4 floats), weight-stationary.
// PT threads per workgroup: 512 (8 waves = two per SIMD, one wave's MFMA phase overlaps another's VALU / memory phase) when the
// activation tiles of 8 waves still fit beside the parameters in LDS, else 256
template <int PT>
__global__ void __launch_bounds__(PT)
k_policy_fwd(const float* __restrict__ obs, const float* __restrict__ hid_in, const float* __restrict__ w1, const float* __restrict__ b1,
             const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
             const float* __restrict__ b_ih, const float* __restrict__ b_hh, const float* __restrict__ w2, const float* __restrict__ b2,
             float* __restrict__ means, float* __restrict__ hid_out, int rows, int n_agents, int o, int ids, float ln_eps, int ids_lds,
             float* __restrict__ x1_out) {
  extern __shared__ float sm[];
  const int KC1 = (o + 15) >> 4;                                  // 16-wide k chunks of fc1 (zero-padded weights)
  f4* sW1 = (f4*)sm;                                              // [4][KC1][64]
  f4* sWih = sW1 + (size_t)4 * KC1 * 64;                          // [12][4][64]
  f4* sWhh = sWih + 12 * 4 * 64;                                  // [12][4][64]
  float* sW1id = (float*)(sWhh + 12 * 4 * 64);                    // [ids][64]
  float* sB1 = sW1id + (size_t)(ids_lds ? ids : 0) * PH;      // (the id columns stay in global memory when LDS is short: 8 waves matter more)
  float* sG = sB1 + PH; float* sB = sG + PH;
  float* sBih = sB + PH; float* sBhh = sBih + 3 * PH; float* sW2 = sBhh + 3 * PH;
  float* sX = sW2 + PH;                                           // [PT / 64 waves][16][68]: C layout -> A layout of the activations
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15, in1 = o + ids;
  for (int i = tid; i < 4 * KC1 * 64; i += PT) {
    const int l = i & 63, c = (i >> 6) % KC1, nt = (i >> 6) / KC1;
    f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = 16 * c + 4 * (l >> 4) + q; v[q] = k < o ? w1[(size_t)(16 * nt + (l & 15)) * in1 + k] : 0.0f; }
    sW1[i] = v;
  }
  for (int i = tid; i < 12 * 4 * 64; i += PT) {
    const int l = i & 63, c = (i >> 6) & 3, nt = i >> 8;
    const size_t base = (size_t)(16 * nt + (l & 15)) * PH + 16 * c + 4 * (l >> 4);
    sWih[i] = *(const f4*)(w_ih + base); sWhh[i] = *(const f4*)(w_hh + base);
  }
  if (ids_lds) for (int i = tid; i < ids * PH; i += PT) sW1id[i] = w1[(size_t)(i % PH) * in1 + o + i / PH];      // [agent][unit]
  for (int i = tid; i < PH; i += PT) { sB1[i] = b1[i]; sG[i] = ln_g[i]; sB[i] = ln_b[i]; sW2[i] = w2[i]; }
  for (int i = tid; i < 3 * PH; i += PT) { sBih[i] = b_ih[i]; sBhh[i] = b_hh[i]; }
  __syncthreads();
  float* xt = sX + (size_t)wave * 16 * 68;
  const float bias2 = b2[0];
  const int n_tiles = (rows + 15) >> 4;
  for (int T = blockIdx.x * (PT / 64) + wave; T < n_tiles; T += gridDim.x * (PT / 64)) {
    const int row0 = T << 4;
    const int arow = min(row0 + j, rows - 1);                     // the row this lane feeds as A operand
    // ---- fc1: X1[16 x 64] = obs[16 x o] W1^T
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    const float* ob = obs + (size_t)arow * o;
    for (int c = 0; c < KC1; ++c) {
      float a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int k = 16 * c + 4 * g + q; a[q] = k < o ? ob[k] : 0.0f; }
      f4 bw[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bw[nt] = sW1[(size_t)(nt * KC1 + c) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bw[nt][q], acc[nt], 0, 0, 0);
    }
    // ---- + bias + one-hot id column, LayerNorm (biased variance, eps inside the root), ReLU; C layout: reg r <-> row 4 g + r
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int agent = min(row0 + 4 * g + r, rows - 1) % n_agents;
      float sum = 0.0f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int u = 16 * nt + j;
        acc[nt][r] += sB1[u] + (ids ? (ids_lds ? sW1id[(size_t)agent * PH + u] : w1[(size_t)u * in1 + o + agent]) : 0.0f);
        if (x1_out && row0 + 4 * g + r < rows) x1_out[(size_t)(row0 + 4 * g + r) * PH + u] = acc[nt][r];   // (training: the LayerNorm input, for the backward)
        sum += acc[nt][r];
      }
      mean[r] = row_sum16(sum) * (1.0f / PH);
      float var = 0.0f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { const float dlt = acc[nt][r] - mean[r]; var = fmaf(dlt, dlt, var); }
      rstd[r] = rsqrtf(row_sum16(var) * (1.0f / PH) + ln_eps);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int u = 16 * nt + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float o = fmaf((acc[nt][r] - mean[r]) * rstd[r], sG[u], sB[u]); xt[(4 * g + r) * 68 + u] = o > 0.0f ? o : (o != o ? o : 0.0f); }   // ReLU, NaN kept (as torch.relu)
    }
    // (one wavefront: its LDS writes are visible to its later reads, no barrier)
    // ---- GRUCell pre-activations: r, z over [x | h] (K = 128), i_n over x, h_n over h — one 16-unit output tile at a time, so that
    // four accumulator tiles are live instead of sixteen (with 512 threads a wave has 256 registers: the sixteen-tile form spilled
    // 24 of them to scratch); the activations of the tile's 16 rows (A operands) are loaded once
    f4 ax[4], ah[4];
    const float* hrow = hid_in + (size_t)arow * PH;
#pragma unroll
    for (int c = 0; c < 4; ++c) { ax[c] = *(const f4*)(xt + j * 68 + 16 * c + 4 * g); ah[c] = *(const f4*)(hrow + 16 * c + 4 * g); }
    float outp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f4 aR = f4{0, 0, 0, 0}, aZ = f4{0, 0, 0, 0}, aIN = f4{0, 0, 0, 0}, aHN = f4{0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f4 wir = sWih[(size_t)(nt * 4 + c) * 64 + lane], whr = sWhh[(size_t)(nt * 4 + c) * 64 + lane];
        const f4 wiz = sWih[(size_t)((4 + nt) * 4 + c) * 64 + lane], whz = sWhh[(size_t)((4 + nt) * 4 + c) * 64 + lane];
        const f4 win = sWih[(size_t)((8 + nt) * 4 + c) * 64 + lane], whn = sWhh[(size_t)((8 + nt) * 4 + c) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          aR = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c][q], wir[q], aR, 0, 0, 0);
          aZ = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c][q], wiz[q], aZ, 0, 0, 0);
          aIN = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c][q], win[q], aIN, 0, 0, 0);
          aR = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[c][q], whr[q], aR, 0, 0, 0);
          aZ = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[c][q], whz[q], aZ, 0, 0, 0);
          aHN = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[c][q], whn[q], aHN, 0, 0, 0);
        }
      }
      // ---- gates (torch.nn.GRUCell), new hidden state, this tile's share of fc2; C layout: reg r <-> row 4 g + r, column j
      const int u = 16 * nt + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        const int rc = min(row, rows - 1);
        const float rg = sigmoidf_(aR[r] + sBih[u] + sBhh[u]);
        const float zg = sigmoidf_(aZ[r] + sBih[PH + u] + sBhh[PH + u]);
        const float nn = tanhf(fmaf(rg, aHN[r] + sBhh[2 * PH + u], aIN[r] + sBih[2 * PH + u]));
        const float hu = hid_in[(size_t)rc * PH + u];
        const float hnew = fmaf(zg, hu - nn, nn);                 // (1 - z) n + z h
        if (hid_out && row < rows) hid_out[(size_t)row * PH + u] = hnew;
        outp[r] = fmaf(sW2[u], hnew, outp[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      const float tot = row_sum16(outp[r]);
      if (j == 0 && row < rows) means[row] = tot + bias2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same forward with every product taken TRANSPOSED (the weight is the MFMA A operand, the activations the B operand), so
// that a lane keeps ONE row of the tile from the observation to the action ("A layout", rowtile.hpp: lane (j, g) = row j, features
// 16 c + 4 g + q): the D registers of x1^T = W1 obs^T and of gates^T = W x^T line up element by element with the row's hidden state,
// which is loaded once (as B operand AND for the gate arithmetic) and stored as float4; LayerNorm is 16 in-lane adds + two row swaps;
// no LDS round trip for the activations and no per-element global accesses (the first form issued 16 scalar loads and 16 scalar
// stores of the hidden state per lane and tile, plus an LDS transposition of x).  Same LDS weight images, same arithmetic per element.
template <int PT>
__global__ void __launch_bounds__(PT)
k_policy_fwd2(const float* __restrict__ obs, const float* __restrict__ hid_in, const float* __restrict__ w1, const float* __restrict__ b1,
              const float* __restrict__ ln_g, const float* __restrict__ ln_b, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
              const float* __restrict__ b_ih, const float* __restrict__ b_hh, const float* __restrict__ w2, const float* __restrict__ b2,
              float* __restrict__ means, float* __restrict__ hid_out, int rows, int n_agents, int o, int ids, float ln_eps, int ids_lds,
              float* __restrict__ x1_out) {
  extern __shared__ float sm[];
  const int KC1 = (o + 15) >> 4;
  f4* sW1 = (f4*)sm;                                              // [4][KC1][64]
  f4* sWih = sW1 + (size_t)4 * KC1 * 64;                          // [12][4][64]
  f4* sWhh = sWih + 12 * 4 * 64;                                  // [12][4][64]
  float* sW1id = (float*)(sWhh + 12 * 4 * 64);                    // [ids][64]
  float* sB1 = sW1id + (size_t)(ids_lds ? ids : 0) * PH;
  float* sG = sB1 + PH; float* sB = sG + PH;
  float* sBih = sB + PH; float* sBhh = sBih + 3 * PH; float* sW2 = sBhh + 3 * PH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15, in1 = o + ids;
  for (int i = tid; i < 4 * KC1 * 64; i += PT) {
    const int l = i & 63, c = (i >> 6) % KC1, nt = (i >> 6) / KC1;
    f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = 16 * c + 4 * (l >> 4) + q; v[q] = k < o ? w1[(size_t)(16 * nt + (l & 15)) * in1 + k] : 0.0f; }
    sW1[i] = v;
  }
  for (int i = tid; i < 12 * 4 * 64; i += PT) {
    const int l = i & 63, c = (i >> 6) & 3, nt = i >> 8;
    const size_t base = (size_t)(16 * nt + (l & 15)) * PH + 16 * c + 4 * (l >> 4);
    sWih[i] = *(const f4*)(w_ih + base); sWhh[i] = *(const f4*)(w_hh + base);
  }
  if (ids_lds) for (int i = tid; i < ids * PH; i += PT) sW1id[i] = w1[(size_t)(i % PH) * in1 + o + i / PH];      // [agent][unit]
  for (int i = tid; i < PH; i += PT) { sB1[i] = b1[i]; sG[i] = ln_g[i]; sB[i] = ln_b[i]; sW2[i] = w2[i]; }
  for (int i = tid; i < 3 * PH; i += PT) { sBih[i] = b_ih[i]; sBhh[i] = b_hh[i]; }
  __syncthreads();
  const float bias2 = b2[0];
  const bool even = (o & 1) == 0;
  const int n_tiles = (rows + 15) >> 4;
  for (int T = blockIdx.x * (PT / 64) + wave; T < n_tiles; T += gridDim.x * (PT / 64)) {
    const int row = (T << 4) + j;
    const bool valid = row < rows;
    const int rc = valid ? row : rows - 1;
    // ---- fc1, transposed: x1^T[u][row] = sum_k W1[u][k] obs[row][k]
    f4 x[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
    const float* ob = obs + (size_t)rc * o;
    for (int c = 0; c < KC1; ++c) {
      const int k0 = 16 * c + 4 * g;
      f4 a;
      if (even && k0 + 3 < o) {                                   // two 8-byte loads (a row of an even width starts 8-byte aligned)
        const float2 lo = *(const float2*)(ob + k0), hi = *(const float2*)(ob + k0 + 2);
        a = f4{lo.x, lo.y, hi.x, hi.y};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = k0 + q < o ? ob[k0 + q] : 0.0f;
      }
      f4 wv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wv[nt] = sW1[(size_t)(nt * KC1 + c) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) x[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][q], a[q], x[nt], 0, 0, 0);
    }
    // ---- + bias + one-hot id column; LayerNorm (biased variance, eps inside the root); ReLU
    const int agent = rc % n_agents;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int u0 = 16 * nt + 4 * g;
      x[nt] += *(const f4*)(sB1 + u0);
      if (ids) {
        if (ids_lds) x[nt] += *(const f4*)(sW1id + (size_t)agent * PH + u0);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) x[nt][r] += w1[(size_t)(u0 + r) * in1 + o + agent];
        }
      }
    }
    if (x1_out && valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *(f4*)(x1_out + (size_t)row * PH + 16 * nt + 4 * g) = x[nt];
    }
    ln_stats(x, ln_eps);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f4 y = x[nt] * *(const f4*)(sG + 16 * nt + 4 * g) + *(const f4*)(sB + 16 * nt + 4 * g);
      x[nt] = f4{relu_nan(y.x), relu_nan(y.y), relu_nan(y.z), relu_nan(y.w)};
    }
    f4 hv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hv[c] = *(const f4*)(hid_in + (size_t)rc * PH + 16 * c + 4 * g);
    // ---- GRUCell pre-activations, transposed: r, z over [x | h], n_input over x, n_hidden over h
    f4 aR[4], aZ[4], aI[4], aH[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) aR[nt] = aZ[nt] = aI[nt] = aH[nt] = f4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f4 wir = sWih[((0 + nt) * 4 + c) * 64 + lane], wiz = sWih[((4 + nt) * 4 + c) * 64 + lane], win = sWih[((8 + nt) * 4 + c) * 64 + lane];
        const f4 whr = sWhh[((0 + nt) * 4 + c) * 64 + lane], whz = sWhh[((4 + nt) * 4 + c) * 64 + lane], whn = sWhh[((8 + nt) * 4 + c) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          aR[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wir[q], x[c][q], aR[nt], 0, 0, 0);
          aZ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wiz[q], x[c][q], aZ[nt], 0, 0, 0);
          aI[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[q], x[c][q], aI[nt], 0, 0, 0);
          aR[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whr[q], hv[c][q], aR[nt], 0, 0, 0);
          aZ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whz[q], hv[c][q], aZ[nt], 0, 0, 0);
          aH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whn[q], hv[c][q], aH[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting all 96 weight loads of the tile)
      }
    // ---- gates (torch.nn.GRUCell), new hidden state, fc2
    float outp = 0.0f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int u0 = 16 * nt + 4 * g;
      const f4 bir = *(const f4*)(sBih + u0), biz = *(const f4*)(sBih + PH + u0), bin = *(const f4*)(sBih + 2 * PH + u0);
      const f4 bhr = *(const f4*)(sBhh + u0), bhz = *(const f4*)(sBhh + PH + u0), bhn = *(const f4*)(sBhh + 2 * PH + u0);
      const f4 w2v = *(const f4*)(sW2 + u0);
      f4 hn4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = fast_sigmoid(aR[nt][r] + bir[r] + bhr[r]);
        const float zg = fast_sigmoid(aZ[nt][r] + biz[r] + bhz[r]);
        const float nn = fast_tanh(fmaf(rg, aH[nt][r] + bhn[r], aI[nt][r] + bin[r]));
        const float hnew = fmaf(zg, hv[nt][r] - nn, nn);              // (1 - z) n + z h
        hn4[r] = hnew;
        outp = fmaf(w2v[r], hnew, outp);
      }
      if (hid_out && valid) *(f4*)(hid_out + (size_t)row * PH + u0) = hn4;
    }
    const float tot = sum_g(outp);
    if (g == 0 && valid) means[row] = tot + bias2;
  }
}

// =====================================================================================================================
// LayerNorm over 64 features (+ optional fused ReLU), forward and backward, for the learner's TRAINING-time passes
// (agents/rnn_agent.py:16-21, critics/mlp_critic.py:22-27: fc1 -> LayerNorm -> ReLU).  At the reference's update intensity a batch
// is 32 steps of every env — 10 M rows of 64 on the 322-bus feeder — and PyTorch's row-per-block LayerNorm kernels run at 0.5 TB/s
// there (9.8 ms forward, 15 ms backward: a third of the update, profiles/e2e/r03_e2e_reference_kernel_stats.txt).  Here a row
// is 16 lanes x float4 (four rows per wavefront), the two moments are DPP-row sums, loads and stores are 16 bytes per lane and
// fully coalesced: both directions stream at HBM rate.  Backward: dx = rstd (a - mean(a) - xhat mean(a xhat)), a = dy gamma
// (dy masked by the ReLU), dgamma / dbeta as per-thread column sums over a grid-stride loop, reduced per block in LDS and
// across blocks by a second tiny launch in a fixed order (no atomics: deterministic).
// =====================================================================================================================
// BC ("broadcast input"): the row is not read but FORMED as x[row / n] + xn[row % n] — the central critic's first layer hands the
// LayerNorm base[b] + id_column[agent] (learner.py::_value_central): the [b, n, 64] tensor is never written or read (2 x 2.7 GB per
// forward at 10 M rows).  One f32 add, the one PyTorch's broadcast add performs: same bits as the materialised route.
template <bool RELU, bool BC>
__global__ void __launch_bounds__(256)
k_ln64_fwd(const f4* __restrict__ x, const f4* __restrict__ xn, int n, const float* __restrict__ gamma, const float* __restrict__ beta,
           f4* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long rows, float eps) {
  const int l16 = threadIdx.x & 15;
  const f4 g = ((const f4*)gamma)[l16], b = ((const f4*)beta)[l16];
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; row < rows; row += ((long)gridDim.x * 256) >> 4) {
    f4 v;
    if (BC) { const long q = row / n; v = x[q * 16 + l16] + xn[(row - q * n) * 16 + l16]; }
    else v = x[row * 16 + l16];
    const float mu = row_sum16((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
    const f4 dv = v - mu;
    const float var = row_sum16((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w)) * (1.0f / 64.0f);
    const float r = rsqrtf(var + eps);
    f4 o = dv * r * g + b;
    // torch.relu propagates NaN (fmaxf would turn it into 0 and hide a diverged update)
    if (RELU) { o.x = o.x > 0.0f ? o.x : (o.x != o.x ? o.x : 0.0f); o.y = o.y > 0.0f ? o.y : (o.y != o.y ? o.y : 0.0f);
                o.z = o.z > 0.0f ? o.z : (o.z != o.z ? o.z : 0.0f); o.w = o.w > 0.0f ? o.w : (o.w != o.w ? o.w : 0.0f); }
    y[row * 16 + l16] = o;
    if (l16 == 0) { mean[row] = mu; rstd[row] = r; }
  }
}

template <bool RELU, bool BC>
__global__ void __launch_bounds__(256)
k_ln64_bwd(const f4* __restrict__ dy, const f4* __restrict__ x, const f4* __restrict__ xn, int n, const float* __restrict__ gamma,
           const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd, f4* __restrict__ dx,
           float* __restrict__ partial, long rows) {
  __shared__ f4 s_g[16][16], s_b[16][16];
  const int l16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const f4 g = ((const f4*)gamma)[l16], b = ((const f4*)beta)[l16];
  f4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; row < rows; row += ((long)gridDim.x * 256) >> 4) {
    f4 v;
    if (BC) { const long q = row / n; v = x[q * 16 + l16] + xn[(row - q * n) * 16 + l16]; }
    else v = x[row * 16 + l16];
    f4 d = dy[row * 16 + l16];
    const float mu = mean[row], r = rstd[row];
    const f4 xh = (v - mu) * r;
    if (RELU) {
      const f4 pre = xh * g + b;
      d.x = pre.x > 0.0f ? d.x : 0.0f; d.y = pre.y > 0.0f ? d.y : 0.0f; d.z = pre.z > 0.0f ? d.z : 0.0f; d.w = pre.w > 0.0f ? d.w : 0.0f;
    }
    const f4 a = d * g;
    const float m1 = row_sum16((a.x + a.y) + (a.z + a.w)) * (1.0f / 64.0f);
    const float m2 = row_sum16((a.x * xh.x + a.y * xh.y) + (a.z * xh.z + a.w * xh.w)) * (1.0f / 64.0f);
    dx[row * 16 + l16] = (a - m1 - xh * m2) * r;
    ag += d * xh; ab += d;
  }
  s_g[rg][l16] = ag; s_b[rg][l16] = ab;
  __syncthreads();
  if (rg == 0) {
    f4 tg = s_g[0][l16], tb = s_b[0][l16];
#pragma unroll
    for (int i = 1; i < 16; ++i) { tg += s_g[i][l16]; tb += s_b[i][l16]; }
    ((f4*)partial)[(size_t)blockIdx.x * 32 + l16] = tg;
    ((f4*)partial)[(size_t)blockIdx.x * 32 + 16 + l16] = tb;
  }
}

// dgamma | dbeta [128] = sum over blocks of partial[block][128], in a fixed order: eight strided sub-sums, then those in order
__global__ void __launch_bounds__(1024) k_ln64_reduce(const float* __restrict__ partial, int nblocks, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_acc[8][128];
  const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
  float acc = 0.0f;
  for (int i = grp; i < nblocks; i += 8) acc += partial[(size_t)i * 128 + c];
  s_acc[grp][c] = acc;
  __syncthreads();
  if (grp == 0) {
    float t = s_acc[0][c];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += s_acc[g][c];
    if (c < 64) dgamma[c] = t; else dbeta[c - 64] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The critic's last two steps on rows of 64: v = relu(pre) . w3 + b3 (critics/mlp_critic.py:22-36: act(fc2(x)) -> fc3) as ONE pass over
// the pre-activation — the [rows, 64] tensor of relu(pre) is neither written nor read back (2 x 2.7 GB per forward at 10 M rows; the
// [rows, 64] x [64, 1] product PyTorch runs for fc3 reads it a third time).  16 lanes per row as in the LayerNorm kernels.
__global__ void __launch_bounds__(256)
k_relu_dot64_fwd(const f4* __restrict__ pre, const float* __restrict__ w, float b3, float* __restrict__ v, long rows) {
  const int l16 = threadIdx.x & 15;
  const f4 wv = ((const f4*)w)[l16];
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; row < rows; row += ((long)gridDim.x * 256) >> 4) {
    f4 h = pre[row * 16 + l16];
    // torch.relu propagates NaN
    h.x = h.x > 0.0f ? h.x : (h.x != h.x ? h.x : 0.0f); h.y = h.y > 0.0f ? h.y : (h.y != h.y ? h.y : 0.0f);
    h.z = h.z > 0.0f ? h.z : (h.z != h.z ? h.z : 0.0f); h.w = h.w > 0.0f ? h.w : (h.w != h.w ? h.w : 0.0f);
    const float s = row_sum16((h.x * wv.x + h.y * wv.y) + (h.z * wv.z + h.w * wv.w));
    if (l16 == 0) v[row] = s + b3;
  }
}
// backward: dpre = [pre > 0] dv w3;  partial[block][0:64] = sum_rows relu(pre) dv (-> dw3), partial[block][64] = sum_rows dv (-> db3)
__global__ void __launch_bounds__(256)
k_relu_dot64_bwd(const float* __restrict__ dv, const f4* __restrict__ pre, const float* __restrict__ w, f4* __restrict__ dpre,
                 float* __restrict__ partial, long rows) {
  __shared__ f4 s_g[16][16]; __shared__ float s_b[16];
  const int l16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const f4 wv = ((const f4*)w)[l16];
  f4 ag = {0.f, 0.f, 0.f, 0.f}; float ab = 0.0f;
  for (long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; row < rows; row += ((long)gridDim.x * 256) >> 4) {
    const f4 p = pre[row * 16 + l16];
    const float d = dv[row];
    f4 o;
    o.x = p.x > 0.0f ? d * wv.x : 0.0f; o.y = p.y > 0.0f ? d * wv.y : 0.0f; o.z = p.z > 0.0f ? d * wv.z : 0.0f; o.w = p.w > 0.0f ? d * wv.w : 0.0f;
    dpre[row * 16 + l16] = o;
    ag.x += (p.x > 0.0f ? p.x : 0.0f) * d; ag.y += (p.y > 0.0f ? p.y : 0.0f) * d; ag.z += (p.z > 0.0f ? p.z : 0.0f) * d; ag.w += (p.w > 0.0f ? p.w : 0.0f) * d;
    ab += d;
  }
  s_g[rg][l16] = ag;
  if (l16 == 0) s_b[rg] = ab;
  __syncthreads();
  if (rg == 0) {
    f4 tg = s_g[0][l16];
#pragma unroll
    for (int i = 1; i < 16; ++i) tg += s_g[i][l16];
    ((f4*)partial)[(size_t)blockIdx.x * 32 + l16] = tg;
    f4 tb = {0.f, 0.f, 0.f, 0.f};
    if (l16 == 0) { float t = s_b[0]; for (int i = 1; i < 16; ++i) t += s_b[i]; tb.x = t; }
    ((f4*)partial)[(size_t)blockIdx.x * 32 + 16 + l16] = tb;
  }
}

}  // namespace mapdn

static int ln64_blocks(long rows) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long need = (rows + 15) / 16;
  return (int)std::max<long>(1, std::min<long>(need, (long)cus * 8));
}

extern "C" int mapdn_layernorm64_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                         int64_t rows, float eps, int32_t relu, void* stream) {
  using namespace mapdn;
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 1) return MAPDN_E_INVALID;
  const int nb = ln64_blocks(rows);
  if (relu) hipLaunchKernelGGL((k_ln64_fwd<true, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)x, (const f4*)nullptr, 1, gamma, beta, (f4*)y, mean, rstd, (long)rows, eps);
  else hipLaunchKernelGGL((k_ln64_fwd<false, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)x, (const f4*)nullptr, 1, gamma, beta, (f4*)y, mean, rstd, (long)rows, eps);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

// ... with the input row formed as base[row / n] + per_n[row % n] (base [rows / n][64], per_n [n][64]); rows must be a multiple of n
extern "C" int mapdn_layernorm64_bc_forward(const float* base, const float* per_n, int32_t n, const float* gamma, const float* beta, float* y,
                                            float* mean, float* rstd, int64_t rows, float eps, int32_t relu, void* stream) {
  using namespace mapdn;
  if (!base || !per_n || n < 1 || !gamma || !beta || !y || !mean || !rstd || rows < 1 || rows % n) return MAPDN_E_INVALID;
  const int nb = ln64_blocks(rows);
  if (relu) hipLaunchKernelGGL((k_ln64_fwd<true, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)base, (const f4*)per_n, (int)n, gamma, beta, (f4*)y, mean, rstd, (long)rows, eps);
  else hipLaunchKernelGGL((k_ln64_fwd<false, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)base, (const f4*)per_n, (int)n, gamma, beta, (f4*)y, mean, rstd, (long)rows, eps);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int mapdn_layernorm64_backward_blocks(int64_t rows) { return rows < 1 ? 0 : ln64_blocks(rows); }

extern "C" int mapdn_layernorm64_backward(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                                          const float* rstd, float* dx, float* dgamma, float* dbeta, float* partial, int64_t rows,
                                          int32_t relu, void* stream) {
  using namespace mapdn;
  if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !partial || rows < 1) return MAPDN_E_INVALID;
  const int nb = ln64_blocks(rows);
  if (relu) hipLaunchKernelGGL((k_ln64_bwd<true, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)dy, (const f4*)x, (const f4*)nullptr, 1, gamma, beta, mean, rstd, (f4*)dx, partial, (long)rows);
  else hipLaunchKernelGGL((k_ln64_bwd<false, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)dy, (const f4*)x, (const f4*)nullptr, 1, gamma, beta, mean, rstd, (f4*)dx, partial, (long)rows);
  hipLaunchKernelGGL(k_ln64_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nb, dgamma, dbeta);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

// backward of mapdn_layernorm64_bc_forward: dx [rows][64] is the gradient wrt the FORMED rows (the caller sums it over n / over rows / n
// for base / per_n), dgamma / dbeta as above
extern "C" int mapdn_layernorm64_bc_backward(const float* dy, const float* base, const float* per_n, int32_t n, const float* gamma, const float* beta,
                                             const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, float* partial,
                                             int64_t rows, int32_t relu, void* stream) {
  using namespace mapdn;
  if (!dy || !base || !per_n || n < 1 || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !partial || rows < 1 || rows % n) return MAPDN_E_INVALID;
  const int nb = ln64_blocks(rows);
  if (relu) hipLaunchKernelGGL((k_ln64_bwd<true, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)dy, (const f4*)base, (const f4*)per_n, (int)n, gamma, beta, mean, rstd, (f4*)dx, partial, (long)rows);
  else hipLaunchKernelGGL((k_ln64_bwd<false, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f4*)dy, (const f4*)base, (const f4*)per_n, (int)n, gamma, beta, mean, rstd, (f4*)dx, partial, (long)rows);
  hipLaunchKernelGGL(k_ln64_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nb, dgamma, dbeta);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int mapdn_relu_dot64_forward(const float* pre, const float* w, float bias, float* v, int64_t rows, void* stream) {
  using namespace mapdn;
  if (!pre || !w || !v || rows < 1) return MAPDN_E_INVALID;
  hipLaunchKernelGGL(k_relu_dot64_fwd, dim3(ln64_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const f4*)pre, w, bias, v, (long)rows);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

// dw [64], db [64] (db[0] is the bias gradient, the rest zero); partial: mapdn_layernorm64_backward_blocks(rows) x 128 floats of scratch
extern "C" int mapdn_relu_dot64_backward(const float* dv, const float* pre, const float* w, float* dpre, float* dw, float* db, float* partial,
                                         int64_t rows, void* stream) {
  using namespace mapdn;
  if (!dv || !pre || !w || !dpre || !dw || !db || !partial || rows < 1) return MAPDN_E_INVALID;
  const int nb = ln64_blocks(rows);
  hipLaunchKernelGGL(k_relu_dot64_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, dv, (const f4*)pre, w, (f4*)dpre, partial, (long)rows);
  hipLaunchKernelGGL(k_ln64_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nb, dw, db);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

// launch shape of k_policy_fwd for an observation width: threads per workgroup (512 / 256), whether the one-hot id columns of
// fc1 are staged in LDS, and the dynamic LDS — or false when no variant fits the 160 KB of a CU (very wide observations)
static bool policy_geometry(int obs_dim, int id_dim, int& pt, int& ids_lds, size_t& lds) {
  using namespace mapdn;
  const int kc1 = (obs_dim + 15) / 16;
  auto lds_for = [&](int p, int il) { return ((size_t)4 * kc1 * 64 + 2 * 12 * 4 * 64) * 16 + ((size_t)(il ? id_dim : 0) * PH + 4 * PH + 2 * 3 * PH + (p / 64) * 16 * 68) * sizeof(float); };
  for (int p : {512, 256})
    for (int il : {1, 0})
      if (lds_for(p, il) <= (size_t)160 * 1024) { pt = p; ids_lds = il; lds = lds_for(p, il); return true; }
  return false;
}

extern "C" int mapdn_policy_forward_fits(int32_t obs_dim, int32_t id_dim) {
  int pt, il; size_t lds;
  return obs_dim >= 1 && id_dim >= 0 && policy_geometry(obs_dim, id_dim, pt, il, lds) ? 1 : 0;
}

static int policy_forward_launch(const float* obs, const float* hid_in, const float* w1, const float* b1, const float* ln_g,
                                 const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                 const float* w2, const float* b2, float* means, float* hid_out, float* x1_out, int32_t rows, int32_t n_agents,
                                 int32_t obs_dim, int32_t id_dim, float ln_eps, void* stream) {
  using namespace mapdn;
  if (!obs || !hid_in || !w1 || !means || rows < 1 || n_agents < 1 || obs_dim < 1 || id_dim < 0) return MAPDN_E_INVALID;
  int pt = 512, ids_lds = 1;
  size_t lds = 0;
  if (!policy_geometry(obs_dim, id_dim, pt, ids_lds, lds)) return MAPDN_E_INVALID;   // callers ask mapdn_policy_forward_fits first
  // (per call, not once per process: the attribute belongs to the current device)
  static const bool v1 = [] { const char* e = getenv("MAPDN_POLICY_FWD_V1"); return e && atoi(e) != 0; }();   // A/B: the rounds-2-5 form
  const void* fn = v1 ? (pt == 512 ? (const void*)k_policy_fwd<512> : (const void*)k_policy_fwd<256>)
                      : (pt == 512 ? (const void*)k_policy_fwd2<512> : (const void*)k_policy_fwd2<256>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MAPDN_E_HIP;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int tiles = (rows + 15) / 16;
  const int blocks = std::min((tiles + pt / 64 - 1) / (pt / 64), cus);   // one resident workgroup per CU (its LDS is the parameter set)
#define MAPDN_POLICY_LAUNCH(K, P)                                                                                                         \
  hipLaunchKernelGGL((K<P>), dim3(blocks), dim3(P), lds, (hipStream_t)stream, obs, hid_in, w1, b1, ln_g, ln_b, w_ih, w_hh, b_ih, b_hh, w2, b2, \
                     means, hid_out, rows, n_agents, obs_dim, id_dim, ln_eps, ids_lds, x1_out)
  if (v1) { if (pt == 512) MAPDN_POLICY_LAUNCH(k_policy_fwd, 512); else MAPDN_POLICY_LAUNCH(k_policy_fwd, 256); }
  else { if (pt == 512) MAPDN_POLICY_LAUNCH(k_policy_fwd2, 512); else MAPDN_POLICY_LAUNCH(k_policy_fwd2, 256); }
#undef MAPDN_POLICY_LAUNCH
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int mapdn_policy_forward(const float* obs, const float* hid_in, const float* w1, const float* b1, const float* ln_g,
                                    const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                    const float* w2, const float* b2, float* means, float* hid_out, int32_t rows, int32_t n_agents,
                                    int32_t obs_dim, int32_t id_dim, float ln_eps, void* stream) {
  return policy_forward_launch(obs, hid_in, w1, b1, ln_g, ln_b, w_ih, w_hh, b_ih, b_hh, w2, b2, means, hid_out, nullptr, rows, n_agents, obs_dim,
                               id_dim, ln_eps, stream);
}

// the same launch for the learner's TRAINING-time forward: also writes x1 [rows, 64] — the LayerNorm input fc1(obs) + b1 + id column, which
// mapdn_policy_backward starts from — and skips the hidden-state output when hid_out is NULL
extern "C" int mapdn_policy_forward_train(const float* obs, const float* hid_in, const float* w1, const float* b1, const float* ln_g,
                                          const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                          const float* w2, const float* b2, float* means, float* hid_out, float* x1_out, int32_t rows,
                                          int32_t n_agents, int32_t obs_dim, int32_t id_dim, float ln_eps, void* stream) {
  if (!x1_out) return MAPDN_E_INVALID;
  return policy_forward_launch(obs, hid_in, w1, b1, ln_g, ln_b, w_ih, w_hh, b_ih, b_hh, w2, b2, means, hid_out, x1_out, rows, n_agents, obs_dim,
                               id_dim, ln_eps, stream);
}
