// critic.hip — the learner's critic trunk on rows of 64 as ONE forward and ONE backward launch, gfx950 (MI355X).
//
// Reference: critics/mlp_critic.py:22-36 (fc1 -> LayerNorm -> ReLU -> fc2 -> ReLU -> fc3) as evaluated by models/maddpg.py:35-79 /
// models/iddpg.py:32-58 on [batch x agents] rows and trained by learning_algorithms/ddpg.py:15-39.  What follows the first layer —
//     v = relu( relu(LayerNorm(x)) W2^T + b2 ) . w3 + b3
// — ran as LayerNorm kernel + hipBLASLt GEMM + relu-dot kernel forward and five launches backward, every one of them streaming
// the [rows, 64] activations (2.7 GB at the end-to-end configuration's 10 M rows) through HBM at K = 64.  Here a wavefront owns
// tiles of 16 rows and keeps them in registers from the LayerNorm input to v (forward) and from dv to dx and the parameter
// gradients (backward, which recomputes the forward instead of reading saved activations): HBM sees x (or, for the central
// critic, only the two small operands the row is FORMED from: base[row / n] + per_n[row % n]), dv and the outputs.
//
// All products run on v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf chain: the reference's fp32 modules).  Layouts, for a tile of 16
// rows and lane l = (j = l & 15, g = l >> 4):
//   "A layout"  lane holds row j, features 16 c + 4 g + q (c, q = 0..3): 4 x float4.  This is the MFMA *B* operand of a product
//               whose OUTPUT is transposed, D[i = feature][j = row]: pre^T = W2 xn^T and dxn^T = W2^T dpre^T take the
//               activations as B and the (LDS- or register-resident) weight as A, and their D registers
//               (row j, feature 16 nt + 4 g + r) are again in A layout — the chain x -> xn -> pre -> dpre -> dxn -> dx never
//               leaves the lane's row, and the LayerNorm / dot reductions are 16 in-lane adds + two row swaps
//               (v_permlane16_swap / v_permlane32_swap).
//   "C layout"  lane holds rows 4 g + s, feature 16 nt + j: needed only by dW2 += dpre^T xn (the contraction runs over ROWS),
//               reached through a per-wave 16 x 64 LDS tile.
// dW2 / dgamma / dbeta / db2 / dw3 / db3 (and, for formed rows, dper_n) are accumulated per wavefront over its contiguous range of
// rows, summed across the wavefronts of a workgroup through LDS and across workgroups by a second small launch, both in a fixed
// order: deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/mapdn.h"
#include "rowtile.hpp"

namespace mapdn {

// the lane's row in A layout: x[row][16 c + 4 g ..] or base[row / n] + per_n[row % n] (one f32 add, as the broadcast add would)
template <bool BC>
__device__ __forceinline__ void load_row(const HeadArgs& p, long row, int g, f4 (&xa)[4]) {
  if (BC) {
    const unsigned q = (unsigned)row / (unsigned)p.n, i = (unsigned)row - q * (unsigned)p.n;
    const float* pb = p.x + (size_t)q * 64 + 4 * g;
    const float* pn = p.per_n + (size_t)i * 64 + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) xa[c] = *(const f4*)(pb + 16 * c) + *(const f4*)(pn + 16 * c);
  } else {
    const float* px = p.x + (size_t)row * 64 + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) xa[c] = *(const f4*)(px + 16 * c);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: v[row] = relu(relu(LN(x)) W2^T + b2) . w3 + b3.  W2 lives in registers as the A operand (64 VGPRs), 64 MFMAs per tile.
template <bool BC>
__global__ void __launch_bounds__(256)
k_head_fwd(HeadArgs p, float* __restrict__ v, long rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  f4 wop[4][4], gam[4], bet[4], b2v[4], w3v[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wop[nt][c] = *(const f4*)(p.w2 + (size_t)(16 * nt + j) * 64 + 16 * c + 4 * g);
    gam[nt] = *(const f4*)(p.gamma + 16 * nt + 4 * g); bet[nt] = *(const f4*)(p.beta + 16 * nt + 4 * g);
    b2v[nt] = *(const f4*)(p.b2 + 16 * nt + 4 * g); w3v[nt] = *(const f4*)(p.w3 + 16 * nt + 4 * g);
  }
  const float b3 = p.b3[0];
  const long n_tiles = (rows + 15) >> 4;
  for (long T = (long)blockIdx.x * 4 + wave; T < n_tiles; T += (long)gridDim.x * 4) {
    const long row = T * 16 + j;
    f4 xa[4];
    load_row<BC>(p, row < rows ? row : rows - 1, g, xa);
    ln_stats(xa, p.eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f4 y = xa[c] * gam[c] + bet[c];
      xa[c] = f4{relu_nan(y.x), relu_nan(y.y), relu_nan(y.z), relu_nan(y.w)};
    }
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wop[nt][c][q], xa[c][q], acc[nt], 0, 0, 0);
    float dot = 0.0f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dot = fmaf(relu_nan(acc[nt][r] + b2v[nt][r]), w3v[nt][r], dot);
    dot = sum_g(dot);
    if (g == 0 && row < rows) v[row] = dot + b3;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward.  MODE 0: dx (or, for formed rows, dbase = sum over the n rows of a group, and dper_n) + every parameter gradient;
//            MODE 1: dx / dbase / dper_n only (the parameters do not require grad);
//            MODE 2: dact[row] = dx[row] . dot_w[row % n] only (the policy update through the central critic: the own-action column
//                    of fc1 — models/maddpg.py:52-58 — is the only path back to the policy);
//            MODE 3: MODE 0 with the value loss fused in (learning_algorithms/ddpg.py:36-38, models/maddpg.py:122-124):
//                    loss = sum_rows w[row] (ret[row] - v[row])^2, w[row] = scale[0] * wrow[row / n] (wrow may be null: 1), and dv is
//                    formed in the kernel as -2 w (ret - v) — the forward launch and its product are not needed at all.
// A wavefront owns a contiguous range of rows (whole groups of n for formed rows: the sum over a group never crosses wavefronts).
// NT threads per workgroup: 256 = one wavefront per SIMD with the whole 512-register file (parameters in registers, the next tile
// prefetched, two staging tiles); 512 = TWO wavefronts per SIMD, each within 256 registers (parameters read from LDS where used, no
// prefetch, one staging tile used twice) — one wavefront's LayerNorm / staging / reduction VALU work then runs under the other's MFMAs.
template <bool BC, int MODE, int NT>
__global__ void __launch_bounds__(NT)
k_head_bwd(HeadArgs p, const float* __restrict__ dv, float* __restrict__ dx, const float* __restrict__ dot_w, float* __restrict__ dact,
           float* __restrict__ partial, int pstride, long rows, const float* __restrict__ wrow, const float* __restrict__ scale) {
  constexpr bool PG = MODE == 0 || MODE == 3;        // parameter gradients wanted
  constexpr int NW = NT / 64, TILES = NT == 512 ? 1 : 2;
  constexpr bool SLIM = NT == 512;
  extern __shared__ float sm[];
  f4* sW = (f4*)sm;                         // [4 nt][4 c][64]: W2[16 nt + j][16 c + 4 g + q]       A operand of pre^T = W2 xn^T
  f4* sWT = sW + 1024;                      // [4 nt][4 c][64]: W2[16 c + 4 g + q][16 nt + j]       A operand of dxn^T = W2^T dpre^T
  float* sP = (float*)(sWT + 1024);         // gamma | beta | b2 | w3 [4][64]
  float* stage = sP + 256 + (size_t)(threadIdx.x >> 6) * TILES * 16 * HS;                // TILES 16 x HS tiles per wavefront
  float* accn = sP + 256 + (size_t)NW * TILES * 16 * HS + (size_t)(threadIdx.x >> 6) * p.n * 64;   // [n][64] per wavefront (BC, MODE != 2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
  if (tid < 64) { sP[tid] = p.gamma[tid]; sP[64 + tid] = p.beta[tid]; sP[128 + tid] = p.b2[tid]; sP[192 + tid] = p.w3[tid]; }
  for (int i = tid; i < 1024; i += NT) {
    const int l = i & 63, c = (i >> 6) & 3, nt = i >> 8, lj = l & 15, lg = l >> 4;
    sW[i] = *(const f4*)(p.w2 + (size_t)(16 * nt + lj) * 64 + 16 * c + 4 * lg);
    f4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = p.w2[(size_t)(16 * c + 4 * lg + q) * 64 + 16 * nt + lj];
    sWT[i] = t;
  }
  if (BC && MODE != 2) for (int i = lane; i < p.n * 64; i += 64) accn[i] = 0.0f;
  f4 gam_[4], bet_[4], b2v_[4], w3v_[4];     // (dead in the SLIM build: the accessors below read LDS there)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    gam_[nt] = *(const f4*)(p.gamma + 16 * nt + 4 * g); bet_[nt] = *(const f4*)(p.beta + 16 * nt + 4 * g);
    b2v_[nt] = *(const f4*)(p.b2 + 16 * nt + 4 * g); w3v_[nt] = *(const f4*)(p.w3 + 16 * nt + 4 * g);
  }
  __syncthreads();
  auto GAM = [&](int c) -> f4 { return SLIM ? *(const f4*)(sP + 16 * c + 4 * g) : gam_[c]; };
  auto BET = [&](int c) -> f4 { return SLIM ? *(const f4*)(sP + 64 + 16 * c + 4 * g) : bet_[c]; };
  auto B2V = [&](int c) -> f4 { return SLIM ? *(const f4*)(sP + 128 + 16 * c + 4 * g) : b2v_[c]; };
  auto W3V = [&](int c) -> f4 { return SLIM ? *(const f4*)(sP + 192 + 16 * c + 4 * g) : w3v_[c]; };

  // this wavefront's range of rows [r0, r1)
  const long W = (long)gridDim.x * NW, w = (long)blockIdx.x * NW + wave;
  long r0, r1;
  if (BC) { const long G = rows / p.n; r0 = (G * w / W) * p.n; r1 = (G * (w + 1) / W) * p.n; }
  else { const long Tn = (rows + 15) >> 4; r0 = (Tn * w / W) * 16; r1 = (Tn * (w + 1) / W) * 16; if (r1 > rows) r1 = rows; }

  f4 accW[4][4];                            // dW2[16 ntu + 4 g + r][16 ntk + j]
  f4 ag[4], ab[4], aw3[4], ab2[4];          // column sums in A layout (this lane's row j only): dgamma, dbeta, dw3, db2
  float ab3 = 0.0f, carry = 0.0f;           // db3; running sum of dx over the rows of the current group (lane = column)
  float aloss = 0.0f;
  const float b3 = MODE == 3 ? p.b3[0] : 0.0f, lscale = MODE == 3 ? scale[0] : 0.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) accW[a][b] = f4{0, 0, 0, 0};
    ag[a] = ab[a] = aw3[a] = ab2[a] = f4{0, 0, 0, 0};
  }

  // the rows of the NEXT tile are requested while the current one is in the matrix cores (one wavefront per SIMD: nobody else hides
  // the HBM / L2 latency)
  f4 xnext[4];
  float dvnext = 0.0f;
  if (!SLIM && r0 < r1) {
    const long row = r0 + j;
    load_row<BC>(p, row < r1 ? row : r1 - 1, g, xnext);
    dvnext = row < r1 ? dv[row] : 0.0f;
  }
  for (long row0 = r0; row0 < r1; row0 += 16) {
    const long row = row0 + j;
    const bool valid = row < r1;
    f4 xh[4];
    float dvcur;
    if (SLIM) {
      load_row<BC>(p, valid ? row : r1 - 1, g, xh);
      dvcur = valid ? dv[row] : 0.0f;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) xh[c] = xnext[c];
      dvcur = dvnext;
      if (row0 + 16 < r1) {
        const long rn = row + 16;
        load_row<BC>(p, rn < r1 ? rn : r1 - 1, g, xnext);
        dvnext = rn < r1 ? dv[rn] : 0.0f;
      }
    }
    const float rs = ln_stats(xh, p.eps);
    f4 xn[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f4 y = xh[c] * GAM(c) + BET(c);
      xn[c] = f4{relu_nan(y.x), relu_nan(y.y), relu_nan(y.z), relu_nan(y.w)};
    }
    // ---- pre^T = W2 xn^T
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f4 wv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wv[nt] = sW[(nt * 4 + c) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][q], xn[c][q], acc[nt], 0, 0, 0);
    }
    // ---- dpre = [pre > 0] dv w3 (in place of acc); dw3 += relu(pre) dv; db2 += dpre; db3 += dv
    float dvr = dvcur;                       // (MODE 3: dv holds the returns)
    if (MODE == 3) {
      float dot = 0.0f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f4 bb = B2V(nt), ww = W3V(nt);
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(relu_nan(acc[nt][r] + bb[r]), ww[r], dot);
      }
      const float err = dvcur - (sum_g(dot) + b3);
      float wgt = lscale;
      if (wrow) { const unsigned rc = (unsigned)(valid ? row : r1 - 1); wgt *= wrow[BC ? rc / (unsigned)p.n : rc]; }
      dvr = valid ? -2.0f * wgt * err : 0.0f;
      if (valid && g == 0) aloss = fmaf(wgt * err, err, aloss);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f4 bb = B2V(nt), ww = W3V(nt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = acc[nt][r] + bb[r];
        const bool pos = pre > 0.0f;
        const float dp = pos ? dvr * ww[r] : 0.0f;
        if (PG) { aw3[nt][r] = fmaf(pos ? pre : 0.0f, dvr, aw3[nt][r]); ab2[nt][r] += dp; }
        acc[nt][r] = dp;
      }
    }
    if (PG && g == 0) ab3 += dvr;
    // ---- dxn^T = W2^T dpre^T
    f4 dxn[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f4 wv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wv[nt] = sWT[(nt * 4 + c) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dxn[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][q], acc[c][q], dxn[nt], 0, 0, 0);
    }
    // ---- dW2 += dpre^T xn: both operands through LDS into C layout (rows on the contraction axis)
    if (PG) {
      float* s0 = stage; float* s1 = SLIM ? stage : stage + 16 * HS;
#pragma unroll
      for (int c = 0; c < 4; ++c) *(f4*)(s0 + j * HS + 16 * c + 4 * g) = acc[c];
      if (!SLIM) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f4*)(s1 + j * HS + 16 * c + 4 * g) = xn[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float dCs[4][4];                      // SLIM: dpre in C layout for all four steps, read before the one tile is reused for xn
      if (SLIM) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) dCs[s][nt] = s0[(4 * g + s) * HS + 16 * nt + j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f4*)(s1 + j * HS + 16 * c + 4 * g) = xn[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float dC[4], xC[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { dC[nt] = SLIM ? dCs[s][nt] : s0[(4 * g + s) * HS + 16 * nt + j]; xC[nt] = s1[(4 * g + s) * HS + 16 * nt + j]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) accW[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(dC[a], xC[b], accW[a][b], 0, 0, 0);
      }
    }
    // ---- LayerNorm backward on the lane's row: d = [y > 0] dxn, a = d gamma, dx = rstd (a - mean(a) - xhat mean(a xhat))
    float s1a = 0.0f, s2a = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f4 d;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = xn[c][q] > 0.0f ? dxn[c][q] : 0.0f;
      if (PG) { ag[c] += d * xh[c]; ab[c] += d; }
      const f4 a = d * GAM(c);
      s1a += hsum(a); s2a += hsum(a * xh[c]);
      dxn[c] = a;
    }
    const float m1 = sum_g(s1a) * (1.0f / 64.0f), m2 = sum_g(s2a) * (1.0f / 64.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c) dxn[c] = (dxn[c] - m1 - xh[c] * m2) * rs;
    // ---- hand dx over
    if (MODE == 2) {
      const unsigned i = BC ? (unsigned)((valid ? row : r1 - 1) - r0) % (unsigned)p.n : 0u;
      const float* pw = dot_w + (size_t)i * 64 + 4 * g;
      float d = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) d += hsum(dxn[c] * *(const f4*)(pw + 16 * c));
      d = sum_g(d);
      if (g == 0 && valid) dact[row] = d;
    } else if (!BC) {
      if (valid) {
        float* px = dx + (size_t)row * 64 + 4 * g;
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f4*)(px + 16 * c) = dxn[c];
      }
    } else {
      // formed rows: dbase[group] = sum of dx over the group's n rows, dper_n[i] += dx — lane = column, the tile's rows in order
      float* s0 = stage;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // (the dW2 reads of this tile are done)
#pragma unroll
      for (int c = 0; c < 4; ++c) *(f4*)(s0 + j * HS + 16 * c + 4 * g) = dxn[c];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int cnt = (int)(r1 - row0 < 16 ? r1 - row0 : 16);
      const unsigned rel = (unsigned)(row0 - r0);
      unsigned grp = rel / (unsigned)p.n, i = rel - grp * (unsigned)p.n;
      float* pbase = dx + ((size_t)(r0 / p.n) + grp) * 64 + lane;
      if (p.n >= 16) {
        unsigned i2 = i;                      // (i stays the tile's first agent index for the slot addresses)
        // the tile's 16 rows belong to 16 DIFFERENT agents: their dper_n slots are independent — 16 reads, 16 adds, 16 writes in flight
        // instead of 16 dependent LDS round trips
        constexpr int H = SLIM ? 4 : 16;      // rows in flight (the SLIM build has 256 registers: four at a time)
#pragma unroll
        for (int h0 = 0; h0 < 16; h0 += H) {
          float val[H], cur[H];
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            unsigned ii = i + h0 + rr; if (ii >= (unsigned)p.n) ii -= p.n;
            val[rr] = h0 + rr < cnt ? s0[(h0 + rr) * HS + lane] : 0.0f;
            cur[rr] = accn[ii * 64 + lane];
          }
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            unsigned ii = i + h0 + rr; if (ii >= (unsigned)p.n) ii -= p.n;
            accn[ii * 64 + lane] = cur[rr] + val[rr];
          }
#pragma unroll
          for (int rr = 0; rr < H; ++rr)
            if (h0 + rr < cnt) {
              carry += val[rr];
              if (++i2 == (unsigned)p.n) { *pbase = carry; carry = 0.0f; i2 = 0; pbase += 64; }
            }
        }
      } else {
        for (int rr = 0; rr < cnt; ++rr) {
          const float val = s0[rr * HS + lane];
          carry += val;
          accn[i * 64 + lane] += val;
          if (++i == (unsigned)p.n) { *pbase = carry; carry = 0.0f; i = 0; pbase += 64; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- partial sums: the wavefronts of a workgroup are summed HERE, through LDS (the weights and staging tiles are dead by now), in a
  // fixed order, so that the second launch reads one partial per workgroup (256) instead of one per wavefront (2048)
  if (MODE != 2) {
    float* pp = partial + (size_t)blockIdx.x * pstride;
    if (BC) {                                 // dper_n: the per-wavefront [n][64] accumulators are already in LDS
      __syncthreads();
      const float* a0 = sP + 256 + (size_t)NW * TILES * 16 * HS;
      for (int i = tid; i < p.n * 64; i += NT) {
        float t = a0[i];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) t += a0[(size_t)ww * p.n * 64 + i];
        pp[HP + i] = t;
      }
    }
    if (PG) {
      __syncthreads();                        // every wavefront is out of its tile loop (and done with accn)
      float* red = sm + (size_t)wave * HP;    // [NW][HP] floats from the start of LDS (the launch sizes LDS for it)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(16 * a + 4 * g + r) * 64 + 16 * b + j] = accW[a][b][r];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tg = sum_j(ag[c][q]), tb = sum_j(ab[c][q]), t2 = sum_j(ab2[c][q]), t3 = sum_j(aw3[c][q]);
          if (j == 0) {
            const int col = 16 * c + 4 * g + q;
            red[4096 + col] = tg; red[4160 + col] = tb; red[4224 + col] = t2; red[4288 + col] = t3;
          }
        }
      const float t = sum_j(ab3), tl = sum_j(aloss);           // (lanes with g != 0 hold 0)
      if (lane == 0) { red[4352] = t; red[4353] = tl; }
      __syncthreads();
      for (int col = tid; col < 4354; col += NT) {
        float v = sm[col];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v += sm[(size_t)ww * HP + col];
        pp[col] = v;
      }
    }
  }
}

// out[col] = sum over wavefronts of partial[w][col] in a fixed order: four strided sub-sums, then those in order
__global__ void __launch_bounds__(256) k_head_reduce(const float* __restrict__ partial, int nw, int pstride, int lo, int hi, float* __restrict__ out) {
  __shared__ float s_acc[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, col = lo + blockIdx.x * 64 + c;
  float acc = 0.0f;
  if (col < hi) for (int i = grp; i < nw; i += 4) acc += partial[(size_t)i * pstride + col];
  s_acc[grp][c] = acc;
  __syncthreads();
  if (grp == 0 && col < hi) out[col] = (s_acc[0][c] + s_acc[1][c]) + (s_acc[2][c] + s_acc[3][c]);
}

}  // namespace mapdn

static int head_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus;
}
// workgroups of the backward launch: one per CU (its wavefronts keep dW2 in registers, the weights in LDS), never more wavefronts
// than units of work (groups of n formed rows / tiles of 16 rows); nw = wavefronts per workgroup (4, or 8 for the 512-thread build)
static int head_bwd_blocks(int64_t rows, int32_t n, bool bc, int nw) {
  const int64_t units = bc ? rows / n : (rows + 15) / 16;
  return (int)std::max<int64_t>(1, std::min<int64_t>((units + nw - 1) / nw, head_cus()));
}
static size_t head_bwd_lds(int threads, int32_t n, bool accn) {
  const int nw = threads / 64, tiles = threads == 512 ? 1 : 2;
  return (size_t)2 * 1024 * 16 + 256 * 4 + (size_t)nw * tiles * 16 * mapdn::HS * 4 + (accn ? (size_t)nw * n * 64 * 4 : 0);
}
// 512 threads (two wavefronts per SIMD) when its LDS fits a CU and the batch is big enough to give every wavefront work;
// MAPDN_HEAD_BWD_THREADS = 256 / 512 forces one (A/B measurements)
static int head_bwd_threads(int64_t rows, int32_t n, bool bc, bool accn) {
  const char* e = getenv("MAPDN_HEAD_BWD_THREADS");
  const bool fits = head_bwd_lds(512, n, accn) <= (size_t)160 * 1024;
  if (e && atoi(e) == 256) return 256;
  if (e && atoi(e) == 512 && fits) return 512;
  const int64_t units = bc ? rows / n : (rows + 15) / 16;
  return fits && units >= (int64_t)head_cus() * 8 ? 512 : 256;
}
static bool head_args_ok(const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta, const float* w2, const float* b2,
                         const float* w3, const float* b3, int64_t rows) {
  if (!x || !gamma || !beta || !w2 || !b2 || !w3 || !b3 || rows < 1 || rows > 0x7fffffff) return false;
  if (per_n && (n < 1 || n > 256 || rows % n)) return false;
  return true;
}

extern "C" int mapdn_critic_head_forward(const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta, float eps,
                                         const float* w2, const float* b2, const float* w3, const float* b3, float* v, int64_t rows,
                                         void* stream) {
  using namespace mapdn;
  if (!head_args_ok(x, per_n, n, gamma, beta, w2, b2, w3, b3, rows) || !v) return MAPDN_E_INVALID;
  const HeadArgs a{x, per_n, per_n ? n : 1, gamma, beta, eps, w2, b2, w3, b3};
  const int64_t tiles = (rows + 15) / 16;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, (int64_t)head_cus() * 2));
  if (per_n) hipLaunchKernelGGL(k_head_fwd<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, v, (long)rows);
  else hipLaunchKernelGGL(k_head_fwd<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, v, (long)rows);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}

extern "C" int64_t mapdn_critic_head_scratch_floats(int64_t rows, int32_t n, int32_t formed) {
  if (rows < 1 || (formed && (n < 1 || rows % n))) return 0;
  const int64_t blocks = std::max<int64_t>(head_bwd_blocks(rows, n, formed != 0, 4), head_bwd_blocks(rows, n, formed != 0, 8));
  return blocks * (mapdn::HP + (formed ? n * 64 : 0));       // one partial per workgroup (whichever launch shape the backward picks)
}

template <bool BC, int MODE, int NT>
static int head_bwd_launch_nt(const mapdn::HeadArgs& a, const float* dv, float* dx, const float* dot_w, float* dact, float* scratch, float* grads,
                              int64_t rows, hipStream_t st, const float* wrow, const float* scale) {
  using namespace mapdn;
  constexpr int NW = NT / 64;
  const int blocks = head_bwd_blocks(rows, a.n, BC, NW), nw = blocks;          // one partial per workgroup
  const int pstride = HP + (BC ? a.n * 64 : 0);
  size_t lds = head_bwd_lds(NT, a.n, BC && MODE != 2);
  if (MODE == 0 || MODE == 3) lds = std::max(lds, (size_t)NW * HP * 4);         // the end-of-kernel reduction across wavefronts
  if (lds > (size_t)160 * 1024) return MAPDN_E_INVALID;
  const void* fn = (const void*)k_head_bwd<BC, MODE, NT>;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MAPDN_E_HIP;
  hipLaunchKernelGGL((k_head_bwd<BC, MODE, NT>), dim3(blocks), dim3(NT), lds, st, a, dv, dx, dot_w, dact, scratch, pstride, (long)rows, wrow, scale);
  if (MODE == 0 || MODE == 3) hipLaunchKernelGGL(k_head_reduce, dim3((HP + 63) / 64), dim3(256), 0, st, scratch, nw, pstride, 0, HP, grads);
  if (BC && MODE != 2) hipLaunchKernelGGL(k_head_reduce, dim3(a.n), dim3(256), 0, st, scratch, nw, pstride, HP, HP + a.n * 64, grads);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}
template <bool BC, int MODE>
static int head_bwd_launch(const mapdn::HeadArgs& a, const float* dv, float* dx, const float* dot_w, float* dact, float* scratch, float* grads,
                           int64_t rows, hipStream_t st, const float* wrow = nullptr, const float* scale = nullptr) {
  if (head_bwd_threads(rows, a.n, BC, BC && MODE != 2) == 512)
    return head_bwd_launch_nt<BC, MODE, 512>(a, dv, dx, dot_w, dact, scratch, grads, rows, st, wrow, scale);
  return head_bwd_launch_nt<BC, MODE, 256>(a, dv, dx, dot_w, dact, scratch, grads, rows, st, wrow, scale);
}

extern "C" int mapdn_critic_head_backward(const float* dv, const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta,
                                          float eps, const float* w2, const float* b2, const float* w3, const float* b3, float* dx, float* grads,
                                          float* scratch, int64_t rows, int32_t param_grads, void* stream) {
  using namespace mapdn;
  if (!head_args_ok(x, per_n, n, gamma, beta, w2, b2, w3, b3, rows) || !dv || !dx || ((param_grads || per_n) && (!grads || !scratch)))
    return MAPDN_E_INVALID;
  const HeadArgs a{x, per_n, per_n ? n : 1, gamma, beta, eps, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  if (per_n) return param_grads ? head_bwd_launch<true, 0>(a, dv, dx, nullptr, nullptr, scratch, grads, rows, st)
                                : head_bwd_launch<true, 1>(a, dv, dx, nullptr, nullptr, scratch, grads, rows, st);
  return param_grads ? head_bwd_launch<false, 0>(a, dv, dx, nullptr, nullptr, scratch, grads, rows, st)
                     : head_bwd_launch<false, 1>(a, dv, dx, nullptr, nullptr, scratch, grads, rows, st);
}

extern "C" int mapdn_critic_head_backward_dot(const float* dv, const float* x, const float* per_n, int32_t n, const float* gamma,
                                              const float* beta, float eps, const float* w2, const float* b2, const float* w3, const float* b3,
                                              const float* dot_w, float* dact, int64_t rows, void* stream) {
  using namespace mapdn;
  if (!head_args_ok(x, per_n, n, gamma, beta, w2, b2, w3, b3, rows) || !dv || !dot_w || !dact) return MAPDN_E_INVALID;
  const HeadArgs a{x, per_n, per_n ? n : 1, gamma, beta, eps, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  return per_n ? head_bwd_launch<true, 2>(a, dv, nullptr, dot_w, dact, nullptr, nullptr, rows, st)
               : head_bwd_launch<false, 2>(a, dv, nullptr, dot_w, dact, nullptr, nullptr, rows, st);
}

// value loss and every gradient of it in one launch (no forward launch): loss = sum_rows scale[0] wrow[row / n] (ret[row] - v[row])^2
// (wrow may be NULL: weight 1) -> grads[4353]; dx / dbase, grads as mapdn_critic_head_backward with param_grads = 1
extern "C" int mapdn_critic_head_mse(const float* ret, const float* wrow, const float* scale, const float* x, const float* per_n, int32_t n,
                                     const float* gamma, const float* beta, float eps, const float* w2, const float* b2, const float* w3,
                                     const float* b3, float* dx, float* grads, float* scratch, int64_t rows, void* stream) {
  using namespace mapdn;
  if (!head_args_ok(x, per_n, n, gamma, beta, w2, b2, w3, b3, rows) || !ret || !scale || !dx || !grads || !scratch) return MAPDN_E_INVALID;
  const HeadArgs a{x, per_n, per_n ? n : 1, gamma, beta, eps, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  return per_n ? head_bwd_launch<true, 3>(a, ret, dx, nullptr, nullptr, scratch, grads, rows, st, wrow, scale)
               : head_bwd_launch<false, 3>(a, ret, dx, nullptr, nullptr, scratch, grads, rows, st, wrow, scale);
}
