// policy_bwd.hip — backward of the shared recurrent agent's trunk for the learner's policy update, gfx950 (MI355X).
//
// Reference: agents/rnn_agent.py:16-32 (LayerNorm -> ReLU -> GRUCell -> fc2) as trained through models/maddpg.py:103-125 /
// learning_algorithms/ddpg.py:15-39 on [batch x agents] rows (10 M at the end-to-end configuration).  The PyTorch route runs the two
// GRU projections as GEMMs that write [rows, 192] each, ATen's fused cell over them (+ a [rows, 320] workspace), and the same again
// backwards: ~100 GB of HBM traffic per policy update.  Here the forward is mapdn_policy_forward_train (policy.hip: one launch, keeps
// only x1 = the LayerNorm input) and the backward is ONE launch that recomputes the trunk from x1 and the stored hidden state and
// leaves exactly what the remaining weight-gradient products need:
//     dG  [rows][256] = d loss / d (gate pre-activations)  r | z | n_input | n_hidden        (torch.nn.GRUCell's decomposition)
//     xn  [rows][64]  = relu(LayerNorm(x1))                                                   (the cell's input)
//     dx1 [rows][64]  = d loss / d x1
// plus, reduced in the kernel in a fixed order (no atomics): the bias gradients (column sums of dG), dgamma, dbeta, dw2, db2.
// dW_ih = dG[:, r|z|n_i]^T xn, dW_hh = dG[:, r|z|n_h]^T h and dW1 = dx1^T [obs | id] are K = rows products over those tensors
// (learner.py runs them block-wise, as _TallLinear does).
// Layout: everything in "A layout" (rowtile.hpp): lane (j, g) holds row j of the tile, features 16 c + 4 g + q.  The gate products are
// taken TRANSPOSED — gates^T = W x^T with the weight as the MFMA A operand from LDS and the activations as B — so that their D
// registers (row j, unit 16 nt + 4 g + r) line up element by element with h, and dxn^T = W_ih^T dG^T needs no transposition either.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/mapdn.h"
#include "rowtile.hpp"

namespace mapdn {

constexpr int PBP = 512;        // partials per workgroup: db_r | db_z | db_ni | db_nh | dgamma | dbeta | dw2 [64 each] | db2 | pad

struct PolBwdArgs {
  const float* x1; const float* h; const float* dmeans;
  const float* gamma; const float* beta; float eps;
  const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh; const float* w2;
  float* dx1; float* dG; float* xn_out; float* partial; long rows;
};

__global__ void __launch_bounds__(256)
k_policy_bwd(PolBwdArgs p) {
  extern __shared__ float sm[];
  f4* sWih = (f4*)sm;                       // [12 = 3 gates x 4 nt][4 c][64]: W_ih[16 ntg + j][16 c + 4 g + q]      A operand of gates^T
  f4* sWhh = sWih + 12 * 4 * 64;            // likewise W_hh
  f4* sWT = sWhh + 12 * 4 * 64;             // [3 gates][4 nt'][4 c][64]: W_ih[64 G + 16 c + 4 g + q][16 nt' + j]    A operand of dxn^T = W_ih^T dG^T
  float* sP = (float*)(sWT + 12 * 4 * 64);  // gamma 64 | beta 64 | b_ih 192 | b_hh 192 | w2 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
  for (int i = tid; i < 12 * 4 * 64; i += 256) {
    const int l = i & 63, c = (i >> 6) & 3, ntg = i >> 8, lj = l & 15, lg = l >> 4;
    const size_t base = (size_t)(16 * ntg + lj) * 64 + 16 * c + 4 * lg;
    sWih[i] = *(const f4*)(p.w_ih + base); sWhh[i] = *(const f4*)(p.w_hh + base);
    const int G = ntg >> 2, ntp = ntg & 3;
    f4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = p.w_ih[(size_t)(64 * G + 16 * c + 4 * lg + q) * 64 + 16 * ntp + lj];
    sWT[i] = t;
  }
  if (tid < 64) { sP[tid] = p.gamma[tid]; sP[64 + tid] = p.beta[tid]; sP[512 + tid] = p.w2[tid]; }
  if (tid < 192) { sP[128 + tid] = p.b_ih[tid]; sP[320 + tid] = p.b_hh[tid]; }
  __syncthreads();

  f4 pbR[4], pbZ[4], pbNI[4], pbNH[4], ag[4], ab[4], aw2[4];      // column sums over this lane's rows
  float ab2 = 0.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a) pbR[a] = pbZ[a] = pbNI[a] = pbNH[a] = ag[a] = ab[a] = aw2[a] = f4{0, 0, 0, 0};

  const long n_tiles = (p.rows + 15) >> 4;
  for (long T = (long)blockIdx.x * 4 + wave; T < n_tiles; T += (long)gridDim.x * 4) {
    const long row = T * 16 + j;
    const bool valid = row < p.rows;
    const long rc = valid ? row : p.rows - 1;
    f4 xh[4], xn[4], hv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { xh[c] = *(const f4*)(p.x1 + (size_t)rc * 64 + 16 * c + 4 * g); hv[c] = *(const f4*)(p.h + (size_t)rc * 64 + 16 * c + 4 * g); }
    const float dm = valid ? p.dmeans[row] : 0.0f;
    const float rs = ln_stats(xh, p.eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f4 y = xh[c] * *(const f4*)(sP + 16 * c + 4 * g) + *(const f4*)(sP + 64 + 16 * c + 4 * g);
      xn[c] = f4{relu_nan(y.x), relu_nan(y.y), relu_nan(y.z), relu_nan(y.w)};
    }
    if (valid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) *(f4*)(p.xn_out + (size_t)row * 64 + 16 * c + 4 * g) = xn[c];
    }
    // ---- gate pre-activations, transposed products: r, z over [x | h], n_input over x, n_hidden over h
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
          aR[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wir[q], xn[c][q], aR[nt], 0, 0, 0);
          aZ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wiz[q], xn[c][q], aZ[nt], 0, 0, 0);
          aI[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(win[q], xn[c][q], aI[nt], 0, 0, 0);
          aR[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whr[q], hv[c][q], aR[nt], 0, 0, 0);
          aZ[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whz[q], hv[c][q], aZ[nt], 0, 0, 0);
          aH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(whn[q], hv[c][q], aH[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting all 96 weight loads of the tile: 384 registers)
      }
    // ---- gates, h', and the gate gradients (torch.nn.GRUCell): dh' = dmeans w2
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int u0 = 16 * nt + 4 * g;
      const f4 bir = *(const f4*)(sP + 128 + u0), biz = *(const f4*)(sP + 192 + u0), bin = *(const f4*)(sP + 256 + u0);
      const f4 bhr = *(const f4*)(sP + 320 + u0), bhz = *(const f4*)(sP + 384 + u0), bhn = *(const f4*)(sP + 448 + u0);
      const f4 w2v = *(const f4*)(sP + 512 + u0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = fast_sigmoid(aR[nt][r] + bir[r] + bhr[r]);
        const float zg = fast_sigmoid(aZ[nt][r] + biz[r] + bhz[r]);
        const float hnb = aH[nt][r] + bhn[r];
        const float ng = fast_tanh(fmaf(rg, hnb, aI[nt][r] + bin[r]));
        const float hu = hv[nt][r];
        const float hnew = fmaf(zg, hu - ng, ng);                 // (1 - z) n + z h
        aw2[nt][r] = fmaf(hnew, dm, aw2[nt][r]);
        const float dh = dm * w2v[r];
        const float dnp = dh * (1.0f - zg) * (1.0f - ng * ng);    // d / d (n pre-activation)
        const float dz = dh * (hu - ng) * zg * (1.0f - zg);       // d / d (z pre-activation)
        const float dr = dnp * hnb * rg * (1.0f - rg);            // d / d (r pre-activation)
        aR[nt][r] = dr; aZ[nt][r] = dz; aI[nt][r] = dnp; aH[nt][r] = dnp * rg;
        pbR[nt][r] += dr; pbZ[nt][r] += dz; pbNI[nt][r] += dnp; pbNH[nt][r] += dnp * rg;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (g == 0) ab2 += dm;
    if (valid) {
      float* pg = p.dG + (size_t)row * 256 + 4 * g;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        *(f4*)(pg + 16 * nt) = aR[nt]; *(f4*)(pg + 64 + 16 * nt) = aZ[nt]; *(f4*)(pg + 128 + 16 * nt) = aI[nt]; *(f4*)(pg + 192 + 16 * nt) = aH[nt];
      }
    }
    // ---- dxn^T = W_ir^T d_r^T + W_iz^T d_z^T + W_in^T d_ni^T
    f4 dxn[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f4 w0 = sWT[((0 + nt) * 4 + c) * 64 + lane], w1 = sWT[((4 + nt) * 4 + c) * 64 + lane], w2 = sWT[((8 + nt) * 4 + c) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dxn[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q], aR[c][q], dxn[nt], 0, 0, 0);
          dxn[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[q], aZ[c][q], dxn[nt], 0, 0, 0);
          dxn[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[q], aI[c][q], dxn[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    // ---- LayerNorm backward on the lane's row
    float s1a = 0.0f, s2a = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f4 d;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = xn[c][q] > 0.0f ? dxn[c][q] : 0.0f;
      ag[c] += d * xh[c]; ab[c] += d;
      const f4 a = d * *(const f4*)(sP + 16 * c + 4 * g);
      s1a += hsum(a); s2a += hsum(a * xh[c]);
      dxn[c] = a;
    }
    const float m1 = sum_g(s1a) * (1.0f / 64.0f), m2 = sum_g(s2a) * (1.0f / 64.0f);
    if (valid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) *(f4*)(p.dx1 + (size_t)row * 64 + 16 * c + 4 * g) = (dxn[c] - m1 - xh[c] * m2) * rs;
    }
  }

  // ---- column sums: lanes of a DPP row -> wavefronts of the workgroup (LDS) -> partial[block]
  __syncthreads();
  float* red = sm + (size_t)wave * PBP;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float t0 = sum_j(pbR[c][q]), t1 = sum_j(pbZ[c][q]), t2 = sum_j(pbNI[c][q]), t3 = sum_j(pbNH[c][q]);
      const float t4 = sum_j(ag[c][q]), t5 = sum_j(ab[c][q]), t6 = sum_j(aw2[c][q]);
      if (j == 0) {
        const int col = 16 * c + 4 * g + q;
        red[col] = t0; red[64 + col] = t1; red[128 + col] = t2; red[192 + col] = t3; red[256 + col] = t4; red[320 + col] = t5; red[384 + col] = t6;
      }
    }
  const float tb = sum_j(ab2);
  if (lane == 0) red[448] = tb;
  __syncthreads();
  for (int col = tid; col < 449; col += 256) p.partial[(size_t)blockIdx.x * PBP + col] = (sm[col] + sm[PBP + col]) + (sm[2 * PBP + col] + sm[3 * PBP + col]);
}

// out[col] = sum over workgroups of partial[b][col]: four strided sub-sums, then those in order
__global__ void __launch_bounds__(256) k_policy_bwd_reduce(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float s_acc[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
  float acc = 0.0f;
  for (int i = grp; i < nb; i += 4) acc += partial[(size_t)i * PBP + col];
  s_acc[grp][c] = acc;
  __syncthreads();
  if (grp == 0) out[col] = (s_acc[0][c] + s_acc[1][c]) + (s_acc[2][c] + s_acc[3][c]);
}

}  // namespace mapdn

static int polbwd_blocks(int64_t rows) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t tiles = (rows + 15) / 16;
  return (int)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, cus));
}

extern "C" int64_t mapdn_policy_backward_scratch_floats(int64_t rows) { return rows < 1 ? 0 : (int64_t)polbwd_blocks(rows) * mapdn::PBP; }

// small [512] = db_r | db_z | db_ni | db_nh | dgamma | dbeta | dw2 [64 each] | db2 [1] (+ pad); scratch: mapdn_policy_backward_scratch_floats(rows)
extern "C" int mapdn_policy_backward(const float* dmeans, const float* x1, const float* hid_in, const float* ln_g, const float* ln_b, float ln_eps,
                                     const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* w2, float* dx1,
                                     float* dgates, float* xn, float* small, float* scratch, int64_t rows, void* stream) {
  using namespace mapdn;
  if (!dmeans || !x1 || !hid_in || !ln_g || !ln_b || !w_ih || !w_hh || !b_ih || !b_hh || !w2 || !dx1 || !dgates || !xn || !small || !scratch ||
      rows < 1 || rows > 0x7fffffff)
    return MAPDN_E_INVALID;
  const PolBwdArgs a{x1, hid_in, dmeans, ln_g, ln_b, ln_eps, w_ih, w_hh, b_ih, b_hh, w2, dx1, dgates, xn, scratch, (long)rows};
  const int blocks = polbwd_blocks(rows);
  const size_t lds = (size_t)3 * 12 * 4 * 64 * 16 + 576 * 4;
  if (hipFuncSetAttribute((const void*)k_policy_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MAPDN_E_HIP;
  hipLaunchKernelGGL(k_policy_bwd, dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
  hipLaunchKernelGGL(k_policy_bwd_reduce, dim3(PBP / 64), dim3(256), 0, (hipStream_t)stream, scratch, blocks, small);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}
