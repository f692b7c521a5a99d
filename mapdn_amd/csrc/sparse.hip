// sparse.hip — k_nr_sparse: the Newton-Raphson power flow for MESHED nets of any size that fits a CU's LDS, gfx950 (MI355X).
//
// pandapower hands the Jacobian of any net to SuperLU at run time (pypower/newtonpf.py: dx = -spsolve(J, F); reference call
// site voltage_control_env.py:557).  Here the symbolic half of that factorisation is done once per topology on the host
// (plan.cpp: minimum-degree order, fill pattern, one slot per structurally non-zero 2x2 block) and compiled into a PROGRAM
// of block operations  B[c] = inv(B[a]) | B[c] = B[a] B[b] | B[c] -= B[a] B[b]  packed into phases of independent
// operations; this kernel is the interpreter.  A workgroup is ONE wavefront serving L envs; its 64 lanes are S = 64 / L
// sub-lanes per env (lane = sub * L + env, as in k_nr_tree) and sub-lane s executes operation s of every phase for its L
// envs.  All blocks of those envs (diagonal, off-diagonal incl. fill, right-hand sides as [b | 0] blocks) live in LDS as
// [slot][env][2x2]; a phase is: operation record (prefetched, broadcast buffer load of a constant table) -> six
// ds_read_b128 -> 8 FMAs (+ the 2x2 inverse) -> two ds_write_b128.  One wavefront needs no barrier: LDS executes its
// instructions in order, so the reads of phase p+1 see the writes of phase p.
//
// Same Newton iteration as the other two solvers (flat start, scaled polar unknowns [dtheta, d|V|/|V|], full Jacobian every
// iteration, ||F||inf < tol, <= max_it iterations, V <- V (1 - z1) e^{-j z0}) and the same fused epilogue (nr_common.hpp).
// Radial feeders keep the specialised tree kernel; MAPDN_NR_SPARSE=1 runs this one on them too (cross-check).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"
#include "nr_common.hpp"

namespace mapdn {

template <int L>
__global__ void __launch_bounds__(64)
k_nr_sparse(Dev d, int mode, double* __restrict__ reward, uint8_t* __restrict__ terminated, double* __restrict__ info) {
  extern __shared__ d2 lds2[];
  constexpr unsigned S = 64u / L;
  const unsigned lane = threadIdx.x, s = lane / L, el = lane % L;
  const unsigned e = blockIdx.x * L + el;
  const int n = d.n;
  const double vroot = d.vroot, tol = d.tol;
  d2* sV = lds2 + el;                                              // sV[k * L] = (e, f); k = n slack, n + 1 trash
  d2* sB = lds2 + (size_t)(n + 2) * L + 2u * el;                   // block b: sB[b * 2L], sB[b * 2L + 1] = rows (a00 a01), (a10 a11)
  double* s_epi = (double*)(lds2 + (size_t)(n + 2) * L + (size_t)d.sp_blocks * 2u * L) + el;   // 10 * S * L doubles (epilogue), also the verdict scratch
  for (unsigned k = s; k < (unsigned)n + 2u; k += S) sV[(size_t)k * L] = d2{vroot, 0.0};        // runpp init="auto": flat start
  const bool act = d.active[e] != 0;
  const int bk_steps = d.steps[e];
  const uint32_t bk_draw = d.draw[e];
  const double bk_sum = d.sum_rewards[e];
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(const_cast<SpOp*>(d.sp_ops), 0, d.sp_ops_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<SpNz*>(d.sp_nz), 0, d.sp_nz_bytes, 0x00020000);
  const int RPS = d.sp_rows_per_sub, MNZ = d.sp_max_nnz, NPH = d.sp_phases;
  const unsigned voO = s * 16u;                                    // this sub-lane's op of phase p: + p * S * 16
  const unsigned voZ = s * (unsigned)RPS * (unsigned)MNZ * 32u;    // the Ybus entries of its assembly rows
  const double2* gsb = (const double2*)((const char*)d.nrbuf + d.sb_off);
  auto blk = [&](unsigned b) { return sB + (size_t)b * (2u * L); };
  const bool nothing_to_solve = __all(!act);
  bool done = !act, conv = false;
  int it = 0;
  while (!nothing_to_solve) {
    // ---- mismatch F = V conj(Ybus V) - Sbus and the Jacobian blocks, row by row (sub-lane s owns rows s, s + S, ...):
    //      dS_i/dtheta_j = -j A_ij, dS_i/dln|V_j| = A_ij (j != i); dS_i/dtheta_i = j (S_i - A_ii), dS_i/dln|V_i| = S_i + A_ii
    for (unsigned q = s; q < (unsigned)d.sp_fill; q += S) {        // blocks that exist only as fill start from zero
      d2* f = blk((unsigned)d.sp_fill_slots[q]);
      f[0] = d2{0.0, 0.0}; f[1] = d2{0.0, 0.0};
    }
    bool ok = true;
    {
      // One stream of RPS * MNZ Ybus entries per sub-lane (row r of sub-lane s is node i = r * S + s; rows are padded to MNZ
      // entries with Y = 0), read through a ring that runs PF entries ahead of the arithmetic: the table is a constant in
      // L2, but a global round trip per entry would otherwise be the whole cost of the assembly.
      constexpr int PF = 8;
      const int T = RPS * MNZ;
      u32x4 zq0[PF]; u32x2 zq1[PF];
      auto zload = [&](int t, u32x4& z0, u32x2& z1) {
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)min(t, T - 1) * 32u);
        z0 = __builtin_amdgcn_raw_buffer_load_b128(rsZ, voZ, so, 0);               // col, slot, Re Y (two dwords)
        z1 = __builtin_amdgcn_raw_buffer_load_b64(rsZ, voZ + 16u, so, 0);          // Im Y
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) zload(u, zq0[u], zq1[u]);
      int r = 0, q = 0;
      unsigned i = s;
      bool live = i < (unsigned)n;
      double2 sb = gsb[(size_t)(live ? i : 0u) * d.Bp + e];
      d2 vi = sV[(size_t)(live ? i : (unsigned)n + 1u) * L];
      double sr = 0.0, si = 0.0, aii_r = 0.0, aii_i = 0.0;
      for (int t0 = 0; t0 < T; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int t = t0 + u;
          if (t < T) {                           // (uniform; no `break`: the loop must unroll for the ring to stay in registers)
          const u32x4 z0 = zq0[u]; const u32x2 z1 = zq1[u];
          zload(t + PF, zq0[u], zq1[u]);
          const double g = __hiloint2double((int)z0.w, (int)z0.z), b = __hiloint2double((int)z1.y, (int)z1.x);
          const unsigned col = z0.x;
          const int slot = (int)z0.y;
          const d2 vj = sV[(size_t)col * L];
          const double ei = vi.x, fi = vi.y;
          const double tr = g * vj.x - b * vj.y, ti = g * vj.y + b * vj.x;
          const double ar = ei * tr + fi * ti, ai = fi * tr - ei * ti;           // A_ij = V_i conj(Y_ij V_j)
          sr += ar; si += ai;
          if (col == i) { aii_r = ar; aii_i = ai; }
          if (slot >= 0) { d2* o = blk((unsigned)slot); o[0] = d2{ai, ar}; o[1] = d2{-ar, ai}; }
          if (++q == MNZ) {                      // (uniform) end of the row: diagonal block, right-hand side, verdict; next row
            const double Fp = sr - sb.x, Fq = si - sb.y;
            if (live) {
              d2* dg = blk(i);
              dg[0] = d2{-(si - aii_i), sr + aii_r}; dg[1] = d2{sr - aii_r, si + aii_i};
              d2* rh = blk((unsigned)n + i);
              rh[0] = d2{Fp, 0.0}; rh[1] = d2{Fq, 0.0};
              ok = ok && (fabs(Fp) < tol) && (fabs(Fq) < tol);
            }
            q = 0; ++r;
            i = (unsigned)r * S + s;
            live = i < (unsigned)n && r < RPS;
            sb = gsb[(size_t)(live ? i : 0u) * d.Bp + e];
            vi = sV[(size_t)(live ? i : (unsigned)n + 1u) * L];
            sr = si = aii_r = aii_i = 0.0;
          }
          }
        }
      }
    }
    {                                                              // AND over the sub-lanes of an env
      s_epi[(size_t)s * L] = ok ? 1.0 : 0.0;
      bool all = true;
#pragma unroll 4
      for (unsigned t = 0; t < S; ++t) all = all && (s_epi[(size_t)t * L] != 0.0);
      ok = all;
    }
    if (!done) {
      conv = ok;
      if (conv || it == d.max_it) done = true;
    }
    if (__all(done)) break;
    // ---- numeric factorisation + forward / backward substitution: the host-compiled program (plan.cpp::sparse_program)
    {
      constexpr int PF = 8;                          // operation records run PF phases ahead (constant table in L2)
      u32x4 oq[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) oq[u] = __builtin_amdgcn_raw_buffer_load_b128(rsO, voO, __builtin_amdgcn_readfirstlane((unsigned)min(u, NPH - 1) * S * 16u), 0);
      for (int p0 = 0; p0 < NPH; p0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int p = p0 + u;
          if (p < NPH) {                           // (uniform)
          const u32x4 op = oq[u];
          oq[u] = __builtin_amdgcn_raw_buffer_load_b128(rsO, voO, __builtin_amdgcn_readfirstlane((unsigned)min(p + PF, NPH - 1) * S * 16u), 0);
          const unsigned type = op.x & 255u;
          const unsigned hint = __builtin_amdgcn_readfirstlane(op.x);     // bits 8 / 9: the phase holds an INV / an UPD (wave-uniform)
          const d2* A = blk(op.z); const d2* B = blk(op.w); d2* C = blk(op.y);
          const d2 a0 = A[0], a1 = A[1], b0 = B[0], b1 = B[1];
          d2 c0 = d2{0.0, 0.0}, c1 = d2{0.0, 0.0};
          if (hint & 512u) { c0 = C[0]; c1 = C[1]; }
          const double p00 = a0.x * b0.x + a0.y * b1.x, p01 = a0.x * b0.y + a0.y * b1.y;
          const double p10 = a1.x * b0.x + a1.y * b1.x, p11 = a1.x * b0.y + a1.y * b1.y;
          d2 n0, n1;
          if (type == 2u) { n0 = d2{p00, p01}; n1 = d2{p10, p11}; }
          else { n0 = d2{c0.x - p00, c0.y - p01}; n1 = d2{c1.x - p10, c1.y - p11}; }
          if (hint & 256u) {
            const double idet = rcp_nr(a0.x * a1.y - a0.y * a1.x);
            if (type == 1u) { n0 = d2{a1.y * idet, -a0.y * idet}; n1 = d2{-a1.x * idet, a0.x * idet}; }
          }
          if (type != 0u && !done) { C[0] = n0; C[1] = n1; }
          }
        }
      }
    }
    // ---- newtonpf update: Va -= z0, Vm -= |V| z1, V = Vm e^{jVa}   =>   V <- V (1 - z1) e^{-j z0}
    for (int r = 0; r < RPS; ++r) {
      const unsigned i = (unsigned)r * S + s;
      if (i < (unsigned)n && !done) {
        const d2* x = blk((unsigned)n + i);
        const double y0 = x[0].x, y1 = x[1].x;
        const d2 vi = sV[(size_t)i * L];
        double sn, cs;
        sincos(-y0, &sn, &cs);
        sV[(size_t)i * L] = nr_rotate(vi, sn, cs, y1);
      }
    }
    if (!done) ++it;
  }
  if (s == 0) { d.iters[e] = it; d.conv[e] = conv ? 1 : 0; }
  nr_epilogue<(unsigned)L, S>(d, mode, reward, terminated, info, sV, s, e, act, conv, bk_steps, bk_draw, bk_sum, s_epi,
                              (const double*)nullptr, [](int) {});
}

size_t nr_sparse_lds_bytes(int n, int n_blocks, int L) {
  return ((size_t)(n + 2) * L + (size_t)n_blocks * 2 * L) * 16 + (size_t)10 * 64 * sizeof(double);
}

#define SP_FOR_EACH(X) X(16) X(8) X(4) X(2)
int nr_sparse_prepare(int L) {
#define X(l) if (L == l) return hipFuncSetAttribute((const void*)k_nr_sparse<l>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
  SP_FOR_EACH(X)
#undef X
  return -2;
}

void launch_nr_sparse(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st) {
  const size_t lds = nr_sparse_lds_bytes(d.n, d.sp_blocks, d.sp_lanes);
#define X(l) if (d.sp_lanes == l) { hipLaunchKernelGGL(k_nr_sparse<l>, dim3(d.Bp / l), dim3(64), lds, st, d, mode, reward, term, info); return; }
  SP_FOR_EACH(X)
#undef X
}

}  // namespace mapdn
