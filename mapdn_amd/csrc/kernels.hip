// kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched VoltageControl hot path.
//
// Mapping: ONE LANE == ONE ENV.  All per-env state is "env-minor" SoA, X[item][Bp] (Bp = B rounded
// up to 64), so the 64 lanes of a wavefront touch 64 consecutive doubles (one 512-B coalesced
// request) for every item they process.  Because all envs share the topology, the elimination
// schedule, the Ybus entries and every index are wave-uniform: they are read through the scalar
// cache (s_load) and all branches on them are scalar branches — no divergence, no LDS traffic and
// no cross-lane exchange in the solve.  Cross-lane work is limited to the wave vote that ends the
// Newton loop and to the LDS-tiled transposes between env-minor state and env-major I/O tensors.
//
// What is computed follows pandapower 2.7.0's runpp (pypower newtonpf; reference call site
// voltage_control_env.py:557) and MAPDN's VoltageControl methods cited at each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mapdn {

// =================================================================================================
// Philox4x32-10 (Salmon et al. SC'11), keyed (seed) / counter (env, draw, stream, block);
// mapping documented in oracle/philox.py (the checker restates the same mapping).
// =================================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  return (double)(((uint64_t)(hi >> 5) << 26) + (uint64_t)(lo >> 6));
}

// =================================================================================================
// K1a  q_new — _clip_reactive_power (voltage_control_env.py:568-572) for step(), or the random
//      initial action of reset() (:120-122, get_action :337).  thread = (sgen j, env e).
// =================================================================================================
template <typename AT>
__global__ void __launch_bounds__(256) k_qnew(Dev d, const AT* __restrict__ actions, int mode) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (e >= d.Bp) return;
  bool act;
  if (e >= d.B) act = false;
  else if (mode == MODE_STEP) act = !d.done[e];
  else act = d.pending[e] != 0;
  if (j == 0) d.active[e] = act ? 1 : 0;
  if (!act) return;
  const size_t o = (size_t)j * d.Bp + e;
  const double p = d.cur_pv[o];
  const double sm = d.smax[j];
  const double lim = sqrt(sm * sm - p * p);
  double a;
  if (mode == MODE_STEP) {
    a = (double)actions[(size_t)e * d.ns + j];
  } else if (d.reset_action) {
    uint32_t x[4];
    philox4x32_10((uint32_t)(d.env_id_offset + e), d.adv_draw[e], STREAM_ACTION, (uint32_t)(j >> 1),
                  d.seed_lo, d.seed_hi, x);
    const double u = ((j & 1) ? u53(x[2], x[3]) : u53(x[0], x[1])) * (1.0 / 9007199254740992.0);
    a = d.action_low + (d.action_high - d.action_low) * u;
  } else {
    d.q_new[o] = 0.0;   // base-net q_mvar (deepcopy of base_powergrid, :106)
    return;
  }
  d.q_new[o] = lim * a;
}

// =================================================================================================
// K1b  Sbus — pandapower build_bus._calc_pq_elements_and_add_on_ppc + makeSbus:
//      Sbus[k] = -(sum load - sum sgen)/sn_mva, by element->bus CSR (no atomics).
//      thread = (position k, env e); writes in elimination-position order for the NR kernel.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_sbus(Dev d, const double* __restrict__ pl, const double* __restrict__ ql, const double* __restrict__ pv,
       const double* __restrict__ q) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (e >= d.Bp) return;
  double P = 0.0, Q = 0.0;
  for (int i = d.load_ptr[k]; i < d.load_ptr[k + 1]; ++i) {
    const size_t o = (size_t)d.load_idx[i] * d.Bp + e;
    P += pl[o]; Q += ql[o];
  }
  for (int i = d.sgen_ptr[k]; i < d.sgen_ptr[k + 1]; ++i) {
    const size_t o = (size_t)d.sgen_idx[i] * d.Bp + e;
    P -= pv[o]; Q -= q[o];
  }
  const size_t o = (size_t)k * d.Bp + e;
  d.Sr[o] = -P / d.sn;
  d.Si[o] = -Q / d.sn;
}

// =================================================================================================
// K2-K5  Newton-Raphson power flow — pandapower/pypower/newtonpf.py (flat start, polar form, full
//        Jacobian every iteration, ||F||inf < tol, <= 10 iterations), for a radial feeder.
//
//  One wavefront = 64 envs; each lane runs the complete solve of its env.  Per iteration:
//   forward sweep over nodes in leaf->root order (parents after children), fusing
//     * I = Ybus V and the mismatch F = V conj(I) - Sbus            (dSbus_dV / _evaluate_Fx)
//     * the four Jacobian entries of every Ybus non-zero              (create_jacobian_matrix)
//     * block-2x2 Gaussian elimination J y = F without fill          (replaces SuperLU spsolve)
//   then, unless converged, a backward sweep (root->leaf) that back-substitutes and applies
//   Va -= y_theta, Vm -= y_V, V = Vm e^{jVa} with the abs/angle re-normalisation of newtonpf.
//  Children send S / Schur contributions to their parent through registers when the parent is the
//  next node of the schedule (feeder chains) and through per-env scratch at junctions.
// =================================================================================================
__global__ void __launch_bounds__(64) k_nr_tree(Dev d) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  const size_t S = (size_t)d.Bp;
  const int n = d.n;
  const double vroot = d.vroot, tol = d.tol;
  const int32_t* __restrict__ par = d.par;
  const uint32_t* __restrict__ flags = d.flags;
  const double* __restrict__ yc = d.yc;
  const double* __restrict__ Sr = d.Sr + e;
  const double* __restrict__ Si = d.Si + e;
  double* __restrict__ Ve = d.Ve + e;
  double* __restrict__ Vf = d.Vf + e;
  double* __restrict__ Vm = d.Vm + e;
  double* __restrict__ Va = d.Va + e;
  double* __restrict__ G = d.G + e;        // [4n]
  double* __restrict__ H = d.H + e;        // [2n]
  double* __restrict__ AS = d.accS + e;    // [2n]
  double* __restrict__ AD = d.accD + e;    // [4n]
  double* __restrict__ AR = d.accR + e;    // [2n]
  double* __restrict__ X = d.X + e;        // [2n]

  bool done = d.active[e] == 0;
  bool conv = false;
  int it = 0;
  if (__all(done)) {   // nothing to solve in this wavefront (e.g. reset retry with no pending env)
    d.iters[e] = 0; d.conv[e] = 0;
    return;
  }

  // flat start: every bus at the ext_grid set-point, angle 0 (runpp init="auto")
  for (int k = 0; k < n; ++k) {
    Ve[k * S] = vroot; Vf[k * S] = 0.0; Vm[k * S] = vroot; Va[k * S] = 0.0;
  }

  for (;;) {
    // ------------------------------------------------------------------ forward sweep
    bool allok = true;
    double cS0 = 0, cS1 = 0, cD0 = 0, cD1 = 0, cD2 = 0, cD3 = 0, cR0 = 0, cR1 = 0;  // register carry
    double ne = Ve[0], nf = Vf[0], nvm = Vm[0];                                       // own V of node k
    for (int k = 0; k < n; ++k) {
      const uint32_t fl = flags[k];
      const int p = par[k];
      const double gkk = yc[6 * k + 0], bkk = yc[6 * k + 1], gkp = yc[6 * k + 2], bkp = yc[6 * k + 3],
                   gpk = yc[6 * k + 4], bpk = yc[6 * k + 5];
      const double ek = ne, fk = nf, vmk = nvm;
      double ep, fp, vmp;
      if (fl & F_PARENT_ROOT) { ep = vroot; fp = 0.0; vmp = vroot; }
      else { ep = Ve[p * S]; fp = Vf[p * S]; vmp = Vm[p * S]; }
      if (k + 1 < n) {
        if (fl & F_PARENT_NEXT) { ne = ep; nf = fp; nvm = vmp; }
        else { ne = Ve[(k + 1) * S]; nf = Vf[(k + 1) * S]; nvm = Vm[(k + 1) * S]; }
      }
      // A_kp = V_k conj(Y_kp V_p),  A_pk = V_p conj(Y_pk V_k),  A_kk = |V_k|^2 conj(Y_kk)
      const double tr = gkp * ep - bkp * fp, ti = gkp * fp + bkp * ep;
      const double akp_r = ek * tr + fk * ti, akp_i = fk * tr - ek * ti;
      const double ur = gpk * ek - bpk * fk, ui = gpk * fk + bpk * ek;
      const double apk_r = ep * ur + fp * ui, apk_i = fp * ur - ep * ui;
      const double v2 = ek * ek + fk * fk;
      const double akk_r = v2 * gkk, akk_i = -v2 * bkk;
      // contributions of the children
      double aS0 = 0, aS1 = 0, aD0 = 0, aD1 = 0, aD2 = 0, aD3 = 0, aR0 = 0, aR1 = 0;
      if (fl & F_SCRATCH_IN) {
        aS0 = AS[(2 * k) * S]; aS1 = AS[(2 * k + 1) * S];
        aD0 = AD[(4 * k) * S]; aD1 = AD[(4 * k + 1) * S]; aD2 = AD[(4 * k + 2) * S]; aD3 = AD[(4 * k + 3) * S];
        aR0 = AR[(2 * k) * S]; aR1 = AR[(2 * k + 1) * S];
      }
      if (fl & F_CARRY_IN) {
        aS0 += cS0; aS1 += cS1; aD0 += cD0; aD1 += cD1; aD2 += cD2; aD3 += cD3; aR0 += cR0; aR1 += cR1;
      }
      // S_k = V_k conj(sum_j Y_kj V_j) and the mismatch F_k = S_k - Sbus_k
      const double sr = akk_r + akp_r + aS0, si = akk_i + akp_i + aS1;
      const double Fp = sr - Sr[k * S], Fq = si - Si[k * S];
      allok = allok && (fabs(Fp) < tol) && (fabs(Fq) < tol);
      // diagonal Jacobian block: dS/dVa = j(S - A_kk); dS/dVm = (S + A_kk)/|V_k|
      const double ivm = 1.0 / vmk;
      const double D0 = -(si - akk_i) - aD0, D1 = (sr + akk_r) * ivm - aD1;
      const double D2 = (sr - akk_r) - aD2, D3 = (si + akk_i) * ivm - aD3;
      const double r0 = Fp - aR0, r1 = Fq - aR1;
      const double idet = 1.0 / (D0 * D3 - D1 * D2);
      const double I0 = D3 * idet, I1 = -D1 * idet, I2 = -D2 * idet, I3 = D0 * idet;
      const double h0 = I0 * r0 + I1 * r1, h1 = I2 * r0 + I3 * r1;
      H[(2 * k) * S] = h0; H[(2 * k + 1) * S] = h1;
      if (!(fl & F_PARENT_ROOT)) {
        // off-diagonal blocks: U = J(k,p) from A_kp, L = J(p,k) from A_pk
        const double ivmp = 1.0 / vmp;
        const double U0 = akp_i, U1 = akp_r * ivmp, U2 = -akp_r, U3 = akp_i * ivmp;
        const double G0 = I0 * U0 + I1 * U2, G1 = I0 * U1 + I1 * U3, G2 = I2 * U0 + I3 * U2, G3 = I2 * U1 + I3 * U3;
        G[(4 * k) * S] = G0; G[(4 * k + 1) * S] = G1; G[(4 * k + 2) * S] = G2; G[(4 * k + 3) * S] = G3;
        const double L0 = apk_i, L1 = apk_r * ivm, L2 = -apk_r, L3 = apk_i * ivm;
        const double s0 = L0 * G0 + L1 * G2, s1 = L0 * G1 + L1 * G3, s2 = L2 * G0 + L3 * G2, s3 = L2 * G1 + L3 * G3;
        const double t0 = L0 * h0 + L1 * h1, t1 = L2 * h0 + L3 * h1;
        if (fl & F_PARENT_NEXT) {
          cS0 = apk_r; cS1 = apk_i; cD0 = s0; cD1 = s1; cD2 = s2; cD3 = s3; cR0 = t0; cR1 = t1;
        } else if (fl & F_SCRATCH_FIRST) {
          AS[(2 * p) * S] = apk_r; AS[(2 * p + 1) * S] = apk_i;
          AD[(4 * p) * S] = s0; AD[(4 * p + 1) * S] = s1; AD[(4 * p + 2) * S] = s2; AD[(4 * p + 3) * S] = s3;
          AR[(2 * p) * S] = t0; AR[(2 * p + 1) * S] = t1;
        } else {
          AS[(2 * p) * S] += apk_r; AS[(2 * p + 1) * S] += apk_i;
          AD[(4 * p) * S] += s0; AD[(4 * p + 1) * S] += s1; AD[(4 * p + 2) * S] += s2; AD[(4 * p + 3) * S] += s3;
          AR[(2 * p) * S] += t0; AR[(2 * p + 1) * S] += t1;
        }
      }
    }
    if (!done) {
      conv = allok;
      if (conv || it == d.max_it) done = true;
    }
    if (__all(done)) break;   // wave vote: the only cross-lane operation of the solve
    // ------------------------------------------------------------------ backward sweep + update
    double x0 = 0, x1 = 0;
    for (int k = n - 1; k >= 0; --k) {
      const uint32_t fl = flags[k];
      double y0 = H[(2 * k) * S], y1 = H[(2 * k + 1) * S];
      if (!(fl & F_PARENT_ROOT)) {
        double p0, p1;
        if (fl & F_PARENT_NEXT) { p0 = x0; p1 = x1; }
        else { const int p = par[k]; p0 = X[(2 * p) * S]; p1 = X[(2 * p + 1) * S]; }
        y0 -= G[(4 * k) * S] * p0 + G[(4 * k + 1) * S] * p1;
        y1 -= G[(4 * k + 2) * S] * p0 + G[(4 * k + 3) * S] * p1;
      }
      x0 = y0; x1 = y1;
      if (fl & F_SCRATCH_IN) { X[(2 * k) * S] = y0; X[(2 * k + 1) * S] = y1; }
      if (!done) {
        // dx = -J^-1 F ; Va += dx_a ; Vm += dx_m ; V = Vm e^{jVa} ; Vm = |V| ; Va = angle(V)
        double va = Va[k * S] - y0;
        double vm = Vm[k * S] - y1;
        if (vm < 0.0) { vm = -vm; va += M_PI; }
        if (va > M_PI) va -= 2.0 * M_PI;
        else if (va <= -M_PI) va += 2.0 * M_PI;
        double s, c;
        sincos(va, &s, &c);
        Va[k * S] = va; Vm[k * S] = vm; Ve[k * S] = vm * c; Vf[k * S] = vm * s;
      }
    }
    if (!done) ++it;
  }
  d.iters[e] = it;
  d.conv[e] = conv ? 1 : 0;
}

// =================================================================================================
// K6/K7  results + reward — pandapower pfsoln/_extract_results (res_bus, res_line) and
//        VoltageControl._calc_reward (voltage_control_env.py:574-623), step() bookkeeping
//        (:185-209) including the unsolvable branch (:188-196).  thread = env.
// =================================================================================================
__device__ __forceinline__ double barrier(int type, double v) {
  switch (type) {
    case MAPDN_BARRIER_L1: return fabs(v - 1.0);                               // l1.py:7
    case MAPDN_BARRIER_L2: return 2.0 * (v - 1.0) * (v - 1.0);                 // l2.py:7
    case MAPDN_BARRIER_COURANT_BELTRAMI: {                                     // courant_beltrami.py:7
      const double a = fmax(0.0, v - 1.05), b = fmax(0.0, 0.95 - v);
      return a * a + b * b;
    }
    case MAPDN_BARRIER_BOWL: {                                                 // bowl.py:6-12
      const double dv = fabs(v - 1.0);
      if (dv > 0.05) return 2.0 * dv - 0.095;
      const double scale = 0.1;
      const double nrm = 1.0 / sqrt(2.0 * M_PI * scale * scale) * exp(-0.5 * (v - 1.0) * (v - 1.0) / (scale * scale));
      return -0.01 * nrm + 0.04;
    }
    default: {                                                                 // bump.py:6-12
      if (fabs(v) < 1.0) return exp(-1.0 / (1.0 - v * v * v * v));
      if (v > 1.0 && v < 3.0) { const double w = v - 2.0; return exp(-1.0 / (1.0 - w * w * w * w)); }
      return 0.0;
    }
  }
}

__global__ void __launch_bounds__(64)
k_commit_reward(Dev d, int mode, int add_noise, double* __restrict__ reward, uint8_t* __restrict__ terminated,
                double* __restrict__ info) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= d.B) return;
  const size_t S = (size_t)d.Bp;
  d.adv_row[e] = -1;
  if (mode == MODE_STEP && d.done[e]) {           // frozen env
    reward[e] = 0.0; terminated[e] = 1;
    for (int c = 0; c < MAPDN_N_INFO; ++c) info[(size_t)e * MAPDN_N_INFO + c] = 0.0;
    return;
  }
  if (mode == MODE_RESET && !d.pending[e]) return;
  const bool ok = d.conv[e] != 0;
  double q_fail = 0.0;
  if (ok) {
    // ---- commit res_bus (vm_pu, va, p_mw, q_mvar), slack injection, res_line.pl_mw, sgen.q_mvar
    const double* Ve = d.Ve + e; const double* Vf = d.Vf + e; const double* Vm = d.Vm + e; const double* Va = d.Va + e;
    double ir = d.yrr0 * d.vroot, ii = d.yrr1 * d.vroot;    // I_root = sum_j Y_rj V_j  (V_root = vroot + 0j)
    for (int k = 0; k < d.n; ++k) {
      const int bus = d.bus_of_pos[k];
      const double vm = Vm[k * S];
      d.vm[(size_t)bus * S + e] = vm;
      d.va[(size_t)bus * S + e] = Va[k * S];
      double P = 0.0, Q = 0.0;
      for (int i = d.load_ptr[k]; i < d.load_ptr[k + 1]; ++i) { const size_t o = (size_t)d.load_idx[i] * S + e; P += d.cur_pl[o]; Q += d.cur_ql[o]; }
      for (int i = d.sgen_ptr[k]; i < d.sgen_ptr[k + 1]; ++i) { const size_t o = (size_t)d.sgen_idx[i] * S + e; P -= d.cur_pv[o]; Q -= d.q_new[o]; }
      P += d.shunt_p[k] * vm * vm; Q += d.shunt_q[k] * vm * vm;
      d.res_p[(size_t)bus * S + e] = P; d.res_q[(size_t)bus * S + e] = Q;
      if (d.flags[k] & F_PARENT_ROOT) {
        const double g = d.yc[6 * k + 4], b = d.yc[6 * k + 5], ek = Ve[k * S], fk = Vf[k * S];
        ir += g * ek - b * fk; ii += g * fk + b * ek;
      }
    }
    {  // slack bus: res_bus = -(V conj(I)) * sn  (+ shunt), consumer sign
      const int bus = d.bus_of_pos[d.n];
      const double sre = d.vroot * ir, sim = -d.vroot * ii;
      d.vm[(size_t)bus * S + e] = d.vroot; d.va[(size_t)bus * S + e] = 0.0;
      d.res_p[(size_t)bus * S + e] = -sre * d.sn + d.shunt_p[d.n] * d.vroot * d.vroot;
      d.res_q[(size_t)bus * S + e] = -sim * d.sn + d.shunt_q[d.n] * d.vroot * d.vroot;
    }
    double loss = 0.0;
    for (int l = 0; l < d.n_line; ++l) {
      const LineFlow L = d.lines[l];
      double pl = 0.0;
      if (L.fpos >= 0) {
        double ef, ff, et, ft;
        if (L.fpos == d.n) { ef = d.vroot; ff = 0.0; } else { ef = Ve[L.fpos * S]; ff = Vf[L.fpos * S]; }
        if (L.tpos == d.n) { et = d.vroot; ft = 0.0; } else { et = Ve[L.tpos * S]; ft = Vf[L.tpos * S]; }
        // Sf = Vf conj(yff Vf + yft Vt),  St = Vt conj(ytf Vf + ytt Vt);  pl = Re(Sf + St) * sn
        const double ifr = L.yff[0] * ef - L.yff[1] * ff + L.yft[0] * et - L.yft[1] * ft;
        const double ifi = L.yff[0] * ff + L.yff[1] * ef + L.yft[0] * ft + L.yft[1] * et;
        const double itr = L.ytf[0] * ef - L.ytf[1] * ff + L.ytt[0] * et - L.ytt[1] * ft;
        const double iti = L.ytf[0] * ff + L.ytf[1] * ef + L.ytt[0] * ft + L.ytt[1] * et;
        pl = ((ef * ifr + ff * ifi) + (et * itr + ft * iti)) * d.sn;
      }
      d.pl[(size_t)l * S + e] = pl;
      loss += pl;
    }
    d.line_loss[e] = loss;
    for (int j = 0; j < d.ns; ++j) d.cur_q[(size_t)j * S + e] = d.q_new[(size_t)j * S + e];
  } else if (mode == MODE_STEP) {
    for (int j = 0; j < d.ns; ++j) q_fail += fabs(d.q_new[(size_t)j * S + e]);     // :189
    q_fail /= (double)d.ns;
  }
  if (mode == MODE_RESET) {
    if (ok) { d.pending[e] = 0; d.done[e] = 0; }
    return;
  }
  // ---- _calc_reward on the committed (or rolled-back == previous) state
  const double vlo = d.v_lower, vhi = d.v_upper, vref = 0.5 * (vlo + vhi);
  int n_lo = 0, n_hi = 0;
  double dev = 0.0, vsum = 0.0, mdrop = 0.0, mrise = 0.0, bar = 0.0;
  for (int b = 0; b < d.nb; ++b) {
    const double v = d.vm[(size_t)b * S + e];
    n_lo += (v < vlo); n_hi += (v > vhi);
    dev += fabs(v - vref); vsum += v;
    mdrop = fmax(mdrop, (v < vlo) ? (vlo - v) : 0.0);
    mrise = fmax(mrise, (v > vhi) ? (v - vhi) : 0.0);
    bar += barrier(d.barrier_type, v);
  }
  const double inv_nb = 1.0 / (double)d.nb;
  const double out = (double)(n_lo + n_hi) / (double)d.nb;
  double q_loss = 0.0;
  for (int j = 0; j < d.ns; ++j) q_loss += fabs(d.cur_q[(size_t)j * S + e]);
  q_loss /= (double)d.ns;
  const double line_loss = d.line_loss[e];
  const double v_loss = bar * inv_nb * d.voltage_weight;
  double loss;
  if (d.use_line_weight) loss = line_loss / (double)d.n_line * d.line_weight + v_loss;   // :612-613
  else loss = q_loss * d.q_weight + v_loss;                                              // :614-615
  double rew = -loss;
  double* inf = info + (size_t)e * MAPDN_N_INFO;
  inf[0] = out; inf[1] = (double)n_lo / (double)d.nb; inf[2] = (double)n_hi / (double)d.nb;
  inf[3] = (out > 1e-3) ? 0.0 : 1.0;
  inf[4] = dev * inv_nb; inf[5] = vsum * inv_nb; inf[6] = mdrop; inf[7] = mrise;
  inf[8] = line_loss; inf[9] = q_loss; inf[10] = 0.0;
  if (!ok) { rew -= 200.0; inf[10] = 1.0; inf[3] = 0.0; inf[9] = q_fail; }              // :192-196
  // ---- bookkeeping: next profile row uses t = steps BEFORE the increment (:199 vs :202)
  const int st = d.steps[e];
  d.adv_row[e] = d.start_row[e] + st;
  d.adv_draw[e] = d.draw[e];
  d.draw[e] += 1;
  d.steps[e] = st + 1;
  d.sum_rewards[e] += rew;
  const bool term = (st + 1 >= d.episode_limit) || !ok;                                   // :204
  d.done[e] = term ? 1 : 0;
  reward[e] = rew; terminated[e] = term ? 1 : 0;
  (void)add_noise;
}

// =================================================================================================
// K9a  reset bookkeeping — reset()/manual_reset() (voltage_control_env.py:96-118, 137-159):
//      steps = 1, pick the episode start row, queue the row/draw for k_advance.  thread = env.
// =================================================================================================
__global__ void __launch_bounds__(256) k_reset_begin(Dev d, const int64_t* __restrict__ start_rows, int first_try) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.B) return;
  if (first_try) d.pending[e] = 1;
  d.adv_row[e] = -1;
  if (!d.pending[e]) return;
  d.steps[e] = 1; d.sum_rewards[e] = 0.0; d.done[e] = 1;   // stays "done" until a solvable start is found
  const uint32_t dr = d.draw[e];
  d.draw[e] = dr + 1;
  int64_t start;
  if (start_rows) start = start_rows[e];
  else {
    uint32_t x[4];
    philox4x32_10((uint32_t)(d.env_id_offset + e), dr, STREAM_START, 0u, d.seed_lo, d.seed_hi, x);
    const int64_t hour = (int64_t)(((uint64_t)x[0] * 24u) >> 32);                        // :384
    const int64_t day = (int64_t)(((uint64_t)x[1] * (uint64_t)d.n_start_days) >> 32);    // :398
    const int64_t interval = (int64_t)(((uint64_t)x[2] * (uint64_t)d.per_hour) >> 32);   // :389
    start = interval + hour * d.per_hour + day * d.per_day;                              // :445
  }
  d.start_row[e] = start;
  d.adv_row[e] = start + 1;   // t = self.steps == 1 (:100, :473)
  d.adv_draw[e] = dr;
  for (int j = 0; j < d.ns; ++j) d.cur_q[(size_t)j * d.Bp + e] = 0.0;
}

// =================================================================================================
// K9b  advance — _set_demand_and_pv (voltage_control_env.py:491-513): next row of the three
//      profile tables + std/100 * |N(0,1)| noise (:498,503,508).  thread = (Philox block, env):
//      one Philox4x32-10 call + one Box-Muller pair serves two adjacent table columns.
// =================================================================================================
__global__ void __launch_bounds__(256) k_advance(Dev d, int add_noise) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.B) return;
  const int64_t row = d.adv_row[e];
  if (row < 0) return;
  int b = blockIdx.y;                  // pair index over [pv pairs | load_p pairs | load_q pairs]
  const int npv = (d.ns + 1) >> 1, npl = (d.nl + 1) >> 1;
  int stream, count, col0;
  double* dst;
  if (b < npv) { stream = STREAM_PV; count = d.ns; col0 = 0; dst = d.cur_pv; }
  else if (b < npv + npl) { b -= npv; stream = STREAM_LOAD_P; count = d.nl; col0 = d.ns; dst = d.cur_pl; }
  else { b -= npv + npl; stream = STREAM_LOAD_Q; count = d.nl; col0 = d.ns + d.nl; dst = d.cur_ql; }
  const int j0 = 2 * b, j1 = 2 * b + 1;
  const double* trow = d.table + (size_t)row * d.ncol + col0;
  double v0 = trow[j0];
  double v1 = (j1 < count) ? trow[j1] : 0.0;
  if (add_noise) {
    uint32_t x[4];
    philox4x32_10((uint32_t)(d.env_id_offset + e), d.adv_draw[e], (uint32_t)stream, (uint32_t)b, d.seed_lo, d.seed_hi, x);
    const double u1 = (u53(x[0], x[1]) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = u53(x[2], x[3]) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(2.0 * M_PI * u2, &s, &c);
    v0 += d.stdv[col0 + j0] * fabs(r * c);
    if (j1 < count) v1 += d.stdv[col0 + j1] * fabs(r * s);
  }
  dst[(size_t)j0 * d.Bp + e] = v0;
  if (j1 < count) dst[(size_t)j1 * d.Bp + e] = v1;
}

// =================================================================================================
// K8  observe — get_obs (voltage_control_env.py:232-274) / get_state (:213-230).
//     k_addback: effective PV add-back onto res_bus p/q at sgen buses (:238-244).
//     k_gather : column descriptors (kind, index) -> env-major output through a 64x64 LDS tile so
//                both the env-minor reads and the env-major writes are coalesced.
// =================================================================================================
__global__ void __launch_bounds__(256) k_addback(Dev d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;   // position, 0..nb-1
  if (e >= d.Bp) return;
  const int bus = d.bus_of_pos[k];
  const size_t o = (size_t)bus * d.Bp + e;
  double P = d.res_p[o], Q = d.res_q[o];
  for (int i = d.sgen_ptr[k]; i < d.sgen_ptr[k + 1]; ++i) {
    const size_t s = (size_t)d.sgen_idx[i] * d.Bp + e;
    P += d.cur_pv[s]; Q += d.cur_q[s];
  }
  d.pb[o] = P; d.qb[o] = Q;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_gather(GatherSrc g, const int32_t* __restrict__ kind, const int32_t* __restrict__ idx, T* __restrict__ out,
         int C, int B, int Bp) {
  __shared__ double tile[64][65];
  const int c0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    double v = 0.0;
    if (c < C) {
      const int kd = kind[c];
      const double* base = g.base[kd];
      if (base) v = base[(size_t)idx[c] * Bp + e0 + tx] * g.scale[kd];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int e = e0 + r, c = c0 + tx;
    if (e < B && c < C) out[(size_t)e * C + c] = (T)tile[tx][r];
  }
}

// env-major [B, n] -> env-minor [n][Bp] (zero-fills the pad lanes)
__global__ void __launch_bounds__(256)
k_to_envminor(const double* __restrict__ src, double* __restrict__ dst, int n, int B, int Bp) {
  __shared__ double tile[64][65];
  const int c0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int e = e0 + r, c = c0 + tx;
    tile[r][tx] = (e < B && c < n) ? src[(size_t)e * n + c] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    if (c < n) dst[(size_t)c * Bp + e0 + tx] = tile[tx][r];
  }
}

// env-minor int32/u8 vectors -> outputs (trivial copies)
__global__ void k_copy_i32(const int32_t* s, int32_t* dd, int B) { int e = blockIdx.x * blockDim.x + threadIdx.x; if (e < B) dd[e] = s[e]; }
__global__ void k_copy_u8(const uint8_t* s, uint8_t* dd, int B) { int e = blockIdx.x * blockDim.x + threadIdx.x; if (e < B) dd[e] = s[e]; }

// identity descriptors for plain transposes
__global__ void k_iota(int32_t* kind, int32_t* idx, int n, int k) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { kind[i] = k; idx[i] = i; } }

// stats: pending count, iteration sum / max over active envs (single block)
__global__ void __launch_bounds__(256) k_stats(Dev d, long long* out) {
  __shared__ long long s_pend[256], s_sum[256], s_cnt[256];
  __shared__ int s_max[256];
  long long pend = 0, sum = 0, cnt = 0; int mx = 0;
  for (int e = threadIdx.x; e < d.B; e += 256) {
    pend += d.pending[e] ? 1 : 0;
    if (d.active[e]) { sum += d.iters[e]; cnt += 1; mx = max(mx, d.iters[e]); }
  }
  s_pend[threadIdx.x] = pend; s_sum[threadIdx.x] = sum; s_cnt[threadIdx.x] = cnt; s_max[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      s_pend[threadIdx.x] += s_pend[threadIdx.x + w]; s_sum[threadIdx.x] += s_sum[threadIdx.x + w];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w]; s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s_pend[0]; out[1] = s_sum[0]; out[2] = s_cnt[0]; out[3] = s_max[0]; }
}

// =================================================================================================
// launchers (host)
// =================================================================================================
static inline dim3 grid_env(const Dev& d, int y) { return dim3((d.Bp + 255) / 256, y); }

void launch_qnew(const Dev& d, const void* actions, int dtype, int mode, hipStream_t st) {
  if (dtype == MAPDN_F32) hipLaunchKernelGGL(k_qnew<float>, grid_env(d, d.ns), dim3(256), 0, st, d, (const float*)actions, mode);
  else hipLaunchKernelGGL(k_qnew<double>, grid_env(d, d.ns), dim3(256), 0, st, d, (const double*)actions, mode);
}
void launch_sbus(const Dev& d, const double* pl, const double* ql, const double* pv, const double* q, hipStream_t st) {
  hipLaunchKernelGGL(k_sbus, grid_env(d, d.n), dim3(256), 0, st, d, pl, ql, pv, q);
}
void launch_nr(const Dev& d, hipStream_t st) {
  hipLaunchKernelGGL(k_nr_tree, dim3(d.Bp / 64), dim3(64), 0, st, d);
}
void launch_commit_reward(const Dev& d, int mode, int add_noise, double* reward, uint8_t* term, double* info, hipStream_t st) {
  hipLaunchKernelGGL(k_commit_reward, dim3(d.Bp / 64), dim3(64), 0, st, d, mode, add_noise, reward, term, info);
}
void launch_reset_begin(const Dev& d, const int64_t* start_rows, int first_try, hipStream_t st) {
  hipLaunchKernelGGL(k_reset_begin, dim3((d.B + 255) / 256), dim3(256), 0, st, d, start_rows, first_try);
}
void launch_advance(const Dev& d, int add_noise, hipStream_t st) {
  const int pairs = ((d.ns + 1) >> 1) + 2 * ((d.nl + 1) >> 1);
  hipLaunchKernelGGL(k_advance, dim3((d.B + 255) / 256, pairs), dim3(256), 0, st, d, add_noise);
}
void launch_addback(const Dev& d, hipStream_t st) {
  hipLaunchKernelGGL(k_addback, grid_env(d, d.nb), dim3(256), 0, st, d);
}
void launch_gather(const Dev& d, const GatherSrc& g, const int32_t* kind, const int32_t* idx, void* out, int dtype, int C, hipStream_t st) {
  dim3 grid((C + 63) / 64, d.Bp / 64);
  if (dtype == MAPDN_F32) hipLaunchKernelGGL(k_gather<float>, grid, dim3(256), 0, st, g, kind, idx, (float*)out, C, d.B, d.Bp);
  else hipLaunchKernelGGL(k_gather<double>, grid, dim3(256), 0, st, g, kind, idx, (double*)out, C, d.B, d.Bp);
}
void launch_to_envminor(const Dev& d, const double* src, double* dst, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_to_envminor, dim3((n + 63) / 64, d.Bp / 64), dim3(256), 0, st, src, dst, n, d.B, d.Bp);
}
void launch_copy_i32(const int32_t* s, int32_t* dd, int B, hipStream_t st) { hipLaunchKernelGGL(k_copy_i32, dim3((B + 255) / 256), dim3(256), 0, st, s, dd, B); }
void launch_copy_u8(const uint8_t* s, uint8_t* dd, int B, hipStream_t st) { hipLaunchKernelGGL(k_copy_u8, dim3((B + 255) / 256), dim3(256), 0, st, s, dd, B); }
void launch_iota(int32_t* kind, int32_t* idx, int n, int k, hipStream_t st) { hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, kind, idx, n, k); }
void launch_stats(const Dev& d, long long* out, hipStream_t st) { hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, st, d, out); }

}  // namespace mapdn
