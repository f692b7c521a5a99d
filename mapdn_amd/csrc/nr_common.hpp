// nr_common.hpp — device code shared by the two power-flow kernels (k_nr_tree: radial feeders, kernels.hip;
// k_nr_dense: any connected topology, dense.hip): small numeric helpers, the voltage barriers and the FUSED EPILOGUE
// that turns the solution a workgroup still holds in LDS into res_line, the reward / info statistics, the unsolvable
// branch and the step bookkeeping of VoltageControl.step().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.hpp"
#include "nrmath.hpp"   // sincos_small / sincos_mid / bowl_inside: host-checkable (tests/test_nrmath.py)

namespace mapdn {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
      const double inside = bowl_inside(v);                                    // nrmath.hpp (branch-free select below)
      return (dv > 0.05) ? 2.0 * dv - 0.095 : inside;
    }
    default: {                                                                 // bump.py:6-12
      if (fabs(v) < 1.0) return exp(-1.0 / (1.0 - v * v * v * v));
      if (v > 1.0 && v < 3.0) { const double w = v - 2.0; return exp(-1.0 / (1.0 - w * w * w * w)); }
      return 0.0;
    }
  }
}

// 1/x from v_rcp_f64 + two Newton steps (<= 1-2 ulp; operands are well-scaled 2x2 determinants)
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double t = fma(-x, r, 1.0);
  r = fma(r, t, r);
  t = fma(-x, r, 1.0);
  return fma(r, t, r);
}

// newtonpf update of one bus in rectangular form: V <- V (1 - y1) e^{j dth} with (s, c) = sincos(dth).  The fused multiply-adds are
// spelled out, so that every kernel and code path rounds alike whatever the compiler would contract.
__device__ __forceinline__ d2 nr_rotate(d2 v, double s, double c, double y1) {
  const double sc = 1.0 - y1;
  return d2{sc * fma(v.x, c, -(v.y * s)), sc * fma(v.x, s, v.y * c)};
}

// =================================================================== fused epilogue
// The workgroup still holds the solution of its L envs in LDS (sV[k * L] = (e, f) of elimination position k, already
// offset by the env's lane; position n = slack), so the rest of the env step happens here, spread over the workgroup's
// Wt workers (strided over buses / lines / sgens):
//   K6  commit  — pandapower pfsoln/_extract_results: res_bus (vm_pu, va, p_mw, q_mvar incl. the slack
//                 injection), res_line.pl_mw, sgen.q_mvar, only for envs whose solve converged
//   K7  reward  — VoltageControl._calc_reward (voltage_control_env.py:574-623) on the committed (or, if
//                 the solve failed, rolled-back == previous) state, the unsolvable branch (:188-196) and
//                 step() bookkeeping (:199-209)
// t = worker id (0 .. Wt-1), e = env; s_epi = LDS for the partial sums (10 * Wt * L doubles, already offset by the env's
// lane); s_lines = LineFlow rows staged in LDS when d.nr_line_lds.  Every thread of the workgroup must call it.
template <unsigned L, unsigned Wt, typename Stamp>
__device__ __forceinline__ void nr_epilogue(const Dev& d, int mode, double* __restrict__ reward, uint8_t* __restrict__ terminated,
                                            double* __restrict__ info, const d2* sV, unsigned t, unsigned e, bool act, bool conv,
                                            int bk_steps, uint32_t bk_draw, double bk_sum, double* s_epi, const double* s_lines,
                                            Stamp stamp) {
  const unsigned n = (unsigned)d.n;
  const double vroot = d.vroot;
  const size_t SB = (size_t)d.Bp;
  if (mode == MODE_SOLVE) {                      // mapdn_solve_only: just the solution
    double* gV = d.nrbuf + (size_t)d.r_vout * SB + e;
    for (unsigned k = t; k < n; k += Wt) {
      const d2 vv = sV[(size_t)k * L];
      const double ek = vv.x, fk = vv.y;
      double* o = gV + (size_t)(VOF * k) * SB;
      o[(size_t)VO_E * SB] = ek; o[(size_t)VO_F * SB] = fk;
      o[(size_t)VO_VM * SB] = sqrt(ek * ek + fk * fk);            // Vm = |V|
      o[(size_t)VO_VA * SB] = atan2(fk, ek);                       // Va = angle(V)
    }
    return;
  }
  const bool commitf = act && conv;              // act: STEP -> not frozen, RESET -> pending
  const bool valid = e < (unsigned)d.B;
  const double vlo = d.v_lower, vhi = d.v_upper, vref = 0.5 * (vlo + vhi);
  double n_lo = 0, n_hi = 0, dev = 0.0, vsum = 0.0, mdrop = 0.0, mrise = 0.0, bar = 0.0, line_loss = 0.0, q_loss = 0.0, q_fail = 0.0;
  // Topology constants are read through the CONSTANT address space: the compiler then knows that no
  // store of this kernel can alias them and batches the loads of the unrolled, branch-free loops
  // (only the stores are predicated).  Envs whose solve failed (rare) take statistics from the
  // rolled-back state in a separate slow path.
  typedef const __attribute__((address_space(4))) double* c_f64;
  const c_f64 linec = (c_f64)(unsigned long long)d.lines;          // LineFlow = {int32 fpos, tpos; double c[4]} = 5 x 8 bytes
  // ---- buses: voltage statistics for the reward; the solution (e, f) goes to Vout, from where the
  // wide k_post kernel (thread per bus x env) commits res_bus — |V|, angle(V), p_mw, q_mvar — for the
  // envs flagged in d.commit: throughput work does not belong in this 512-wave kernel
  double* gV = d.nrbuf + (size_t)d.r_vout * SB + e;
  // this worker's first sgen q values (written by the inject kernel, cold in the cache): requested now, used after the two loops
  constexpr int QPRE = 3;
  double q_pre[QPRE];
#pragma unroll
  for (int i = 0; i < QPRE; ++i) { const unsigned j = t + (unsigned)i * Wt; q_pre[i] = (j < (unsigned)d.ns) ? d.q_new[(size_t)j * SB + e] : 0.0; }
  // the barrier type is hoisted out of the loop (one branch-free instance per type), so that the unrolled
  // iterations — sqrt and exp chains of different buses — can be interleaved by the scheduler
  auto bus_loop = [&](auto type_tag) {
    constexpr int BT = decltype(type_tag)::value;
#pragma unroll 4
    for (unsigned k = t; k < n; k += Wt) {
      const d2 vv = sV[(size_t)k * L];
      const double ek = vv.x, fk = vv.y;
      const double v = sqrt(ek * ek + fk * fk);                      // res_bus.vm_pu = |V|
      if (commitf) { gV[((size_t)VOF * k + VO_E) * SB] = ek; gV[((size_t)VOF * k + VO_F) * SB] = fk; }
      n_lo += (v < vlo) ? 1.0 : 0.0; n_hi += (v > vhi) ? 1.0 : 0.0;
      dev += fabs(v - vref); vsum += v;
      mdrop = fmax(mdrop, (v < vlo) ? (vlo - v) : 0.0);
      mrise = fmax(mrise, (v > vhi) ? (v - vhi) : 0.0);
      bar += barrier(BT, v);
    }
  };
  switch (d.barrier_type) {
    case MAPDN_BARRIER_L1: bus_loop(std::integral_constant<int, MAPDN_BARRIER_L1>{}); break;
    case MAPDN_BARRIER_L2: bus_loop(std::integral_constant<int, MAPDN_BARRIER_L2>{}); break;
    case MAPDN_BARRIER_COURANT_BELTRAMI: bus_loop(std::integral_constant<int, MAPDN_BARRIER_COURANT_BELTRAMI>{}); break;
    case MAPDN_BARRIER_BOWL: bus_loop(std::integral_constant<int, MAPDN_BARRIER_BOWL>{}); break;
    default: bus_loop(std::integral_constant<int, MAPDN_BARRIER_BUMP>{}); break;
  }
  if (t == n % Wt) {                              // the slack bus: committed voltage never changes
    const double v = vroot;
    n_lo += (v < vlo) ? 1.0 : 0.0; n_hi += (v > vhi) ? 1.0 : 0.0;
    dev += fabs(v - vref); vsum += v;
    mdrop = fmax(mdrop, (v < vlo) ? (vlo - v) : 0.0);
    mrise = fmax(mrise, (v > vhi) ? (v - vhi) : 0.0);
    bar += barrier(d.barrier_type, v);
  }
  if (d.n_alias) {                                // bus fusion (wave-uniform; no fused buses: not taken): a fused bus is a row of res_bus
    for (unsigned a = t; a < (unsigned)d.n_alias; a += Wt) {       // too, so the reward statistics count its node's voltage once more
      const unsigned k = (unsigned)d.alias_pos[a];
      double v = vroot;
      if (k < n) { const d2 vv = sV[(size_t)k * L]; v = sqrt(vv.x * vv.x + vv.y * vv.y); }
      n_lo += (v < vlo) ? 1.0 : 0.0; n_hi += (v > vhi) ? 1.0 : 0.0;
      dev += fabs(v - vref); vsum += v;
      mdrop = fmax(mdrop, (v < vlo) ? (vlo - v) : 0.0);
      mrise = fmax(mrise, (v > vhi) ? (v - vhi) : 0.0);
      bar += barrier(d.barrier_type, v);
    }
  }
  stamp(22);
  if (t == 0 && valid) d.commit[e] = commitf ? 1 : 0;
  // ---- res_line.pl_mw = Re(Sf + St) * sn   (out-of-service rows: fpos = tpos = slack, all-zero admittances)
  // LineFlow = {int32 fpos, tpos; double c[4]}: from LDS when staged there, else through the constant address space
  auto line_loop = [&](auto Lbase) {
#pragma unroll 4
    for (unsigned l = t; l < (unsigned)d.n_line; l += Wt) {
      const auto Lc = Lbase + (size_t)l * 5;
      const double ab = Lc[0];
      const unsigned a = (unsigned)__double2loint(ab), b = (unsigned)__double2hiint(ab);
      const double gff = Lc[1], gtt = Lc[2], gs = Lc[3], bd = Lc[4];
      const d2 vf = sV[(size_t)a * L], vt = sV[(size_t)b * L];
      const double ef = vf.x, ff = vf.y, et = vt.x, ft = vt.y;
      // Re(Vf conj(If) + Vt conj(It)) with the products of the two ends collected: w = Vf conj(Vt) = wr + j wi
      const double wr = ef * et + ff * ft, wi = ff * et - ef * ft;
      const double pl = ((gff * (ef * ef + ff * ff) + gtt * (et * et + ft * ft)) + (gs * wr + bd * wi)) * d.sn;
      if (commitf) d.pl[(size_t)l * SB + e] = pl;
      line_loss += pl;
    }
  };
  if (d.nr_line_lds) line_loop((const double*)s_lines); else line_loop(linec);
  stamp(23);
  // ---- sgen.q_mvar of the accepted solve; q statistics
  {
    int i = 0;
    for (unsigned j = t; j < (unsigned)d.ns; j += Wt, ++i) {
      const size_t o = (size_t)j * SB + e;
      const double qn = i < QPRE ? (i == 0 ? q_pre[0] : i == 1 ? q_pre[1] : q_pre[2]) : d.q_new[o];
      if (commitf) d.cur_q[o] = qn;
      q_loss += fabs(qn * d.sgen_scale[j]); q_fail += fabs(qn);     // res_sgen.q_mvar (:604-606); the raw table (:189)
    }
  }
  // ---- slow path (wave-uniform, rare): an env of this wave is active but did not converge -> its
  // statistics come from the PREVIOUS committed state (voltage_control_env.py:190 restores last_powergrid)
  if (mode == MODE_STEP && __any(act && !conv)) {
    double a_lo = 0, a_hi = 0, a_dev = 0, a_sum = 0, a_drop = 0, a_rise = 0, a_bar = 0, a_ll = 0, a_ql = 0;
    for (unsigned k = t; k < (unsigned)d.nbo; k += Wt) {           // every row of res_bus (original buses)
      const double v = valid ? d.vm[(size_t)k * SB + e] : 1.0;
      a_lo += (v < vlo) ? 1.0 : 0.0; a_hi += (v > vhi) ? 1.0 : 0.0;
      a_dev += fabs(v - vref); a_sum += v;
      a_drop = fmax(a_drop, (v < vlo) ? (vlo - v) : 0.0);
      a_rise = fmax(a_rise, (v > vhi) ? (v - vhi) : 0.0);
      a_bar += barrier(d.barrier_type, v);
    }
    for (unsigned l = t; l < (unsigned)d.n_line; l += Wt) a_ll += valid ? d.pl[(size_t)l * SB + e] : 0.0;
    for (unsigned j = t; j < (unsigned)d.ns; j += Wt) a_ql += fabs(d.cur_q[(size_t)j * SB + e] * d.sgen_scale[j]);
    if (!commitf) { n_lo = a_lo; n_hi = a_hi; dev = a_dev; vsum = a_sum; mdrop = a_drop; mrise = a_rise; bar = a_bar; line_loss = a_ll; q_loss = a_ql; }
  }
  if (mode == MODE_RESET) {
    if (t == 0 && valid) {
      d.adv_row[e] = -1;
      if (act && conv) { d.pending[e] = 0; d.done[e] = 0; }
    }
    return;
  }
  stamp(21);
  // ---- combine the workers' partials through LDS, fixed order
  double* sm = s_epi;                             // sm[(q*Wt + worker)*L]
  const double part[10] = {n_lo, n_hi, dev, vsum, mdrop, mrise, bar, line_loss, q_loss, q_fail};
#pragma unroll
  for (int q = 0; q < 10; ++q) sm[(size_t)(q * Wt + t) * L] = part[q];
  __syncthreads();
  // worker q (and q + Wt, ...) forms the total of quantity q over the workers, in the fixed order 0, 1, 2, ..., and leaves it in
  // entry 0 of its row (only this lane reads that row); worker 0 then picks up ten totals instead of summing 10 x Wt values itself
  constexpr unsigned UNR = Wt <= 16 ? Wt : 8;     // (the general solvers have up to 128 workers)
  for (unsigned q = t; q < 10u; q += Wt) {
    double a = sm[(size_t)(q * Wt) * L];
#pragma unroll UNR
    for (unsigned ww = 1; ww < Wt; ++ww) { const double b = sm[(size_t)(q * Wt + ww) * L]; a = (q == 4u || q == 5u) ? fmax(a, b) : a + b; }
    sm[(size_t)(q * Wt) * L] = a;
  }
  __syncthreads();
  if (t != 0 || !valid) return;
  if (!act) {                                     // frozen env: terminated earlier in this episode
    reward[e] = 0.0; terminated[e] = 1;
    for (int c = 0; c < MAPDN_N_INFO; ++c) info[(size_t)e * MAPDN_N_INFO + c] = 0.0;
    return;
  }
  if (d.resetting[e]) {                           // auto_reset: this call was the env's reset(); no transition to report
    reward[e] = 0.0;
    for (int c = 0; c < MAPDN_N_INFO; ++c) info[(size_t)e * MAPDN_N_INFO + c] = 0.0;
    d.draw[e] = bk_draw + 1;                      // (a start that does not solve is re-drawn by the next call, :108)
    if (conv) { terminated[e] = 0; d.steps[e] = 1; d.sum_rewards[e] = 0.0; d.done[e] = 0; }
    else { terminated[e] = 1; d.resetting[e] = 0; }   // the restart did not solve: still frozen, NOT reported as restarted
    return;
  }
  double tot[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) tot[q] = sm[(size_t)(q * Wt) * L];
  {
    const bool ok = conv;
    const double inv_nb = 1.0 / (double)d.nbo;     // all rows of res_bus (voltage_control_env.py:584-589)
    const double out = (tot[0] + tot[1]) * inv_nb;
    const double ql = tot[8] / (double)d.ns, qf = tot[9] / (double)d.ns;
    const double v_loss = tot[6] * inv_nb * d.voltage_weight;
    double loss;
    if (d.use_line_weight) loss = tot[7] / (double)d.n_line * d.line_weight + v_loss;   // :612-613
    else loss = ql * d.q_weight + v_loss;                                                // :614-615
    double rew = -loss;
    double* inf = info + (size_t)e * MAPDN_N_INFO;
    inf[0] = out; inf[1] = tot[0] * inv_nb; inf[2] = tot[1] * inv_nb;
    inf[3] = (out > 1e-3) ? 0.0 : 1.0;
    inf[4] = tot[2] * inv_nb; inf[5] = tot[3] * inv_nb; inf[6] = tot[4]; inf[7] = tot[5];
    inf[8] = tot[7]; inf[9] = ql; inf[10] = 0.0;
    if (!ok) { rew -= 200.0; inf[10] = 1.0; inf[3] = 0.0; inf[9] = qf; }                 // :192-196
    // ---- bookkeeping (the next profile row was queued by k_inject from the pre-increment counters)
    const int st = bk_steps;
    d.draw[e] = bk_draw + 1;
    d.steps[e] = st + 1;
    d.sum_rewards[e] = bk_sum + rew;
    const bool term = (st + 1 >= d.episode_limit) || !ok;                                 // :204
    d.done[e] = term ? 1 : 0;
    reward[e] = rew; terminated[e] = term ? 1 : 0;
  }
  stamp(24);
}

}  // namespace mapdn
