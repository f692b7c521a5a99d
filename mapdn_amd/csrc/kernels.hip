// kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched VoltageControl hot path.
//
// Layout: all per-env state is "env-minor" SoA, X[item][Bp] (Bp = B rounded up to 64), so the lanes that
// serve consecutive envs touch consecutive doubles for every item they process; the LDS-tiled transposes
// convert between env-minor state and env-major I/O tensors.  The wide kernels (inject, advance, gather) are
// one thread per (item, env).  The NR kernel gives a workgroup L envs and splits each of its waves into
// 64/L lane groups ("workers") that eliminate different nodes of the feeder tree for those envs at the same
// time, following a host-built schedule whose per-step records sit in LDS together with the whole solve
// state; see the block comment at K2-K5.
//
// What is computed follows pandapower 2.7.0's runpp (pypower newtonpf; reference call site
// voltage_control_env.py:557) and MAPDN's VoltageControl methods cited at each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.hpp"
#include "philox.hpp"     // Philox4x32-10, u53 (host-checkable: tests/test_philox.py)
#include "nr_common.hpp"
#include "nr_inst_list.hpp"

namespace mapdn {

// =================================================================================================
// K1  inject — _clip_reactive_power (voltage_control_env.py:568-572) for step(), or the random
//     initial action of reset() (:120-122, get_action :337), fused with pandapower
//     build_bus._calc_pq_elements_and_add_on_ppc + makeSbus: Sbus[k] = -(sum load - sum sgen)/sn_mva
//     by element->bus CSR (no atomics).  thread = (position k, env e); the thread of a bus computes
//     the q of the sgens on that bus, so q_new is written exactly once.  MODE_SOLVE takes all four
//     element-power arrays as given (mapdn_solve_only).
// =================================================================================================
// episode start row of reset(): hour, day, interval sampled in the order of voltage_control_env.py:111-113 from the
// env's Philox stream (mapping in oracle/philox.py)
__device__ __forceinline__ int64_t sample_start_row(const Dev& d, int e, uint32_t draw) {
  uint32_t x[4];
  philox4x32_10((uint32_t)(d.env_id_offset + e), draw, STREAM_START, 0u, d.seed_lo, d.seed_hi, x);
  const int64_t hour = (int64_t)(((uint64_t)x[0] * 24u) >> 32);                        // :384
  const int64_t day = (int64_t)(((uint64_t)x[1] * (uint64_t)d.n_start_days) >> 32);    // :398
  const int64_t interval = (int64_t)(((uint64_t)x[2] * (uint64_t)d.per_hour) >> 32);   // :389
  return interval + hour * d.per_hour + day * d.per_day;                                // :445
}
// one column of _set_demand_and_pv (:491-513): table value + std/100 * |N(0,1)|; column j of a stream takes the cosine
// (even j) or sine (odd j) Box-Muller branch of Philox block j >> 1 — the same numbers k_advance produces pairwise
__device__ __forceinline__ double profile_value(const Dev& d, int e, int64_t row, uint32_t draw, int stream, int j, int col0, int add_noise) {
  double v = d.table[(size_t)row * d.ncol + col0 + j];
  if (add_noise) {
    uint32_t x[4];
    philox4x32_10((uint32_t)(d.env_id_offset + e), draw, (uint32_t)stream, (uint32_t)(j >> 1), d.seed_lo, d.seed_hi, x);
    const double u1 = (u53(x[0], x[1]) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = u53(x[2], x[3]) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(2.0 * M_PI * u2, &sn, &cs);
    v += d.stdv[col0 + j] * fabs(r * ((j & 1) ? sn : cs));
  }
  return v;
}

template <typename AT>
__global__ void __launch_bounds__(256)
k_inject(Dev d, int mode, const AT* __restrict__ actions, const double* __restrict__ pl, const double* __restrict__ ql,
         const double* __restrict__ pv, const double* __restrict__ qin, int add_noise) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;                      // 0..n (n == slack: only its sgens' q, no Sbus row)
  if (e >= d.Bp) return;
  bool act = true;
  // auto_reset: an env that terminated in an earlier call starts its next episode in this one (reset(), :96-135, for
  // this env alone): new start row, first profile row of the window (+ noise), random initial action — computed here
  // element by element by the thread of the element's bus, so that every cur_* entry still has exactly one writer
  bool ar = false;
  int64_t ar_row = 0;
  uint32_t draw = 0;
  if (mode != MODE_SOLVE) {
    if (e >= d.B) act = false;
    else if (mode == MODE_STEP) {
      const bool dn = d.done[e] != 0;
      ar = dn && d.auto_reset;
      act = !dn || ar;
    } else act = d.pending[e] != 0;
    draw = mode == MODE_STEP ? d.draw[e] : d.adv_draw[e];
    if (ar) {
      const int64_t start = sample_start_row(d, e, draw);
      ar_row = start + 1;                        // t = self.steps == 1 (:100, :473)
      if (k == 0) d.start_row[e] = start;
    }
    if (k == 0) {
      d.active[e] = act ? 1 : 0;
      if (mode == MODE_STEP) {
        d.resetting[e] = ar ? 1 : 0;
        // step(): the next profile row is row `steps` (before the increment, :199 vs :202) of the episode window and
        // uses the env's current draw counter; queued here, before the solve (frozen / restarting envs do not advance)
        d.adv_row[e] = (act && !ar) ? d.start_row[e] + d.steps[e] : -1; d.adv_draw[e] = draw;
      }
    }
  }
  double P = 0.0, Q = 0.0;
  for (int i = d.load_ptr[k]; i < d.load_ptr[k + 1]; ++i) {
    const int li = d.load_idx[i];
    const size_t o = (size_t)li * d.Bp + e;
    if (ar) {
      const double p = profile_value(d, e, ar_row, draw, STREAM_LOAD_P, li, d.ns, add_noise);
      const double q = profile_value(d, e, ar_row, draw, STREAM_LOAD_Q, li, d.ns + d.nl, add_noise);
      d.cur_pl[o] = p; d.cur_ql[o] = q;
      P += p * d.load_scale[li]; Q += q * d.load_scale[li];
    } else { P += pl[o] * d.load_scale[li]; Q += ql[o] * d.load_scale[li]; }   // pd2ppc: PD = sum p_mw * scaling
  }
  if (mode != MODE_SOLVE && d.sgb_of_pos[k] >= 0)   // load part of the injection at a PV bus, kept for k_inject_sgen
    ((double2*)d.bus_ld)[(size_t)d.sgb_of_pos[k] * d.Bp + e] = make_double2(P, Q);
  for (int i = d.sgen_ptr[k]; i < d.sgen_ptr[k + 1]; ++i) {
    const int j = d.sgen_idx[i];
    const size_t o = (size_t)j * d.Bp + e;
    double p;
    if (ar) { p = profile_value(d, e, ar_row, draw, STREAM_PV, j, 0, add_noise); d.cur_pv[o] = p; d.cur_q[o] = 0.0; }
    else p = pv[o];
    double q;
    if (mode == MODE_SOLVE) q = qin[o];
    else if (!act) q = d.q_new[o];
    else {
      const double sm = d.smax[j];
      const double lim = sqrt(sm * sm - p * p);
      if (mode == MODE_STEP && !ar) q = lim * (double)actions[(size_t)e * d.ns + j];
      else if (d.reset_action) {
        uint32_t x[4];
        philox4x32_10((uint32_t)(d.env_id_offset + e), draw, STREAM_ACTION, (uint32_t)(j >> 1), d.seed_lo, d.seed_hi, x);
        const double u = ((j & 1) ? u53(x[2], x[3]) : u53(x[0], x[1])) * (1.0 / 9007199254740992.0);
        q = lim * (d.action_low + (d.action_high - d.action_low) * u);
      } else q = 0.0;                            // base-net q_mvar (deepcopy of base_powergrid, :106)
      d.q_new[o] = q;
    }
    P -= p * d.sgen_scale[j]; Q -= q * d.sgen_scale[j];
  }
  if (k < d.n) {   // scheduled injection as an (re, im) pair, stored in the order the NR workers consume it
    const size_t o = (size_t)d.sb_index[k] * d.Bp + e;
    const double2 v = make_double2(-P / d.sn, -Q / d.sn);
    ((double2*)((char*)d.nrbuf + d.sb_off))[o] = v;
    if (mode != MODE_SOLVE) ((double2*)((char*)d.nrbuf + d.sb_off_alt))[o] = v;   // both Sbus buffers (see k_advance) are current afterwards
  }
}

// K1'  inject, step()/reset() form.  Between two steps only the sgen q changes: the loads of the next step are written by
//      k_advance at the end of the previous one, which also leaves the finished Sbus entry of every bus WITHOUT sgens and,
//      for the buses with sgens, the load part of the injection (bus_ld).  So this launch has one thread per
//      (PV bus, env) — 22 rows instead of 141 on the 141-bus feeder: q = a sqrt(s_max^2 - p^2) of the bus's sgens and
//      Sbus = -((loads) - (sgens)) / sn, the same expressions in the same order as k_inject (bit-identical).
//      An env that auto-resets in this call (rare) refreshes ALL its loads here, its threads striding over the load buses.
//      (n_sgb >= 1 always: build_plan refuses a net without sgens — no agents — so row 0, which does the step bookkeeping, exists.)
//      Round 4: in step() of a handle without auto_reset these rows run in the prologue of k_nr_tree instead (nr_tree.hpp).
template <typename AT>
__global__ void __launch_bounds__(256)
k_inject_sgen(Dev d, int mode, const AT* __restrict__ actions, int add_noise) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int jb = blockIdx.y;                     // PV-bus number; rows beyond: buses with SEVERAL loads and no sgens
  if (e >= d.B) return;
  const bool mlo_row = jb >= d.n_sgb;
  const int k = mlo_row ? d.mlo_pos[jb - d.n_sgb] : d.sgb_pos[jb];   // elimination position (n == slack: q only, no Sbus row)
  bool act, ar = false;
  if (mode == MODE_STEP) {
    const bool dn = d.done[e] != 0;
    ar = dn && d.auto_reset;
    act = !dn || ar;
  } else act = d.pending[e] != 0;
  const uint32_t draw = mode == MODE_STEP ? d.draw[e] : d.adv_draw[e];
  int64_t ar_row = 0;
  if (ar) {
    const int64_t start = sample_start_row(d, e, draw);
    ar_row = start + 1;                          // t = self.steps == 1 (:100, :473)
    if (jb == 0) d.start_row[e] = start;
  }
  if (jb == 0) {
    d.active[e] = act ? 1 : 0;
    if (mode == MODE_STEP) {
      d.resetting[e] = ar ? 1 : 0;
      d.adv_row[e] = (act && !ar) ? d.start_row[e] + d.steps[e] : -1; d.adv_draw[e] = draw;
    }
  }
  if (!act) return;                              // frozen: q_new, Sbus stay as they are
  double2* const sbp = (double2*)((char*)d.nrbuf + d.sb_off) + e;
  // the load sum of a bus with several loads needs ONE thread (canonical order of its CSR list): formed here from the load
  // values k_advance stored for this step (a restarting env forms all its sums below)
  auto stored_load_sum = [&](int kk, double& Ps, double& Qs) {
    Ps = 0.0; Qs = 0.0;
    for (int i = d.load_ptr[kk]; i < d.load_ptr[kk + 1]; ++i) {
      const int li = d.load_idx[i];
      Ps += d.cur_pl[(size_t)li * d.Bp + e] * d.load_scale[li]; Qs += d.cur_ql[(size_t)li * d.Bp + e] * d.load_scale[li];
    }
  };
  if (mlo_row) {
    if (ar || k >= d.n) return;
    double Ps, Qs;
    stored_load_sum(k, Ps, Qs);
    sbp[(size_t)d.sb_index[k] * d.Bp] = make_double2(-Ps / d.sn, -Qs / d.sn);
    return;
  }
  double2* const bl = (double2*)d.bus_ld + (size_t)jb * d.Bp + e;
  double P, Q;
  if (ar) {
    auto load_sum = [&](int kk, double& Ps, double& Qs) {
      Ps = 0.0; Qs = 0.0;
      for (int i = d.load_ptr[kk]; i < d.load_ptr[kk + 1]; ++i) {
        const int li = d.load_idx[i];
        const size_t o = (size_t)li * d.Bp + e;
        const double p = profile_value(d, e, ar_row, draw, STREAM_LOAD_P, li, d.ns, add_noise);
        const double q = profile_value(d, e, ar_row, draw, STREAM_LOAD_Q, li, d.ns + d.nl, add_noise);
        d.cur_pl[o] = p; d.cur_ql[o] = q;
        Ps += p * d.load_scale[li]; Qs += q * d.load_scale[li];
      }
    };
    load_sum(k, P, Q);
    *bl = make_double2(P, Q);
    for (int i = jb; i < d.n_lb; i += d.n_sgb) {   // the buses with loads but no sgens, shared out over this env's threads
      const int kk = d.lb_pos[i];
      double Ps, Qs;
      load_sum(kk, Ps, Qs);
      if (kk < d.n) {                             // a restarting env is not advanced by this call's k_advance: both Sbus buffers
        const double2 v = make_double2(-Ps / d.sn, -Qs / d.sn);
        sbp[(size_t)d.sb_index[kk] * d.Bp] = v;
        ((double2*)((char*)d.nrbuf + d.sb_off_alt))[(size_t)d.sb_index[kk] * d.Bp + e] = v;
      }
    }
  } else if (d.load_ptr[k + 1] - d.load_ptr[k] > 1) stored_load_sum(k, P, Q);
  else { const double2 v = *bl; P = v.x; Q = v.y; }
  for (int i = d.sgen_ptr[k]; i < d.sgen_ptr[k + 1]; ++i) {
    const int j = d.sgen_idx[i];
    const size_t o = (size_t)j * d.Bp + e;
    double p;
    if (ar) { p = profile_value(d, e, ar_row, draw, STREAM_PV, j, 0, add_noise); d.cur_pv[o] = p; d.cur_q[o] = 0.0; }
    else p = d.cur_pv[o];
    const double sm = d.smax[j];
    const double lim = sqrt(sm * sm - p * p);
    double q;
    if (mode == MODE_STEP && !ar) q = lim * (double)actions[(size_t)e * d.ns + j];
    else if (d.reset_action) {
      uint32_t x[4];
      philox4x32_10((uint32_t)(d.env_id_offset + e), draw, STREAM_ACTION, (uint32_t)(j >> 1), d.seed_lo, d.seed_hi, x);
      const double u = ((j & 1) ? u53(x[2], x[3]) : u53(x[0], x[1])) * (1.0 / 9007199254740992.0);
      q = lim * (d.action_low + (d.action_high - d.action_low) * u);
    } else q = 0.0;                              // base-net q_mvar (deepcopy of base_powergrid, :106)
    d.q_new[o] = q;
    P -= p * d.sgen_scale[j]; Q -= q * d.sgen_scale[j];
  }
  if (k < d.n) sbp[(size_t)d.sb_index[k] * d.Bp] = make_double2(-P / d.sn, -Q / d.sn);
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
  else start = sample_start_row(d, e, dr);
  // A start whose episode window leaves the table (the reference slices past the end and raises IndexError on the empty
  // row, :446-447,473-475) never becomes steppable: not pending, stays done, counted as a reset failure by k_stats.
  const bool bad = start < 0 || start + (int64_t)d.episode_limit + 1 >= d.T;
  d.bad_start[e] = bad ? 1 : 0;
  if (bad) { d.pending[e] = 0; d.start_row[e] = 0; return; }
  d.start_row[e] = start;
  d.adv_row[e] = start + 1;   // t = self.steps == 1 (:100, :473)
  d.adv_draw[e] = dr;
  for (int j = 0; j < d.ns; ++j) d.cur_q[(size_t)j * d.Bp + e] = 0.0;
}

// =================================================================================================
// K9b  advance (+ K6 bus commit) — _set_demand_and_pv (voltage_control_env.py:491-513): next row of the three
//      profile tables + std/100 * |N(0,1)| noise (:498,503,508), and the commit of res_bus for the solve that just finished.
//      rows [0, npv): thread = (Philox block of the PV table, env): one Philox4x32-10 call + one Box-Muller pair serves two
//                     adjacent PV columns;
//      load rows:     thread = (pair of loads, env): their P and Q values (one Philox block of each of the two load streams);
//                     a load that is alone on its bus IS that bus's load sum, so the thread also stores what the next
//                     k_inject_sgen / solve needs — the finished Sbus entry of a bus without sgens, the load part (bus_ld) of
//                     a PV bus — as one 16-byte (P, Q) store;
//                     (the load sum of a bus with SEVERAL loads is formed by the next k_inject_sgen from the stored values);
//      then nb rows:  thread = (bus position, env): the K6 commit.
//      Sbus is DOUBLE-BUFFERED: the commit reads the buffer the solve used (d.sb_off) while the load rows of the same launch
//      write the buffer of the next solve (`sb_write_off`: the other one in step(), the same one in reset(), where the
//      advance comes before the solve); the host flips the buffers after every step.
// =================================================================================================
// the pieces of k_advance as device functions:
// Box-Muller pair of Philox block b of a stream: the two half-normal noise factors of columns 2b, 2b + 1
__device__ __forceinline__ void noise_pair(const Dev& d, int e, uint32_t draw, int stream, int b, double& n0, double& n1) {
  uint32_t x[4];
  philox4x32_10((uint32_t)(d.env_id_offset + e), draw, (uint32_t)stream, (uint32_t)b, d.seed_lo, d.seed_hi, x);
  const double u1 = (u53(x[0], x[1]) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = u53(x[2], x[3]) * (1.0 / 9007199254740992.0);
  const double r = sqrt(-2.0 * log(u1));
  double sn_, cs_;
  sincos(2.0 * M_PI * u2, &sn_, &cs_);
  n0 = fabs(r * cs_); n1 = fabs(r * sn_);
}
// K6 commit of one res_bus row (ORIGINAL bus b_; bus fusion: several buses may share an electrical node) of env e, for envs whose
// solve was accepted (pandapower pfsoln / _extract_results): vm_pu = |V|, va = angle(V), p_mw / q_mvar = bus demand (-Sbus sn) +
// shunt |V|^2, slack = -(V conj(I)) sn.  WANT: the caller needs the row as res_bus holds it after this call in any case
// (a fused commit + obs launch was tried and removed: profiles/r04_wide_kernel_parts_experiment.txt).
template <bool WANT>
__device__ __forceinline__ void commit_bus(const Dev& d, int b_, int e, size_t S, double& v, double& va, double& P, double& Q) {
  const size_t o = (size_t)b_ * S + e;
  if (!d.commit[e]) {
    if (WANT) { v = d.vm[o]; va = d.va[o]; P = d.res_p[o]; Q = d.res_q[o]; }
    return;
  }
  const int k = d.pos_of_obus[b_];               // elimination position of its node, n == slack
  if (k < d.n) {
    const double* vo = d.nrbuf + ((size_t)d.r_vout + (size_t)VOF * k) * S + e;
    const double2 sb = ((const double2*)((const char*)d.nrbuf + d.sb_off))[(size_t)d.sb_index[k] * S + e];
    const double ek = vo[(size_t)VO_E * S], fk = vo[(size_t)VO_F * S];
    v = sqrt(ek * ek + fk * fk);
    va = atan2(fk, ek);
    P = -sb.x * d.sn; Q = -sb.y * d.sn;
  } else {
    v = d.vroot; va = 0.0;
    double ir = d.yrr0 * d.vroot, ii = d.yrr1 * d.vroot;     // I = Y_rr V_r + sum_neighbours Y_rk V_k
    for (int j = 0; j < d.n_root_children; ++j) {
      const double* cb = d.nrbuf + ((size_t)d.r_vout + (size_t)VOF * d.root_children[j]) * S + e;
      const double g = d.root_y[2 * j], b = d.root_y[2 * j + 1], ec = cb[(size_t)VO_E * S], fc = cb[(size_t)VO_F * S];
      ir += g * ec - b * fc; ii += g * fc + b * ec;
    }
    P = -(d.vroot * ir) * d.sn; Q = (d.vroot * ii) * d.sn;
  }
  d.va[o] = va;
  d.vm[o] = v;
  if (d.cm_kind[b_] == 0) { P = P + d.shunt_p[k] * v * v; Q = Q + d.shunt_q[k] * v * v; d.res_p[o] = P; d.res_q[o] = Q; }
  // (a bus of a fused group reports its OWN elements: k_commit_fused wrote p_mw / q_mvar before this launch)
  else if (WANT) { P = d.res_p[o]; Q = d.res_q[o]; }
}
// PV columns 2b, 2b + 1 of the next profile row (one Philox block + one Box-Muller pair)
template <bool WANT>
__device__ __forceinline__ void advance_pv_pair(const Dev& d, int e, int b, int add_noise, size_t S, double& v0, double& v1) {
  const int j0 = 2 * b, j1 = 2 * b + 1;
  const int64_t row = d.adv_row[e];
  if (row < 0 || row >= d.T) {                     // never read outside the table
    if (WANT) { v0 = d.cur_pv[(size_t)j0 * S + e]; v1 = (j1 < d.ns) ? d.cur_pv[(size_t)j1 * S + e] : 0.0; }
    return;
  }
  const double* trow = d.table + (size_t)row * d.ncol;
  v0 = trow[j0]; v1 = (j1 < d.ns) ? trow[j1] : 0.0;
  if (add_noise) {
    double n0, n1;
    noise_pair(d, e, d.adv_draw[e], STREAM_PV, b, n0, n1);
    v0 += d.stdv[j0] * n0;
    if (j1 < d.ns) v1 += d.stdv[j1] * n1;
  }
  d.cur_pv[(size_t)j0 * S + e] = v0;
  if (j1 < d.ns) d.cur_pv[(size_t)j1 * S + e] = v1;
}
// loads 2b, 2b + 1: their P (stream LOAD_P) and Q (stream LOAD_Q) values.  A load that is ALONE on its bus is the bus's whole
// load sum (0 + p * scaling, as k_inject forms it), so the (P, Q) pair goes straight to where the next k_inject_sgen / solve
// reads it, as one 16-byte store — ld_dest[li] = entry << 2 | kind: kind 0 Sbus entry of the next solve, 1 bus_ld row of a
// PV bus, 2 nothing (bus with several loads: the rows above; loads on the slack bus)
__device__ __forceinline__ void advance_load_pair(const Dev& d, int e, int b, int add_noise, size_t S, uint32_t sb_write_off) {
  const int64_t row = d.adv_row[e];
  if (row < 0 || row >= d.T) return;
  double2* const sbw = (double2*)((char*)d.nrbuf + sb_write_off) + e;
  const int j0 = 2 * b, j1 = 2 * b + 1;
  const bool has1 = j1 < d.nl;
  const double* trp = d.table + (size_t)row * d.ncol + d.ns;
  const double* trq = trp + d.nl;
  double p0 = trp[j0], q0 = trq[j0], p1 = has1 ? trp[j1] : 0.0, q1 = has1 ? trq[j1] : 0.0;
  if (add_noise) {
    const uint32_t draw = d.adv_draw[e];
    double n0, n1, m0, m1;
    noise_pair(d, e, draw, STREAM_LOAD_P, b, n0, n1);
    noise_pair(d, e, draw, STREAM_LOAD_Q, b, m0, m1);
    p0 += d.stdv[d.ns + j0] * n0; q0 += d.stdv[d.ns + d.nl + j0] * m0;
    if (has1) { p1 += d.stdv[d.ns + j1] * n1; q1 += d.stdv[d.ns + d.nl + j1] * m1; }
  }
  auto put = [&](int li, double p, double q) {
    d.cur_pl[(size_t)li * S + e] = p; d.cur_ql[(size_t)li * S + e] = q;
    const int dd = d.ld_dest[li];
    const double sc = d.load_scale[li];
    const double ps = p * sc, qs = q * sc;
    if ((dd & 3) == 0) sbw[(size_t)(dd >> 2) * S] = make_double2(-ps / d.sn, -qs / d.sn);
    else if ((dd & 3) == 1) ((double2*)d.bus_ld)[(size_t)(dd >> 2) * S + e] = make_double2(ps, qs);
  };
  put(j0, p0, q0);
  if (has1) put(j1, p1, q1);
}

__device__ __forceinline__ void advance_body(const Dev& d, int add_noise, int do_profiles, int do_commit, uint32_t sb_write_off,
                                             unsigned blk_x, unsigned blk_y) {
  const int e = d.xcd_lanes ? xcd_env(blk_x, threadIdx.x, 256u, (unsigned)d.xcd_lanes, (unsigned)(d.Bp / d.xcd_lanes)) : (int)(blk_x * 256u + threadIdx.x);
  if (e < 0 || e >= d.B) return;
  const int npv = (d.ns + 1) >> 1, npl = (d.nl + 1) >> 1;
  const int npairs = do_profiles ? npv + npl : 0;
  const size_t S = (size_t)d.Bp;
  if ((int)blk_y >= npairs) {                      // rows [npairs, npairs + nbo): thread = (original bus, env)
    double v, va, P, Q;
    commit_bus<false>(d, (int)blk_y - npairs, e, S, v, va, P, Q);
    return;
  }
  if ((int)blk_y < npv) { double v0, v1; advance_pv_pair<false>(d, e, (int)blk_y, add_noise, S, v0, v1); }
  else advance_load_pair(d, e, (int)blk_y - npv, add_noise, S, sb_write_off);
}

__global__ void __launch_bounds__(256) k_advance(Dev d, int add_noise, int do_profiles, int do_commit, uint32_t sb_write_off) {
  advance_body(d, add_noise, do_profiles, do_commit, sb_write_off, blockIdx.x, blockIdx.y);
}

// K6'  res_bus p_mw / q_mvar of the buses of FUSED groups (closed bus-bus switches): pandapower reports, per pandapower bus, the bus's
//      own loads - sgens + shunt |V|^2 (results_bus._get_p_q_results), the ext_grid's injection on the ext_grid's own bus.  thread =
//      (fused bus, env), launched after the solve and BEFORE the profile advance overwrites the element tables.
__global__ void __launch_bounds__(256) k_commit_fused(Dev d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.B || !d.commit[e]) return;
  const size_t S = (size_t)d.Bp;
  auto own = [&](int i, double& P, double& Q) {      // own demand of fused bus i: loads - sgens (the q the accepted solve used)
    P = 0.0; Q = 0.0;
    for (int q = d.ob_load_ptr[i]; q < d.ob_load_ptr[i + 1]; ++q) {
      const int li = d.ob_load_idx[q];
      P += d.cur_pl[(size_t)li * S + e] * d.load_scale[li]; Q += d.cur_ql[(size_t)li * S + e] * d.load_scale[li];
    }
    for (int q = d.ob_sgen_ptr[i]; q < d.ob_sgen_ptr[i + 1]; ++q) {
      const int j = d.ob_sgen_idx[q];
      P -= d.cur_pv[(size_t)j * S + e] * d.sgen_scale[j]; Q -= d.cur_q[(size_t)j * S + e] * d.sgen_scale[j];
    }
  };
  const int i = blockIdx.y;
  const int b_ = d.fused_obus[i], k = d.pos_of_obus[b_];
  double v = d.vroot;
  if (k < d.n) {
    const double* vo = d.nrbuf + ((size_t)d.r_vout + (size_t)VOF * k) * S + e;
    const double ek = vo[(size_t)VO_E * S], fk = vo[(size_t)VO_F * S];
    v = sqrt(ek * ek + fk * fk);
  }
  double P, Q;
  own(i, P, Q);
  P += d.ob_shunt_p[i] * v * v; Q += d.ob_shunt_q[i] * v * v;
  if (d.cm_kind[b_] == 2) {                          // the ext_grid's own bus: minus the ext_grid's injection = (demand of the whole
    double Dp = 0.0, Dq = 0.0;                       // group) + (power the slack node feeds into the network)
    for (int m = 0; m < d.n_slack_group; ++m) { double p_, q_; own(d.slack_group[m], p_, q_); Dp += p_; Dq += q_; }
    double ir = d.yrr0 * d.vroot, ii = d.yrr1 * d.vroot;
    for (int j = 0; j < d.n_root_children; ++j) {
      const double* cb = d.nrbuf + ((size_t)d.r_vout + (size_t)VOF * d.root_children[j]) * S + e;
      const double g = d.root_y[2 * j], b = d.root_y[2 * j + 1], ec = cb[(size_t)VO_E * S], fc = cb[(size_t)VO_F * S];
      ir += g * ec - b * fc; ii += g * fc + b * ec;
    }
    P -= Dp + (d.vroot * ir) * d.sn; Q -= Dq - (d.vroot * ii) * d.sn;
  }
  d.res_p[(size_t)b_ * S + e] = P; d.res_q[(size_t)b_ * S + e] = Q;
}

// =================================================================================================
// K8  observe — get_obs (voltage_control_env.py:232-274) / get_state (:213-230).
//     The effective PV add-back onto res_bus p/q at sgen buses (:238-244) is a per-column list of
//     extra rows to add (x_ptr/x_row): only columns of PV buses have any.
//     k_gather : per output column a precomputed source row of the state block (-1 = zero pad) and
//                a scale -> env-major output through a 64x64 LDS tile so
//                both the env-minor reads and the env-major writes are coalesced.
// =================================================================================================
#define GATHER_HAS_EXTRA 0x40000000   // flag bit in a row descriptor: the column has add-back rows (x_ptr/x_row)
template <typename T, bool XM>
__device__ __forceinline__ void gather_body(const double* __restrict__ base, const int32_t* rows_g, const double* scales_g,
                                            double scale_all, const int32_t* x_ptr_g, const int32_t* x_row_g,
                                            T* __restrict__ out, int C, int B, int Bp, unsigned blk_x, unsigned blk_y, unsigned grd_x, unsigned grd_y, int xl) {
  __shared__ T tile[64][65];                      // output-typed tile: 16.6 KB for f32 -> 8 workgroups per CU
  // descriptors are wave-uniform (a wave handles whole columns): read them through the constant
  // address space -> s_load on the scalar unit, no VMEM round trip ahead of the data loads
  typedef const __attribute__((address_space(4))) int32_t* c_i32;
  typedef const __attribute__((address_space(4))) double* c_f64;
  const c_i32 rows = (c_i32)(unsigned long long)rows_g, x_ptr = (c_i32)(unsigned long long)x_ptr_g, x_row = (c_i32)(unsigned long long)x_row_g;
  const c_f64 scales = (c_f64)(unsigned long long)scales_g;
  // XCD-aware tile order: consecutive workgroups land on consecutive XCDs (block b on XCD b % 8, observed; only speed depends
  // on it), and a source row of 64 envs is read by every column tile whose zones contain that bus (2.4 of them on average) —
  // so all column tiles of an env tile go to ONE XCD, whose L2 then serves the re-reads: XCD k takes env tiles k, k + 8, ...
  unsigned bx = blk_x, by = blk_y;
  if ((grd_y & 7u) == 0u) {
    const unsigned lin = blk_y * grd_x + blk_x, xcd = lin & 7u, slot = lin >> 3;
    by = (slot / grd_x) * 8u + xcd; bx = slot % grd_x;
  }
  const int c0 = (int)bx * 64, e0 = (int)by * 64;
  const int tx = threadIdx.x & 63, ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the 64 envs of this tile: consecutive, or (xl: XCD-aligned order) the tile's groups of xl envs of XCD by % 8
  auto env_of = [&](int r) { if constexpr (XM) return xcd_env(by, (unsigned)r, 64u, (unsigned)xl, (unsigned)(Bp / xl)); else return e0 + r; };
  const int etx = max(env_of(tx), 0);                // (a slot beyond the batch reads env 0 and is never written)
  int rw[16]; double sc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {                 // all descriptor reads first, then the data
    const int c = c0 + ty + 4 * i;
    rw[i] = (c < C) ? rows[c] : -1;
    sc[i] = (scales_g && c < C) ? scales[c] : scale_all;
  }
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    v[i] = (rw[i] >= 0) ? base[(size_t)(rw[i] & ~GATHER_HAS_EXTRA) * Bp + etx] * sc[i] : 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (rw[i] >= 0 && (rw[i] & GATHER_HAS_EXTRA)) {   // wave-uniform, only the PV-bus columns
      const int c = c0 + ty + 4 * i;
      for (int q = x_ptr[c]; q < x_ptr[c + 1]; ++q) v[i] += base[(size_t)x_row[q] * Bp + etx];
    }
    tile[ty + 4 * i][tx] = (T)v[i];
  }
  __syncthreads();
  // write phase: 16 threads x 4 consecutive columns per env row -> 16-byte (f32) / 32-byte (f64) stores
  const int q4 = (threadIdx.x & 15) * 4, er = threadIdx.x >> 4;
  // the 4-column vector store needs C % 4 == 0 AND a base pointer aligned to the vector (callers may hand in a view into a packed buffer)
  const bool vec = (C % 4) == 0 && ((unsigned long long)out % (sizeof(T) * 4)) == 0;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = er + 16 * pass, e = env_of(r), c = c0 + q4;
    if (e < 0 || e >= B || c >= C) continue;
    T* o = out + (size_t)e * C + c;
    if (vec) {                                   // C % 4 == 0 and c % 4 == 0: the 4 columns exist and are aligned
      struct alignas(sizeof(T) * 4) V4 { T a, b, c, d; };
      *reinterpret_cast<V4*>(o) = V4{tile[q4][r], tile[q4 + 1][r], tile[q4 + 2][r], tile[q4 + 3][r]};
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (c + u < C) o[u] = tile[q4 + u][r];
    }
  }
}

template <typename T, bool XM>
__global__ void __launch_bounds__(256)
k_gather(const double* __restrict__ base, const int32_t* rows_g, const double* scales_g,
         double scale_all, const int32_t* x_ptr_g, const int32_t* x_row_g,
         T* __restrict__ out, int C, int B, int Bp, int xl) {
  gather_body<T, XM>(base, rows_g, scales_g, scale_all, x_ptr_g, x_row_g, out, C, B, Bp, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, xl);
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

// stats: pending count, iteration sum / max over active envs (single block)
__global__ void __launch_bounds__(256) k_stats(Dev d, long long* out) {
  __shared__ long long s_pend[256], s_sum[256], s_cnt[256];
  __shared__ int s_max[256];
  long long pend = 0, sum = 0, cnt = 0; int mx = 0;
  for (int e = threadIdx.x; e < d.B; e += 256) {
    pend += (d.pending[e] || d.bad_start[e]) ? 1 : 0;
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
// Calibration of the rocprofv3 HBM counters on THIS library's global access pattern (the MI355X guide: FETCH_SIZE reports half the
// bytes of a wide 16 B / lane streaming read on gfx950; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern").  k_nr_tree moves its scratch as raw-buffer 16-byte loads / stores of env-minor
// pair rows: the 16 lanes of a worker touch 256 contiguous bytes, the four workers of a wave four different rows.
//   pattern 0: that one (workgroup = 16 envs x 16 workers, worker w of pass i takes pair row 16 i + w);
//   pattern 1: a whole wave on one row (64 lanes x 16 B = 1 KB contiguous: the guide's "wide coalesced streaming read").
// Copies rows x Bp x 16 bytes from src to dst (known byte count each way).  tools/calibrate_traffic.py
// =================================================================================================
__global__ void __launch_bounds__(256) k_calib_stream(const double* __restrict__ src, double* __restrict__ dst, int rows, int Bp, int pattern) {
  const unsigned pb = (unsigned)Bp * 16u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, (unsigned)rows * pb, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (unsigned)rows * pb, 0x00020000);
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  if (pattern == 0) {
    const unsigned el = threadIdx.x & 15u, w = threadIdx.x >> 4, e = blockIdx.x * 16u + el;
    for (unsigned r = w; r < (unsigned)rows; r += 16u) {
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 16u, r * pb, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v, rd, e * 16u, r * pb, 0);
    }
  } else {
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    for (unsigned r = blockIdx.y; r < (unsigned)rows; r += gridDim.y) {
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 16u, r * pb, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v, rd, e * 16u, r * pb, 0);
    }
  }
}
}  // namespace mapdn
extern "C" int mapdn_debug_stream(const double* src, double* dst, int32_t rows, int32_t Bp, int32_t pattern, void* stream) {
  if (!src || !dst || rows < 1 || Bp < 256 || (Bp & 255) || (size_t)rows * Bp * 16 >= 0xffffffffull) return MAPDN_E_INVALID;
  if (pattern == 0) hipLaunchKernelGGL(mapdn::k_calib_stream, dim3(Bp / 16), dim3(256), 0, (hipStream_t)stream, src, dst, rows, Bp, 0);
  else hipLaunchKernelGGL(mapdn::k_calib_stream, dim3(Bp / 256, 64), dim3(256), 0, (hipStream_t)stream, src, dst, rows, Bp, 1);
  return hipGetLastError() == hipSuccess ? MAPDN_OK : MAPDN_E_HIP;
}
namespace mapdn {

// =================================================================================================
// launchers (host)
// =================================================================================================

void launch_inject(const Dev& d, int mode, const void* actions, int dtype, const double* pl, const double* ql,
                   const double* pv, const double* q, int add_noise, hipStream_t st) {
  const dim3 grid((d.Bp + 255) / 256, d.nb);
  if (dtype == MAPDN_F32) hipLaunchKernelGGL(k_inject<float>, grid, dim3(256), 0, st, d, mode, (const float*)actions, pl, ql, pv, q, add_noise);
  else hipLaunchKernelGGL(k_inject<double>, grid, dim3(256), 0, st, d, mode, (const double*)actions, pl, ql, pv, q, add_noise);
}
// k_nr_tree instantiations: tables exported by the parts of nr_inst.hip (nr_inst_list.hpp).  RES is the residency of the step
// records / flat-start constants as a compile-time fact (1 both, 2 neither, 3 records only); 0 = the generic body.
static const NrInst* nr_find(int W, int L, bool hl, bool gl, int res) {
  const NrInst* tabs[NR_INST_PARTS] = {nr_insts_0, nr_insts_1, nr_insts_2, nr_insts_3};
  const int cnt[NR_INST_PARTS] = {nr_n_insts_0, nr_n_insts_1, nr_n_insts_2, nr_n_insts_3};
  for (int p = 0; p < NR_INST_PARTS; ++p)
    for (int i = 0; i < cnt[p]; ++i) {
      const NrInst& I = tabs[p][i];
      if (I.W == W && I.L == L && (I.HL != 0) == hl && (I.GL != 0) == gl && I.RES == res) return &I;
    }
  return nullptr;
}
static int nr_res_of(int rec_lds, int flat_lds) { return (rec_lds && flat_lds) ? 1 : (!rec_lds && !flat_lds) ? 2 : (rec_lds && !flat_lds) ? 3 : 0; }
// the instantiation a handle with this geometry / residency runs: the specialised one when compiled in, else the generic body
static const NrInst* nr_pick(int W, int L, int h_lds, int g_lds, int rec_lds, int flat_lds) {
  const bool hl = h_lds != 0, gl = hl && g_lds != 0;
  const int res = nr_res_of(rec_lds, flat_lds);
  const NrInst* I = res ? nr_find(W, L, hl, gl, res) : nullptr;
  return I ? I : nr_find(W, L, hl, gl, 0);
}
#ifdef MAPDN_NR_STAMPS
int nr_debug_stamps_0(unsigned long long*, int); int nr_debug_stamps_1(unsigned long long*, int);
int nr_debug_stamps_2(unsigned long long*, int); int nr_debug_stamps_3(unsigned long long*, int);
static int g_last_part = 0;
}  // namespace mapdn
extern "C" int mapdn_debug_stamps(unsigned long long* out, int n) {   // the stamps of the most recent k_nr_tree launch
  using namespace mapdn;
  switch (g_last_part) { case 0: return nr_debug_stamps_0(out, n); case 1: return nr_debug_stamps_1(out, n);
                         case 2: return nr_debug_stamps_2(out, n); default: return nr_debug_stamps_3(out, n); }
}
namespace mapdn {
#endif
void launch_nr(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st, const void* fused_actions, int fused_dtype) {
  if (d.dense) { launch_nr_dense(d, mode, reward, term, info, st); return; }
  if (d.sparse) { launch_nr_sparse(d, mode, reward, term, info, st); return; }
  const size_t lds = nr_lds_bytes(d.nr_waves, d.nr_lanes, d.n, d.nr_cslots, d.nr_xslots, d.nr_nclist, d.nr_h_lds, d.nr_g_lds,
                                  d.nr_line_lds ? d.n_line : 0, d.nr_rec_lds ? d.nr_rows : 0, d.nr_flat_lds ? d.nr_rows : 0);
  const NrInst* I = nr_pick(d.nr_waves, d.nr_lanes, d.nr_h_lds, d.nr_g_lds, d.nr_rec_lds, d.nr_flat_lds);
  if (!I) return;                                   // (mapdn_create refuses such a geometry: nr_set_lds_limit)
#ifdef MAPDN_NR_STAMPS
  { const NrInst* tabs[NR_INST_PARTS] = {nr_insts_0, nr_insts_1, nr_insts_2, nr_insts_3};
    const int cnt[NR_INST_PARTS] = {nr_n_insts_0, nr_n_insts_1, nr_n_insts_2, nr_n_insts_3};
    for (int p = 0; p < NR_INST_PARTS; ++p) if (I >= tabs[p] && I < tabs[p] + cnt[p]) g_last_part = p; }
#endif
  Dev dd = d;
  dd.fi_actions = (mode == MODE_STEP) ? fused_actions : nullptr; dd.fi_dtype = fused_dtype;
  void* args[] = {(void*)&dd, (void*)&mode, (void*)&reward, (void*)&term, (void*)&info};
  (void)hipLaunchKernel(I->fn, dim3(d.Bp / d.nr_lanes), dim3(64 * d.nr_waves), args, lds, st);
}
// raises the dynamic-LDS limit of the instantiation this geometry runs; -2: the geometry is not compiled in
int nr_set_lds_limit(int waves, int lanes, int h_lds, int g_lds, int rec_lds, int flat_lds, size_t bytes) {
  const NrInst* I = nr_pick(waves, lanes, h_lds, g_lds, rec_lds, flat_lds);
  if (!I) return -2;
  return hipFuncSetAttribute(I->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
// 1 when some instantiation serves (waves, lanes) at all (host-side check, no device call)
// ... and 2 when that instantiation is a specialised one (residency known at compile time: ~6 % faster than the generic body)
int nr_geometry_compiled(int waves, int lanes, int h_lds, int g_lds, int rec_lds, int flat_lds) {
  const NrInst* I = nr_pick(waves, lanes, h_lds, g_lds, rec_lds, flat_lds);
  return I ? (I->RES ? 2 : 1) : 0;
}
void launch_reset_begin(const Dev& d, const int64_t* start_rows, int first_try, hipStream_t st) {
  hipLaunchKernelGGL(k_reset_begin, dim3((d.B + 255) / 256), dim3(256), 0, st, d, start_rows, first_try);
}
// do_profiles: next profile row + noise for the envs queued in adv_row; do_commit: res_bus commit of
// the envs flagged by the preceding k_nr_tree launch
void launch_advance(const Dev& d, int add_noise, int do_profiles, int do_commit, uint32_t sb_write_off, hipStream_t st) {
  const int rows = (do_profiles ? ((d.ns + 1) >> 1) + ((d.nl + 1) >> 1) : 0) + (do_commit ? d.nbo : 0);
  if (rows == 0) return;
  const unsigned gx = d.xcd_lanes ? xcd_blocks((unsigned)d.Bp, 256u, (unsigned)d.xcd_lanes) : (unsigned)((d.B + 255) / 256);
  hipLaunchKernelGGL(k_advance, dim3(gx, rows), dim3(256), 0, st, d, add_noise, do_profiles, do_commit, sb_write_off);
}
void launch_commit_fused(const Dev& d, hipStream_t st) {
  if (d.n_fused > 0) hipLaunchKernelGGL(k_commit_fused, dim3((d.B + 255) / 256, d.n_fused), dim3(256), 0, st, d);
}
void launch_inject_sgen(const Dev& d, int mode, const void* actions, int dtype, int add_noise, hipStream_t st) {
  const dim3 grid((d.B + 255) / 256, d.n_sgb + d.n_mlo);
  if (dtype == MAPDN_F32) hipLaunchKernelGGL(k_inject_sgen<float>, grid, dim3(256), 0, st, d, mode, (const float*)actions, add_noise);
  else hipLaunchKernelGGL(k_inject_sgen<double>, grid, dim3(256), 0, st, d, mode, (const double*)actions, add_noise);
}
void launch_gather(const Dev& d, const double* base, const int32_t* rows, const double* scales, double scale_all,
                   const int32_t* x_ptr, const int32_t* x_row, void* out, int dtype, int C, hipStream_t st) {
  const int xl = d.xcd_lanes;
  dim3 grid((C + 63) / 64, xl ? xcd_blocks((unsigned)d.Bp, 64u, (unsigned)xl) : (unsigned)(d.Bp / 64));
  if (dtype == MAPDN_F32) {
    if (xl) hipLaunchKernelGGL((k_gather<float, true>), grid, dim3(256), 0, st, base, rows, scales, scale_all, x_ptr, x_row, (float*)out, C, d.B, d.Bp, xl);
    else hipLaunchKernelGGL((k_gather<float, false>), grid, dim3(256), 0, st, base, rows, scales, scale_all, x_ptr, x_row, (float*)out, C, d.B, d.Bp, 0);
  } else {
    if (xl) hipLaunchKernelGGL((k_gather<double, true>), grid, dim3(256), 0, st, base, rows, scales, scale_all, x_ptr, x_row, (double*)out, C, d.B, d.Bp, xl);
    else hipLaunchKernelGGL((k_gather<double, false>), grid, dim3(256), 0, st, base, rows, scales, scale_all, x_ptr, x_row, (double*)out, C, d.B, d.Bp, 0);
  }
}
void launch_to_envminor(const Dev& d, const double* src, double* dst, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_to_envminor, dim3((n + 63) / 64, d.Bp / 64), dim3(256), 0, st, src, dst, n, d.B, d.Bp);
}
void launch_copy_i32(const int32_t* s, int32_t* dd, int B, hipStream_t st) { hipLaunchKernelGGL(k_copy_i32, dim3((B + 255) / 256), dim3(256), 0, st, s, dd, B); }
void launch_copy_u8(const uint8_t* s, uint8_t* dd, int B, hipStream_t st) { hipLaunchKernelGGL(k_copy_u8, dim3((B + 255) / 256), dim3(256), 0, st, s, dd, B); }
void launch_stats(const Dev& d, long long* out, hipStream_t st) { hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, st, d, out); }

}  // namespace mapdn
