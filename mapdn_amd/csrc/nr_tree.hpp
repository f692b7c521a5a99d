// nr_tree.hpp — k_nr_tree, the Newton-Raphson power-flow kernel for radial feeders (device code, header-only template).
//
// The template is instantiated in nr_inst.hip, which the build compiles once per PART (-DNR_INST_PART=0..3, in parallel):
// every part takes the address of its share of the (W, L, HL, GL, RES) list in nr_inst_list.hpp and exports it as a table that
// launch_nr (kernels.hip) searches.  What is computed follows pandapower 2.7.0's runpp (pypower newtonpf; reference call site
// voltage_control_env.py:557).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.hpp"
#include "nr_common.hpp"

namespace mapdn {

// =================================================================================================
// K2-K5  Newton-Raphson power flow — pandapower/pypower/newtonpf.py (flat start, polar form, full
//        Jacobian every iteration, ||F||inf < tol, <= 10 iterations), for a radial feeder.
//
//  Per iteration:
//   forward sweep over the feeder tree in leaf->root order, fusing
//     * I = Ybus V and the mismatch F = V conj(I) - Sbus            (dSbus_dV / _evaluate_Fx)
//     * the four Jacobian entries of every Ybus non-zero              (create_jacobian_matrix)
//     * block-2x2 Gaussian elimination J y = F without fill          (replaces SuperLU spsolve)
//   then, unless converged, a backward sweep (root->leaf) that back-substitutes and applies
//   Va += dx_a, Vm += dx_m, V = Vm e^{jVa} with the abs/angle re-normalisation of newtonpf.
//  The first forward sweep uses the host's factorisation of the flat-start Jacobian (the same for every
//  env) and a sweep that is expected to find convergence first runs mismatch-only; see `first` / `light`.
//
//  Parallelism: a workgroup = L envs x W wavefronts; each wavefront is split into 64/L lane groups
//  ("workers": lane = worker * L + env) and the Wt = W * 64/L workers follow a host-built Hu schedule
//  (plan.cpp::build_schedule): in every row each worker eliminates one node of an independent subtree
//  (or idles) for its L envs, so the critical path per sweep is ~the tree radius instead of n.
//
//  Data movement (the solve state is kept ON CHIP; HBM sees one Sbus read, one solution write and
//  the LU factors):
//   * node voltages (e,f) and Sbus are LDS-resident for the whole solve, [node][L envs] per
//     workgroup; a child reads its parent's voltage straight from LDS.
//   * everything that crosses workers inside a sweep — a child's S/Schur contribution to its parent
//     (forward), a parent's x (backward) — goes through LDS slots [slot][item][env] allocated by the
//     host with interval colouring; values stay in registers instead when the same worker handles
//     the parent in the adjacent row.  Rows are separated by an LDS-only barrier when W > 1
//     (s_waitcnt lgkmcnt(0); s_barrier) and by nothing at all when one wave holds all workers.
//   * only the LU factors G (4 doubles per node; with h, 6, when h does not fit in LDS) go to global
//     scratch: every (worker,row) step owns one FACTOR BLOCK addressed as block(worker,row) + field through
//     one buffer resource (scalar row offset, loop-invariant lane VGPR offsets); the backward sweep
//     prefetches them two rows ahead.  Every step issues the same VMEM instructions (prefetches past the
//     ends are clamped, never skipped), so vmcnt waits are exact.
//   * step constants (Y entries, flags, slot ids) are 96-byte records staged once in LDS.
//   * the solution (e, f) goes to the Vout region; |V| and angle are formed once, by the commit part of
//     k_advance (Vm = |V|, Va = angle(V) as newtonpf) — or here, in mapdn_solve_only's MODE_SOLVE.
//  The linear system is solved for z = [dtheta ; d|V|/|V|] (|V| columns scaled by |V_k|): the
//  entries are j(S - A_kk), S + A_kk on the diagonal and -jA_ik, A_ik off it — no division by |V|;
//  1/det uses v_rcp_f64 + two Newton steps; the update rotates V by the small step angle
//  (polynomial sin/cos, full sincos only when a lane diverges).  Children are summed in a canonical
//  order, so results are bit-identical for every W and L.
// =================================================================================================

#ifdef MAPDN_NR_STAMPS
// debug build only (-DMAPDN_NR_STAMPS): cycle stamps of workgroup 0 / wave 0 at phase and row boundaries
static __device__ unsigned long long g_stamps[4096];   // one copy per translation unit (nr_inst.hip part)
#define STAMP(id) do { if (stamp_on && ns < 4095) g_stamps[++ns] = ((unsigned long long)(id) << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull); } while (0)
#ifdef MAPDN_NR_STAMPS_FINE
#define STAMP2(id) STAMP(id)
#else
#define STAMP2(id) do { } while (0)
#endif
#else
#define STAMP(id) do { } while (0)
#define STAMP2(id) do { } while (0)
#endif


#if defined(MAPDN_EXP) || defined(MAPDN_NO_FUSE_PROLOGUE)
#error "MAPDN_EXP / MAPDN_NO_FUSE_PROLOGUE were timing-experiment switches of rounds 3-4 (results under profiles/); they produced wrong results by design and are gone"
#endif
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}


// buffer_load/store_dwordx4 v, v_lane_offset, s[rsrc], s_row_offset offen : zero VALU address math
__device__ __forceinline__ d2 bld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4 bldu4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void bst2(d2 x, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), r, voff, soff, 0);
}

// Per-env max over the S = 64 / L workers of a wavefront (lane = worker * L + env), every lane receiving the result — VALU-rate
// lane exchanges instead of an LDS round trip.  The four 16-lane rows of the wave are combined by two gfx950 row swaps
// (v_permlane16_swap / v_permlane32_swap); with fewer than 16 envs per workgroup a row holds 16 / L workers, combined first by DPP
// rotations within the row (row_ror:L, row_ror:2L ...).  max is exact, so the order of the combination does not matter.
template <int CTRL>
__device__ __forceinline__ double dpp_row_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int L>
__device__ __forceinline__ double grp_max(double v) {
  static_assert(L == 4 || L == 8 || L == 16 || L == 32, "envs per workgroup");
  constexpr int ROW_ROR = 0x120;                   // DPP control: rotate right within each row of 16 lanes
  if constexpr (L == 4) v = fmax(v, dpp_row_f64<ROW_ROR + 4>(v));
  if constexpr (L <= 8) v = fmax(v, dpp_row_f64<ROW_ROR + 8>(v));
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u32x2 a, b;
  if constexpr (L <= 16) {
    a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  }
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return fmax(__hiloint2double((int)b.x, (int)a.x), __hiloint2double((int)b.y, (int)a.y));
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}
// ... and downwards: f(N - 1), ..., f(0)
template <int N, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
  if constexpr (N > 0) { f(std::integral_constant<int, N - 1>{}); static_for_down<N - 1>(f); }
}

struct Rec { u32x4 ix; d2 ykk, ykp, ypk, cks, sb; };    // per-(worker,row) constants + the env's scheduled injection
struct RecF { u32x4 ix; d2 s, i01, i23, ap, sb; };      // flat-start form: host-factorised constants (Schedule::flat)
struct BwdF { d2 h, g01, g23; };                        // factors of one step when they come from global memory

// HL / GL: the h / G factors live in LDS (when they fit) instead of global scratch.  RES: 1 = step records and flat-start
// constants are LDS-resident (compile-time: the "fat" geometry), 2 = neither is (the "lean" one), 3 = the records are, the
// flat-start constants are not (the 322-bus feeder: W = 4, L = 8), 0 = per handle (d.nr_*_lds)
template <int W, int L, bool HL, bool GL, int RES = 0>
__global__ void __launch_bounds__(64 * W)
k_nr_tree(Dev d, int mode, double* __restrict__ reward, uint8_t* __restrict__ terminated, double* __restrict__ info) {
  extern __shared__ d2 lds2[];
  // worker = (wave w, lane group s): S = 64/L sub-workers per wave, each serving the same L envs of
  // this workgroup but eliminating a DIFFERENT node per row.  The step body is branch-free per lane (zero / trash
  // slots instead of predicated LDS traffic); whole groups of LDS instructions that would only move zeros or
  // trash for every worker of the wave are skipped on the wave-uniform hints the host put into the records.
  constexpr unsigned S = 64u / L, Wt = (unsigned)W * S;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned w = threadIdx.x >> 6;
  const unsigned sw = lane / L, el = lane % L;
  const unsigned t = w * S + sw;                 // worker id, 0 .. Wt-1
  const unsigned e = blockIdx.x * L + el;
  const int R = d.nr_rows;
  const unsigned n = (unsigned)d.n;
  const double vroot = d.vroot, tol = d.tol;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d.nrbuf, 0, d.nrbuf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(d.flat), 0, d.flat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<StepRec*>(d.sched), 0, d.sched_bytes, 0x00020000);
  const unsigned pb = (unsigned)d.Bp * 16u;      // bytes per pair row (one d2 per env)
  constexpr unsigned TB = (unsigned)sizeof(StepRec);        // bytes per step record
  constexpr unsigned FB = (unsigned)FLAT_N * 8u;            // bytes per flat-start step
  const unsigned bb = (unsigned)NBP * pb;        // bytes per factor block
  const unsigned voT = t * (unsigned)R * TB;     // this worker's records (same address for its L lanes)
  const unsigned voF = t * (unsigned)R * FB;     // this worker's flat-start steps
  // factor blocks are addressed by NODE (block k of env e at e*16 + k*bb, field offset on the scalar side): the idle steps of
  // all workers share the trash node's block, so a sweep moves n blocks per env, not workers x rows
  const unsigned voE = e * 16u;
  const unsigned sF_H = __builtin_amdgcn_readfirstlane(NB_H * pb), sF_G01 = __builtin_amdgcn_readfirstlane(NB_G01 * pb),
                 sF_G23 = __builtin_amdgcn_readfirstlane(NB_G23 * pb);
  const unsigned voS = d.sb_off + e * 16u;       // the injection Sbus, one (re, im) pair row per NODE (entry n + 1, the trash node idle steps work on, stays 0)
  // ---- fused PV-bus injection (step() of a handle without auto_reset; otherwise k_inject_sgen ran as a launch of its own).
  // _clip_reactive_power (voltage_control_env.py:568-572): q = a sqrt(s_max^2 - p^2) of the sgens of every PV bus of this workgroup's
  // envs, Sbus = -((loads) - (sgens)) / sn with the load part k_advance left in bus_ld — the expressions of k_inject_sgen for an env
  // that is not restarting, in the same order (bit-identical: test_fused_injection_equals_the_injection_launch) — and the step
  // bookkeeping its first row does.  One item per (PV bus, env), env fastest; a launch of its own cost 6.8 us for 90 k such
  // items (launch + a chain of dependent loads); here the chain runs once per workgroup beside the LDS initialisation.  The
  // Sbus entries are read back by this workgroup only, after the barrier that ends the prologue.
  if (d.fi_actions) {
    // Items are taken IT at a time per thread with ALL loads of a level issued before any is used (record + flags, then the
    // values): the chain is paid once per batch, not once per item.  Loads are unconditional with clamped indices, only the
    // stores are predicated.  (A first version with one item per loop turn cost 3.5 us on the 141-bus feeder: two turns of a
    // three-deep chain of ~2 k-cycle HBM round trips — the operands were written by the previous launches.)
    constexpr unsigned IT = 3, NT = 64u * W;
    const unsigned nitems = (unsigned)(d.n_sgb + d.n_mlo) * L;
    const size_t S_ = (size_t)d.Bp;
    const unsigned Bm1 = (unsigned)d.B - 1u;
    const bool a32 = d.fi_dtype == MAPDN_F32;
    for (unsigned base = threadIdx.x; base < nitems; base += IT * NT) {
      int4 rec[IT]; int2 rl[IT]; unsigned jb[IT], e2[IT]; bool ok[IT]; uint8_t dn[IT];
#pragma unroll
      for (unsigned it = 0; it < IT; ++it) {
        const unsigned i = base + it * NT;
        ok[it] = i < nitems;
        const unsigned ii = ok[it] ? i : 0u;
        jb[it] = ii / L;
        const unsigned ee = blockIdx.x * L + (ii % L);
        ok[it] = ok[it] && ee < (unsigned)d.B;
        e2[it] = min(ee, Bm1);
        rec[it] = ((const int4*)d.sgb_rec)[2 * jb[it]];
        rl[it] = ((const int2*)d.sgb_rec)[4 * jb[it] + 2];
        dn[it] = d.done[e2[it]];
      }
      double2 blv[IT]; double pv0[IT], sm0[IT], sc0[IT], a0[IT];
      int bk_st[IT]; int64_t bk_row[IT]; uint32_t bk_dr[IT];
#pragma unroll
      for (unsigned it = 0; it < IT; ++it) {
        const int j0c = max(rec[it].z, 0);
        blv[it] = ((const double2*)d.bus_ld)[(size_t)min(jb[it], (unsigned)d.n_sgb - 1u) * S_ + e2[it]];
        pv0[it] = d.cur_pv[(size_t)j0c * S_ + e2[it]];
        sm0[it] = d.smax[j0c]; sc0[it] = d.sgen_scale[j0c];
        const size_t ai = (size_t)e2[it] * d.ns + j0c;
        a0[it] = a32 ? (double)((const float*)d.fi_actions)[ai] : ((const double*)d.fi_actions)[ai];
        bk_st[it] = 0; bk_row[it] = 0; bk_dr[it] = 0;
        if (jb[it] == 0) { bk_st[it] = d.steps[e2[it]]; bk_row[it] = d.start_row[e2[it]]; bk_dr[it] = d.draw[e2[it]]; }
      }
      // buses with several loads (none on the 33- / 141-bus feeders, 37 on the 322-bus one): the stored values of their first
      // two loads, requested with the rest (wave-uniform branch)
      double lp0[IT], lq0[IT], ls0[IT], lp1[IT], lq1[IT], ls1[IT];
      bool multi = false;
#pragma unroll
      for (unsigned it = 0; it < IT; ++it) multi = multi || (ok[it] && (rec[it].w & 255) > 1);
      if (__any(multi)) {
#pragma unroll
        for (unsigned it = 0; it < IT; ++it) {
          const int l0 = max(rl[it].x, 0), l1 = max(rl[it].y, 0);
          lp0[it] = d.cur_pl[(size_t)l0 * S_ + e2[it]]; lq0[it] = d.cur_ql[(size_t)l0 * S_ + e2[it]]; ls0[it] = d.load_scale[l0];
          lp1[it] = d.cur_pl[(size_t)l1 * S_ + e2[it]]; lq1[it] = d.cur_ql[(size_t)l1 * S_ + e2[it]]; ls1[it] = d.load_scale[l1];
        }
      } else {
#pragma unroll
        for (unsigned it = 0; it < IT; ++it) { lp0[it] = lq0[it] = ls0[it] = lp1[it] = lq1[it] = ls1[it] = 0.0; }
      }
#pragma unroll
      for (unsigned it = 0; it < IT; ++it) {
        if (!ok[it]) continue;
        const unsigned ee = e2[it];
        const bool act2 = dn[it] == 0;
        if (jb[it] == 0) {
          d.active[ee] = act2 ? 1 : 0; d.resetting[ee] = 0;
          d.adv_row[ee] = act2 ? bk_row[it] + bk_st[it] : -1; d.adv_draw[ee] = bk_dr[it];
        }
        if (!act2) continue;                       // frozen: q_new, Sbus stay as they are
        const int sbi = rec[it].x, k = rec[it].y, j0 = rec[it].z, nsg = rec[it].w >> 8, nld = rec[it].w & 255;
        double P, Q;
        if (nld > 1) {                             // several loads on the bus: the sum of the stored values, in CSR order
          P = 0.0; Q = 0.0;
          P += lp0[it] * ls0[it]; Q += lq0[it] * ls0[it];
          P += lp1[it] * ls1[it]; Q += lq1[it] * ls1[it];
#pragma unroll 1
          for (int q = d.load_ptr[k] + 2; q < d.load_ptr[k + 1]; ++q) {     // a third, fourth ... load (rare)
            const int li = d.load_idx[q];
            P += d.cur_pl[(size_t)li * S_ + ee] * d.load_scale[li]; Q += d.cur_ql[(size_t)li * S_ + ee] * d.load_scale[li];
          }
        } else if (j0 >= 0) { P = blv[it].x; Q = blv[it].y; }
        else { P = 0.0; Q = 0.0; }
        if (nsg > 0) {
          {
            const double p = pv0[it], sm = sm0[it];
            const double lim = sqrt(sm * sm - p * p);
            const double qv = lim * a0[it];
            d.q_new[(size_t)j0 * S_ + ee] = qv;
            P -= p * sc0[it]; Q -= qv * sc0[it];
          }
#pragma unroll 1
          for (int q = 1; q < nsg; ++q) {          // further sgens on the same bus (rare)
            const int j = d.sgen_idx[d.sgen_ptr[k] + q];
            const size_t o = (size_t)j * S_ + ee;
            const double p = d.cur_pv[o];
            const double sm = d.smax[j];
            const double lim = sqrt(sm * sm - p * p);
            const size_t ai = (size_t)ee * d.ns + j;
            const double qv = lim * (a32 ? (double)((const float*)d.fi_actions)[ai] : ((const double*)d.fi_actions)[ai]);
            d.q_new[o] = qv;
            P -= p * d.sgen_scale[j]; Q -= qv * d.sgen_scale[j];
          }
        }
        if (sbi >= 0) ((double2*)((char*)d.nrbuf + d.sb_off))[(size_t)sbi * S_ + ee] = make_double2(-P / d.sn, -Q / d.sn);
      }
    }
  }
  // LDS map, in pair rows (L x 16 bytes: one d2 per env; a worker's 16 lanes read 256 contiguous bytes with one
  // conflict-free ds_read_b128):  V [n+2] | h [n+2] if HL | G [2(n+2)] if GL | contribution slots x 4 | x slots x 1
  //   then verdict bytes [64 W], step sizes [64 W doubles], overflow child list, net.line constants (when they fit)
  // node n = slack, n+1 = trash
  d2* sV = lds2 + el;                                        // sV[k*L] = (e, f)
  d2* sH = sV + (size_t)(n + 2) * L;                         // sH[k*L] = (h0, h1)
  d2* sG = sH + (HL ? (size_t)(n + 2) * L : 0);              // sG[(2k+j)*L] = (G0,G1), (G2,G3)
  d2* cs = sG + (GL ? (size_t)2 * (n + 2) * L : 0);          // cs[(slot*4 + j)*L] = (S0,S1) (D0,D1) (D2,D3) (R0,R1)
  d2* xs = cs + (size_t)d.nr_cslots * 4 * L;                 // xs[slot*L] = (x0, x1)
  uint8_t* s_ok = (uint8_t*)(xs - el + (size_t)d.nr_xslots * L);   // [Wt][L], Wt*L = 64*W
  double* s_dx = (double*)(s_ok + 64 * W) + el;              // step-size partials: s_dx[worker*L], 64*W doubles
  double* s_epi = (double*)(cs - el) + el;                   // epilogue partials s_epi[(q*Wt + worker)*L], 10*64*W doubles: they re-use the
                                                             // contribution slots, dead once the solve is over (host: cslots >= nr_min_cslots)
  int32_t* s_clist = (int32_t*)(s_ok + 64 * W + 64 * W * sizeof(double));
  double* s_lines = (double*)(s_clist + ((d.nr_nclist + 3) & ~3));   // LineFlow rows of net.line (5 doubles each) when d.nr_line_lds
  // step records (80 B) and flat-start constants (96 B) of all workers, when they fit (d.nr_rec_lds / d.nr_flat_lds): a
  // worker's 16 lanes read the same 16 bytes (LDS broadcast)
  char* s_rec = (char*)s_lines + (d.nr_line_lds ? nr_line_bytes(d.n_line) : 0);
  char* s_flat = s_rec + (d.nr_rec_lds ? (size_t)Wt * R * sizeof(StepRec) : 0);
  {  // LDS init: flat start (runpp init="auto": every bus at the slack set-point), ZERO slots, small tables
    const d2 v0 = {vroot, 0.0}, z2 = {0.0, 0.0};
    for (unsigned k = t; k < n + 2; k += Wt) sV[(size_t)k * L] = v0;
    for (unsigned i = threadIdx.x; i < (unsigned)d.nr_nclist; i += 64u * W) s_clist[i] = d.clist[i];
    if (HL && t == 0) sH[(size_t)n * L] = z2;    // the slack entry of the h / x array: the x an elimination root reads
    if (t == 0) {                                // the ZERO slots (second-to-last of each kind) read as 0 forever
#pragma unroll
      for (int i = 0; i < 4; ++i) cs[((size_t)(d.nr_cslots - 2) * 4 + i) * L] = z2;
      xs[(size_t)(d.nr_xslots - 2) * L] = z2;
    }
    if (d.nr_line_lds) {                         // res_line constants for the epilogue (uniform branch)
      const uint4* ls = (const uint4*)d.lines;
      uint4* ld = (uint4*)s_lines;
      const unsigned nl4 = (unsigned)(nr_line_bytes(d.n_line) / 16);
      for (unsigned base = threadIdx.x; base < nl4; base += 4 * 64u * W) {
        const uint4 a0 = ls[base], a1 = ls[min(base + 64u * W, nl4 - 1)], a2 = ls[min(base + 2 * 64u * W, nl4 - 1)],
                    a3 = ls[min(base + 3 * 64u * W, nl4 - 1)];
        ld[base] = a0;
        if (base + 64u * W < nl4) ld[base + 64u * W] = a1;
        if (base + 2 * 64u * W < nl4) ld[base + 2 * 64u * W] = a2;
        if (base + 3 * 64u * W < nl4) ld[base + 3 * 64u * W] = a3;
      }
    }
    // Global -> LDS staging in batches: the loads of a batch are unconditional (index clamped) and issued back to back,
    // only the LDS stores are predicated — a load-per-iteration loop would pay the full memory latency once per element
    auto stage = [&](const void* src_, char* dst_, unsigned n16) {
      const uint4* src = (const uint4*)src_;
      uint4* dst = (uint4*)dst_;
      constexpr unsigned NT = 64u * W;
      for (unsigned base = threadIdx.x; base < n16; base += 8 * NT) {
        const uint4 r0 = src[base], r1 = src[min(base + NT, n16 - 1)], r2 = src[min(base + 2 * NT, n16 - 1)], r3 = src[min(base + 3 * NT, n16 - 1)],
                    r4 = src[min(base + 4 * NT, n16 - 1)], r5 = src[min(base + 5 * NT, n16 - 1)], r6 = src[min(base + 6 * NT, n16 - 1)],
                    r7 = src[min(base + 7 * NT, n16 - 1)];
        dst[base] = r0;
        if (base + NT < n16) dst[base + NT] = r1;
        if (base + 2 * NT < n16) dst[base + 2 * NT] = r2;
        if (base + 3 * NT < n16) dst[base + 3 * NT] = r3;
        if (base + 4 * NT < n16) dst[base + 4 * NT] = r4;
        if (base + 5 * NT < n16) dst[base + 5 * NT] = r5;
        if (base + 6 * NT < n16) dst[base + 6 * NT] = r6;
        if (base + 7 * NT < n16) dst[base + 7 * NT] = r7;
      }
    };
    if (d.nr_rec_lds) stage(d.sched, s_rec, Wt * (unsigned)R * (unsigned)(sizeof(StepRec) / 16));
    if (d.nr_flat_lds) stage(d.flat, s_flat, Wt * (unsigned)R * (unsigned)(FLAT_N * 8 / 16));
  }
#ifdef MAPDN_NR_STAMPS
  const bool stamp_on = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned ns = 0;
#endif
  STAMP(1);
  // this env takes part in the solve (same for all its workers); with the fused injection the flag is formed as the prologue formed it
  const bool act = d.fi_actions ? (e < (unsigned)d.B && d.done[e] == 0) : (d.active[e] != 0);
  // step() bookkeeping inputs, fetched now so that their latency is not paid at the very end
  const int bk_steps = d.steps[e];
  const uint32_t bk_draw = d.draw[e];
  const double bk_sum = d.sum_rewards[e];
  __syncthreads();
  STAMP(2);
  bool done = !act;
  bool conv = false;
  int it = 0;
  const bool nothing_to_solve = __all(done);      // identical in all waves of the group

  bool allok;
  double fmx;                                      // largest mismatch component seen by this worker in the current sweep
  double Fprev = 0.0, Fcur = 0.0;                  // per env: ||F||inf of the two most recent accepted sweeps
  double cS0, cS1, cD0, cD1, cD2, cD3, cR0, cR1;   // register carry child -> parent (same worker, next row)
  // G factors of the first KR rows stay in REGISTERS between a forward sweep and its backward sweep: a (worker, row) step
  // is the same lane's in both, and with one wave per SIMD (the fat layouts) most of the register file — 512 VGPRs + AGPRs
  // per lane — is unused.  Registers cannot be indexed by a loop counter for free (s_set_gpr_idx per dword was measured:
  // +9 % kernel time), so those rows are PEELED: rows 0 .. KR-1 of the full forward sweep and of its backward sweep are
  // straight-line code with the row number a compile-time constant, and the row's G is eight named 32-bit values that live
  // in AGPRs (v_accvgpr_write / _read through the "a" constraint, so that they never compete for arch VGPRs).  The Hu
  // schedule fills the leaf-side rows first — 94 of the 140 nodes of the 141-bus feeder sit in rows 0..5, 120 in rows
  // 0..8 — so only the sparse root-side rows still send G through global scratch: HBM-side traffic of the launch
  // 77 -> 30 MB (case141 x 4096), 152 -> 78 MB (case322 x 4096) at unchanged kernel time (+-1 %).  The peeled code is
  // instruction-cache footprint (its first execution in a launch is cold), which is what limits KR.
  // When h lives in global scratch too (!HL: the lean layouts, the 322-bus feeder at 16 envs per workgroup), the peeled rows keep
  // h AND G in registers (12 AGPRs per row) and there are NR_HG_REG_ROWS of them.
  constexpr int KR = GL ? 0 : (HL ? NR_G_REG_ROWS : NR_HG_REG_ROWS);
  constexpr bool RH = !HL;                         // the peeled rows' h is in registers as well
  uint32_t Ga[KR > 0 ? KR : 1][8];                 // AGPR-class values: written / read only by the two helpers below
  uint32_t Ha[(KR > 0 && RH) ? KR : 1][4];
  auto a_put = [](double v, uint32_t& lo, uint32_t& hi) {
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(v)));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(v)));
  };
  // a forward sweep that computes no G still DEFINES the registers (a distinct number in each statement's comment, so that they are not merged):
  // every path from a forward to a backward sweep then carries defined values and nothing is live around the iteration loop
  auto a_def1 = [](uint32_t& x, auto qc) { asm("v_accvgpr_write_b32 %0, 0 ; def %1" : "=a"(x) : "n"(decltype(qc)::value)); };
  auto a_define = [&]() {
    static_for<KR * 8>([&](auto ic) { constexpr int q = decltype(ic)::value; a_def1(Ga[q / 8][q % 8], ic); });
    if constexpr (RH) static_for<KR * 4>([&](auto ic) { constexpr int q = decltype(ic)::value; a_def1(Ha[q / 4][q % 4], std::integral_constant<int, 1000 + q>{}); });
  };
  auto a_get = [](uint32_t lo, uint32_t hi) -> double {
    int l, h;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
    return __hiloint2double(h, l);
  };
  double x0, x1;                                   // register carry parent -> child in the backward sweep

  // ---- per-row constants: all addressed by (worker, row) only, so they are fetched PF rows ahead with no dependent
  // address; the scalar row offset is the only thing that changes (prefetches past the ends are clamped, never
  // skipped: every step issues the same VMEM instructions and the compiler's s_waitcnt counts stay exact)
  auto row_s = [&](int row, unsigned stride) { return __builtin_amdgcn_readfirstlane((unsigned)row * stride); };
  const bool recL = (RES == 1 || RES == 3) ? true : RES == 2 ? false : d.nr_rec_lds != 0;      // wave-uniform; compile-time when RES != 0
  const bool flatL = RES == 1 ? true : (RES == 2 || RES == 3) ? false : d.nr_flat_lds != 0;
  const char* recT = s_rec + voT;                  // this worker's records / flat steps in LDS
  const char* flatT = s_flat + voF;
  auto load_ix = [&](int row) -> u32x4 {
    if (recL) return *(const u32x4*)(recT + (unsigned)row * TB);
    return bldu4(rsT, voT, row_s(row, TB));
  };
  auto load_rec = [&](int row, Rec& o) {
    if (recL) {
      const char* p = recT + (unsigned)row * TB;
      o.ix = *(const u32x4*)p;
      o.ykk = *(const d2*)(p + 16); o.ykp = *(const d2*)(p + 32); o.ypk = *(const d2*)(p + 48); o.cks = *(const d2*)(p + 64);
    } else {
      const unsigned st = row_s(row, TB);
      o.ix = bldu4(rsT, voT, st);
      o.ykk = bld2(rsT, voT + 16u, st); o.ykp = bld2(rsT, voT + 32u, st); o.ypk = bld2(rsT, voT + 48u, st); o.cks = bld2(rsT, voT + 64u, st);
    }
  };
  auto load_recf = [&](int row, RecF& o) {
    o.ix = load_ix(row);
    if (flatL) {
      const char* p = flatT + (unsigned)row * FB;
      o.s = *(const d2*)(p + FL_SR * 8); o.i01 = *(const d2*)(p + FL_I0 * 8); o.i23 = *(const d2*)(p + FL_I2 * 8); o.ap = *(const d2*)(p + FL_APR * 8);
    } else {
      const unsigned sf = row_s(row, FB);
      o.s = bld2(rsF, voF + FL_SR * 8u, sf); o.i01 = bld2(rsF, voF + FL_I0 * 8u, sf); o.i23 = bld2(rsF, voF + FL_I2 * 8u, sf);
      o.ap = bld2(rsF, voF + FL_APR * 8u, sf);
    }
  };
  // the env's injection at a node: addressed by node, so that the Sbus array is n pair rows, not workers x rows (its re-reads in
  // every sweep then stay in L2 together with the G factor scratch); requested one row ahead, when the next row's node is known
  auto load_sb = [&](uint32_t kp) -> d2 { return bld2(rs, voS + (kp & 0xffffu) * pb, 0u); };
  auto uni = [&](unsigned x) { return __builtin_amdgcn_readfirstlane(x); };

  // ---------------------------------------------------------------------------------------------------------------
  // Row anatomy.  With 16 envs per CU a SIMD runs ONE wave, so nothing hides latency but the order of the wave's own
  // instructions.  Every row is therefore laid out by hand (sched_barrier fences keep the compiler from undoing it):
  //   (1) issue the LDS reads that depend on the previous row (children's contributions / parent's x) and the prefetches
  //   (2) SHADOW: work that does not depend on them — this row's child-independent Jacobian terms, the convergence
  //       bookkeeping DEFERRED from the previous row — runs while those reads are in flight
  //   (3) the dependent chain (sums -> pivot -> factors -> contribution) and its LDS write, then the row barrier
  // Only (3) sits between two barriers of the inter-row dependency chain.
  // Absent children / parents are the ZERO slot, unread outputs go to the TRASH slot / node; whole groups of LDS
  // instructions that would only move zeros or trash for every worker of the wave are skipped on the wave-uniform hints.
  // ---------------------------------------------------------------------------------------------------------------
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
  // mismatch bookkeeping of one step (deferred into the next row's shadow)
  auto note_mismatch = [&](double Fp, double Fq, bool live) {
    // (bitwise on purpose: the short-circuit form compiled to two exec-mask branches per row; same booleans)
    const bool okp = fabs(Fp) < tol, okq = fabs(Fq) < tol;
    allok = allok & (!live | (okp & okq));
    fmx = fmax(fmx, live ? fmax(fabs(Fp), fabs(Fq)) : 0.0);
  };
  auto clist_ptr = [&](uint32_t fl, uint32_t slots, uint32_t chs) { return (fl >> 24) | ((slots >> 30) << 8) | ((chs >> 30) << 10); };

  // forward sweep.  K 0: full (mismatch + Jacobian + block elimination), 1: mismatch only (no Jacobian, no elimination, no
  // factor stores; only the S part of the contributions, pair 0, travels — same expressions, same order as the full
  // step, so the verdict is the one the full step would reach).  Records PF rows ahead (global), own / parent voltage
  // one row ahead (LDS: V is constant during a forward sweep); unrolled by 3 so that ring indices are compile-time.
  auto fwd_sweep = [&](auto kind) {
    constexpr int K = decltype(kind)::value;
    Rec Tq[3]; d2 vkq[3], vpq[3];
    load_rec(0, Tq[0]); load_rec(min(1, R - 1), Tq[1]);
    { const unsigned kp = Tq[0].ix.w; vkq[0] = sV[(size_t)(kp & 0xffffu) * L]; vpq[0] = sV[(size_t)(kp >> 16) * L]; Tq[0].sb = load_sb(kp); }
    double pFp = 0.0, pFq = 0.0; bool pLive = false;       // deferred mismatch bookkeeping of the previous row
    // one row; u = ring position (compile-time), RS = the row number when it is a compile-time constant (peeled rows), else -1
    auto fwd_row = [&](auto uc, auto rsc, int r) {
      constexpr int u = decltype(uc)::value, RS = decltype(rsc)::value;
      {
        const Rec& T = Tq[u % 3];
        const d2 vk = vkq[u % 3], vp = vpq[u % 3];
        const uint32_t fl = T.ix.x, slots = T.ix.y, chs = T.ix.z, kp = T.ix.w;
        const uint32_t flu = uni(fl);              // the wave-uniform hints, once per row on the scalar unit
        const unsigned gmax = (flu >> SU_GMAX_SHIFT) & 3u;
        // (1) gathers first: they depend on the previous row's writes and head the critical path
        d2 g0[4], g1[4];                           // loaded iff gmax >= 1 / 2 (never read otherwise)
        constexpr int NP = K == 0 ? 4 : 1;         // pairs that travel
        if (gmax >= 1u) {
          const d2* c0 = cs + (size_t)(chs & 1023u) * (4 * L);
#pragma unroll
          for (int i = 0; i < NP; ++i) g0[i] = c0[i * L];
        }
        if (gmax >= 2u) {
          const d2* c1 = cs + (size_t)((chs >> 10) & 1023u) * (4 * L);
#pragma unroll
          for (int i = 0; i < NP; ++i) g1[i] = c1[i * L];
        }
        {                                          // next row's operands (LDS) and the record two rows ahead (global)
          const unsigned kpn = Tq[(u + 1) % 3].ix.w;
          vkq[(u + 1) % 3] = sV[(size_t)(kpn & 0xffffu) * L]; vpq[(u + 1) % 3] = sV[(size_t)(kpn >> 16) * L];
          Tq[(u + 1) % 3].sb = load_sb(kpn);
          load_rec(min(r + 2, R - 1), Tq[(u + 2) % 3]);
        }
        SCHED_FENCE();
        STAMP2(200 + 10 * K);
        // (2) shadow: previous row's bookkeeping; this row's child-independent part
        //     A_kp = V_k conj(Y_kp V_p), A_pk = V_p conj(Y_pk V_k), A_kk = |V_k|^2 conj(Y_kk), A_ks = V_k conj(Y_k,slack V_slack)
        note_mismatch(pFp, pFq, pLive);
        const double gkk = T.ykk.x, bkk = T.ykk.y, gkp = T.ykp.x, bkp = T.ykp.y, gpk = T.ypk.x, bpk = T.ypk.y;
        const double ek = vk.x, fk = vk.y, ep = vp.x, fp = vp.y;
        const double tr = gkp * ep - bkp * fp, ti = gkp * fp + bkp * ep;
        const double akp_r = ek * tr + fk * ti, akp_i = fk * tr - ek * ti;
        const double ur = gpk * ek - bpk * fk, ui = gpk * fk + bpk * ek;
        const double apk_r = ep * ur + fp * ui, apk_i = fp * ur - ep * ui;
        const double v2 = ek * ek + fk * fk;
        const double akk_r = v2 * gkk, akk_i = -v2 * bkk;
        double aks_r = 0.0, aks_i = 0.0;           // a constant-voltage neighbour only feeds S_k
        if (flu & SU_SLACK_ANY) { aks_r = ek * T.cks.x + fk * T.cks.y; aks_i = fk * T.cks.x - ek * T.cks.y; }
        const double base_r = (akk_r + aks_r) + akp_r, base_i = (akk_i + aks_i) + akp_i;
        const double m = (fl & S_CARRY_IN) ? 1.0 : 0.0;    // register carry (same worker, previous row) masked by a 0/1 factor
        SCHED_FENCE();
        STAMP2(201 + 10 * K);
        // (3) the dependent chain: children (carry + LDS slots, canonical order) -> S, mismatch -> pivot -> factors -> contribution
        double aS0, aS1, aD0 = 0.0, aD1 = 0.0, aD2 = 0.0, aD3 = 0.0, aR0 = 0.0, aR1 = 0.0;
        if (gmax == 0u) {
          aS0 = m * cS0; aS1 = m * cS1;
          if constexpr (K == 0) { aD0 = m * cD0; aD1 = m * cD1; aD2 = m * cD2; aD3 = m * cD3; aR0 = m * cR0; aR1 = m * cR1; }
        } else {
          aS0 = fma(m, cS0, g0[0].x); aS1 = fma(m, cS1, g0[0].y);
          if constexpr (K == 0) {
            aD0 = fma(m, cD0, g0[1].x); aD1 = fma(m, cD1, g0[1].y); aD2 = fma(m, cD2, g0[2].x); aD3 = fma(m, cD3, g0[2].y);
            aR0 = fma(m, cR0, g0[3].x); aR1 = fma(m, cR1, g0[3].y);
          }
          if (gmax >= 2u) {
            aS0 += g1[0].x; aS1 += g1[0].y;
            if constexpr (K == 0) { aD0 += g1[1].x; aD1 += g1[1].y; aD2 += g1[2].x; aD3 += g1[2].y; aR0 += g1[3].x; aR1 += g1[3].y; }
          }
          if (gmax >= 3u) {                        // rare: junctions with more than two slot children
            const int nch = (int)((fl >> 16) & 255u);
            if (nch > 2) {
              auto gather = [&](unsigned slot) {
                const d2* c = cs + (size_t)slot * (4 * L);
                const d2 a = c[0];
                aS0 += a.x; aS1 += a.y;
                if constexpr (K == 0) {
                  const d2 b = c[L], cc = c[2 * L], dd = c[3 * L];
                  aD0 += b.x; aD1 += b.y; aD2 += cc.x; aD3 += cc.y; aR0 += dd.x; aR1 += dd.y;
                }
              };
              gather((chs >> 20) & 1023u);
              const unsigned cptr = clist_ptr(fl, slots, chs);
#pragma unroll 1                             // rare path: keep it small, the rows' code size is instruction-cache footprint
              for (int j = 3; j < nch; ++j) gather((unsigned)s_clist[cptr + j - 3]);
            }
          }
        }
        // S_k = V_k conj(sum_j Y_kj V_j), mismatch F_k = S_k - Sbus_k
        const double sr = base_r + aS0, si = base_i + aS1;
        const double Fp = sr - T.sb.x, Fq = si - T.sb.y;
        pFp = Fp; pFq = Fq; pLive = (fl & S_LIVE) != 0;
        if constexpr (K == 0) {
          const double D0 = -(si - akk_i) - aD0, D1 = (sr + akk_r) - aD1;
          const double D2 = (sr - akk_r) - aD2, D3 = (si + akk_i) - aD3;
          const double r0 = Fp - aR0, r1 = Fq - aR1;
          const double idet = rcp_nr(D0 * D3 - D1 * D2);
          const double I0 = D3 * idet, I1 = -D1 * idet, I2 = -D2 * idet, I3 = D0 * idet;
          const double h0 = I0 * r0 + I1 * r1, h1 = I2 * r0 + I3 * r1;
          // U = J'(k,p) = [[Im A_kp, Re A_kp], [-Re A_kp, Im A_kp]],  L = J'(p,k) likewise from A_pk
          const double G0 = I0 * akp_i - I1 * akp_r, G1 = I0 * akp_r + I1 * akp_i;
          const double G2 = I2 * akp_i - I3 * akp_r, G3 = I2 * akp_r + I3 * akp_i;
          const double s0 = apk_i * G0 + apk_r * G2, s1 = apk_i * G1 + apk_r * G3;
          const double s2 = apk_i * G2 - apk_r * G0, s3 = apk_i * G3 - apk_r * G1;
          const double t0 = apk_i * h0 + apk_r * h1, t1 = apk_i * h1 - apk_r * h0;
          // contribution to the parent: registers (consumed only if the next step has S_CARRY_IN) and the
          // LDS slot (TRASH unless S_SCRATCH_OUT; skipped when no worker of the wave has a real slot)
          cS0 = apk_r; cS1 = apk_i; cD0 = s0; cD1 = s1; cD2 = s2; cD3 = s3; cR0 = t0; cR1 = t1;
          if (flu & SU_W_ANY) {
            d2* c = cs + (size_t)(slots & 1023u) * (4 * L);
            c[0] = d2{apk_r, apk_i}; c[L] = d2{s0, s1}; c[2 * L] = d2{s2, s3}; c[3 * L] = d2{t0, t1};
          }
          SCHED_FENCE();                           // the factors leave after the contribution is on its way
          const unsigned k = kp & 0xffffu, voN = voE + k * bb;
          if constexpr (HL) sH[(size_t)k * L] = d2{h0, h1};
          else if constexpr (RS >= 0) { a_put(h0, Ha[RS][0], Ha[RS][1]); a_put(h1, Ha[RS][2], Ha[RS][3]); }
          else bst2(d2{h0, h1}, rs, voN, sF_H);
          if (GL) { sG[(size_t)(2 * k) * L] = d2{G0, G1}; sG[(size_t)(2 * k + 1) * L] = d2{G2, G3}; }
          else if constexpr (RS >= 0) {
            a_put(G0, Ga[RS][0], Ga[RS][1]); a_put(G1, Ga[RS][2], Ga[RS][3]); a_put(G2, Ga[RS][4], Ga[RS][5]); a_put(G3, Ga[RS][6], Ga[RS][7]);
          }
          else { bst2(d2{G0, G1}, rs, voN, sF_G01); bst2(d2{G2, G3}, rs, voN, sF_G23); }
        } else {
          cS0 = apk_r; cS1 = apk_i;
          if (flu & SU_W_ANY) cs[(size_t)(slots & 1023u) * (4 * L)] = d2{apk_r, apk_i};
        }
        STAMP2(202 + 10 * K);
        if (W > 1) lds_barrier();
        STAMP(100 + K);
      }
    };
    constexpr int KP = (K == 0) ? KR : 0;          // peeled rows (their G stays in registers)
    static_for<KP>([&](auto ic) {                  // (the host pads every schedule to R >= NR_G_REG_ROWS)
      constexpr int i = decltype(ic)::value;
      fwd_row(std::integral_constant<int, i % 3>{}, ic, i);
    });
    static_assert(KP % 3 == 0, "the peeled rows must leave the record ring at position 0");
    int r = KP;
    while (r < R) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (r >= R) break;
        if (u == 0) fwd_row(std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{}, r);
        else if (u == 1) fwd_row(std::integral_constant<int, 1>{}, std::integral_constant<int, -1>{}, r);
        else fwd_row(std::integral_constant<int, 2>{}, std::integral_constant<int, -1>{}, r);
        ++r;
      }
    }
    note_mismatch(pFp, pFq, pLive);
  };
  // First iteration: every V is the flat start, so D^-1 (I), G and the link terms are the host-made
  // constants of Schedule::flat; per env there is only the forward substitution of the right-hand side
  // r = F - sum of the children's t (pair 3 of the contribution slots), h = I r, t = L h.
  auto fwd_sweep_flat = [&]() {
    RecF Tq[3];
    load_recf(0, Tq[0]); load_recf(min(1, R - 1), Tq[1]);
    Tq[0].sb = load_sb(Tq[0].ix.w);
    int r = 0;
    while (r < R) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (r >= R) break;
        const RecF& T = Tq[u % 3];
        const uint32_t fl = T.ix.x, slots = T.ix.y, chs = T.ix.z, kp = T.ix.w;
        const uint32_t flu = uni(fl);
        const unsigned gmax = (flu >> SU_GMAX_SHIFT) & 3u;
        d2 g0, g1;
        if (gmax >= 1u) g0 = cs[((size_t)(chs & 1023u) * 4 + 3) * L];
        if (gmax >= 2u) g1 = cs[((size_t)((chs >> 10) & 1023u) * 4 + 3) * L];
        Tq[(u + 1) % 3].sb = load_sb(Tq[(u + 1) % 3].ix.w);
        load_recf(min(r + 2, R - 1), Tq[(u + 2) % 3]);
        SCHED_FENCE();
        STAMP2(230);
        const double s_r = T.s.x, s_i = T.s.y, i0 = T.i01.x, i1 = T.i01.y, i2 = T.i23.x, i3 = T.i23.y, ap_r = T.ap.x, ap_i = T.ap.y;
        const double Fp = s_r - T.sb.x, Fq = s_i - T.sb.y;           // shadow: the flat-start mismatch does not depend on children
        note_mismatch(Fp, Fq, (fl & S_LIVE) != 0);
        const double m = (fl & S_CARRY_IN) ? 1.0 : 0.0;
        SCHED_FENCE();
        double aR0, aR1;
        if (gmax == 0u) { aR0 = m * cR0; aR1 = m * cR1; }
        else {
          aR0 = fma(m, cR0, g0.x); aR1 = fma(m, cR1, g0.y);
          if (gmax >= 2u) { aR0 += g1.x; aR1 += g1.y; }
          if (gmax >= 3u) {
            const int nch = (int)((fl >> 16) & 255u);
            if (nch > 2) {
              auto gather = [&](unsigned slot) { const d2 a = cs[((size_t)slot * 4 + 3) * L]; aR0 += a.x; aR1 += a.y; };
              gather((chs >> 20) & 1023u);
              const unsigned cptr = clist_ptr(fl, slots, chs);
#pragma unroll 1                             // rare path: keep it small, the rows' code size is instruction-cache footprint
              for (int j = 3; j < nch; ++j) gather((unsigned)s_clist[cptr + j - 3]);
            }
          }
        }
        const double r0 = Fp - aR0, r1 = Fq - aR1;
        const double h0 = i0 * r0 + i1 * r1, h1 = i2 * r0 + i3 * r1;
        const double t0 = ap_i * h0 + ap_r * h1, t1 = ap_i * h1 - ap_r * h0;
        cR0 = t0; cR1 = t1;
        if (flu & SU_W_ANY) cs[((size_t)(slots & 1023u) * 4 + 3) * L] = d2{t0, t1};
        SCHED_FENCE();
        const unsigned k = kp & 0xffffu;
        if (HL) sH[(size_t)k * L] = d2{h0, h1}; else bst2(d2{h0, h1}, rs, voE + k * bb, sF_H);
        STAMP2(232);
        if (W > 1) lds_barrier();
        STAMP(102);
        ++r;
      }
    }
  };
  double dxm = 0.0;                                // largest Newton step component of this worker's nodes, per env
  // newtonpf update of one node: Va += dx_a, Vm += dx_m, V = Vm e^{jVa}, with dx_a = -y0, dx_m = -|V| y1
  //   =>  V <- V (1 - y1) e^{-j y0}   (rotation by the small step; |V|, angle are formed at the end)
  auto apply_update = [&](double y0, double y1, d2 vk, unsigned k, bool live) {
    dxm = fmax(dxm, live ? fmax(fabs(y0), fabs(y1)) : 0.0);
    double s, c;
    const double dth = -y0;
    sincos_small(dth, &s, &c);
    if (__any(live && !(fabs(dth) <= 0.5))) sincos_mid(dth, &s, &c);   // wave-uniform, only when a lane takes a large step (nr_common.hpp)
    sV[(size_t)k * L] = done ? vk : nr_rotate(vk, s, c, y1);   // converged envs keep their state; idle steps hit the trash node
  };
  // ... of three nodes at once: ONE wave-uniform check for the rare large-angle case instead of one per node, so that the three
  // polynomial chains are a single basic block the scheduler can interleave (same expressions per node: same bits)
  constexpr int UPN = NR_UPDATE_GROUP;            // nodes per turn of the update pass
  auto apply_update3 = [&](const d2 (&xx)[UPN], const d2 (&vv)[UPN], const unsigned (&kk)[UPN], const bool (&lv)[UPN]) {
    double s[UPN], c[UPN];
    bool big = false;
#pragma unroll
    for (int i = 0; i < UPN; ++i) {
      dxm = fmax(dxm, lv[i] ? fmax(fabs(xx[i].x), fabs(xx[i].y)) : 0.0);
      sincos_small(-xx[i].x, &s[i], &c[i]);
      big = big | (lv[i] & !(fabs(xx[i].x) <= 0.5));
    }
    if (__any(big)) {
#pragma unroll
      for (int i = 0; i < UPN; ++i) if (__any(lv[i] && !(fabs(xx[i].x) <= 0.5))) sincos_mid(-xx[i].x, &s[i], &c[i]);   // as apply_update decides, node by node
    }
#pragma unroll
    for (int i = 0; i < UPN; ++i) sV[(size_t)kk[i] * L] = done ? vv[i] : nr_rotate(vv[i], s[i], c[i], xx[i].y);
  };
  // ---- backward sweep, h in LDS (HL): x-propagation + parallel update.
  // Only x_k = h_k - G_k x_parent is a chain down the tree; the voltage update of a node needs nothing but its own x.  So the
  // rows of the backward sweep carry the x recurrence alone — x_k replaces h_k IN PLACE in the LDS h array, a child reads
  // its parent's x from there (no x slots; the slack entry of the array holds the 0 that elimination roots read) — and
  // the update V <- V (1 - x1) e^{-j x0} of all nodes follows as a barrier-free pass in which every worker takes every
  // Wt-th node: n / Wt steps at full occupancy of the lanes instead of R rows at the schedule's ~50 %.
  // SRC 0: first iteration, G from the flat-start table; 1: G from LDS (GL) or from the factor blocks in global scratch,
  // prefetched two rows ahead through a static register ring (their address needs the row's node, read three rows ahead).
  auto bwd_xprop = [&](auto src) {
    constexpr int SRC = decltype(src)::value;
    constexpr bool gG = (SRC == 1) && !GL;         // G comes from the factor blocks in global memory
    u32x4 ixq[4]; d2 g01q[4], g23q[4];
    auto load_g = [&](int row, const u32x4& ixr, int slot) {
      if constexpr (SRC == 0) {
        if (flatL) { const char* p = flatT + (unsigned)row * FB; g01q[slot] = *(const d2*)(p + FL_G0 * 8); g23q[slot] = *(const d2*)(p + FL_G2 * 8); }
        else { const unsigned sf = row_s(row, FB); g01q[slot] = bld2(rsF, voF + FL_G0 * 8u, sf); g23q[slot] = bld2(rsF, voF + FL_G2 * 8u, sf); }
      } else if constexpr (gG) {
        const unsigned voN = voE + (ixr.w & 0xffffu) * bb;
        g01q[slot] = bld2(rs, voN, sF_G01); g23q[slot] = bld2(rs, voN, sF_G23);
      }
    };
    // rows below KP are peeled (straight-line code, G in registers: see Gs); their index words are fetched up front
    constexpr int KP = gG ? KR : 0;
    u32x4 ixs[KP > 0 ? KP : 1];
    static_for<KP>([&](auto ic) { constexpr int i = decltype(ic)::value; ixs[i] = load_ix(i); });
    ixq[0] = load_ix(R - 1); ixq[1] = load_ix(max(R - 2, 0)); ixq[2] = load_ix(max(R - 3, 0));
    load_g(R - 1, ixq[0], 0); load_g(max(R - 2, 0), ixq[1], 1);
    int r = R - 1;
    while (r >= KP) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r < KP) break;
        const uint32_t fl = ixq[u % 4].x, kp = ixq[u % 4].w;
        const unsigned k = kp & 0xffffu, p = kp >> 16;
        // (1) the parent's x (its h slot, already overwritten; the slack entry holds the 0 that elimination roots read), this node's
        // h and G.  The parent's entry is read unconditionally, together with h (round 4): behind the SU_XR_ANY hint the read sat in a
        // scalar branch of its own and its wait came before the other requests of the row were issued — one more exposed LDS round trip
        // in a row that consists of three.
        const d2 q = sH[(size_t)p * L];
        const d2 hh = sH[(size_t)k * L];
        d2 g01, g23;
        if constexpr (SRC == 1 && GL) { g01 = sG[(size_t)(2 * k) * L]; g23 = sG[(size_t)(2 * k + 1) * L]; }
        ixq[(u + 3) % 4] = load_ix(max(r - 3, KP));
        load_g(max(r - 2, KP), ixq[(u + 2) % 4], (u + 2) % 4);
        SCHED_FENCE();
        STAMP2(240);
        if constexpr (SRC == 0 || gG) { g01 = g01q[u % 4]; g23 = g23q[u % 4]; }
        // (2) x_k = h_k - G_k x_parent
        const bool cout = (fl & S_CARRY_OUT) != 0;
        const double p0 = cout ? x0 : q.x, p1 = cout ? x1 : q.y;
        const double y0 = hh.x - (g01.x * p0 + g01.y * p1);
        const double y1 = hh.y - (g23.x * p0 + g23.y * p1);
        x0 = y0; x1 = y1;
        sH[(size_t)k * L] = d2{y0, y1};            // (idle steps: the trash node)
        STAMP2(242);
        if (W > 1) lds_barrier();
        STAMP(110 + SRC);
        --r;
      }
    }
    static_for_down<KP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      {
        const uint32_t fl = ixs[i].x, kp = ixs[i].w;
        const unsigned k = kp & 0xffffu, p = kp >> 16;
        const d2 q = sH[(size_t)p * L];
        const d2 hh = sH[(size_t)k * L];
        SCHED_FENCE();
        const bool cout = (fl & S_CARRY_OUT) != 0;
        const double p0 = cout ? x0 : q.x, p1 = cout ? x1 : q.y;
        const double G0 = a_get(Ga[i][0], Ga[i][1]), G1 = a_get(Ga[i][2], Ga[i][3]), G2 = a_get(Ga[i][4], Ga[i][5]), G3 = a_get(Ga[i][6], Ga[i][7]);
        const double y0 = hh.x - (G0 * p0 + G1 * p1);
        const double y1 = hh.y - (G2 * p0 + G3 * p1);
        x0 = y0; x1 = y1;
        sH[(size_t)k * L] = d2{y0, y1};
        if (W > 1) lds_barrier();
        STAMP(110 + SRC);
      }
    });
    // (3) update of every node from its x, three nodes per pass (loads first)
    for (unsigned kb = t; kb < n; kb += (unsigned)UPN * Wt) {
      unsigned kk[UPN]; d2 xx[UPN], vv[UPN]; bool lv[UPN];
#pragma unroll
      for (int i = 0; i < UPN; ++i) {
        const unsigned kx = kb + (unsigned)i * Wt;
        lv[i] = kx < n; kk[i] = lv[i] ? kx : n + 1u;
        xx[i] = sH[(size_t)kk[i] * L]; vv[i] = sV[(size_t)kk[i] * L];
      }
      apply_update3(xx, vv, kk, lv);
      STAMP(30);
    }
  };
  // ---- mismatch-only evaluation as a PASS (HL): what the K = 1 forward sweep computes — S_k = V_k conj(sum_j Y_kj V_j), F_k = S_k -
  // Sbus_k and the verdict — has no elimination in it, so it needs no tree order: (1) every node's term in its PARENT's sum,
  // A_pk = V_p conj(Y_pk V_k), goes to the node's entry of the h array (dead between a backward and a forward sweep); barrier;
  // (2) every node adds its own terms and its children's entries in the sweeps' canonical order (chain child first, then
  // ascending: the same expressions in the same order as fwd_sweep, so the verdict is bit-identical to the sweep's).  Every
  // worker takes every Wt-th node: ~2 x n / Wt barrier-free steps instead of R rows.
  auto mismatch_pass = [&]() {
    // records of this worker, one per turn, streamed from global memory (L2) one turn ahead: no dependent address chains
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<StepRec*>(d.mm_recs), 0, d.mm_recs_bytes, 0x00020000);
    const int NPs = d.mm_np;
    const unsigned voP = t * (unsigned)NPs * TB;
    // (1) A_pk = V_p conj(Y_pk V_k) of every node -> its entry of the h array
    {
      u32x4 ixn = bldu4(rsP, voP, 0u); d2 ypn = bld2(rsP, voP + 48u, 0u);
      for (int j = 0; j < NPs; ++j) {
        const u32x4 ix = ixn; const d2 ypk = ypn;
        const unsigned sn = row_s(min(j + 1, NPs - 1), TB);
        ixn = bldu4(rsP, voP, sn); ypn = bld2(rsP, voP + 48u, sn);
        const unsigned k = ix.w & 0xffffu, pp = ix.w >> 16;
        const d2 vk = sV[(size_t)k * L], vp = sV[(size_t)pp * L];
        const double ek = vk.x, fk = vk.y, ep = vp.x, fp = vp.y, gpk = ypk.x, bpk = ypk.y;
        const double ur = gpk * ek - bpk * fk, ui = gpk * fk + bpk * ek;
        const double apk_r = ep * ur + fp * ui, apk_i = fp * ur - ep * ui;
        sH[(size_t)k * L] = d2{apk_r, apk_i};
      }
    }
    STAMP(31);
    if (W > 1) lds_barrier();
    // (2) S_k, F_k, verdict.  The index words run two turns ahead, the constants and the Sbus entry (whose address needs the index
    // word) one turn ahead.
    {
      const unsigned voSb = d.sb_off + e * 16u;
      auto cl = [&](int j) { return row_s(min(j, NPs - 1), TB); };
      u32x4 ixA = bldu4(rsP, voP, 0u), ixB = bldu4(rsP, voP, cl(1));
      d2 ykkN = bld2(rsP, voP + 16u, 0u), ykpN = bld2(rsP, voP + 32u, 0u), cksN = bld2(rsP, voP + 64u, 0u);
      d2 sbN = bld2(rs, voSb + (ixA.w & 0xffffu) * pb, 0u);
      for (int j = 0; j < NPs; ++j) {
        const u32x4 ix = ixA;
        const d2 ykk = ykkN, ykp = ykpN, cks = cksN, sb = sbN;
        const unsigned k = ix.w & 0xffffu, pp = ix.w >> 16;
        const int nch = (int)((ix.x >> 8) & 255u);
        const d2 vk = sV[(size_t)k * L], vp = sV[(size_t)pp * L];
        const d2 a0 = sH[(size_t)(ix.y & 0xffffu) * L], a1 = sH[(size_t)(ix.y >> 16) * L], a2 = sH[(size_t)(ix.z & 0xffffu) * L];
        {
          const unsigned s1 = cl(j + 1);
          ykkN = bld2(rsP, voP + 16u, s1); ykpN = bld2(rsP, voP + 32u, s1); cksN = bld2(rsP, voP + 64u, s1);
          sbN = bld2(rs, voSb + (ixB.w & 0xffffu) * pb, 0u);
          ixA = ixB; ixB = bldu4(rsP, voP, cl(j + 2));
        }
        const double gkk = ykk.x, bkk = ykk.y, gkp = ykp.x, bkp = ykp.y;
        const double ek = vk.x, fk = vk.y, ep = vp.x, fp = vp.y;
        const double tr = gkp * ep - bkp * fp, ti = gkp * fp + bkp * ep;
        const double akp_r = ek * tr + fk * ti, akp_i = fk * tr - ek * ti;
        const double v2 = ek * ek + fk * fk;
        const double akk_r = v2 * gkk, akk_i = -v2 * bkk;
        const double aks_r = ek * cks.x + fk * cks.y, aks_i = fk * cks.x - ek * cks.y;
        const double base_r = (akk_r + aks_r) + akp_r, base_i = (akk_i + aks_i) + akp_i;
        double aS0 = nch > 0 ? a0.x : 0.0, aS1 = nch > 0 ? a0.y : 0.0;   // the children in canonical order, the first one opening the sum
        if (nch > 1) { aS0 += a1.x; aS1 += a1.y; }
        if (nch > 2) { aS0 += a2.x; aS1 += a2.y; }
        if (__any(nch > 3)) {                                            // rare: junctions with more than three children
          const int c_lo = d.mm_ptr[k];
#pragma unroll 1
          for (int q = 3; q < nch; ++q) { const d2 a = sH[(size_t)d.mm_child[c_lo + q] * L]; aS0 += a.x; aS1 += a.y; }
        }
        const double sr = base_r + aS0, si = base_i + aS1;
        note_mismatch(sr - sb.x, si - sb.y, (ix.x & 1u) != 0);
        STAMP(120);
      }
    }
  };
  // backward sweep when h lives in global scratch (the lean layouts).  The update of a node is DEFERRED into the shadow of
  // the next row (only the x chain is between the row barriers).  SRC as above.
  auto bwd_sweep = [&](auto src) {
    constexpr int SRC = decltype(src)::value;
    constexpr bool gG = (SRC == 0) || !GL;         // G comes from global memory (flat table or factor block)
    constexpr bool gH = !HL;
    // rings of 4 (static indices): the record of a row is read THREE rows ahead, so that its node number is there when the
    // factor loads of that row are issued TWO rows ahead (factor blocks are addressed by node)
    u32x4 ixq[4]; BwdF fq[4];
    auto load_f = [&](int row, const u32x4& ix, BwdF& f) {
      const unsigned voN = voE + (ix.w & 0xffffu) * bb;
      if (gH) f.h = bld2(rs, voN, sF_H);
      if (gG) {
        if (SRC == 0) {
          if (flatL) { const char* p = flatT + (unsigned)row * FB; f.g01 = *(const d2*)(p + FL_G0 * 8); f.g23 = *(const d2*)(p + FL_G2 * 8); }
          else { const unsigned sf = row_s(row, FB); f.g01 = bld2(rsF, voF + FL_G0 * 8u, sf); f.g23 = bld2(rsF, voF + FL_G2 * 8u, sf); }
        }
        else { f.g01 = bld2(rs, voN, sF_G01); f.g23 = bld2(rs, voN, sF_G23); }
      }
    };
    // rows below KP are peeled (straight-line code, h and G in registers: see Ga / Ha); their index words are fetched up front
    constexpr int KP = (SRC == 1 && !GL) ? KR : 0;
    u32x4 ixs[KP > 0 ? KP : 1];
    static_for<KP>([&](auto ic) { constexpr int i = decltype(ic)::value; ixs[i] = load_ix(i); });
    ixq[0] = load_ix(R - 1); ixq[1] = load_ix(max(R - 2, 0)); ixq[2] = load_ix(max(R - 3, 0));
    load_f(R - 1, ixq[0], fq[0]); load_f(max(R - 2, 0), ixq[1], fq[1]);
    double py0 = 0.0, py1 = 0.0; d2 pvk = sV[(size_t)(n + 1) * L]; unsigned pk = n + 1; bool pLive = false;   // deferred update of the previous row
    int r = R - 1;
    while (r >= KP) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r < KP) break;
        const u32x4 ix = ixq[u % 4];
        const uint32_t fl = ix.x, slots = ix.y;
        const uint32_t flu = uni(fl);
        const unsigned k = ix.w & 0xffffu;
        // (1) the parent's x (ZERO slot for slack parents, carried parents and idle steps: read unconditionally, so that the request
        // leaves with the row's other LDS reads instead of waiting in a hint branch of its own), this node's factors and voltage
        const d2 q = xs[(size_t)((slots >> 20) & 1023u) * L];
        const d2 hh = gH ? fq[u % 4].h : sH[(size_t)k * L];
        const d2 g01 = gG ? fq[u % 4].g01 : sG[(size_t)(2 * k) * L];
        const d2 g23 = gG ? fq[u % 4].g23 : sG[(size_t)(2 * k + 1) * L];
        const d2 vk = sV[(size_t)k * L];           // (only this node's own deferred update ever writes it)
        ixq[(u + 3) % 4] = load_ix(max(r - 3, KP));
        load_f(max(r - 2, KP), ixq[(u + 2) % 4], fq[(u + 2) % 4]);
        SCHED_FENCE();
        // (2) shadow: the previous row's voltage update
        apply_update(py0, py1, pvk, pk, pLive);
        SCHED_FENCE();
        // (3) x_k = h_k - G_k x_parent
        const bool cout = (fl & S_CARRY_OUT) != 0;
        const double p0 = cout ? x0 : q.x, p1 = cout ? x1 : q.y;
        const double y0 = hh.x - (g01.x * p0 + g01.y * p1);
        const double y1 = hh.y - (g23.x * p0 + g23.y * p1);
        x0 = y0; x1 = y1;
        if (flu & SU_XW_ANY) xs[(size_t)((slots >> 10) & 1023u) * L] = d2{y0, y1};   // TRASH unless S_X_OUT
        py0 = y0; py1 = y1; pvk = vk; pk = k; pLive = (fl & S_LIVE) != 0;
        if (W > 1) lds_barrier();
        STAMP(110 + SRC);
        --r;
      }
    }
    static_for_down<KP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const u32x4 ix = ixs[i];
      const uint32_t fl = ix.x, slots = ix.y;
      const uint32_t flu = uni(fl);
      const unsigned k = ix.w & 0xffffu;
      const d2 q = xs[(size_t)((slots >> 20) & 1023u) * L];
      const d2 vk = sV[(size_t)k * L];
      SCHED_FENCE();
      apply_update(py0, py1, pvk, pk, pLive);
      SCHED_FENCE();
      const double hh0 = a_get(Ha[i][0], Ha[i][1]), hh1 = a_get(Ha[i][2], Ha[i][3]);
      const double G0 = a_get(Ga[i][0], Ga[i][1]), G1 = a_get(Ga[i][2], Ga[i][3]), G2 = a_get(Ga[i][4], Ga[i][5]), G3 = a_get(Ga[i][6], Ga[i][7]);
      const bool cout = (fl & S_CARRY_OUT) != 0;
      const double p0 = cout ? x0 : q.x, p1 = cout ? x1 : q.y;
      const double y0 = hh0 - (G0 * p0 + G1 * p1);
      const double y1 = hh1 - (G2 * p0 + G3 * p1);
      x0 = y0; x1 = y1;
      if (flu & SU_XW_ANY) xs[(size_t)((slots >> 10) & 1023u) * L] = d2{y0, y1};
      py0 = y0; py1 = y1; pvk = vk; pk = k; pLive = (fl & S_LIVE) != 0;
      if (W > 1) lds_barrier();
      STAMP(110 + SRC);
    });
    apply_update(py0, py1, pvk, pk, pLive);
  };
#undef SCHED_FENCE

  // Sweep forms.  `first`: the flat-start iteration (host-factorised constants).  `light`: the previous Newton
  // step of every unfinished env of this workgroup was tiny, so convergence is expected and the sweep is
  // first run mismatch-only; if some env then fails the test after all, the sweep is redone in full (its
  // factors are needed for another iteration).  All three flags are uniform over the workgroup: every
  // wave holds the same envs and derives them from the same per-env values.
  bool first = true, light = false;
  while (!nothing_to_solve) {
    // ------------------------------------------------------------------ forward sweep
    allok = true; fmx = 0.0;
    cS0 = cS1 = cD0 = cD1 = cD2 = cD3 = cR0 = cR1 = 0.0;
    STAMP(10);
    if (first) { fwd_sweep_flat(); a_define(); }
    else if (light) { if (HL && d.nr_mm_pass) mismatch_pass(); else fwd_sweep(std::integral_constant<int, 1>{}); a_define(); }
    else fwd_sweep(std::integral_constant<int, 0>{});
    {                                            // AND of the workers' verdicts, per env: in the wave by lane exchanges, across
      fmx = grp_max<L>(fmx);                     // the W waves through W LDS entries (instead of Wt)
      allok = grp_max<L>(allok ? 0.0 : 1.0) == 0.0;
      if (W > 1) {
        s_ok[w * L + el] = allok ? 1 : 0;
        s_dx[(size_t)w * L] = fmx;               // (free here: the step sizes it holds were consumed before this sweep)
        lds_barrier();
#pragma unroll
        // (& not &&: the short-circuit form put every s_ok read behind an exec-mask branch with its own lgkmcnt(0) — W serialised LDS round trips)
        for (unsigned ww = 0; ww < (unsigned)W; ++ww) { allok = allok & (s_ok[ww * L + el] != 0); fmx = fmax(fmx, s_dx[(size_t)ww * L]); }
      }
    }
    if (light) {
      light = false;
      if (__any(!done && !allok && it < d.max_it)) {   // mispredicted: this env iterates on and needs the factors
        if (W > 1) lds_barrier();                      // s_ok is rewritten by the redone sweep's verdict
        continue;
      }
    }
    STAMP(11);
    Fprev = Fcur; Fcur = fmx;                    // this sweep stands (a redone mismatch-only sweep never gets here)
    if (!done) {
      conv = allok;
      if (conv || it == d.max_it) done = true;
    }
    if (__all(done)) break;                      // same envs, same values in every wave of the group
    // ------------------------------------------------------------------ backward sweep + update
    x0 = x1 = 0.0;
    dxm = 0.0;
    STAMP(12);
    if constexpr (HL) { if (first) bwd_xprop(std::integral_constant<int, 0>{}); else bwd_xprop(std::integral_constant<int, 1>{}); }
    else { if (first) bwd_sweep(std::integral_constant<int, 0>{}); else bwd_sweep(std::integral_constant<int, 1>{}); }
    first = false;
    if (!done) ++it;
    {                                            // size of the step just taken, per env: max over the workers
      double dxe = grp_max<L>(dxm);
      if (W > 1) {
        s_dx[(size_t)w * L] = dxe;
        lds_barrier();
#pragma unroll
        for (unsigned ww = 0; ww < (unsigned)W; ++ww) dxe = fmax(dxe, s_dx[(size_t)ww * L]);
      }
      // convergence is predicted from a tiny step, or — scale-free — from quadratic convergence of the mismatch:
      // ||F_next|| ~ ||F||^3 / ||F_prev||^2 (two sweeps of history needed)
      const bool quad = it >= 2 && Fcur * Fcur * Fcur * d.nr_check_quad < tol * Fprev * Fprev;
      light = __all(done || dxe < d.nr_check_dx || quad);
    }
  }
  STAMP(20);
#ifdef MAPDN_NR_STAMPS
  if (stamp_on) g_stamps[0] = ns;
#endif
  if (t == 0) { d.iters[e] = it; d.conv[e] = conv ? 1 : 0; }
  nr_epilogue<(unsigned)L, Wt>(d, mode, reward, terminated, info, sV, t, e, act, conv, bk_steps, bk_draw, bk_sum, s_epi, s_lines,
                               [&](int id) { STAMP(id);
#ifdef MAPDN_NR_STAMPS
                                             if (stamp_on) g_stamps[0] = ns;
#endif
                               });
}


}  // namespace mapdn
