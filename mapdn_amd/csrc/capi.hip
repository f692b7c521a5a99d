// capi.hip — C ABI of libmapdn_hip.so (declared in include/mapdn.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "colstats.hpp"
#include "plan.hpp"

using namespace mapdn;

struct mapdn_handle {
  Plan plan;
  Schedule sched;
  SparseProg sprog;
  int solver = 0;                 // 0 tree (radial), 1 general sparse (k_nr_sparse), 2 general dense (k_nr_dense)
  int sp_lanes = 0;
  mapdn_env_config cfg;
  Dev d;
  int device = 0;
  std::string err;
  std::vector<void*> allocs;
  bool have_profiles = false, was_reset = false, host_only = false;
  bool sbus_stale = true;         // Sbus / bus_ld do not reflect cur_pl / cur_ql (fresh handle, after mapdn_solve_only): the next
                                  // injection runs the all-bus kernel; MAPDN_INJECT_FULL=1 keeps it that way (A/B, tests)
  bool inject_full = false;
  bool fuse_inject = false;       // step(): the PV-bus injection runs in the prologue of k_nr_tree instead of as k_inject_sgen (tree solver, no auto_reset)
  bool overlap = false;           // step(): the profile rows of k_advance run on a side stream beside the solver launch (experiment)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  uint32_t sb_base = 0, sb_bytes = 0;   // the two Sbus buffers of nrbuf: d.sb_off / d.sb_off_alt alternate between them
  std::vector<int32_t> ld_dest_host;
  size_t lds_bytes = 0;
  // what mapdn_create settled on (mapdn_get_nr_geometry); tree solver only: W .. mm_pass
  struct Geo { int W = 0, L = 0, lean = 0, rows = 0, h_lds = 0, g_lds = 0, rec_lds = 0, flat_lds = 0, line_lds = 0, mm_pass = 0, ncl = 0;
               size_t lds = 0; long wgs = 0; int resident = 0, rounds = 0; double model_ns = 0.0; } geo;
  int32_t *obs_rows = nullptr, *state_rows = nullptr, *iota_idx = nullptr, *vm_row = nullptr, *va_row = nullptr;
  int32_t *obs_xptr = nullptr, *obs_xrow = nullptr;
  double *obs_scale = nullptr, *state_scale = nullptr;
  double *t_pl = nullptr, *t_ql = nullptr, *t_pv = nullptr, *t_q = nullptr;
  double* table = nullptr; double* stdv = nullptr; double* smax = nullptr;
  std::vector<double> stdv_host, smax_host;     // what set_profiles computed (mapdn_get_profile_stats)
  long long* stats_dev = nullptr;
  // NR kernel timing
  bool timing = false;
  std::vector<hipEvent_t> ev;   // pairs
  size_t ev_used = 0;
  double acc_ms = 0.0;
  int64_t acc_launches = 0;
};

static std::string g_create_err;

// Nothing may leave an extern "C" entry point by exception: every one of them is a function-try-block ending in
// MAPDN_CATCH, which turns std::bad_alloc into MAPDN_E_NOMEM and anything else into MAPDN_E_INTERNAL, with the text in mapdn_last_error.
static int api_fail(const mapdn_handle* h, int code, const char* what) noexcept {
  try { (h ? const_cast<mapdn_handle*>(h)->err : g_create_err) = what; } catch (...) {}      // (the message itself may not fit any more)
  return code;
}
#define MAPDN_CATCH(h)                                                                                       \
  catch (const std::bad_alloc&) { return api_fail((h), MAPDN_E_NOMEM, "out of host memory (std::bad_alloc)"); } \
  catch (const std::exception& e) { return api_fail((h), MAPDN_E_INTERNAL, e.what()); }                      \
  catch (...) { return api_fail((h), MAPDN_E_INTERNAL, "unknown C++ exception"); }

#define NEEDDEV(h) do { if ((h)->host_only) { (h)->err = "host-only handle (device == -1): no device entry points"; return MAPDN_E_STATE; } } while (0)
#define HIPCHK(h, call)                                                                         \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_e);                             \
      return MAPDN_E_HIP;                                                                       \
    }                                                                                           \
  } while (0)

template <typename T>
static int dalloc(mapdn_handle* h, T** p, size_t count) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIPCHK(h, hipMalloc(&q, bytes));
  h->allocs.push_back(q);
  HIPCHK(h, hipMemset(q, 0, bytes));
  *p = (T*)q;
  return MAPDN_OK;
}

template <typename T>
static int dupload(mapdn_handle* h, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  int rc = dalloc(h, &q, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(h, hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return MAPDN_OK;
}

// ---- tuning knobs: a field of mapdn_env_config (0 = automatic), overridden by an environment variable when one is set (tools,
// A/B runs).  Read once, in mapdn_create; nothing is process-global afterwards.
static int knob_int(int cfg_value, const char* env) { const char* s = getenv(env); return s ? atoi(s) : cfg_value; }
// tri-state residency / mode switches: cfg 0 auto | 1 on | 2 off;  env NAME=1 -> on, NAME=0 -> off
static int knob_tri(int cfg_value, const char* env) { const char* s = getenv(env); return s ? (atoi(s) ? 1 : 2) : cfg_value; }
static double knob_f64(double cfg_value, const char* env, double dflt) {
  const char* s = getenv(env);
  if (s) return atof(s);
  return cfg_value != 0.0 ? cfg_value : dflt;
}

// ---- k_nr_tree launch geometry.
// A workgroup = W waves serving L envs; each wave carries 64/L lane-group workers, so Wt = W*64/L workers eliminate independent
// subtrees of every env concurrently (schedule rows R(Wt) >= the radius of the feeder).  Small batches are latency-bound: spread
// each env over many workers and keep everything the solve touches in LDS ("fat", one workgroup per CU); batches that would need
// several ROUNDS of such workgroups take fewer workers / less LDS per env instead so that more envs are resident ("lean", or 16
// instead of 8 envs per workgroup on the 322-bus class).  The choice is the minimum of a launch-time model over the candidate
// (W, L, lean) triples, each with its own schedule and LDS residency settled first:
//     t = rounds x [ c0 + b n/Wt + row(W) R (1 + p [h not in LDS]) ] x load x share
//   rounds = ceil(workgroups / (CUs x resident));  resident = workgroups per CU by LDS (160 KB) and registers (one wave per SIMD:
//            every instantiation uses more than half of the register file)
//   row(W) = 2.37 / 2.90 / 3.00 us per schedule row (all sweeps of one solve together) with 1 / 2 / 4 waves: the row barrier
//            spans the workgroup;  c0 = 26 us, b = 0.36 us per node and worker (passes, epilogue)
//   load   = 1 + 0.21 [h not in LDS] x min(1, workgroups / CUs): factors through L2 / HBM slow down as the chip fills
//   share  = 1 + 0.5 x max(0, waves per SIMD - 1)
// Least-squares fit (relative error, 25 points, residuals <= 10 %, tools/nr_geometry_fit.py) to the launch times measured for
// case33 / case141 / case141_deep / case322 at 1024 ... 16384 envs in fat and lean layouts (profiles/r02_nr_geometry_case141.txt,
// profiles/r03_geometry_case322.txt, profiles/r03_final_*_kernel_stats.txt); it ranks every measured pair in the measured order.
// mapdn_env_config.nr_waves / nr_lanes / nr_lean pin a choice.
static double nr_model_ns(int W, int L, int n, int R, int h_lds, long wgs, int resident, int n_cu) {
  const double Wt = (double)W * (64 / L);
  const long per_round = (long)n_cu * std::max(resident, 1);
  const double rounds = (double)((wgs + per_round - 1) / per_round);
  const double row = W == 1 ? 2374.0 : (W == 2 ? 2903.0 : 3001.0);
  const double fill = (double)wgs / (double)n_cu;                           // workgroups per CU wanted
  const double conc = std::min((double)std::max(resident, 1), std::max(1.0, fill));   // ... sharing a CU
  const double base = 26032.0 + 363.0 * (double)n / Wt + row * (double)R * (h_lds ? 1.0 : 1.023);
  const double load = 1.0 + (h_lds ? 0.0 : 0.21) * std::min(1.0, fill);
  const double share = 1.0 + 0.5 * std::max(0.0, conc * W / 4.0 - 1.0);
  return rounds * base * load * share;
}

// Settles the k_nr_tree geometry of a handle: (W, L, lean) — pinned by the config / environment or chosen by the model above —
// then the schedule for Wt workers and the LDS residents.  Works without a device (host-only handles assume n_cu CUs).
static int settle_tree_geometry(mapdn_handle* h, int Bp, int n_cu) {
  const Plan& P = h->plan;
  const mapdn_env_config& c = h->cfg;
  const size_t LDS_MAX = 160 * 1024;
  if (P.n + 1 > 0xffff) { h->err = "networks with more than 65534 buses are not supported (16-bit node positions in the NR step records)"; return MAPDN_E_INVALID; }
  const int f_h = knob_tri(c.nr_h_lds, "MAPDN_NR_H_LDS"), f_g = knob_tri(c.nr_g_lds, "MAPDN_NR_G_LDS"), f_rec = knob_tri(c.nr_rec_lds, "MAPDN_NR_REC_LDS"),
            f_flat = knob_tri(c.nr_flat_lds, "MAPDN_NR_FLAT_LDS"), f_line = knob_tri(c.nr_line_lds, "MAPDN_NR_LINE_LDS");
  auto tri = [](int forced, bool dflt) { return forced == 1 ? 1 : (forced == 2 ? 0 : (dflt ? 1 : 0)); };
  // Optional LDS residents, in order of benefit: the h factors, the step records, the flat-start constants, the net.line
  // constants of the fused res_line epilogue, then the G factors (with everything resident the solve state never leaves
  // the chip; what does not fit stays in / goes to L2-resident global memory).  In lean mode only the voltages and hand-off
  // slots are resident, so that several workgroups share a CU.
  // The kernel peels the first rows of its full sweeps (their G — and h, when h is not in LDS — stay in registers): a schedule
  // shorter than that is rebuilt with idle rows appended; one that needs no peeled rows (G in LDS) is left alone.
  auto settle = [&](int W, int L, int lean, Schedule& S, mapdn_handle::Geo& g) -> int {   // 0 ok, 1 does not fit / not compiled, <0 error
    const int Wt = W * (64 / L);
    int min_rows = 0;
    for (int pass = 0; pass < 4; ++pass) {
      build_schedule(P, Wt, S, nr_min_cslots(W, L), 64 / L, min_rows);
      if (S.n_cslots > 1023 || S.n_xslots > 1023) { h->err = "NR schedule needs more than 1023 LDS slots"; return MAPDN_E_INVALID; }
      g.ncl = (int)S.clist.size();
      if (g.ncl > 4095) { h->err = "NR schedule: more than 4095 overflow children (junctions with > 3 non-chain children)"; return MAPDN_E_INVALID; }
      if ((long)S.R * Wt > 0xffff) { h->err = "NR schedule has more than 65535 steps"; return MAPDN_E_INVALID; }
      const int R_ = S.R;
      auto lds_for = [&](int hl, int gl, int ll, int rl, int fl) {
        return nr_lds_bytes(W, L, P.n, S.n_cslots, S.n_xslots, g.ncl, hl, gl, ll ? P.n_line : 0, rl ? R_ : 0, fl ? R_ : 0); };
      g.h_lds = tri(f_h, !lean && lds_for(1, 0, 0, 0, 0) <= LDS_MAX);
      g.rec_lds = tri(f_rec, !lean && lds_for(g.h_lds, 0, 0, 1, 0) <= LDS_MAX);
      g.flat_lds = tri(f_flat, !lean && lds_for(g.h_lds, 0, 0, g.rec_lds, 1) <= LDS_MAX);
      g.line_lds = (tri(f_line, !lean && P.n_line > 0 && lds_for(g.h_lds, 0, 1, g.rec_lds, g.flat_lds) <= LDS_MAX) && P.n_line > 0) ? 1 : 0;
      g.g_lds = (tri(f_g, !lean && g.h_lds && lds_for(1, 1, g.line_lds, g.rec_lds, g.flat_lds) <= LDS_MAX) && g.h_lds) ? 1 : 0;
      g.lds = lds_for(g.h_lds, g.g_lds, g.line_lds, g.rec_lds, g.flat_lds);
      if (g.lds > LDS_MAX) return 1;
      const int need = g.g_lds ? 0 : (g.h_lds ? NR_G_REG_ROWS : NR_HG_REG_ROWS);
      if (R_ >= need) { min_rows = -1; break; }
      min_rows = need;
    }
    if (min_rows >= 0) { h->err = "NR schedule: could not settle the number of peeled rows"; return MAPDN_E_INVALID; }
    const int compiled = nr_geometry_compiled(W, L, g.h_lds, g.g_lds, g.rec_lds, g.flat_lds);
    if (!compiled) return 1;
    g.W = W; g.L = L; g.lean = lean; g.rows = S.R;
    g.wgs = Bp / L;
    // workgroups resident per CU: LDS, and the register file — every k_nr_tree instantiation keeps factors of its peeled rows in
    // AGPRs and uses 284 ... 452 of a SIMD's 512 registers per lane: one wave per SIMD, i.e. 4 / W workgroups per CU
    g.resident = (int)std::max<size_t>(1, std::min<size_t>(LDS_MAX / std::max<size_t>(g.lds, 1), (size_t)std::max(4 / W, 1)));
    const long per_round = (long)n_cu * g.resident;
    g.rounds = (int)((g.wgs + per_round - 1) / per_round);
    g.model_ns = nr_model_ns(W, L, P.n, S.R, g.h_lds, g.wgs, g.resident, n_cu) * (compiled == 2 ? 1.0 : 1.05);   // generic body: the
                                                                   // per-row residency branches cost ~6 %
    return 0;
  };
  int W = knob_int(c.nr_waves, "MAPDN_NR_WAVES"), L = knob_int(c.nr_lanes, "MAPDN_NR_LANES");
  const int lean_k = knob_tri(c.nr_lean, "MAPDN_NR_LEAN");          // 0 auto, 1 lean, 2 fat
  if ((W != 0 && W != 1 && W != 2 && W != 4 && W != 8) || (L != 0 && L != 4 && L != 8 && L != 16 && L != 32)) {
    h->err = "nr_waves (MAPDN_NR_WAVES) must be 1/2/4/8 and nr_lanes (MAPDN_NR_LANES) 4/8/16/32 (0 = automatic)"; return MAPDN_E_INVALID; }
  mapdn_handle::Geo best; Schedule bestS; bool have = false;
  const bool forced = W != 0 && L != 0 && lean_k != 0;
  // automatic candidates: the pairs the model was fitted on; the other compiled pairs (nr_inst_list.hpp) only when pinned
  static const int cand[][2] = {{1, 16}, {2, 16}, {4, 16}, {4, 8}};
  for (const auto& wl : cand) {
    if ((W && wl[0] != W) || (L && wl[1] != L)) continue;
    for (int lean = 0; lean < 2; ++lean) {
      if ((lean_k == 1 && !lean) || (lean_k == 2 && lean)) continue;
      mapdn_handle::Geo g; Schedule S;
      const int r = settle(wl[0], wl[1], lean, S, g);
      if (r < 0) return r;
      if (r > 0) continue;
      if (!have || g.model_ns < best.model_ns) { best = g; bestS = std::move(S); have = true; }
    }
  }
  if (!have && W && L) {          // a pinned pair outside the candidate list (e.g. tests): no model, just settle it
    mapdn_handle::Geo g; Schedule S;
    const int r = settle(W, L, lean_k == 1 ? 1 : 0, S, g);
    if (r < 0) return r;
    if (r == 0) { best = g; bestS = std::move(S); have = true; }
    else if (g.lds > LDS_MAX) { h->err = "NR schedule needs more LDS than one CU has (160 KB); use fewer envs per workgroup (nr_lanes / MAPDN_NR_LANES) or fewer waves"; return MAPDN_E_INVALID; }
  }
  if (!have) {
    h->err = (W || L) ? "this (nr_waves, nr_lanes) combination is not compiled in or does not fit the 160 KB LDS of a CU (csrc/nr_inst_list.hpp)"
                      : "no compiled k_nr_tree geometry fits this network into the 160 KB LDS of a CU";
    return MAPDN_E_INVALID; }
  if (forced) best.model_ns = 0.0;
  const int mm = knob_tri(c.nr_mm_pass, "MAPDN_NR_MM_PASS");
  best.mm_pass = (best.h_lds && mm != 2) ? 1 : 0;    // the pass needs the h array in LDS
  h->geo = best; h->sched = std::move(bestS); h->lds_bytes = best.lds;
  if (knob_int(c.debug_geometry, "MAPDN_DEBUG_GEOMETRY"))
    fprintf(stderr, "[mapdn] k_nr_tree geometry: W %d L %d lean %d rows %d cslots %d | LDS: h %d rec %d flat %d line %d G %d = %zu B | "
                    "%ld workgroups, %d per CU, %d round(s), model %.1f us\n",
            best.W, best.L, best.lean, best.rows, h->sched.n_cslots, best.h_lds, best.rec_lds, best.flat_lds, best.line_lds, best.g_lds,
            best.lds, best.wgs, best.resident, best.rounds, best.model_ns * 1e-3);
  return MAPDN_OK;
}

extern "C" {

const char* mapdn_last_error(const mapdn_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

#ifndef MAPDN_SRC_HASH
#define MAPDN_SRC_HASH "unknown"
#endif
// "MAPDN_SRC_HASH=<sha256 of the sources and flags this library was built from>" (mapdn_amd/build.py): the loader compares it with the
// sources on disk, so that a prebuilt library can never silently disagree with them
const char* mapdn_build_info(void) { return "MAPDN_SRC_HASH=" MAPDN_SRC_HASH; }

static int create_impl(mapdn_handle* h, const mapdn_netspec* net, const mapdn_env_config* cfg, int32_t B, int32_t device) {
  if (!net || !cfg) { h->err = "null netspec/config"; return MAPDN_E_INVALID; }
  if (B < 1) { h->err = "n_envs must be >= 1"; return MAPDN_E_INVALID; }
  if (cfg->barrier_type < 0 || cfg->barrier_type > MAPDN_BARRIER_BUMP) { h->err = "unknown voltage_barrier_type"; return MAPDN_E_INVALID; }
  if (!cfg->use_line_weight && !cfg->use_q_weight) {   // voltage_control_env.py:616-617
    h->err = "NotImplementedError: Please at least give one weight, either q_weight or line_weight."; return MAPDN_E_INVALID; }
  if (cfg->episode_limit < 2) { h->err = "episode_limit must be >= 2"; return MAPDN_E_INVALID; }
  if (cfg->nr_init != 0) {   // reserved: see include/mapdn.h (the gating study found the warm start safe but without effect on the launch time)
    h->err = "nr_init != 0 (warm start, runpp init=\"results\") is not built: tools/warm_start_study.py / profiles/r04_warm_start_study_*.json "
             "show no iteration saved per 16-env workgroup; every solve starts flat like the reference's";
    return MAPDN_E_INVALID; }
  int rc = build_plan(*net, *cfg, h->plan, h->err);
  if (rc) return rc;
  {
    // Solver choice.  pp.runpp (voltage_control_env.py:557) solves any connected net: radial feeders take the fill-free
    // tree kernel; meshed nets the general sparse kernel (host symbolic factorisation with fill + a block program, all
    // blocks of L envs in LDS); MAPDN_NR_DENSE=1 selects the dense LDS-resident LU with f64 MFMA (<= 65 buses),
    // MAPDN_NR_SPARSE=1 the sparse kernel on a radial net (cross-checks).
    const Plan& P0 = h->plan;
    int pick = cfg->nr_solver;
    if (const char* fs = getenv("MAPDN_NR_SPARSE")) if (atoi(fs)) pick = 1;
    if (const char* fd = getenv("MAPDN_NR_DENSE")) if (atoi(fd)) pick = 2;
    if (pick < 0 || pick > 2) { h->err = "nr_solver must be 0 (auto), 1 (sparse) or 2 (dense)"; return MAPDN_E_INVALID; }
    const bool want_dense = pick == 2;
    const bool want_sparse = !want_dense && (pick == 1 || !P0.radial);
    if (want_dense) {
      if (2 * P0.n > 1024) { h->err = "nr_solver = dense (MAPDN_NR_DENSE): the dense general-topology solver handles at most 513 buses (one thread per Jacobian row)"; return MAPDN_E_TOPOLOGY; }
      h->solver = 2;
    } else if (want_sparse) {
      SparseProg g0;
      sparse_symbolic(P0, g0);
      int Lc = 0;
      for (int l : {16, 8, 4, 2}) if (nr_sparse_lds_bytes(P0.n, g0.n_blocks, l) <= 160 * 1024) { Lc = l; break; }
      if (!Lc) {
        h->err = "topology: meshed network with " + std::to_string(P0.nb) + " buses needs " + std::to_string(g0.n_blocks) +
                 " Jacobian blocks after fill; two envs of it do not fit the 160 KB LDS of a CU";
        return MAPDN_E_TOPOLOGY; }
      h->solver = 1; h->sp_lanes = Lc;
    }
  }
  h->cfg = *cfg;
  h->device = device;
  std::memset(&h->d, 0, sizeof(h->d));
  h->d.B = B;
  if (device == -1) {                                            // plan only (CPU tests): no device work
    h->host_only = true;
    if (h->solver == 0) return settle_tree_geometry(h, (B + 63) / 64 * 64, 256);   // (an MI355X has 256 CUs)
    return MAPDN_OK;
  }
  int ndev = 0;
  HIPCHK(h, hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { h->err = "device index out of range"; return MAPDN_E_HIP; }
  HIPCHK(h, hipSetDevice(device));
  const Plan& P = h->plan;
  Dev& d = h->d;
  d.B = B; d.Bp = (B + 63) / 64 * 64; d.nb = P.nb; d.n = P.n; d.nl = P.nl; d.ns = P.ns; d.n_line = P.n_line;
  d.nbo = P.nbo;                                 // original buses (> nb with fused buses): the rows of res_bus / obs / state
  d.ncol = P.ns + 2 * P.nl;
  d.vroot = P.vroot; d.sn = P.sn_mva; d.tol = P.tol; d.max_it = 10;   // runpp max_iteration="auto" -> 10
  d.yrr0 = P.yrr[0]; d.yrr1 = P.yrr[1];
  d.barrier_type = cfg->barrier_type; d.use_line_weight = cfg->use_line_weight; d.episode_limit = cfg->episode_limit;
  d.reset_action = cfg->reset_action; d.auto_reset = cfg->auto_reset ? 1 : 0; d.voltage_weight = cfg->voltage_weight; d.q_weight = cfg->q_weight;
  d.line_weight = cfg->line_weight; d.v_lower = cfg->v_lower; d.v_upper = cfg->v_upper;
  d.action_low = cfg->action_low; d.action_high = cfg->action_high;
  d.seed_lo = (uint32_t)(cfg->seed & 0xffffffffull); d.seed_hi = (uint32_t)(cfg->seed >> 32);
  d.env_id_offset = cfg->env_id_offset;
#define UP(field, vec) do { rc = dupload(h, &d.field, vec); if (rc) return rc; } while (0)
  UP(bus_of_pos, P.bus_of_pos); UP(root_children, P.root_children); UP(root_y, P.root_y);
  UP(pos_of_obus, P.pos_of_obus); UP(cm_kind, P.cm_kind);
  d.n_fused = (int32_t)P.fused_obus.size(); d.n_alias = (int32_t)P.alias_pos.size(); d.n_slack_group = (int32_t)P.slack_group.size();
  if (d.n_fused) {
    UP(fused_obus, P.fused_obus); UP(ob_load_ptr, P.ob_load_ptr); UP(ob_load_idx, P.ob_load_idx); UP(ob_sgen_ptr, P.ob_sgen_ptr);
    UP(ob_sgen_idx, P.ob_sgen_idx); UP(ob_shunt_p, P.ob_shunt_p); UP(ob_shunt_q, P.ob_shunt_q);
    { std::vector<int32_t> sg(P.slack_group); if (sg.empty()) sg.push_back(0); UP(slack_group, sg); }
    { std::vector<int32_t> ap(P.alias_pos); if (ap.empty()) ap.push_back(0); UP(alias_pos, ap); }
  }
  d.n_root_children = (int32_t)P.root_children.size();
  UP(load_ptr, P.load_ptr); UP(load_idx, P.load_idx); UP(sgen_ptr, P.sgen_ptr); UP(sgen_idx, P.sgen_idx);
  UP(shunt_p, P.shunt_p); UP(shunt_q, P.shunt_q); UP(load_scale, P.load_scale); UP(sgen_scale, P.sgen_scale);
  {
    std::vector<int32_t> sgb, sgb_of(P.nb, -1), lb, mlo;
    for (int k = 0; k < P.nb; ++k) {
      if (P.sgen_ptr[k + 1] > P.sgen_ptr[k]) { sgb_of[k] = (int32_t)sgb.size(); sgb.push_back(k); }
      else if (P.load_ptr[k + 1] > P.load_ptr[k]) {
        lb.push_back(k);
        if (P.load_ptr[k + 1] - P.load_ptr[k] > 1 && k < P.n) mlo.push_back(k);
      }
    }
    d.n_sgb = (int32_t)sgb.size(); d.n_lb = (int32_t)lb.size(); d.n_mlo = (int32_t)mlo.size();
    if (lb.empty()) lb.push_back(0);
    if (mlo.empty()) mlo.push_back(0);
    {   // per (PV bus | load-only bus with several loads) row of the injection: what the fused prologue of k_nr_tree needs in one 16-byte load
      std::vector<int32_t> rec;
      auto put = [&](int k, bool pv) {
        const int nsg = pv ? P.sgen_ptr[k + 1] - P.sgen_ptr[k] : 0, nld = P.load_ptr[k + 1] - P.load_ptr[k];
        rec.push_back(k < P.n ? k : -1);            // Sbus is stored by node position (sb_index[k] == k, see alloc_nrbuf below)
        rec.push_back(k);
        rec.push_back(pv ? P.sgen_idx[P.sgen_ptr[k]] : -1);
        rec.push_back((nsg << 8) | std::min(nld, 2));
        rec.push_back(nld > 0 ? P.load_idx[P.load_ptr[k]] : -1);        // the first two loads of the bus (CSR order)
        rec.push_back(nld > 1 ? P.load_idx[P.load_ptr[k] + 1] : -1);
        rec.push_back(0); rec.push_back(0);
      };
      for (int i = 0; i < d.n_sgb; ++i) put(sgb[i], true);
      for (int i = 0; i < d.n_mlo; ++i) put(mlo[i], false);
      UP(sgb_rec, rec);
    }
    UP(sgb_pos, sgb); UP(sgb_of_pos, sgb_of); UP(lb_pos, lb); UP(mlo_pos, mlo);
    h->ld_dest_host.assign(std::max(P.nl, 1), 2);             // filled once the Sbus order (sb_index) is known, see alloc_nrbuf
    rc = dalloc(h, &d.bus_ld, (size_t)2 * d.n_sgb * d.Bp); if (rc) return rc;
    h->inject_full = knob_int(cfg->inject_full, "MAPDN_INJECT_FULL") != 0;
  }
  {
    std::vector<LineFlow> padded(P.lines);               // read 16 bytes at a time by the NR kernel's LDS staging
    if (padded.size() % 2) padded.push_back(LineFlow{});
    UP(lines, padded);
  }
  const size_t Bp = d.Bp;
#define AL(field, rows) do { rc = dalloc(h, &d.field, (size_t)(rows) * Bp); if (rc) return rc; } while (0)
  AL(q_new, d.ns); AL(cur_pl, d.nl); AL(cur_ql, d.nl); AL(pl, d.n_line);
  // gatherable state block: cur_pv cur_q [ns] | vm va res_p res_q [nb]
  const int r_pv = 0, r_q = r_pv + d.ns, r_vm = r_q + d.ns, r_va = r_vm + d.nbo,
            r_rp = r_va + d.nbo, r_rq = r_rp + d.nbo, g_rows = r_rq + d.nbo;
  rc = dalloc(h, &d.gbuf, (size_t)g_rows * Bp); if (rc) return rc;
  d.cur_pv = d.gbuf + (size_t)r_pv * Bp;
  d.cur_q = d.gbuf + (size_t)r_q * Bp; d.vm = d.gbuf + (size_t)r_vm * Bp; d.va = d.gbuf + (size_t)r_va * Bp;
  d.res_p = d.gbuf + (size_t)r_rp * Bp; d.res_q = d.gbuf + (size_t)r_rq * Bp;
  AL(sum_rewards, 1); AL(steps, 1); AL(start_row, 1); AL(draw, 1); AL(done, 1); AL(pending, 1);
  AL(active, 1); AL(commit, 1); AL(bad_start, 1); AL(resetting, 1); AL(adv_row, 1); AL(adv_draw, 1); AL(iters, 1); AL(conv, 1);
  {
    std::vector<uint8_t> ones(Bp, 1);
    HIPCHK(h, hipMemcpy(d.done, ones.data(), Bp, hipMemcpyHostToDevice));   // nothing is steppable before reset
  }
  const int32_t* tmp;
  {  // obs / state columns -> (source row of gbuf, scale, extra rows to add); -1 = zero padding.
     // P/Q columns with the effective PV add-back (voltage_control_env.py:238-244) = res_bus row + the
     // sgen.p_mw / q_mvar rows of the sgens on that bus.
    std::vector<std::vector<int>> sgens_at(P.nbo);
    for (int j = 0; j < P.ns; ++j) sgens_at[P.sgen_bus[j]].push_back(j);
    auto tables = [&](const std::vector<int32_t>& kind, const std::vector<int32_t>& idx, std::vector<int32_t>& rows,
                      std::vector<double>& scale, std::vector<int32_t>& xptr, std::vector<int32_t>& xrow) {
      rows.resize(kind.size()); scale.assign(kind.size(), 1.0); xptr.assign(kind.size() + 1, 0); xrow.clear();
      for (size_t c = 0; c < kind.size(); ++c) {
        switch (kind[c]) {
          case G_P_ADDBACK: rows[c] = r_rp + idx[c]; for (int j : sgens_at[idx[c]]) xrow.push_back(r_pv + j); break;
          case G_Q_ADDBACK: rows[c] = r_rq + idx[c]; for (int j : sgens_at[idx[c]]) xrow.push_back(r_q + j); break;
          case G_SGEN_P: rows[c] = r_pv + idx[c]; break;
          case G_SGEN_Q: rows[c] = r_q + idx[c]; break;
          case G_VM: rows[c] = r_vm + idx[c]; break;
          case G_VA_RAD: rows[c] = r_va + idx[c]; break;
          case G_P: rows[c] = r_rp + idx[c]; break;
          case G_Q: rows[c] = r_rq + idx[c]; break;
          case G_VA_DEG: rows[c] = r_va + idx[c]; scale[c] = 180.0 / M_PI; break;
          default: rows[c] = -1; break;
        }
        xptr[c + 1] = (int32_t)xrow.size();
        if (xptr[c + 1] > xptr[c]) rows[c] |= 0x40000000;   // GATHER_HAS_EXTRA
      }
      if (xrow.empty()) xrow.push_back(0);
    };
    std::vector<int32_t> rows, xptr, xrow; std::vector<double> scale; const double* dt;
    tables(P.obs_kind, P.obs_idx, rows, scale, xptr, xrow);
    rc = dupload(h, &tmp, rows); if (rc) return rc; h->obs_rows = (int32_t*)tmp;
    rc = dupload(h, &dt, scale); if (rc) return rc; h->obs_scale = (double*)dt;
    rc = dupload(h, &tmp, xptr); if (rc) return rc; h->obs_xptr = (int32_t*)tmp;
    rc = dupload(h, &tmp, xrow); if (rc) return rc; h->obs_xrow = (int32_t*)tmp;
    tables(P.state_kind, P.state_idx, rows, scale, xptr, xrow);     // get_state has no add-back
    rc = dupload(h, &tmp, rows); if (rc) return rc; h->state_rows = (int32_t*)tmp;
    rc = dupload(h, &dt, scale); if (rc) return rc; h->state_scale = (double*)dt;
  }
  const int maxn = std::max(std::max(d.nbo, d.n_line), std::max(d.nl, d.ns));
  std::vector<int32_t> io(maxn);
  for (int i = 0; i < maxn; ++i) io[i] = i;
  rc = dupload(h, &tmp, io); if (rc) return rc; h->iota_idx = (int32_t*)tmp;
  rc = dalloc(h, &h->t_pl, (size_t)d.nl * Bp); if (rc) return rc;
  rc = dalloc(h, &h->t_ql, (size_t)d.nl * Bp); if (rc) return rc;
  rc = dalloc(h, &h->t_pv, (size_t)d.ns * Bp); if (rc) return rc;
  rc = dalloc(h, &h->t_q, (size_t)d.ns * Bp); if (rc) return rc;
  rc = dalloc(h, &h->stats_dev, 4); if (rc) return rc;
  // NR scratch `nrbuf`: factor blocks (fb_rows pair rows) | Sbus (nblk pair rows, entry sbi[k] for position k) | Vout;
  // a single buffer resource addresses it
  auto alloc_nrbuf = [&](size_t fb_rows, size_t nblk, const std::vector<int32_t>& sbi) -> int {
    const size_t sb_off = fb_rows * Bp * 16;
    const size_t vout_off = sb_off + 2 * nblk * Bp * 16;        // two Sbus buffers
    const size_t bytes = vout_off + (size_t)VOF * (P.n + 1) * Bp * sizeof(double);
    if (bytes >= (size_t)0xFFFFFFFFu) { h->err = "env batch too large: NR scratch exceeds the 4 GiB one buffer resource addresses; use fewer envs per handle"; return MAPDN_E_INVALID; }
    rc = dalloc(h, &d.nrbuf, bytes / sizeof(double)); if (rc) return rc;
    d.nrbuf_bytes = (uint32_t)bytes;
    d.sb_off = (uint32_t)sb_off; d.sb_off_alt = (uint32_t)(sb_off + nblk * Bp * 16);
    h->sb_base = d.sb_off; h->sb_bytes = (uint32_t)(nblk * Bp * 16);
    d.r_vout = (uint32_t)(vout_off / (Bp * sizeof(double)));
    rc = dupload(h, &d.sb_index, sbi); if (rc) return rc;
    {   // ld_dest: a load alone on its bus goes straight to the Sbus entry (kind 0) / bus_ld row (kind 1) of that bus
      std::vector<int32_t> sgb_of(P.nb, -1); int nsg = 0;
      for (int k = 0; k < P.nb; ++k) if (P.sgen_ptr[k + 1] > P.sgen_ptr[k]) sgb_of[k] = nsg++;
      for (int k = 0; k < P.nb; ++k) {
        if (P.load_ptr[k + 1] - P.load_ptr[k] != 1) continue;
        const int li = P.load_idx[P.load_ptr[k]];
        if (sgb_of[k] >= 0) h->ld_dest_host[li] = (sgb_of[k] << 2) | 1;
        else if (k < P.n) h->ld_dest_host[li] = (sbi[k] << 2) | 0;
      }
      rc = dupload(h, &d.ld_dest, h->ld_dest_host); if (rc) return rc;
    }
    std::vector<double> row(Bp, d.vroot);   // slack entry of Vout: V = vroot + 0j (angle 0 from the memset)
    double* rootv = d.nrbuf + ((size_t)d.r_vout + (size_t)VOF * P.n) * Bp;
    HIPCHK(h, hipMemcpy(rootv + (size_t)VO_E * Bp, row.data(), Bp * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(rootv + (size_t)VO_VM * Bp, row.data(), Bp * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int32_t> vmrow(P.nbo), varow(P.nbo);
    for (int b = 0; b < P.nbo; ++b) { const int k = P.pos_of_obus[b]; vmrow[b] = (int)d.r_vout + VOF * k + VO_VM; varow[b] = (int)d.r_vout + VOF * k + VO_VA; }
    rc = dupload(h, &tmp, vmrow); if (rc) return rc; h->vm_row = (int32_t*)tmp;
    rc = dupload(h, &tmp, varow); if (rc) return rc; h->va_row = (int32_t*)tmp;
    return MAPDN_OK;
  };
  // ---- step() composition switches that only exist on the tree solver: checked BEFORE the solver-specific set-up returns
  // (mapdn.h: a pinned switch that cannot take effect is MAPDN_E_INVALID, never silently ignored)
  {
    const int fi = knob_tri(cfg->fuse_inject, "MAPDN_FUSE_INJECT");
    const bool ov = knob_int(cfg->overlap_advance, "MAPDN_OVERLAP_ADVANCE") != 0, xm = knob_tri(cfg->xcd_map, "MAPDN_XCD_MAP") == 1;
    if (h->solver != 0 && (fi == 1 || ov || xm)) {
      h->err = "fuse_inject = 1 / overlap_advance / xcd_map exist on the tree solver only (this handle runs the general sparse / dense solver)";
      return MAPDN_E_INVALID; }
    if (ov && fi != 2 && !cfg->auto_reset && !knob_int(cfg->inject_full, "MAPDN_INJECT_FULL")) {
      h->err = "overlap_advance = 1 has no effect while the PV-bus injection runs in the solver's prologue (the prologue reads what the side "
               "stream's profile rows write): set fuse_inject = 2 as well";
      return MAPDN_E_INVALID; }
  }
  // ---- general-topology paths (see the solver choice above)
  if (h->solver == 2) {                           // k_nr_dense (dense.hip): one env per workgroup, dense Jacobian in LDS, f64 MFMA
    d.dense = 1; d.dn_N = (2 * P.n + 15) / 16 * 16; d.dn_lda = d.dn_N + 2;
    if (d.dn_N > 128) {                            // beyond 65 buses the Jacobian of every env lives in a slab of global memory
      const size_t per_env = (size_t)d.dn_N * d.dn_lda;
      rc = dalloc(h, &d.dn_A, per_env * d.Bp); if (rc) return rc;
    }
    UP(gy_ptr, P.gy_ptr); UP(gy_col, P.gy_col); UP(gy_val, P.gy_val);
    std::vector<int32_t> sbi(P.n);
    for (int k = 0; k < P.n; ++k) sbi[k] = k;
    rc = alloc_nrbuf(0, (size_t)P.n, sbi); if (rc) return rc;
    if (nr_dense_prepare(d) != 0) { (void)hipGetLastError(); h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_nr_dense"; return MAPDN_E_HIP; }
    h->lds_bytes = nr_dense_lds_bytes(d);
    return MAPDN_OK;
  }
  if (h->solver == 1) {                           // k_nr_sparse (sparse.hip): host-compiled block elimination program
    // Envs per (one-wave) workgroup: fewer envs = more sub-lanes per env = fewer phases, and a smaller LDS tile = more
    // resident waves per CU to hide each other's LDS latency.  Score = envs in flight per CU / length of the per-iteration
    // instruction stream (phases + assembly entries); measured on case33 / case141 / case322 with tie lines closed
    // (profiles/r02_sparse_lanes_sweep.txt): the score ranks the four geometries in the measured order.
    {
      double best = -1.0;
      for (int l : {16, 8, 4, 2}) {
        SparseProg g;
        sparse_program(P, 64 / l, g);
        const size_t lds = nr_sparse_lds_bytes(P.n, g.n_blocks, l);
        if (lds > 160 * 1024) continue;
        const double waves = (double)std::min<size_t>(160 * 1024 / lds, 8);
        const double score = waves * l / ((double)g.n_phases + 0.5 * g.rows_per_sub * g.max_nnz);
        if (score > best) { best = score; h->sp_lanes = l; }
      }
    }
    {                                                  // pinned envs per workgroup (16 / 8 / 4 / 2)
      const int l = knob_int(cfg->sp_lanes, "MAPDN_SP_LANES");
      if (l == 16 || l == 8 || l == 4 || l == 2) h->sp_lanes = l;
      else if (l != 0) { h->err = "sp_lanes (MAPDN_SP_LANES) must be 16, 8, 4 or 2 (0 = automatic)"; return MAPDN_E_INVALID; }
    }
    sparse_program(P, 64 / h->sp_lanes, h->sprog);
    const SparseProg& G = h->sprog;
    if (nr_sparse_lds_bytes(P.n, G.n_blocks, h->sp_lanes) > 160 * 1024) { h->err = "sp_lanes (MAPDN_SP_LANES): does not fit in LDS"; return MAPDN_E_INVALID; }
    d.sparse = 1; d.sp_lanes = h->sp_lanes; d.sp_blocks = G.n_blocks; d.sp_fill = (int32_t)G.fill_slots.size();
    d.sp_phases = G.n_phases; d.sp_rows_per_sub = G.rows_per_sub; d.sp_max_nnz = G.max_nnz;
    UP(sp_ops, G.ops); d.sp_ops_bytes = (uint32_t)(G.ops.size() * sizeof(SpOp));
    UP(sp_nz, G.nz); d.sp_nz_bytes = (uint32_t)(G.nz.size() * sizeof(SpNz));
    { std::vector<int32_t> fs(G.fill_slots); if (fs.empty()) fs.push_back(G.n_blocks - 1); UP(sp_fill_slots, fs); }
    std::vector<int32_t> sbi(P.n);
    for (int k = 0; k < P.n; ++k) sbi[k] = k;
    rc = alloc_nrbuf(0, (size_t)P.n, sbi); if (rc) return rc;
    if (nr_sparse_prepare(h->sp_lanes) != 0) { (void)hipGetLastError(); h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_nr_sparse"; return MAPDN_E_HIP; }
    h->lds_bytes = nr_sparse_lds_bytes(P.n, G.n_blocks, h->sp_lanes);
    return MAPDN_OK;
  }
  // ---- NR launch geometry (settle_tree_geometry above)
  {
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    rc = settle_tree_geometry(h, d.Bp, prop.multiProcessorCount); if (rc) return rc;
  }
  const mapdn_handle::Geo& G_ = h->geo;
  const int ncl = G_.ncl, h_lds = G_.h_lds, g_lds = G_.g_lds;
  d.nr_waves = G_.W; d.nr_lanes = G_.L; d.nr_h_lds = G_.h_lds; d.nr_g_lds = G_.g_lds; d.nr_line_lds = G_.line_lds; d.nr_rec_lds = G_.rec_lds; d.nr_flat_lds = G_.flat_lds;
  // 1e-7: with quadratic convergence the mismatch after such a step is ~|Y| dx^2 << tol, so a wrong prediction
  // (which costs one extra mismatch-only sweep for that workgroup) practically never happens
  d.nr_check_dx = knob_f64(cfg->nr_check_dx, "MAPDN_NR_CHECK_DX", 1e-7);
  // quadratic extrapolation of the mismatch norm, no safety margin: in 4096-env samples of all three cases it
  // predicts the last sweep of 95-100 % of the workgroups and never a non-final one (tools/predictor_study.py)
  d.nr_check_quad = knob_f64(cfg->nr_check_quad, "MAPDN_NR_CHECK_QUAD", 1.0);
  d.nr_rows = h->sched.R; d.nr_cslots = h->sched.n_cslots; d.nr_xslots = h->sched.n_xslots; d.nr_nclist = ncl;
  {
    // the attribute is per kernel function, not per handle: always raise it to the full 160 KB so that
    // handles with different LDS needs can share an instantiation
    const int lr = nr_set_lds_limit(G_.W, G_.L, G_.h_lds, G_.g_lds, G_.rec_lds, G_.flat_lds, 160 * 1024);
    if (lr == -2) { h->err = "this (nr_waves, nr_lanes) combination is not compiled in (csrc/nr_inst_list.hpp)"; return MAPDN_E_INVALID; }
    if (lr != 0) { (void)hipGetLastError(); h->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"; return MAPDN_E_HIP; }
  }
  UP(sched, h->sched.steps); d.sched_bytes = (uint32_t)(h->sched.steps.size() * sizeof(StepRec));
  UP(clist, h->sched.clist);
  UP(mm_ptr, h->sched.mm_ptr); UP(mm_child, h->sched.mm_child);
  UP(mm_recs, h->sched.mm_recs); d.mm_recs_bytes = (uint32_t)(h->sched.mm_recs.size() * sizeof(StepRec)); d.mm_np = h->sched.mm_np;
  d.nr_mm_pass = G_.mm_pass;   // the predicted-final mismatch evaluation as a barrier-free pass over all nodes instead of a tree sweep
  UP(flat, h->sched.flat); d.flat_bytes = (uint32_t)(h->sched.flat.size() * sizeof(double));
  {  // NR scratch: factor blocks (one per node) | 2 x Sbus (one entry per node) | Vout
    const size_t nblk = (size_t)P.n + 2;           // Sbus by node position (+ slack, + the trash node of idle steps: stays 0)
    const size_t fb_rows = (h_lds && g_lds) ? 0 : (size_t)(P.n + 2) * NBP;   // pair rows of Bp x 16 bytes: one block per node (+ slack, trash)
    std::vector<int32_t> sbi(P.n);
    for (int k = 0; k < P.n; ++k) sbi[k] = k;
    rc = alloc_nrbuf(fb_rows, nblk, sbi); if (rc) return rc;
  }
#undef UP
  {   // composition of the step() launches
    const int fi = knob_tri(cfg->fuse_inject, "MAPDN_FUSE_INJECT");
    const bool can = !d.auto_reset && !h->inject_full;
    if (fi == 1 && !can) { h->err = "fuse_inject = 1 needs a handle without auto_reset and without inject_full"; return MAPDN_E_INVALID; }
    h->fuse_inject = can && fi != 2;
    // XCD-aligned env order of the wide kernels (tree solver; the group size is the solver's envs per workgroup)
    const int xm = knob_tri(cfg->xcd_map, "MAPDN_XCD_MAP");
    d.xcd_lanes = (xm == 1 && (64 % d.nr_lanes) == 0) ? d.nr_lanes : 0;     // opt-in: measured a wash (profiles/r04_xcd_map_ab.txt)
    h->overlap = knob_int(cfg->overlap_advance, "MAPDN_OVERLAP_ADVANCE") != 0;
    if (h->overlap && d.n_fused) { h->err = "overlap_advance is not available on a net with fused buses (bus_alias)"; return MAPDN_E_INVALID; }
    if (h->overlap) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
  }
  return MAPDN_OK;
}

int mapdn_create(const mapdn_netspec* net, const mapdn_env_config* cfg, int32_t n_envs, int32_t device, mapdn_handle** out) {
  if (!out) { g_create_err = "null out pointer"; return MAPDN_E_INVALID; }
  *out = nullptr;
  mapdn_handle* h = nullptr;
  try {
    h = new mapdn_handle();
    const int rc = create_impl(h, net, cfg, n_envs, device);        // plan.cpp: dozens of std::vector / std::string allocations
    if (rc) { g_create_err = h->err; mapdn_destroy(h); return rc; }
    *out = h;
    return MAPDN_OK;
  } catch (...) {
    mapdn_destroy(h);                                                // (frees whatever the half-built handle owns; null is fine)
    try { throw; } MAPDN_CATCH(nullptr)
  }
}

void mapdn_destroy(mapdn_handle* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

int mapdn_dims(const mapdn_handle* h, mapdn_dims_t* out) try {
  if (!h || !out) return MAPDN_E_INVALID;
  const Plan& P = h->plan;
  out->n_envs = h->d.B; out->n_bus = P.nbo; out->n_line = P.n_line; out->n_load = P.nl; out->n_sgen = P.ns;
  out->n_agents = P.n_agents; out->n_actions = 1; out->obs_size = P.obs_size; out->state_size = P.state_size;
  out->n_info = MAPDN_N_INFO; out->is_radial = P.radial ? 1 : 0; out->max_zone_size = P.max_zone;
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_set_profiles(mapdn_handle* h, const double* pv, const double* load_p, const double* load_q,
                       int64_t T, int32_t time_delta_min, int32_t days) try {
  if (!h) return MAPDN_E_INVALID;
  if (!pv || !load_p || !load_q || T < 2) { h->err = "set_profiles: null table or too few rows"; return MAPDN_E_INVALID; }
  if (time_delta_min <= 0 || 60 % time_delta_min) { h->err = "set_profiles: time_delta_min must divide 60"; return MAPDN_E_INVALID; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  Dev& d = h->d;
  const int ns = d.ns, nl = d.nl, ncol = d.ncol;
  d.per_hour = 60 / time_delta_min; d.per_day = 24 * d.per_hour;
  const int episode_days = d.episode_limit / d.per_day + 1;          // voltage_control_env.py:397
  d.n_start_days = days - episode_days;                              // :398
  if (d.n_start_days < 1) { h->err = "set_profiles: table too short for one episode (pv_days - episode_days < 1)"; return MAPDN_E_INVALID; }
  const int64_t max_start = (d.per_hour - 1) + 23 * d.per_hour + (int64_t)(d.n_start_days - 1) * d.per_day;
  if (max_start + d.episode_limit + 1 >= T) { h->err = "set_profiles: sampled episodes could run past the end of the table"; return MAPDN_E_INVALID; }
  std::vector<double> tab((size_t)T * ncol), stdv(ncol), smax(ns);
  for (int64_t t = 0; t < T; ++t) {
    double* r = &tab[(size_t)t * ncol];
    std::memcpy(r, pv + (size_t)t * ns, ns * sizeof(double));
    std::memcpy(r + ns, load_p + (size_t)t * nl, nl * sizeof(double));
    std::memcpy(r + ns + nl, load_q + (size_t)t * nl, nl * sizeof(double));
  }
  // population std over the whole table / 100 (:70-72) in numpy's own summation order (colstats.hpp: pairwise along every column, as
  // `DataFrame.values.std(axis=0)` does on its F-ordered block), read row-major; s_max = 1.2 * max_t pv (:518-520)
  column_std(tab.data(), T, ncol, 100.0, stdv);
  for (int j = 0; j < ns; ++j) smax[(size_t)j] = tab[(size_t)j];
  for (int64_t t = 1; t < T; ++t) {
    const double* r = &tab[(size_t)t * ncol];
    for (int j = 0; j < ns; ++j) smax[(size_t)j] = std::max(smax[(size_t)j], r[j]);
  }
  for (int j = 0; j < ns; ++j) smax[(size_t)j] *= 1.2;
  if (h->have_profiles) {   // replace: free the old table
    for (double* p : {h->table, h->stdv, h->smax}) {
      auto it = std::find(h->allocs.begin(), h->allocs.end(), (void*)p);
      if (it != h->allocs.end()) { (void)hipFree(*it); h->allocs.erase(it); }
    }
  }
  const double* tmp;
  int rc = dupload(h, &tmp, tab); if (rc) return rc; h->table = (double*)tmp;
  rc = dupload(h, &tmp, stdv); if (rc) return rc; h->stdv = (double*)tmp;
  rc = dupload(h, &tmp, smax); if (rc) return rc; h->smax = (double*)tmp;
  d.table = h->table; d.stdv = h->stdv; d.smax = h->smax; d.T = T;
  h->stdv_host = stdv; h->smax_host = smax;
  h->have_profiles = true;
  return MAPDN_OK;
} MAPDN_CATCH(h)

static void nr_launch(mapdn_handle* h, int mode, double* reward, uint8_t* term, double* info, hipStream_t st,
                      const void* fused_actions = nullptr, int fused_dtype = 0) {
  if (!h->timing) { launch_nr(h->d, mode, reward, term, info, st, fused_actions, fused_dtype); return; }
  if (h->ev_used + 2 > h->ev.size()) {
    if (h->ev.size() >= 2 * 8192) {   // pool full: drain
      double ms; int64_t n; mapdn_nr_time_ms(h, &ms, &n); h->acc_ms = ms; h->acc_launches = n;
    } else {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      h->ev.push_back(a); h->ev.push_back(b);
    }
  }
  hipEvent_t a = h->ev[h->ev_used], b = h->ev[h->ev_used + 1];
  h->ev_used += 2;
  (void)hipEventRecord(a, st);
  launch_nr(h->d, mode, reward, term, info, st, fused_actions, fused_dtype);
  (void)hipEventRecord(b, st);
}

// the injection of a step() / reset() call: PV buses only when Sbus and bus_ld are known to reflect the current loads
static void inject_launch(mapdn_handle* h, int mode, const void* actions, int dtype, int add_noise, hipStream_t st) {
  const Dev& d = h->d;
  if (h->sbus_stale || h->inject_full) {
    launch_inject(d, mode, actions, dtype, d.cur_pl, d.cur_ql, d.cur_pv, nullptr, add_noise, st);
    h->sbus_stale = false;
  } else launch_inject_sgen(d, mode, actions, dtype, add_noise, st);
}

// The launches of one step(): injection (its own launch, or the prologue of k_nr_tree: fuse_inject) -> solve + reward ->
// profile advance (-> the Sbus buffer of the next solve) + res_bus commit.  With `overlap` the profile rows, which do not
// depend on the solve, run on the side stream beside the solver launch.
static int step_launches(mapdn_handle* h, const void* actions, int32_t actions_dtype, int32_t add_noise, double* reward,
                         uint8_t* terminated, double* info, hipStream_t st) {
  const Dev& d = h->d;
  const bool fused = h->fuse_inject && h->solver == 0 && !h->sbus_stale;
  if (!fused) inject_launch(h, MODE_STEP, actions, actions_dtype, add_noise, st);
  if (h->overlap && !fused) {
    // fork after the injection (it queues the row / draw the advance uses), join before the commit rows.  The side stream's
    // profile rows write cur_pv / cur_pl / bus_ld while the solver runs: safe only because the NON-fused k_nr_tree never reads them
    // (mapdn_create refuses overlap together with the fused prologue).  On an error between fork and join the side stream is
    // drained and the Sbus buffers are declared stale, so that the next call rebuilds them instead of reading a half-written one.
    auto bail = [&](const char* what, hipError_t e) {
      (void)hipStreamSynchronize(h->side);
      h->sbus_stale = true;
      h->err = std::string(what) + ": " + hipGetErrorString(e);
      return MAPDN_E_HIP;
    };
    hipError_t e_;
    if ((e_ = hipEventRecord(h->ev_fork, st)) != hipSuccess) return bail("hipEventRecord(fork)", e_);
    if ((e_ = hipStreamWaitEvent(h->side, h->ev_fork, 0)) != hipSuccess) return bail("hipStreamWaitEvent(side, fork)", e_);
    launch_advance(d, add_noise, 1, 0, d.sb_off_alt, h->side);
    if ((e_ = hipEventRecord(h->ev_join, h->side)) != hipSuccess) return bail("hipEventRecord(join)", e_);
    nr_launch(h, MODE_STEP, reward, terminated, info, st);
    if ((e_ = hipStreamWaitEvent(st, h->ev_join, 0)) != hipSuccess) return bail("hipStreamWaitEvent(join)", e_);
    launch_commit_fused(d, st);     // (with fused buses the side stream's profile rows race with this: overlap is an experiment switch)
    launch_advance(d, 0, 0, 1, d.sb_off_alt, st);
  } else {
    nr_launch(h, MODE_STEP, reward, terminated, info, st, fused ? actions : nullptr, actions_dtype);
    launch_commit_fused(d, st);     // (only with fused buses: their own p_mw / q_mvar, before the element tables advance)
    launch_advance(d, add_noise, 1, 1, d.sb_off_alt, st);   // next profile row (-> the Sbus buffer of the next solve) + res_bus commit
  }
  std::swap(h->d.sb_off, h->d.sb_off_alt);
  return MAPDN_OK;
}

int mapdn_reset(mapdn_handle* h, const int64_t* start_rows, int32_t add_noise, int32_t max_tries, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  if (!h->have_profiles) { h->err = "reset before set_profiles"; return MAPDN_E_STATE; }
  if (max_tries < 1) max_tries = 1;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const Dev& d = h->d;
  for (int t = 0; t < max_tries; ++t) {
    launch_reset_begin(d, start_rows, t == 0, st);
    launch_advance(d, add_noise, 1, 0, d.sb_off, st);   // the advance precedes the solve here: it fills the buffer the solve reads
    inject_launch(h, MODE_RESET, nullptr, MAPDN_F64, add_noise, st);
    nr_launch(h, MODE_RESET, nullptr, nullptr, nullptr, st);
    launch_commit_fused(d, st);
    launch_advance(d, 0, 0, 1, d.sb_off, st);    // res_bus commit of the envs that found a solvable start
  }
  HIPCHK(h, hipGetLastError());
  h->was_reset = true;
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_step(mapdn_handle* h, const void* actions, int32_t actions_dtype, int32_t add_noise, double* reward,
               uint8_t* terminated, double* info, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  if (!h->was_reset) { h->err = "step before reset"; return MAPDN_E_STATE; }
  if (!actions || !reward || !terminated || !info) { h->err = "step: null buffer"; return MAPDN_E_INVALID; }
  if (actions_dtype != MAPDN_F32 && actions_dtype != MAPDN_F64) { h->err = "step: bad actions dtype"; return MAPDN_E_INVALID; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const Dev& d = h->d;
  { const int rc = step_launches(h, actions, actions_dtype, add_noise, reward, terminated, info, st); if (rc) return rc; }
  (void)d;
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_step_obs(mapdn_handle* h, const void* actions, int32_t actions_dtype, int32_t add_noise, double* reward,
                   uint8_t* terminated, double* info, void* obs, int32_t obs_dtype, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  if (!h->was_reset) { h->err = "step before reset"; return MAPDN_E_STATE; }
  if (!actions || !reward || !terminated || !info || !obs) { h->err = "step_obs: null buffer"; return MAPDN_E_INVALID; }
  if (actions_dtype != MAPDN_F32 && actions_dtype != MAPDN_F64) { h->err = "step_obs: bad actions dtype"; return MAPDN_E_INVALID; }
  if (obs_dtype != MAPDN_F32 && obs_dtype != MAPDN_F64) { h->err = "step_obs: bad obs dtype"; return MAPDN_E_INVALID; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const Dev& d = h->d;
  const int C = h->plan.n_agents * h->plan.obs_size;
  { const int rc = step_launches(h, actions, actions_dtype, add_noise, reward, terminated, info, st); if (rc) return rc; }
  launch_gather(d, d.gbuf, h->obs_rows, h->obs_scale, 1.0, h->obs_xptr, h->obs_xrow, obs, obs_dtype, C, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_obs(mapdn_handle* h, void* obs, int32_t dtype, void* stream) try {
  if (!h || !obs) return MAPDN_E_INVALID;
  if (!h->was_reset) { h->err = "get_obs before reset"; return MAPDN_E_STATE; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  launch_gather(h->d, h->d.gbuf, h->obs_rows, h->obs_scale, 1.0, h->obs_xptr, h->obs_xrow, obs, dtype,
                h->plan.n_agents * h->plan.obs_size, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_state(mapdn_handle* h, void* state, int32_t dtype, void* stream) try {
  if (!h || !state) return MAPDN_E_INVALID;
  if (!h->was_reset) { h->err = "get_state before reset"; return MAPDN_E_STATE; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  launch_gather(h->d, h->d.gbuf, h->state_rows, h->state_scale, 1.0, nullptr, nullptr, state, dtype, h->plan.state_size, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

static void transpose_out(mapdn_handle* h, const double* src, double scale, const int32_t* rows, double* out, int n, hipStream_t st) {
  launch_gather(h->d, src, rows, nullptr, scale, nullptr, nullptr, out, MAPDN_F64, n, st);
}

int mapdn_get_results(mapdn_handle* h, double* vm_pu, double* va_degree, double* p_mw, double* q_mvar, double* pl_mw,
                      double* sgen_p, double* sgen_q, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  if (!h->was_reset) { h->err = "get_results before reset"; return MAPDN_E_STATE; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const Dev& d = h->d;
  if (vm_pu) transpose_out(h, d.vm, 1.0, h->iota_idx, vm_pu, d.nbo, st);
  if (va_degree) transpose_out(h, d.va, 180.0 / M_PI, h->iota_idx, va_degree, d.nbo, st);
  if (p_mw) transpose_out(h, d.res_p, 1.0, h->iota_idx, p_mw, d.nbo, st);
  if (q_mvar) transpose_out(h, d.res_q, 1.0, h->iota_idx, q_mvar, d.nbo, st);
  if (pl_mw) transpose_out(h, d.pl, 1.0, h->iota_idx, pl_mw, d.n_line, st);
  if (sgen_p) transpose_out(h, d.cur_pv, 1.0, h->iota_idx, sgen_p, d.ns, st);
  if (sgen_q) transpose_out(h, d.cur_q, 1.0, h->iota_idx, sgen_q, d.ns, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_loads(mapdn_handle* h, double* load_p, double* load_q, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  if (load_p) transpose_out(h, h->d.cur_pl, 1.0, h->iota_idx, load_p, h->d.nl, st);
  if (load_q) transpose_out(h, h->d.cur_ql, 1.0, h->iota_idx, load_q, h->d.nl, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_start_rows(mapdn_handle* h, int64_t* start_rows, void* stream) try {
  if (!h || !start_rows) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(start_rows, h->d.start_row, (size_t)h->d.B * sizeof(int64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_auto_reset_mask(mapdn_handle* h, uint8_t* mask, void* stream) try {
  if (!h || !mask) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(mask, h->d.resetting, (size_t)h->d.B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_returns(mapdn_handle* h, double* returns, void* stream) try {
  if (!h || !returns) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(returns, h->d.sum_rewards, (size_t)h->d.B * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_solve_only(mapdn_handle* h, const double* p_load, const double* q_load, const double* p_sgen, const double* q_sgen,
                     double* vm_pu, double* va_degree, int32_t* iterations, uint8_t* converged, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  if (!p_load || !q_load || !p_sgen || !q_sgen) { h->err = "solve_only: null input"; return MAPDN_E_INVALID; }
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const Dev& d = h->d;
  launch_to_envminor(d, p_load, h->t_pl, d.nl, st);
  launch_to_envminor(d, q_load, h->t_ql, d.nl, st);
  launch_to_envminor(d, p_sgen, h->t_pv, d.ns, st);
  launch_to_envminor(d, q_sgen, h->t_q, d.ns, st);
  HIPCHK(h, hipMemsetAsync(d.active, 1, d.B, st));
  if (d.Bp > d.B) HIPCHK(h, hipMemsetAsync(d.active + d.B, 0, d.Bp - d.B, st));
  launch_inject(d, MODE_SOLVE, nullptr, MAPDN_F64, h->t_pl, h->t_ql, h->t_pv, h->t_q, 0, st);
  h->sbus_stale = true;                          // Sbus now holds the caller's loads, not cur_pl / cur_ql
  nr_launch(h, MODE_SOLVE, nullptr, nullptr, nullptr, st);
  if (vm_pu) transpose_out(h, d.nrbuf, 1.0, h->vm_row, vm_pu, d.nbo, st);
  if (va_degree) transpose_out(h, d.nrbuf, 180.0 / M_PI, h->va_row, va_degree, d.nbo, st);
  if (iterations) launch_copy_i32(d.iters, iterations, d.B, st);
  if (converged) launch_copy_u8(d.conv, converged, d.B, st);
  HIPCHK(h, hipGetLastError());
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_profile_stats(const mapdn_handle* h, double* stdv, double* smax) try {
  if (!h) return MAPDN_E_INVALID;
  if (!h->have_profiles) { const_cast<mapdn_handle*>(h)->err = "get_profile_stats before set_profiles"; return MAPDN_E_STATE; }
  if (stdv) std::memcpy(stdv, h->stdv_host.data(), h->stdv_host.size() * sizeof(double));
  if (smax) std::memcpy(smax, h->smax_host.data(), h->smax_host.size() * sizeof(double));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_ybus_dense(const mapdn_handle* h, double* out) try {
  if (!h || !out) return MAPDN_E_INVALID;
  const Plan& P = h->plan;
  for (size_t i = 0; i < P.ybus.size(); ++i) { out[2 * i] = P.ybus[i].real(); out[2 * i + 1] = P.ybus[i].imag(); }
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_obs_index(const mapdn_handle* h, int32_t* kind, int32_t* index) try {
  if (!h || !kind || !index) return MAPDN_E_INVALID;
  std::memcpy(kind, h->plan.obs_kind.data(), h->plan.obs_kind.size() * sizeof(int32_t));
  std::memcpy(index, h->plan.obs_idx.data(), h->plan.obs_idx.size() * sizeof(int32_t));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_nr_geometry(const mapdn_handle* h, int32_t* out) try {
  if (!h || !out) return MAPDN_E_INVALID;
  const mapdn_handle::Geo& g = h->geo;
  const bool fuse_possible = h->solver == 0 && !h->cfg.auto_reset && !knob_int(h->cfg.inject_full, "MAPDN_INJECT_FULL") &&
                             knob_tri(h->cfg.fuse_inject, "MAPDN_FUSE_INJECT") != 2;
  const int32_t v[20] = {h->solver, g.W, g.L, g.lean, g.rows, g.h_lds, g.g_lds, g.rec_lds, g.flat_lds, g.line_lds, g.mm_pass,
                         (int32_t)h->lds_bytes, (int32_t)g.wgs, g.resident, g.rounds, (int32_t)std::min(g.model_ns, 2.0e9),
                         (h->host_only ? fuse_possible : h->fuse_inject) ? 1 : 0, (int32_t)h->plan.fused_obus.size(), h->plan.nb, 0};
  std::memcpy(out, v, sizeof(v));
  if (h->solver == 1) out[2] = h->sp_lanes;
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_flat_factors(const mapdn_handle* h, double* factors, int32_t* bus_of_pos) try {
  if (!h || !factors) return MAPDN_E_INVALID;
  if (!h->plan.radial) return MAPDN_E_TOPOLOGY;   // the tree factorisation exists for radial feeders only
  Schedule S;
  build_schedule(h->plan, 1, S);                 // one worker: row r of the schedule is one node
  const int n = h->plan.n;
  for (int r = 0; r < S.R; ++r) {
    const StepRec& T = S.steps[(size_t)r];
    if (T.flags & S_LIVE) std::memcpy(factors + (size_t)(T.kp & 0xffffu) * FLAT_N, &S.flat[(size_t)r * FLAT_N], FLAT_N * sizeof(double));
  }
  if (bus_of_pos) std::memcpy(bus_of_pos, h->plan.bus_of_pos.data(), (size_t)(n + 1) * sizeof(int32_t));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_schedule(const mapdn_handle* h, int32_t W, int32_t* n_rows, int32_t* rows, int32_t* parent) try {
  if (!h || !n_rows || W < 1 || W > 16) return MAPDN_E_INVALID;
  if (!h->plan.radial) return MAPDN_E_TOPOLOGY;
  Schedule S;
  build_schedule(h->plan, W, S);
  *n_rows = S.R;
  if (rows) for (size_t i = 0; i < S.steps.size(); ++i) rows[i] = (S.steps[i].flags & S_LIVE) ? (int32_t)(S.steps[i].kp & 0xffffu) : -1;
  if (parent) std::memcpy(parent, h->plan.par.data(), h->plan.par.size() * sizeof(int32_t));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_get_sparse_program(const mapdn_handle* h, int32_t S, int32_t* dims, int32_t* ops, int32_t* order, int32_t* slots_ij) try {
  if (!h || !dims || S < 1 || S > 64) return MAPDN_E_INVALID;
  SparseProg G;
  sparse_program(h->plan, S, G);
  dims[0] = G.n_blocks; dims[1] = G.n_fill; dims[2] = G.n_phases; dims[3] = G.rows_per_sub; dims[4] = G.max_nnz;
  dims[5] = (int32_t)(G.slots_ij.size() / 3);
  if (ops) std::memcpy(ops, G.ops.data(), G.ops.size() * sizeof(SpOp));
  if (order) std::memcpy(order, G.order.data(), G.order.size() * sizeof(int32_t));
  if (slots_ij) std::memcpy(slots_ij, G.slots_ij.data(), G.slots_ij.size() * sizeof(int32_t));
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_dense_solve(const double* a, const double* b, double* x, int32_t n, int32_t batch, void* stream) try {
  if (!a || !b || !x) return MAPDN_E_INVALID;
  const int rc = dense_solve_debug(a, b, x, n, batch, (hipStream_t)stream);
  return rc == 0 ? MAPDN_OK : (rc == -1 ? MAPDN_E_INVALID : MAPDN_E_HIP);
} MAPDN_CATCH(nullptr)

int mapdn_stats(mapdn_handle* h, int64_t* reset_failures, double* mean_iters, int32_t* max_iters, void* stream) try {
  if (!h) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  launch_stats(h->d, h->stats_dev, st);
  long long host[4];
  HIPCHK(h, hipMemcpyAsync(host, h->stats_dev, sizeof(host), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (reset_failures) *reset_failures = host[0];
  if (mean_iters) *mean_iters = host[2] ? (double)host[1] / (double)host[2] : 0.0;
  if (max_iters) *max_iters = (int32_t)host[3];
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_nr_timing(mapdn_handle* h, int32_t enable) try {
  if (!h) return MAPDN_E_INVALID;
  h->timing = enable != 0;
  return MAPDN_OK;
} MAPDN_CATCH(h)

int mapdn_nr_time_ms(mapdn_handle* h, double* total_ms, int64_t* launches) try {
  if (!h) return MAPDN_E_INVALID;
  NEEDDEV(h);
  HIPCHK(h, hipSetDevice(h->device));
  double ms = h->acc_ms; int64_t n = h->acc_launches;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    HIPCHK(h, hipEventSynchronize(h->ev[i + 1]));
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
    ms += t; n += 1;
  }
  h->ev_used = 0; h->acc_ms = 0.0; h->acc_launches = 0;
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return MAPDN_OK;
} MAPDN_CATCH(h)

}  // extern "C"
