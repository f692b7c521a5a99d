/*
 * mapdn.h — C ABI of libmapdn_hip.so: batched MI355X (gfx950) implementation of the hot path of
 * Future-Power-Networks/MAPDN's VoltageControl environment.
 *
 * The reference has no FFI: the path sits behind a Python class
 *   VoltageControl(MultiAgentEnv)   environments/var_voltage_control/voltage_control_env.py:24
 * whose numeric work is `pp.runpp(self.powergrid)` (voltage_control_env.py:557, pandapower 2.7.0).
 * Each entry point below names the reference method(s) whose arithmetic it replaces, batched over
 * B independent env instances that share one network topology.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (MAPDN_E_*); mapdn_last_error() gives text;
 *     nothing throws across the boundary.
 *   - `mapdn_handle` is opaque; one handle == one (device, env batch).  The library owns topology
 *     constants, profile tables and workspace; THE CALLER OWNS EVERY I/O BUFFER and passes raw
 *     device pointers (e.g. torch.Tensor.data_ptr()) plus the hipStream_t to enqueue on
 *     (`void* stream`, NULL = default stream).
 *   - step-path calls only enqueue kernels; they never allocate and never synchronise.
 *   - batched I/O tensors are env-major C-contiguous: actions [B, ns], obs [B, n_agents, obs_size],
 *     state [B, state_size], voltages [B, nb], reward [B], terminated [B], info [B, 11].
 *   - one handle is used by one host thread at a time; handles are independent of each other.
 */
#ifndef MAPDN_H
#define MAPDN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPDN_OK 0
#define MAPDN_E_INVALID (-1)   /* bad argument / inconsistent netspec                        */
#define MAPDN_E_TOPOLOGY (-2)  /* net not connected, or meshed and too large for a CU's LDS        */
#define MAPDN_E_HIP (-3)       /* HIP runtime error (no device, OOM, launch failure)           */
#define MAPDN_E_STATE (-4)     /* call order violated (e.g. step before set_profiles/reset)    */
#define MAPDN_E_NOMEM (-5)     /* host allocation failed (std::bad_alloc caught at the boundary)  */
#define MAPDN_E_INTERNAL (-6)  /* any other C++ exception caught at the boundary (text: mapdn_last_error) */

#define MAPDN_N_INFO 11        /* keys of `info`, voltage_control_env.py:586-606,621 — column order:
                                  percentage_of_v_out_of_control, percentage_of_lower_than_lower_v,
                                  percentage_of_higher_than_upper_v, totally_controllable_ratio,
                                  average_voltage_deviation, average_voltage, max_voltage_drop_deviation,
                                  max_voltage_rise_deviation, total_line_loss, q_loss, destroy */

/* voltage_barrier/voltage_barrier_registry.py:9-15 */
enum { MAPDN_BARRIER_L1 = 0, MAPDN_BARRIER_L2 = 1, MAPDN_BARRIER_COURANT_BELTRAMI = 2,
       MAPDN_BARRIER_BOWL = 3, MAPDN_BARRIER_BUMP = 4 };

/* bits of `state_space` (args/env_args/var_voltage_control.yaml:11) */
enum { MAPDN_SS_PV = 1, MAPDN_SS_DEMAND = 2, MAPDN_SS_REACTIVE = 4, MAPDN_SS_VM_PU = 8,
       MAPDN_SS_VA_DEGREE = 16, MAPDN_SS_ALL = 31 };

enum { MAPDN_F32 = 0, MAPDN_F64 = 1 };

/* The columns of the pandapower net (`model.p`, voltage_control_env.py:400-405) that runpp and the
 * env read.  Host pointers, copied during mapdn_create.  Bus ids are 0..n_bus-1. */
typedef struct mapdn_netspec {
  int32_t n_bus;
  const double* bus_vn_kv;          /* [n_bus]                                             */
  const int32_t* bus_zone;          /* [n_bus] 0 = "main", k = "zone{k}"  (bus.zone)       */
  int32_t n_line;                   /* net.line                                            */
  const int32_t* line_from_bus;
  const int32_t* line_to_bus;
  const double* line_r_ohm_per_km;
  const double* line_x_ohm_per_km;
  const double* line_c_nf_per_km;
  const double* line_g_us_per_km;
  const double* line_length_km;
  const int32_t* line_parallel;
  const uint8_t* line_in_service;
  int32_t n_branch_pu;              /* generic per-unit pi branches (e.g. trafos after T->pi) */
  const int32_t* br_from_bus;
  const int32_t* br_to_bus;
  const double* br_r_pu;
  const double* br_x_pu;
  const double* br_b_pu;
  const double* br_ratio;           /* 0 => 1                                              */
  const double* br_shift_deg;
  int32_t n_shunt;                  /* net.shunt (consumer sign, MW/MVAr at 1 p.u.)        */
  const int32_t* shunt_bus;
  const double* shunt_p_mw;
  const double* shunt_q_mvar;
  int32_t n_load;
  const int32_t* load_bus;          /* net.load.bus                                        */
  int32_t n_sgen;
  const int32_t* sgen_bus;          /* net.sgen.bus                                        */
  const int32_t* sgen_zone;         /* net.sgen.name as zone id (voltage_control_env.py:532) */
  int32_t ext_grid_bus;
  double ext_grid_vm_pu;
  double sn_mva;
  double f_hz;
  /* optional columns (NULL = default), appended so that zero-initialised older callers keep working */
  const double* br_g_pu;            /* [n_branch_pu] shunt conductance of the pi branch: ppc BR_B = br_b_pu - 1j*br_g_pu
                                       (a transformer's iron losses after pandapower's T->pi conversion); NULL = 0   */
  const double* load_scaling;       /* [n_load] net.load.scaling * in_service (pd2ppc: PD = sum p_mw * scaling); NULL = 1 */
  const double* sgen_scaling;       /* [n_sgen] net.sgen.scaling * in_service; runpp and res_sgen see p, q * scaling, the env
                                       (obs, q clip, :189) the raw table values; NULL = 1                               */
  const int32_t* bus_alias;         /* [n_bus] bus fusion — closed bus-bus switches (net.switch, et == "b"), which pd2ppc's bus lookup
                                       merges into ONE ppc bus: bus_alias[b] = the representative of b's group (itself for an unfused
                                       bus; representatives represent themselves).  Fused buses share the solved voltage but stay
                                       rows of everything the env reads: res_bus vm_pu / va_degree (equal within a group), their
                                       OWN p_mw / q_mvar (the ext_grid's injection on the ext_grid's own bus), zone frames,
                                       get_state, the reward's averages over all buses.  A branch between two buses of one group
                                       is refused.  NULL = no fusion                                                          */
} mapdn_netspec;

/* Constructor kwargs of VoltageControl (args/env_args/var_voltage_control.yaml:3-20). */
typedef struct mapdn_env_config {
  int32_t barrier_type;             /* MAPDN_BARRIER_*                                     */
  double voltage_weight;
  double q_weight;
  double line_weight;
  int32_t use_line_weight;          /* line_weight != None   (voltage_control_env.py:612)  */
  int32_t use_q_weight;             /* q_weight != None      (voltage_control_env.py:614)  */
  double v_lower, v_upper;
  int32_t episode_limit;
  double action_low, action_high;   /* bias -/+ scale (voltage_control_env.py:76)          */
  int32_t reset_action;
  int32_t state_space;              /* MAPDN_SS_* mask                                     */
  uint64_t seed;
  int64_t env_id_offset;            /* global id of local env 0 (multi-GPU sharding)       */
  int32_t auto_reset;               /* 0: a terminated env stays frozen (reward 0, terminated 1) until mapdn_reset — the
                                       batch analogue of "the caller resets" (models/model.py:204); 1: a terminated env
                                       starts its next episode by itself on the FOLLOWING mapdn_step call: that call
                                       ignores its action, performs reset() for it (sampled start, first profile row,
                                       random initial action; voltage_control_env.py:96-135) and reports reward 0,
                                       terminated 0, info 0 and mapdn_get_auto_reset_mask() == 1; `terminated` is
                                       therefore reported exactly once per episode                                  */
  /* ---- launch / solver tuning.  Appended fields, every one 0 = automatic, so that zero-initialised older callers keep
   * working and two handles in one process can differ (nothing below is process-global).  Results do not depend on the
   * GEOMETRY fields: bus voltages, iteration counts and observations are bit-identical in every geometry; reward / info are
   * sums over buses and lines formed from per-worker partials and agree to the last ulp.  For tools and A/B runs every
   * field has an environment-variable override that is read once, inside mapdn_create (named per field). */
  int32_t nr_solver;                /* 0 auto: tree kernel (k_nr_tree) for radial feeders, sparse block program (k_nr_sparse)
                                       for meshed nets; 1: k_nr_sparse on a radial net too (cross-checks); 2: k_nr_dense, the
                                       dense LU with f64 MFMA trailing updates (any topology; the Jacobian LDS-resident up to 65 buses,
                                       in per-env slabs of global memory up to 513 buses — an exhibit, not a fast path).
                                       env: MAPDN_NR_SPARSE=1 / MAPDN_NR_DENSE=1                                         */
  int32_t nr_waves;                 /* k_nr_tree: waves per workgroup, 0 auto | 1 | 2 | 4              env: MAPDN_NR_WAVES */
  int32_t nr_lanes;                 /* k_nr_tree: envs per workgroup,  0 auto | 4 | 8 | 16 | 32           env: MAPDN_NR_LANES
                                       (a (waves, lanes) pair that is not compiled in — csrc/nr_inst_list.hpp — is refused) */
  int32_t nr_lean;                  /* 0 auto | 1 lean: only voltages + hand-off slots in LDS, several workgroups per CU |
                                       2 fat: whatever fits of h, step records, flat-start constants, net.line constants, G
                                                                                                     env: MAPDN_NR_LEAN=1/0 */
  int32_t nr_h_lds, nr_g_lds, nr_rec_lds, nr_flat_lds, nr_line_lds;
                                    /* LDS residency of the h / G factors, the step records, the flat-start constants, the
                                       net.line constants: 0 auto (whatever fits, in that order) | 1 resident | 2 not resident
                                       env: MAPDN_NR_H_LDS, MAPDN_NR_G_LDS, MAPDN_NR_REC_LDS, MAPDN_NR_FLAT_LDS,
                                       MAPDN_NR_LINE_LDS = 1/0                                                             */
  int32_t nr_mm_pass;               /* the predicted-final mismatch evaluation: 0 auto (barrier-free pass over all nodes when h
                                       is LDS-resident) | 1 pass | 2 mismatch-only tree sweep (same bits) env: MAPDN_NR_MM_PASS */
  int32_t sp_lanes;                 /* k_nr_sparse: envs per workgroup, 0 auto (scored per net) | 16 | 8 | 4 | 2
                                                                                                       env: MAPDN_SP_LANES */
  int32_t inject_full;              /* 1: step() / reset() rebuild Sbus on every bus (k_inject) instead of on the PV buses only
                                       (k_inject_sgen); same bits, for A/B runs and tests               env: MAPDN_INJECT_FULL */
  double nr_check_dx;               /* Newton-step size below which the next sweep is first tried mismatch-only; 0 = 1e-7
                                       (never changes results)                                         env: MAPDN_NR_CHECK_DX */
  double nr_check_quad;             /* safety factor of the second predictor, ||F||^3 / ||F_prev||^2 < tol / factor; 0 = 1,
                                       +inf disables it                                              env: MAPDN_NR_CHECK_QUAD */
  int32_t debug_geometry;           /* 1: print the chosen NR geometry and LDS residency to stderr   env: MAPDN_DEBUG_GEOMETRY */
  /* ---- numerical options of runpp that DO change results (pp.runpp keyword arguments the reference leaves at their
   * defaults, voltage_control_env.py:557) */
  double tolerance_mva;             /* runpp(tolerance_mva=...): 0 = 1e-8 (the pandapower default)                          */
  int32_t tolerance_is_pu;          /* How the stopping rule of newtonpf is formed from tolerance_mva.  0 (default): ||F||inf <
                                       tolerance_mva / sn_mva, F in per unit — the rule as restated in oracle/pp_restated.py from
                                       pandapower 2.7.0 (`ppci_variables` / `_run_newton_raphson_pf`; UNPINNED: pandapower is not
                                       installable here).  1: ||F||inf < tolerance_mva with F in per unit (no division).  The two
                                       coincide on every net with sn_mva == 1 (all MAPDN scenarios and bench nets);
                                       tests/test_pandapower_pin.py::test_tolerance_rule_on_sn_mva_not_one decides which one
                                       pandapower uses wherever pandapower is installed                                    */
  int32_t nr_init;                  /* runpp(init=...): 0 "auto" / "flat" — every solve starts at the slack set-point, what the
                                       reference does (exact pandapower iterates).  1 ("results": start a step() solve from the
                                       env's last accepted voltages, fall back to the flat start after 3 iterations) is RESERVED
                                       and refused (MAPDN_E_INVALID): the study that was to gate it (tools/history/warm_start_study.py,
                                       profiles/r04_warm_start_study_*.json: 3.6 M solves on the oracle, 15 % of them stressed
                                       to and beyond voltage collapse) found it SAFE — never "converged" where the flat start
                                       reports LoadflowNotConverged, never another root, |dV| < 1e-9 — but USELESS for this
                                       kernel: under i.i.d. actions (bench.py's workload) the previous voltages are no closer in
                                       Newton iterations than the flat start (mean 4.12 -> 4.23 iterations on the 141-bus feeder,
                                       4.56 -> 6.34 on the 33-bus one, with the fallback), and under a smooth policy only 43 % /
                                       79 % of the envs (141 / 322 buses) save an iteration, while a workgroup's 16 envs finish
                                       together: P(all 16 save one) ~ 0.  Not built.                                           */
  /* ---- composition of the step() launches (same results; A/B switches) */
  int32_t fuse_inject;              /* step(): 0 auto — the PV-bus injection (_clip_reactive_power, Sbus of the buses with sgens)
                                       runs inside the prologue of k_nr_tree when the handle uses the tree solver and has no
                                       auto_reset (one launch less per step); 1 same, refused (MAPDN_E_INVALID) when impossible;
                                       2 always its own launch (k_inject_sgen)                        env: MAPDN_FUSE_INJECT=1/0 */
  int32_t overlap_advance;          /* step(): 1 — the profile rows of the advance (next row of the tables + noise, independent of
                                       the solve) run on an internal side stream beside the solver launch, fork / join by events;
                                       0 (default) everything on the caller's stream.              env: MAPDN_OVERLAP_ADVANCE=1/0 */
  int32_t xcd_map;                  /* 1: the wide kernels (profile advance + commit, obs gather) serve the envs in the order that keeps an
                                       env on the XCD its solver workgroup runs on (env e of an L-env workgroup i = e / L: XCD i % 8), so
                                       that what one launch writes the next reads from that XCD's L2; 0 (default) / 2: plain order.
                                       Same results; measured a wash (solver -1 us, gather +1 us), so it is opt-in.   env: MAPDN_XCD_MAP */
} mapdn_env_config;

typedef struct mapdn_dims_t {
  int32_t n_envs, n_bus, n_line, n_load, n_sgen, n_agents, n_actions, obs_size, state_size, n_info;
  int32_t is_radial;
  int32_t max_zone_size;
} mapdn_dims_t;

typedef struct mapdn_handle mapdn_handle;

/* last error text: of `h`, or of the last failed mapdn_create when h == NULL */
const char* mapdn_last_error(const mapdn_handle* h);

/* VoltageControl.__init__ up to (not including) data loading — voltage_control_env.py:36-94.
 * Builds per-unit Ybus (pandapower pd2ppc/makeYbus), the elimination order and all gather index
 * tables on the host, uploads them to `device`, allocates state for n_envs envs.
 * Topology: pp.runpp (voltage_control_env.py:557) solves any connected net.  Radial feeders (all three MAPDN
 * scenarios) take the fill-free tree solver; a meshed net (closed tie switches, loops) takes the general sparse solver:
 * symbolic factorisation with fill on the host, the numeric part as a program of 2x2 block operations interpreted with
 * all blocks of a few envs in LDS (nets up to ~2400 blocks after fill: the 322-bus case with tie lines fits).
 * Nets beyond that and nets with buses not connected to the ext_grid return MAPDN_E_TOPOLOGY.
 * device == -1 builds a host-only handle (plan, dims, ybus/obs-index export; no device calls).
 * Launch geometry and solver choice: automatic per topology and batch size; mapdn_env_config's tuning fields (and, for tools,
 * their environment-variable overrides) select them explicitly.  mapdn_get_nr_geometry() reports what was chosen. */
int mapdn_create(const mapdn_netspec* net, const mapdn_env_config* cfg, int32_t n_envs,
                 int32_t device, mapdn_handle** out);
void mapdn_destroy(mapdn_handle* h);
int mapdn_dims(const mapdn_handle* h, mapdn_dims_t* out);

/* _load_pv_data/_load_active_demand_data/_load_reactive_demand_data (voltage_control_env.py:407-438)
 * after CSV parsing and *_scale: HOST row-major tables pv [T, ns], load_p [T, nl], load_q [T, nl].
 * Also derives the per-column std/100 (:70-72) and s_max = 1.2*max (:515-520).  Synchronous. */
int mapdn_set_profiles(mapdn_handle* h, const double* pv, const double* load_p, const double* load_q,
                       int64_t n_rows, int32_t time_delta_min, int32_t days);

/* Host-side export of what mapdn_set_profiles derived from the tables (parity tests): stdv [n_sgen + 2 n_load] = the per-column noise
 * scale `values.std(axis=0) / 100.0` in table order pv | load_p | load_q (voltage_control_env.py:70-72, numpy's own summation order:
 * csrc/colstats.hpp), smax [n_sgen] = 1.2 * max_t pv (:518-520).  HOST pointers; either may be NULL. */
int mapdn_get_profile_stats(const mapdn_handle* h, double* stdv, double* smax);

/* reset() / manual_reset() — voltage_control_env.py:96-176.  start_rows: device int64 [B] giving
 * `start` of :445 per env, or NULL to sample (hour, day, interval) per env from the keyed RNG.
 * Retries unsolvable initial states up to `max_tries` times (reference: unbounded loop, :108). */
int mapdn_reset(mapdn_handle* h, const int64_t* start_rows, int32_t add_noise, int32_t max_tries,
                void* stream);

/* step(actions, add_noise) — voltage_control_env.py:178-211 (= _take_action :548-566 with the
 * pandapower runpp inside, _calc_reward :574-623, _set_demand_and_pv :491-513).
 * actions: device [B, ns] of actions_dtype; reward f64 [B]; terminated u8 [B]; info f64 [B, 11].
 * Envs already terminated are frozen: reward 0, terminated 1, info unchanged (zeros). */
int mapdn_step(mapdn_handle* h, const void* actions, int32_t actions_dtype, int32_t add_noise,
               double* reward, uint8_t* terminated, double* info, void* stream);

/* step() immediately followed by get_obs(), the pair every caller of the reference issues (models/model.py:216,219;
 * utilities/tester.py:48-49) as one call: the same four launches as mapdn_step + mapdn_get_obs, one host entry.
 * (A single fused post-solve kernel — commit + profile advance + gather out of an LDS tile — was built and measured:
 * 29.5 us against 26.3 us for the two wide launches on case141 x 4096; not kept, see DESIGN.md.)
 * obs: device [B, n_agents, obs_size] of obs_dtype. */
int mapdn_step_obs(mapdn_handle* h, const void* actions, int32_t actions_dtype, int32_t add_noise,
                   double* reward, uint8_t* terminated, double* info, void* obs, int32_t obs_dtype, void* stream);

/* get_obs() — voltage_control_env.py:232-316 (distributed mode): [B, n_agents, obs_size]. */
int mapdn_get_obs(mapdn_handle* h, void* obs, int32_t dtype, void* stream);
/* get_state() — voltage_control_env.py:213-230: [B, state_size]. */
int mapdn_get_state(mapdn_handle* h, void* state, int32_t dtype, void* stream);

/* tester getters — voltage_control_env.py:625-647; any pointer may be NULL.
 * vm_pu, va_degree, p_mw, q_mvar: f64 [B, nb]; pl_mw: f64 [B, n_line]; sgen_p, sgen_q: f64 [B, ns] */
int mapdn_get_results(mapdn_handle* h, double* vm_pu, double* va_degree, double* p_mw, double* q_mvar,
                      double* pl_mw, double* sgen_p, double* sgen_q, void* stream);
/* current load table values f64 [B, nl] (net.load.p_mw / q_mvar) */
int mapdn_get_loads(mapdn_handle* h, double* load_p, double* load_q, void* stream);

/* `start` of voltage_control_env.py:445 for every env: device int64 [B] */
int mapdn_get_start_rows(mapdn_handle* h, int64_t* start_rows, void* stream);

/* sum_rewards of the running episode (voltage_control_env.py:203) for every env: device f64 [B] */
/* uint8 [n_envs]: 1 for the envs that the last mapdn_step call (re)started (auto_reset == 1), else 0.  An env whose restart
 * power flow does not solve is NOT reported here: it stays frozen (reward 0, terminated 1) and draws a new start on the next
 * call, like the reference's `while not solvable` loop (voltage_control_env.py:108); until then its load / PV tables already
 * hold the rejected start's values while res_bus still holds the previous episode's voltages. */
int mapdn_get_auto_reset_mask(mapdn_handle* h, uint8_t* mask, void* stream);

int mapdn_get_returns(mapdn_handle* h, double* returns, void* stream);

/* Pure power flow == pp.runpp(net) (voltage_control_env.py:557) on explicit element powers:
 * p_load, q_load f64 [B, nl]; p_sgen, q_sgen f64 [B, ns] (MW / MVAr) ->
 * vm_pu, va_degree f64 [B, nb]; iterations i32 [B]; converged u8 [B].  Does not touch env state. */
int mapdn_solve_only(mapdn_handle* h, const double* p_load, const double* q_load, const double* p_sgen,
                     const double* q_sgen, double* vm_pu, double* va_degree, int32_t* iterations,
                     uint8_t* converged, void* stream);

/* Host-side debug export of the per-unit admittance matrix the library built (for parity tests):
 * dense row-major complex [nb, nb] as (re, im) pairs (with fused buses — bus_alias — over the merged nodes: nb = the number of
 * representatives, in ascending order of their bus index). */
int mapdn_get_ybus_dense(const mapdn_handle* h, double* ybus_re_im);
/* Host-side export of the integer gather tables of get_obs: kind/index per obs column
 * [n_agents*obs_size] (kinds: 0 zero pad, 1 p_mw, 2 q_mvar, 3 pv, 4 q, 5 vm_pu, 6 va rad). */
int mapdn_get_obs_index(const mapdn_handle* h, int32_t* kind, int32_t* index);

/* Host-side export of the NR elimination schedule for W cooperating waves (plan check, CPU tests):
 * rows [W * (*n_rows)] node position per (wave, row) or -1, parent [n] parent position per node
 * (n = n_bus-1 means the slack).  Call with rows == NULL to query *n_rows first. */
int mapdn_get_schedule(const mapdn_handle* h, int32_t n_waves, int32_t* n_rows, int32_t* rows, int32_t* parent);

/* Host-side export of the launch geometry mapdn_create settled on (also for device == -1 handles, which assume a 256-CU
 * device): out[20] = solver (0 tree, 1 sparse, 2 dense), waves, lanes (envs per workgroup), lean, schedule rows, h_lds, g_lds,
 * rec_lds, flat_lds, line_lds, mm_pass, dynamic LDS bytes per workgroup, workgroups, workgroups resident per CU (model),
 * rounds of workgroups, modelled launch time in ns (the chooser's score; 0 when the geometry was forced), fuse_inject (1: step()
 * performs the PV-bus injection in the solver's prologue), number of buses in fused groups (bus_alias), electrical nodes, 0. */
int mapdn_get_nr_geometry(const mapdn_handle* h, int32_t* out20);

/* Host-side export of the flat-start factorisation the NR kernel's first iteration uses (plan check, CPU tests):
 * factors [n][12] per elimination position = S_calc (re, im), D^-1 (4, row-major), A_pk (re, im), G (4, row-major)
 * of the block LU of the Jacobian at V = ext_grid vm_pu everywhere, unknowns [dtheta, d|V|/|V|];
 * bus_of_pos [n+1] (optional) = bus id of every elimination position (position n is the slack). */
int mapdn_get_flat_factors(const mapdn_handle* h, double* factors, int32_t* bus_of_pos);

/* Host-side export of the general sparse solver's elimination PROGRAM for S sub-lanes (plan check, CPU tests): dims [6] =
 * block slots, fill-only blocks, phases, assembly rows per sub-lane, max row entries, number of off-diagonal blocks;
 * ops [phases * S * 4] = (type, c, a, b) with (type & 255) 0 NOP, 1 B[c] = inv(B[a]), 2 B[c] = B[a] B[b], 3 B[c] -= B[a] B[b]
 * (bits 8 / 9 of type: wave-uniform hints "this phase holds an INV / an UPD")
 * (block slots: diagonal of node i = i, right-hand side of node i = n + i as the block [b | 0]); order [n] = minimum-degree
 * elimination order (positions); slots_ij [3 * dims[5]] = (i, j, slot) of every off-diagonal block incl. fill.  Executing the
 * ops in phase order on the assembled blocks solves J x = b — tests/test_general_topology.py does it in numpy against
 * scipy's spsolve (what pandapower calls in pypower/newtonpf.py).  Any pointer but dims may be NULL. */
int mapdn_get_sparse_program(const mapdn_handle* h, int32_t sub_lanes, int32_t* dims, int32_t* ops, int32_t* order, int32_t* slots_ij);

/* Debug / pin export of the general solver's linear algebra: solves `batch` independent dense systems A x = b
 * (device pointers; A row-major [batch, n, n], b and x [batch, n]; n even, <= 1024; LDS-resident up to n = 128, in global scratch —
 * synchronous — beyond) with the blocked LU
 * (2x2 block pivots, v_mfma_f64_16x16x4_f64 trailing updates) that k_nr_dense runs on the Jacobian — what pandapower
 * does with SuperLU in pypower/newtonpf.py (dx = -spsolve(J, F)).  Tests compare with numpy.linalg.solve. */
int mapdn_dense_solve(const double* a, const double* b, double* x, int32_t n, int32_t batch, void* stream);

/* Rollout-side forward of the shared-parameter recurrent agent (agents/rnn_agent.py:5-32 as called by
 * models/model.py:101-139): fc1 (+ one-hot agent-id column) -> LayerNorm -> ReLU -> GRUCell -> fc2, one launch, inference only.
 * Device pointers, fp32, contiguous: obs [rows, obs_dim] (rows = envs x agents, agent = row % n_agents), hid_in / hid_out
 * [rows, 64], w1 [64, obs_dim + id_dim] (id_dim = n_agents or 0), w_ih / w_hh [192, 64] (torch.nn.GRUCell layout: r, z, n),
 * w2 [1, 64]; means [rows].  Hidden size 64, action_dim 1 (the reference's defaults).  hid_out may be NULL (the new hidden state is not stored). */
int mapdn_policy_forward(const float* obs, const float* hid_in, const float* w1, const float* b1, const float* ln_g,
                         const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                         const float* w2, const float* b2, float* means, float* hid_out, int32_t rows, int32_t n_agents,
                         int32_t obs_dim, int32_t id_dim, float ln_eps, void* stream);

/* The same launch for the learner's TRAINING-time forward (models/maddpg.py:103-125 through agents/rnn_agent.py:16-32): also writes
 * x1 [rows, 64] = fc1(obs) + b1 + id column (the LayerNorm input, all the backward keeps); hid_out may be NULL (not stored). */
int mapdn_policy_forward_train(const float* obs, const float* hid_in, const float* w1, const float* b1, const float* ln_g,
                               const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                               const float* w2, const float* b2, float* means, float* hid_out, float* x1_out, int32_t rows,
                               int32_t n_agents, int32_t obs_dim, int32_t id_dim, float ln_eps, void* stream);
/* Backward of the agent's trunk LayerNorm -> ReLU -> GRUCell -> fc2 (agents/rnn_agent.py:16-32) for d loss / d means [rows], one launch
 * (csrc/policy_bwd.hip) that recomputes the trunk from x1 and hid_in and writes what the remaining weight-gradient products need:
 *   dx1 [rows, 64] = d loss / d x1 (-> fc1: dW1 = dx1^T [obs | id], db1);
 *   dgates [rows, 256] = d loss / d gate pre-activations  r | z | n_input | n_hidden  (torch.nn.GRUCell: dW_ih = dgates[:, 0:192]^T xn,
 *                        dW_hh = dgates[:, [0:128, 192:256]]^T hid_in);  xn [rows, 64] = relu(LayerNorm(x1));
 *   small [512] = db_r | db_z | db_ni | db_nh | dgamma | dbeta | dw2 (64 each) | db2 (1) + pad, reduced in a fixed order (no atomics):
 *                 b_ih gradient = db_r | db_z | db_ni, b_hh gradient = db_r | db_z | db_nh;
 *   scratch: mapdn_policy_backward_scratch_floats(rows) floats.  Hidden size 64, one action output; fp32, contiguous device pointers. */
int64_t mapdn_policy_backward_scratch_floats(int64_t rows);
int mapdn_policy_backward(const float* dmeans, const float* x1, const float* hid_in, const float* ln_g, const float* ln_b, float ln_eps,
                          const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* w2, float* dx1,
                          float* dgates, float* xn, float* small, float* scratch, int64_t rows, void* stream);

/* 1 when mapdn_policy_forward has a launch shape for this observation width (the parameter set and a tile's activations must
 * fit the 160 KB LDS of a CU: obs_dim up to ~2 400 columns without ids), else 0 — callers then keep their own forward. */
int mapdn_policy_forward_fits(int32_t obs_dim, int32_t id_dim);

/* LayerNorm over 64 features (+ fused ReLU when relu != 0) for the learner's training-time passes (agents/rnn_agent.py:16-21,
 * critics/mlp_critic.py:22-27: fc1 -> LayerNorm -> ReLU), forward and backward; device pointers, fp32, contiguous [rows, 64].
 * forward also writes the row statistics (mean, rstd [rows]) the backward needs.  backward: dx [rows, 64], dgamma, dbeta [64];
 * `partial` is scratch of mapdn_layernorm64_backward_blocks(rows) x 128 floats (block partial sums, reduced in a fixed order:
 * deterministic, no atomics). */
int mapdn_layernorm64_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              int64_t rows, float eps, int32_t relu, void* stream);
int mapdn_layernorm64_backward_blocks(int64_t rows);
int mapdn_layernorm64_backward(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, float* dx, float* dgamma, float* dbeta, float* partial, int64_t rows,
                               int32_t relu, void* stream);
/* ... with every input row FORMED as base[row / n] + per_n[row % n] instead of read (base [rows / n][64], per_n [n][64], rows a multiple
 * of n): the central critic's first layer feeds the LayerNorm  W_obs·obs_all + b  per batch element plus the id column of the agent
 * (critics/mlp_critic.py:22-27 behind models/maddpg.py:35-79), and the [batch, agents, 64] tensor of their sum is never materialised.
 * _bc_backward returns dx per formed row; the caller reduces it over the agents / over the batch. */
int mapdn_layernorm64_bc_forward(const float* base, const float* per_n, int32_t n, const float* gamma, const float* beta, float* y,
                                 float* mean, float* rstd, int64_t rows, float eps, int32_t relu, void* stream);
int mapdn_layernorm64_bc_backward(const float* dy, const float* base, const float* per_n, int32_t n, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, float* partial,
                                  int64_t rows, int32_t relu, void* stream);
/* The critic's last two steps on rows of 64 as one pass: v[row] = relu(pre[row, :]) . w + bias  (critics/mlp_critic.py:22-36: the
 * activation after fc2, then fc3 with one output) — relu(pre) is not materialised.  Backward: dpre = [pre > 0] dv w; dw [64]; db [64]
 * with the bias gradient in db[0]; `partial` = mapdn_layernorm64_backward_blocks(rows) x 128 floats of scratch (fixed-order reduction). */
int mapdn_relu_dot64_forward(const float* pre, const float* w, float bias, float* v, int64_t rows, void* stream);
int mapdn_relu_dot64_backward(const float* dv, const float* pre, const float* w, float* dpre, float* dw, float* db, float* partial,
                              int64_t rows, void* stream);

/* The critic trunk behind its first layer as ONE launch each way (critics/mlp_critic.py:22-36 as trained by
 * learning_algorithms/ddpg.py:15-39 / models/maddpg.py:103-125):   v[row] = relu( relu(LayerNorm(x[row])) W2^T + b2 ) . w3 + b3
 * on rows of 64, fp32 on v_mfma_f32_16x16x4_f32 (csrc/critic.hip).  x is read ([rows][64], per_n == NULL) or FORMED as
 * base[row / n] + per_n[row % n] (x = base [rows / n][64], per_n [n][64], rows a multiple of n, n <= 256: the central critic).
 * Device pointers, contiguous; gamma, beta, b2, w3 [64], w2 [64][64] (out, in), b3 [1]; rows < 2^31.
 * backward recomputes the forward from the same inputs (nothing is saved but them):
 *   dx      [rows][64]                       (x read)   — or dbase [rows / n][64] = sum of dx over every group of n rows (x formed);
 *   grads   [4416 (+ n * 64)] when param_grads != 0 or x is formed:  dW2 [64][64] | dgamma | dbeta | db2 | dw3 [64 each] | db3 [1] + pad to
 *           4416 | dper_n [n][64] (formed rows only; with param_grads == 0 only dper_n is written);
 *   scratch mapdn_critic_head_scratch_floats(rows, n, formed) floats (per-workgroup partial sums — the wavefronts of a workgroup are summed through LDS —, reduced in a fixed order: deterministic).
 * _backward_dot: only dact[row] = dx[row] . dot_w[row % n] (dot_w [n][64]; n = 1 when x is read) — the policy update through the
 * central critic, whose own-action column of fc1 (models/maddpg.py:52-58) is the only gradient path back to the policy. */
int mapdn_critic_head_forward(const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta, float eps,
                              const float* w2, const float* b2, const float* w3, const float* b3, float* v, int64_t rows, void* stream);
int64_t mapdn_critic_head_scratch_floats(int64_t rows, int32_t n, int32_t formed);
int mapdn_critic_head_backward(const float* dv, const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta,
                               float eps, const float* w2, const float* b2, const float* w3, const float* b3, float* dx, float* grads,
                               float* scratch, int64_t rows, int32_t param_grads, void* stream);
/* the value loss of the DDPG family (learning_algorithms/ddpg.py:36-38) and every gradient of it in ONE launch, without a forward:
 * loss = sum_rows scale[0] * wrow[row / n] * (ret[row] - v[row])^2 (wrow NULL: 1; wrow [rows / n], or [rows] when x is read) -> grads[4353];
 * dx / dbase and grads as mapdn_critic_head_backward(param_grads = 1) for dv = d loss / d v. */
int mapdn_critic_head_mse(const float* ret, const float* wrow, const float* scale, const float* x, const float* per_n, int32_t n,
                          const float* gamma, const float* beta, float eps, const float* w2, const float* b2, const float* w3,
                          const float* b3, float* dx, float* grads, float* scratch, int64_t rows, void* stream);
int mapdn_critic_head_backward_dot(const float* dv, const float* x, const float* per_n, int32_t n, const float* gamma, const float* beta,
                                   float eps, const float* w2, const float* b2, const float* w3, const float* b3, const float* dot_w,
                                   float* dact, int64_t rows, void* stream);

/* The glue of one batched rollout step (models/model.py:197-262) as three launches instead of ~45 one-line PyTorch kernels
 * (csrc/rollout.hip).  Device pointers, contiguous.
 * mapdn_explore_actions: action = tanh(mean + std * eps) (utilities/util.py:57-66; no tanh when tanh_bound == 0), action_pol =
 *   (avail != 0) * action (maddpg.py:92-93; avail / action_pol may be NULL), actual = translate_action(action) = 0.5 (clamp(action, -1,
 *   1) + 1) (high - low) + low with low / high = bias -/+ scale (utilities/util.py:123-132); f32, the PyTorch chain's operations in
 *   its order, bit-identical.
 * mapdn_rollout_stats: sums[0..10] += sum over live envs of info[e][k], sums[11] += of reward[e], sums[12] += live envs; then
 *   alive_out[e] = alive[e] & !done[e] (models/model.py:243-248, 225; alive_out may be alive); info [n_envs][11], reward [n_envs] f64; alive / done one byte per env.
 * mapdn_copy_segments: dst[i][0 .. nbytes[i]) = src[i][...] for up to 48 segments in one launch (the fields of a transition into
 *   their replay-ring positions, utilities/replay_buffer.py:25-29); HOST arrays of device pointers; 16-byte aligned, nbytes % 16 == 0. */
int mapdn_explore_actions(const float* mean, const float* eps, const float* avail, float stdv, int32_t tanh_bound, double action_scale,
                          double action_bias, float* action, float* action_pol, float* actual, int64_t n, void* stream);
int mapdn_rollout_stats(const double* info, const double* reward, const uint8_t* alive, const uint8_t* done, uint8_t* alive_out, double* sums,
                        int32_t n_envs, void* stream);
int mapdn_copy_segments(const void* const* src, void* const* dst, const int64_t* nbytes, int32_t n_segments, void* stream);

/* Calibration aid for the HBM counters (tools/calibrate_traffic.py): copies rows x Bp x 16 bytes from src to dst (device pointers) with
 * the solver's own global access pattern — raw-buffer 16-byte loads / stores of env-minor pair rows, 256 contiguous bytes per
 * 16-lane worker (pattern 0) — or with whole waves on one row (pattern 1), so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be read
 * against a known byte count.  Bp a multiple of 256, rows x Bp x 16 < 4 GiB. */
int mapdn_debug_stream(const double* src, double* dst, int32_t rows, int32_t Bp, int32_t pattern, void* stream);

/* "MAPDN_SRC_HASH=<hex>": sha256 of the sources + flags the library was built from (mapdn_amd/build.py::source_hash).  The Python
 * loader refuses (or rebuilds) a library whose hash differs from the sources beside it.  No reference counterpart: build hygiene. */
const char* mapdn_build_info(void);

/* counters (host, synchronises the given stream): number of envs whose last reset exhausted
 * max_tries; mean / max NR iterations of the last solve */
int mapdn_stats(mapdn_handle* h, int64_t* reset_failures, double* mean_iters, int32_t* max_iters,
                void* stream);

/* seconds-resolution device timing of the dominant kernel for bench.py: enables hipEvent pairs
 * around every NR-solve launch on its stream; mapdn_nr_time_ms returns the accumulated time and
 * launch count since the last call and resets them (synchronises). */
int mapdn_nr_timing(mapdn_handle* h, int32_t enable);
int mapdn_nr_time_ms(mapdn_handle* h, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MAPDN_H */
