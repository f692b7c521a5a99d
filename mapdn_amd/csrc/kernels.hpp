// kernels.hpp — device-visible argument block + host launchers of kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"

namespace mapdn {

enum { MODE_STEP = 0, MODE_RESET = 1, MODE_SOLVE = 2 };
enum { STREAM_PV = 0, STREAM_LOAD_P = 1, STREAM_LOAD_Q = 2, STREAM_ACTION = 3, STREAM_START = 4 };

// NR global scratch `nrbuf` (env-minor), three regions:
//   factor blocks  [n+2][NBP] pair rows (Bp x 16 bytes)  one block per NODE (position; n+1 = the trash node every idle
//                                step works on): the LU factors (G0,G1) (G2,G3) (h0,h1) written in the forward sweep and
//                                read back in the backward sweep — only for the factors that do not fit in LDS
//                                (nr_h_lds / nr_g_lds)
//   Sbus           2 x [nblk] pair rows   (Re, Im) of the scheduled injection, stored in SCHEDULE order (k_inject writes
//                                entry sb_index[k]; the NR workers prefetch theirs by (worker,row)); idle steps stay 0;
//                                two buffers, flipped by the host after every step (k_advance fills the next one)
//   Vout           [n+1][4] rows of Bp doubles   per position (n == slack): e f |V| angle — the solution (k_nr_tree)
// Voltages and everything that crosses workers live in LDS during the solve.
enum { NB_G01 = 0, NB_G23, NB_H, NBP };
enum { VO_E = 0, VO_F, VO_VM, VO_VA, VOF };

// Everything a kernel needs, passed by value (kernarg segment -> scalar loads).
// Layout rule: per-env arrays are env-minor, X[item][Bp]; Bp = B rounded up to 64.
struct Dev {
  int32_t B, Bp, nb, n, nl, ns, n_line, ncol;
  double vroot, sn, tol;
  int32_t max_it;
  // ---- topology plan (shared by all envs; wave-uniform reads)
  const int32_t* bus_of_pos;
  const int32_t *load_ptr, *load_idx, *sgen_ptr, *sgen_idx;
  const double *shunt_p, *shunt_q;
  const double *load_scale, *sgen_scale;     // [nl], [ns] element scaling * in_service (runpp sees p, q * scaling)
  // buses with sgens ("PV buses", n_sgb of them: positions sgb_pos, inverse sgb_of_pos[nb] or -1) and buses with loads but no
  // sgens (lb_pos, n_lb): k_inject_sgen works on the former; bus_ld [n_sgb][Bp] pairs = load part (P, Q) of their injection
  const int32_t *sgb_pos, *sgb_of_pos, *lb_pos, *mlo_pos, *ld_dest; int32_t n_sgb, n_lb, n_mlo;   // mlo_pos: buses with several loads and no sgens;
                                                   // ld_dest[nl]: where a load that is alone on its bus goes (see k_advance)
  double* bus_ld;
  const LineFlow* lines;
  const int32_t* root_children; const double* root_y; int32_t n_root_children;   // children of the slack: position, Y_root,k
  double yrr0, yrr1;
  // ---- profile tables: [T][ncol], columns = pv | load_p | load_q
  const double* table; const double* stdv; const double* smax;
  int64_t T; int32_t n_start_days, per_hour, per_day;
  // ---- config
  int32_t barrier_type, use_line_weight, episode_limit, reset_action, auto_reset;
  double voltage_weight, q_weight, line_weight, v_lower, v_upper, action_low, action_high;
  uint32_t seed_lo, seed_hi; int64_t env_id_offset;
  // ---- env state
  double *cur_pv, *cur_q, *cur_pl, *cur_ql, *q_new;   // [ns|nl][Bp]  MW / MVAr
  // gatherable state lives in ONE block `gbuf` (rows of Bp doubles) so obs/state columns are plain row
  // numbers: cur_pv cur_q [ns] | vm va res_p res_q [nb]   (pointers below alias into it)
  double* gbuf;
  double *vm, *va, *res_p, *res_q;                    // [nb][Bp] by bus id; va in rad
  double *pl;                                         // [n_line][Bp] res_line.pl_mw
  double *sum_rewards;                                // [Bp]
  int32_t* steps; int64_t* start_row; uint32_t* draw;
  uint8_t *done, *pending, *active, *commit, *bad_start, *resetting;   // resetting: (re)started by this step call (auto_reset)
  int64_t* adv_row; uint32_t* adv_draw;
  // ---- NR scratch (see NB_* / VO_*): row offsets of the Sbus and Vout regions
  double* nrbuf; uint32_t nrbuf_bytes; uint32_t sb_off, r_vout;   // sb_off: byte offset of the Sbus region the solve reads
  uint32_t sb_off_alt;                                            // ... of the other Sbus buffer (double-buffered, see k_advance)
  const int32_t* sb_index;                                           // [n] Sbus entry (schedule step) of elimination position k
  int32_t* iters; uint8_t* conv;
  // ---- NR schedule (k_nr_tree): W waves per workgroup, L envs per workgroup (64/L lane-group workers per wave), R rows
  int32_t nr_waves, nr_lanes, nr_rows, nr_cslots, nr_xslots, nr_nclist, nr_h_lds, nr_g_lds, nr_line_lds, nr_rec_lds, nr_flat_lds;
  const StepRec* sched; uint32_t sched_bytes; const int32_t* clist;
  const double* flat; uint32_t flat_bytes;   // Schedule::flat, [Wt][R][FLAT_N]
  // mismatch pass (k_nr_tree): per (worker, turn) a record with the node, its parent, its first three children in canonical
  // order, its Sbus entry and its Y constants (Schedule::mm_recs); further children of a junction from mm_ptr / mm_child
  const int32_t *mm_ptr, *mm_child; int32_t nr_mm_pass, mm_np;
  const StepRec* mm_recs; uint32_t mm_recs_bytes;   // Schedule::mm_recs
  // a Newton step whose largest component (|dtheta|, |d|V|/|V||) is below this predicts convergence: the next
  // forward sweep is first run in its mismatch-only form (k_nr_tree)
  double nr_check_dx;
  double nr_check_quad;      // safety factor of the quadratic-convergence predictor (inf disables it, tiny = always predict)
  // ---- general-topology solve (k_nr_dense, dense.hip): Ybus rows by position in CSR form (columns are positions,
  // n == slack; values (re, im)); the dense Jacobian is N x N (2n rounded up to 16) with row stride dn_lda in LDS
  // ---- general sparse path (k_nr_sparse, sparse.hip): host-compiled elimination program (plan.hpp SparseProg)
  int32_t sparse, sp_lanes, sp_blocks, sp_fill, sp_phases, sp_rows_per_sub, sp_max_nnz;
  const SpOp* sp_ops; uint32_t sp_ops_bytes; const SpNz* sp_nz; uint32_t sp_nz_bytes; const int32_t* sp_fill_slots;
  int32_t dense, dn_N, dn_lda;
  double* dn_A;              // k_nr_dense beyond 65 buses: per-env slabs [Bp][dn_N][dn_lda] of global memory for the Jacobian (else nullptr: LDS)
  const int32_t *gy_ptr, *gy_col; const double* gy_val;
  // ---- XCD-aligned env order of the wide kernels (0 = off: block b serves envs 256 b ...): a k_nr_tree workgroup i serves envs
  // [L i, L i + L) and lands on XCD i % 8 (observed), so env e "lives" on XCD (e / L) % 8; with xcd_lanes = L the wide kernels give
  // block b the L-env groups of XCD b % 8, and what one kernel writes the next reads from the same XCD's L2 (the L2s of the eight XCDs
  // are not coherent with each other: a line written on another XCD comes back from memory)
  int32_t xcd_lanes;
  // ---- bus fusion (plan.hpp): original buses, nbo of them, vs electrical nodes (nb); all nullptr / 0 / nbo == nb without fusion
  int32_t nbo, n_fused, n_alias, n_slack_group;
  const int32_t *pos_of_obus, *cm_kind, *fused_obus, *ob_load_ptr, *ob_load_idx, *ob_sgen_ptr, *ob_sgen_idx, *slack_group, *alias_pos;
  const double *ob_shunt_p, *ob_shunt_q;
  // ---- PV-bus injection fused into the k_nr_tree prologue (step(), handles without auto_reset): per launch, set by launch_nr.
  // sgb_rec [n_sgb + n_mlo][8] = Sbus entry of the bus (-1: slack bus, q only) | elimination position | first sgen on the bus (-1:
  // a load-only bus with several loads) | (number of sgens << 8) | min(number of loads, 2) || first load | second load | 0 | 0
  const int32_t* sgb_rec;
  const void* fi_actions; int32_t fi_dtype;      // actions [B, ns] of MAPDN_F32 / MAPDN_F64; nullptr: the injection ran as its own launch
};

void launch_inject(const Dev& d, int mode, const void* actions, int dtype, const double* pl, const double* ql,
                   const double* pv, const double* q, int add_noise, hipStream_t st);
// step()/reset() form of the injection: PV buses only (the rest of Sbus is kept up to date by k_advance)
void launch_inject_sgen(const Dev& d, int mode, const void* actions, int dtype, int add_noise, hipStream_t st);
// fused_actions != nullptr (MODE_STEP only): k_nr_tree's prologue performs the PV-bus injection itself (no k_inject_sgen launch)
void launch_nr(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st,
               const void* fused_actions = nullptr, int fused_dtype = 0);
int nr_set_lds_limit(int waves, int lanes, int h_lds, int g_lds, int rec_lds, int flat_lds, size_t bytes);   // -2: geometry not instantiated
int nr_geometry_compiled(int waves, int lanes, int h_lds, int g_lds, int rec_lds, int flat_lds);
// dynamic LDS of k_nr_tree (W waves, L envs per workgroup => Wt = W*64/L workers), in pair rows of L x 16 bytes:
// node voltages (n+2: nodes, slack, trash), h (n+2) and G (2(n+2)) when resident, contribution slots (4 rows each),
// x slots (1 row each); then verdict bytes, step-size partials (64*W doubles), overflow child list (padded to
// 16 bytes), and — when they fit — the LineFlow constants of net.line for the fused res_line epilogue
__host__ __device__ static inline size_t nr_line_bytes(int n_line) { return ((size_t)n_line * sizeof(LineFlow) + 15) & ~(size_t)15; }
// The epilogue's partial sums (10 x 64*W doubles) re-use the contribution slots, so cslots >= nr_min_cslots(W, L).
static inline int nr_min_cslots(int W, int L) { return (10 * 64 * W + 8 * L - 1) / (8 * L); }
// rec_rows / flat_rows: R when the step records / flat-start constants of all Wt workers are staged in LDS, else 0
static inline size_t nr_lds_bytes(int W, int L, int n, int cslots, int xslots, int nclist, int h_lds, int g_lds, int n_line_lds,
                                  int rec_rows, int flat_rows) {
  const size_t rows = (size_t)(n + 2) * (1 + (h_lds ? 1 : 0) + (g_lds ? 2 : 0)) + (size_t)cslots * 4 + (size_t)xslots;
  const size_t Wt = (size_t)W * (64 / L);
  return rows * (size_t)L * 16 + (size_t)W * 64 + (size_t)64 * W * sizeof(double) + (size_t)((nclist + 3) & ~3) * sizeof(int32_t) +
         nr_line_bytes(n_line_lds) + Wt * rec_rows * sizeof(StepRec) + Wt * flat_rows * FLAT_N * sizeof(double);
}
// general sparse kernel (sparse.hip): L envs per one-wave workgroup; prepare returns -2 for an L that is not instantiated
size_t nr_sparse_lds_bytes(int n, int n_blocks, int L);
int nr_sparse_prepare(int L);
void launch_nr_sparse(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st);
// general-topology kernel (dense.hip): nr_dense_prepare returns -2 when the net is too large for the LDS-resident Jacobian
size_t nr_dense_lds_bytes(const Dev& d);
int nr_dense_prepare(const Dev& d);
void launch_nr_dense(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st);
int dense_solve_debug(const double* A, const double* b, double* x, int n, int batch, hipStream_t st);
void launch_reset_begin(const Dev& d, const int64_t* start_rows, int first_try, hipStream_t st);
// slot t of block `blk` (T slots per block) of a wide kernel -> env (XCD-aligned order, see Dev::xcd_lanes); -1 beyond the batch
__host__ __device__ static inline int xcd_env(unsigned blk, unsigned t, unsigned T, unsigned L, unsigned NG) {
  const unsigned c = blk & 7u, chunk = blk >> 3, gpb = T / L;
  const unsigned gid = (chunk * gpb + t / L) * 8u + c;
  return gid < NG ? (int)(gid * L + t % L) : -1;
}
// blocks of T env slots that cover a padded batch of Bp envs in that order (a multiple of 8)
static inline unsigned xcd_blocks(unsigned Bp, unsigned T, unsigned L) { const unsigned ng = Bp / L, gpb = T / L; return 8u * (((ng + 7u) / 8u + gpb - 1u) / gpb); }
void launch_advance(const Dev& d, int add_noise, int do_profiles, int do_commit, uint32_t sb_write_off, hipStream_t st);
// res_bus p_mw / q_mvar of the buses of fused groups (their OWN elements), after the solve and before the profile advance
void launch_commit_fused(const Dev& d, hipStream_t st);
void launch_gather(const Dev& d, const double* base, const int32_t* rows, const double* scales, double scale_all,
                   const int32_t* x_ptr, const int32_t* x_row, void* out, int dtype, int C, hipStream_t st);
void launch_to_envminor(const Dev& d, const double* src, double* dst, int n, hipStream_t st);
void launch_copy_i32(const int32_t* s, int32_t* dd, int B, hipStream_t st);
void launch_copy_u8(const uint8_t* s, uint8_t* dd, int B, hipStream_t st);
void launch_stats(const Dev& d, long long* out, hipStream_t st);

}  // namespace mapdn
