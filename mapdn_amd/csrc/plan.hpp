// plan.hpp — host-side, topology-only precomputation shared by every env of a batch.
//
// Replaces what pandapower rebuilds on every runpp call (pd2ppc -> ppc branch table -> makeYbus;
// reference call site voltage_control_env.py:557) by a one-time build: per-unit pi-branches, Ybus,
// a leaf-to-root elimination order of the radial feeder (so the block-2x2 LU of the NR Jacobian has
// zero fill), element->bus CSR lists, and the integer gather tables of get_obs / get_state
// (voltage_control_env.py:213-316).
#pragma once
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mapdn.h"

namespace mapdn {

using cplx = std::complex<double>;

// node flags of the elimination schedule (wave-uniform control flow in the NR kernel)
enum : uint32_t {
  F_PARENT_ROOT = 1u,    // elimination root (no parent): no off-diagonal Jacobian block
  F_PARENT_NEXT = 2u,    // parent is node k+1: Schur update / S contribution carried in registers
  F_CARRY_IN = 4u,       // node k-1 is a child of k (its contribution arrives in registers)
  F_SCRATCH_IN = 8u,     // k has children other than k-1: their contributions sit in scratch
  F_SCRATCH_FIRST = 16u, // k is the first scratch-child of its parent: store, don't add
};

// gather source kinds (obs/state column descriptors)
enum : int32_t {
  G_ZERO = 0, G_P_ADDBACK = 1, G_Q_ADDBACK = 2, G_SGEN_P = 3, G_SGEN_Q = 4, G_VM = 5, G_VA_RAD = 6,
  G_P = 7, G_Q = 8, G_VA_DEG = 9, G_NKIND = 10
};

// One step of one NR worker: 80 bytes in global memory, addressed by (worker, row) alone and therefore prefetched
// by the kernel two rows ahead with no dependent address (5 x buffer_load_dwordx4, the 16 lanes of a worker read the
// same record).  Every slot / node index is always valid: the host substitutes a ZERO slot for absent children /
// parents and a TRASH slot or node for outputs nobody reads, so the kernel's step body is branch-free per lane.
//   flags : S_* bits 0-5 | wave-uniform hints bits 6-10 (see SU_*) | number of slot children << 16 | cptr bits 0-7 << 24
//   slots : oslot | xslot << 10 | pxslot << 20 (contribution / x slots this node writes, parent's x slot) | cptr bits 8-9 << 30
//   chs   : ch0 | ch1 << 10 | ch2 << 20 (contribution slots of the first three slot children, canonical order) | cptr bits 10-11 << 30
//   kp    : node position | parent position << 16   (idle step: trash node n+1; no parent: slack position n with Y = 0)
// cptr = start of the overflow child list (slot children 3..) in Schedule::clist.
struct StepRec {
  uint32_t flags, slots, chs, kp;
  double ykk[2], ykp[2], ypk[2];   // Y_kk, Y_k,parent, Y_parent,k (elimination parent; 0 for elimination roots)
  double cks[2];                   // Y_k,slack * V_slack (0 unless k neighbours the slack)
};
static_assert(sizeof(StepRec) == 80, "StepRec must be 80 bytes (5 x dwordx4)");

// schedule-step flags
enum : uint32_t {
  S_PARENT_ROOT = 1u,    // elimination root (no parent): no off-diagonal Jacobian block
  S_CARRY_OUT = 2u,      // the same worker processes the parent in the next row: contribution stays in registers
  S_CARRY_IN = 4u,       // the chain child's contribution arrives in registers
  S_SCRATCH_OUT = 8u,    // own contribution goes to a real LDS slot (parent gathers it)
  S_X_OUT = 16u,         // backward sweep: x_k goes to a real LDS slot (some child reads it)
  S_LIVE = 32u,          // not an idle step
  // wave-uniform hints: the same value in the records of all workers of one wavefront in one row, so the kernel can
  // skip LDS traffic that would only move zeros / trash (it reads them with readfirstlane)
  SU_GMAX_SHIFT = 6u,    // bits 6-7: max number of slot children over the wave's workers, saturated at 3
  SU_W_ANY = 256u,       // some worker of the wave writes a real contribution slot
  SU_XW_ANY = 512u,      // backward: some worker writes a real x slot
  SU_XR_ANY = 1024u,     // backward: some worker reads its parent's x from a slot
  SU_SLACK_ANY = 2048u,  // some worker's node neighbours the slack (cks != 0)
};

struct Schedule {
  int32_t W = 1, R = 0;             // workers per env group, rows
  int32_t S = 1;                    // workers per wavefront (64 / envs per workgroup): granularity of the SU_* hints
  std::vector<StepRec> steps;       // [W][R]
  std::vector<int32_t> step_of_node;  // [n + 1]: index w * R + r of the step that eliminates node k (slack: -1); the
                                      // scheduled injection Sbus is stored in this order (k_inject writes, the NR kernel
                                      // prefetches it by (worker, row) like the records)
  std::vector<int32_t> clist;       // LDS slots of the children to gather (canonical order: chain child first, then ascending)
  // LDS slots (reused by interval colouring): 4 resp. 1 pairs of doubles per env each; the last two of each kind
  // are the ZERO slot (n-2) and the TRASH slot (n-1)
  int32_t n_cslots = 0, n_xslots = 0;
  // Flat-start constants, [W][R][FLAT_N] doubles: at the flat start every voltage equals the slack set-point, so
  // the Jacobian of the FIRST Newton iteration — and with it the whole block LU — is the same for all envs
  // and is factorised here once; only the right-hand side (mismatch against the env's Sbus) is per env.
  std::vector<double> flat;
  // Mismatch-only evaluation without a tree sweep (k_nr_tree's mismatch pass): per node the children in the CANONICAL sum
  // order of the sweeps (chain child first, then ascending position), CSR by node position
  std::vector<int32_t> mm_ptr, mm_child;   // [n + 1], [n - #roots]
  // ... and the same as 80-byte records (StepRec layout) in the order the pass consumes them: worker t takes nodes t, t + W,
  // t + 2W, ... (record [t][j] = node t + j W):  flags = live | number of children << 8,  slots = child 0 | child 1 << 16,
  // chs = child 2 | Sbus entry << 16,  kp = node | parent << 16; ykk, ykp, ypk, cks as in a step record.  Children 3.. of a
  // junction come from mm_child.
  int32_t mm_np = 0;                       // records per worker
  std::vector<StepRec> mm_recs;            // [W][mm_np]
};
// per-step layout of Schedule::flat
enum { FL_SR = 0, FL_SI, FL_I0, FL_I1, FL_I2, FL_I3, FL_APR, FL_API, FL_G0, FL_G1, FL_G2, FL_G3, FLAT_N };

struct LineFlow {      // res_line.pl_mw of one net.line row from the pi model: pl / sn = Re(Sf + St) with If = yff Vf + yft Vt,
  int32_t fpos, tpos;  // It = ytf Vf + ytt Vt  ==  gff |Vf|^2 + gtt |Vt|^2 + (gft + gtf) a + (bft - btf) b,  a + jb = Vf conj(Vt)
  double c[4];         // gff, gtt, gft + gtf, bft - btf  (elimination positions, n == root; out of service: both n, all zero)
};

struct Plan {
  int32_t nb = 0, n = 0;              // buses, non-slack nodes
  int32_t nl = 0, ns = 0, n_line = 0; // loads, sgens, net.line rows
  int32_t root_bus = 0;
  double vroot = 1.0, sn_mva = 1.0, tol = 1e-8;
  bool radial = false;

  std::vector<int32_t> bus_of_pos, pos_of_bus;  // [nb]; pos n == root
  std::vector<int32_t> par;                     // [n] parent position
  std::vector<uint32_t> flags;                  // [n]
  std::vector<double> yc;                       // [n*8] ykk, ykp, ypk, Y_k,slack*V_slack (re, im)
  double yrr[2] = {0, 0};
  std::vector<int32_t> root_children;           // positions of the slack's neighbours
  std::vector<double> root_y;                   // [2*len] Y[slack, neighbour] (re, im)
  std::vector<cplx> ybus;                       // dense [nb*nb], debug export only
  std::vector<int32_t> gy_ptr, gy_col;          // Ybus rows of the non-slack buses by position, CSR (columns: positions, n == slack)
  std::vector<double> gy_val;                   // (re, im) per entry — read by the general-topology kernel (dense.hip)

  // ---- bus fusion (mapdn_netspec.bus_alias: closed bus-bus switches).  nb / n / positions / every array above and the element CSRs
  // below describe the ELECTRICAL nodes (one per group of fused buses).  The env's tables are per ORIGINAL bus (nbo of them):
  int32_t nbo = 0;                              // original buses (== nb without fusion)
  std::vector<int32_t> pos_of_obus;             // [nbo] elimination position of the electrical node of original bus b
  std::vector<int32_t> cm_kind;                 // [nbo] res_bus p / q of bus b: 0 = the node's (-Sbus sn + shunt |V|^2; slack: its injection),
                                                //       1 = the bus's OWN elements (fused group), 2 = own elements - ext_grid (its own bus)
  std::vector<int32_t> fused_obus;              // original buses with cm_kind >= 1
  std::vector<int32_t> ob_load_ptr, ob_load_idx, ob_sgen_ptr, ob_sgen_idx;   // own elements of the fused buses, CSR over fused_obus
  std::vector<double> ob_shunt_p, ob_shunt_q;   // [fused] own shunts (MW / MVAr at 1 p.u.)
  std::vector<int32_t> slack_group;             // indices into fused_obus of the members of the slack's group (kind 2 needs their sum)
  std::vector<int32_t> alias_pos;               // position of the node of every NON-representative bus (reward statistics count it again)

  std::vector<LineFlow> lines;                  // [n_line]
  // element -> bus CSR by position (0..nb-1, root last)
  std::vector<int32_t> load_ptr, load_idx, sgen_ptr, sgen_idx;
  std::vector<double> shunt_p, shunt_q;         // [nb] by position, MW/MVAr at 1 p.u.
  std::vector<int32_t> sgen_bus;                // [ns] bus ids
  std::vector<double> load_scale, sgen_scale;   // [nl], [ns] scaling * in_service (pd2ppc sums p, q * scaling)

  // get_obs / get_state tables
  int32_t n_agents = 0, obs_size = 0, state_size = 0, max_zone = 0;
  std::vector<int32_t> obs_kind, obs_idx;       // [n_agents*obs_size]
  std::vector<int32_t> state_kind, state_idx;   // [state_size]
};

// -------------------------------------------------------------------------------------------------------------------
// General sparse path (k_nr_sparse, sparse.hip): for a MESHED net the block-2x2 LU of the Jacobian has fill, so the host
// does the symbolic work once per topology — minimum-degree order on the bus graph, fill pattern, a slot for every
// structurally non-zero 2x2 block — and compiles the numeric factorisation + forward / backward substitution into a
// PROGRAM of block operations  B[c] = inv(B[a]) | B[c] = B[a] B[b] | B[c] -= B[a] B[b]  that a list scheduler packs
// into phases of at most S independent operations (S = sub-lanes per env in the kernel).  Right-hand sides live in the
// block array as [b | 0] blocks, so substitution uses the same three operations.  What pandapower leaves to SuperLU
// at run time (pypower newtonpf.py: dx = -spsolve(J, F)) is thus decided here, off the hot path.
struct SpOp { uint32_t type, c, a, b; };                                  // type: 0 NOP, 1 INV, 2 MUL, 3 UPD; block slots
struct SpNz { uint32_t col; int32_t slot; double y[2]; uint32_t pad[2]; };   // Ybus entry of a row: column position (n = slack), block slot (-1: none), Y
struct SpRow { uint32_t node, nnz, sb, live; };                           // assembly row: position, entries, Sbus entry, 1 if a real node
static_assert(sizeof(SpOp) == 16 && sizeof(SpNz) == 32 && sizeof(SpRow) == 16, "sparse program records");
struct SparseProg {
  int32_t S = 0;                    // sub-lanes the program is packed for
  int32_t n_blocks = 0;             // block slots: diag [0, n) | rhs [n, 2n) | off-diagonal incl. fill [2n, ..) | one scratch slot
  int32_t n_fill = 0, n_phases = 0, rows_per_sub = 0, max_nnz = 0;
  std::vector<int32_t> order;       // elimination order (positions)
  std::vector<int32_t> fill_slots;  // blocks that exist only as fill: zeroed before every assembly (padded with the scratch slot to a multiple of S)
  std::vector<SpRow> rows;          // [S][rows_per_sub]
  std::vector<SpNz> nz;             // [S][rows_per_sub][max_nnz]
  std::vector<SpOp> ops;            // [n_phases][S]
  std::vector<int32_t> slots_ij;    // (i, j, slot) of every off-diagonal block incl. fill (host only: plan checks)
};
// symbolic factorisation only (n_blocks etc.; S-independent), then the packed program for S sub-lanes
void sparse_symbolic(const Plan& P, SparseProg& out);
void sparse_program(const Plan& P, int S, SparseProg& out);

// returns MAPDN_OK or an error code, filling `err`
int build_plan(const mapdn_netspec& net, const mapdn_env_config& cfg, Plan& out, std::string& err);

// Hu's level algorithm (optimal for unit-time in-trees) on W workers, with chain affinity so that a
// parent scheduled right after its canonical chain child (node k-1 -> k when F_PARENT_NEXT) runs on
// the same wave and receives the contribution through registers.  The sum order of children is
// canonical (chain child first, then ascending position) and therefore independent of W.
// min_cslots: lower bound on the number of contribution slots (the kernel's epilogue re-uses that LDS region)
// S: workers per wavefront (only sets the granularity of the wave-uniform SU_* hints).
// leaf-side rows of the tree kernel whose G factors stay in registers (k_nr_tree peels them: a multiple of 3; 8 AGPRs each);
// the schedules the kernel runs are padded with idle rows to at least this many (min_rows)
#ifndef NR_G_REG_ROWS
#define NR_G_REG_ROWS 6
#endif
// ... and, in the layouts whose h factors are in global scratch too, rows whose h AND G stay in registers (12 AGPRs each)
// nodes per turn of k_nr_tree's update pass (their loads up front, their arithmetic one basic block)
#ifndef NR_UPDATE_GROUP
#define NR_UPDATE_GROUP 3
#endif
#ifndef NR_HG_REG_ROWS
#define NR_HG_REG_ROWS 12
#endif
void build_schedule(const Plan& P, int W, Schedule& out, int min_cslots = 0, int S = 1, int min_rows = 0);

}  // namespace mapdn
