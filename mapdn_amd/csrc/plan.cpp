// plan.cpp — see plan.hpp.  Host only (no HIP).
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <map>

namespace mapdn {

namespace {

struct PiBranch {
  int32_t f, t;
  cplx yff, yft, ytf, ytt;
};

// pandapower build_branch._calc_line_parameter + pypower makeYbus, one branch at a time
PiBranch pi_from_line(const mapdn_netspec& net, int l) {
  const int f = net.line_from_bus[l], t = net.line_to_bus[l];
  const double vn = net.bus_vn_kv[f];
  const double base_r = vn * vn / net.sn_mva;
  const double len = net.line_length_km[l];
  const double par = (double)net.line_parallel[l];
  const double r = net.line_r_ohm_per_km[l] * len / base_r / par;
  const double x = net.line_x_ohm_per_km[l] * len / base_r / par;
  const double b = 2.0 * net.f_hz * M_PI * net.line_c_nf_per_km[l] * 1e-9 * base_r * len * par;
  const double g = net.line_g_us_per_km[l] * 1e-6 * base_r * len * par;
  const cplx bc(b, -g);                       // ppc BR_B = b - 1j*g
  const cplx ys = 1.0 / cplx(r, x);
  const cplx ytt = ys + cplx(0, 1) * bc / 2.0;
  return PiBranch{f, t, ytt, -ys, -ys, ytt};  // tap == 1
}

PiBranch pi_from_pu(const mapdn_netspec& net, int k) {
  const double ratio = net.br_ratio[k] == 0.0 ? 1.0 : net.br_ratio[k];
  const cplx tap = std::polar(ratio, net.br_shift_deg[k] * M_PI / 180.0);
  const cplx ys = 1.0 / cplx(net.br_r_pu[k], net.br_x_pu[k]);
  const cplx ytt = ys + cplx(0, 1) * cplx(net.br_b_pu[k], net.br_g_pu ? -net.br_g_pu[k] : 0.0) / 2.0;   // BR_B = b - 1j*g
  return PiBranch{net.br_from_bus[k], net.br_to_bus[k], ytt / (tap * std::conj(tap)),
                  -ys / std::conj(tap), -ys / tap, ytt};
}

}  // namespace

static int build_plan_nodes(const mapdn_netspec& net, const mapdn_netspec& net_o, const mapdn_env_config& cfg, Plan& P, std::string& err);

// pd2ppc's bus lookup: the buses joined by closed bus-bus switches (mapdn_netspec.bus_alias) are ONE electrical node.  The plan is
// built on the merged net (nodes = the representatives, in ascending order); what the env reads per pandapower bus — zone frames,
// get_state, res_bus rows — keeps the original buses (Plan::nbo, pos_of_obus, cm_kind ...).
int build_plan(const mapdn_netspec& net_o, const mapdn_env_config& cfg, Plan& P, std::string& err) {
  const int nbo = net_o.n_bus;
  if (nbo < 2) { err = "netspec: need at least 2 buses"; return MAPDN_E_INVALID; }
  std::vector<int32_t> alias(nbo);
  bool fused = false;
  for (int b = 0; b < nbo; ++b) {
    alias[b] = net_o.bus_alias ? net_o.bus_alias[b] : b;
    if (alias[b] < 0 || alias[b] >= nbo) { err = "netspec: bus_alias out of range"; return MAPDN_E_INVALID; }
    fused = fused || alias[b] != b;
  }
  for (int b = 0; b < nbo; ++b) if (alias[alias[b]] != alias[b]) { err = "netspec: bus_alias must map every bus to a representative that represents itself"; return MAPDN_E_INVALID; }
  if (!fused) {
    int rc = build_plan_nodes(net_o, net_o, cfg, P, err);
    if (rc) return rc;
    P.nbo = nbo; P.pos_of_obus = P.pos_of_bus; P.cm_kind.assign(nbo, 0);
    return MAPDN_OK;
  }
  // ---- the merged net: node ids = the representatives in ascending order
  std::vector<int32_t> eid(nbo, -1);
  int nbe = 0;
  for (int b = 0; b < nbo; ++b) if (alias[b] == b) eid[b] = nbe++;
  for (int b = 0; b < nbo; ++b) eid[b] = eid[alias[b]];
  auto bad = [&](int b) { return b < 0 || b >= nbo; };
  auto remap = [&](const int32_t* a, int n, std::vector<int32_t>& out, const char* what) {
    out.resize(std::max(n, 1));
    for (int i = 0; i < n; ++i) { if (bad(a[i])) { err = std::string("netspec: ") + what + " bus out of range"; return false; } out[i] = eid[a[i]]; }
    return true;
  };
  std::vector<int32_t> lf, lt, bf, bt, lb, sb, hb, zone_e(nbe, 0);
  std::vector<double> vn_e(nbe, 0.0);
  if (!remap(net_o.line_from_bus, net_o.n_line, lf, "line") || !remap(net_o.line_to_bus, net_o.n_line, lt, "line") ||
      !remap(net_o.br_from_bus, net_o.n_branch_pu, bf, "branch") || !remap(net_o.br_to_bus, net_o.n_branch_pu, bt, "branch") ||
      !remap(net_o.load_bus, net_o.n_load, lb, "load") || !remap(net_o.sgen_bus, net_o.n_sgen, sb, "sgen") ||
      !remap(net_o.shunt_bus, net_o.n_shunt, hb, "shunt")) return MAPDN_E_INVALID;
  if (bad(net_o.ext_grid_bus)) { err = "netspec: ext_grid_bus out of range"; return MAPDN_E_INVALID; }
  for (int l = 0; l < net_o.n_line; ++l) if (lf[l] == lt[l]) {
    if (net_o.line_in_service[l]) { err = "netspec: line " + std::to_string(l) + " joins two buses of one fused group (bus_alias): not supported"; return MAPDN_E_INVALID; }
    lt[l] = (lf[l] + 1) % nbe;                    // out of service: any valid pair of ends
  }
  for (int k = 0; k < net_o.n_branch_pu; ++k) if (bf[k] == bt[k]) { err = "netspec: branch " + std::to_string(k) + " joins two buses of one fused group (bus_alias): not supported"; return MAPDN_E_INVALID; }
  for (int b = 0; b < nbo; ++b) if (alias[b] == b) { vn_e[eid[b]] = net_o.bus_vn_kv[b]; zone_e[eid[b]] = net_o.bus_zone[b]; }
  mapdn_netspec E = net_o;
  E.n_bus = nbe; E.bus_vn_kv = vn_e.data(); E.bus_zone = zone_e.data(); E.line_from_bus = lf.data(); E.line_to_bus = lt.data();
  E.br_from_bus = bf.data(); E.br_to_bus = bt.data(); E.load_bus = lb.data(); E.sgen_bus = sb.data(); E.shunt_bus = hb.data();
  E.ext_grid_bus = eid[net_o.ext_grid_bus]; E.bus_alias = nullptr;
  int rc = build_plan_nodes(E, net_o, cfg, P, err);
  if (rc) return rc;
  // ---- the reporting layer
  P.nbo = nbo;
  P.pos_of_obus.resize(nbo);
  std::vector<int> gsize(nbe, 0);
  for (int b = 0; b < nbo; ++b) { P.pos_of_obus[b] = P.pos_of_bus[eid[b]]; gsize[eid[b]]++; }
  P.cm_kind.assign(nbo, 0);
  P.fused_obus.clear(); P.alias_pos.clear(); P.slack_group.clear();
  for (int b = 0; b < nbo; ++b) {
    if (gsize[eid[b]] > 1) {
      P.cm_kind[b] = b == net_o.ext_grid_bus ? 2 : 1;
      if (eid[b] == E.ext_grid_bus) P.slack_group.push_back((int32_t)P.fused_obus.size());
      P.fused_obus.push_back(b);
    }
    if (alias[b] != b) P.alias_pos.push_back(P.pos_of_obus[b]);
  }
  const int nf = (int)P.fused_obus.size();
  std::vector<int> fidx(nbo, -1);
  for (int i = 0; i < nf; ++i) fidx[P.fused_obus[i]] = i;
  auto own = [&](int n_el, const int32_t* el_bus, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
    ptr.assign(nf + 1, 0); idx.clear();
    for (int i = 0; i < nf; ++i) {
      for (int j = 0; j < n_el; ++j) if (el_bus[j] == P.fused_obus[i]) idx.push_back(j);      // ascending element index
      ptr[i + 1] = (int32_t)idx.size();
    }
    if (idx.empty()) idx.push_back(0);
  };
  own(net_o.n_load, net_o.load_bus, P.ob_load_ptr, P.ob_load_idx);
  own(net_o.n_sgen, net_o.sgen_bus, P.ob_sgen_ptr, P.ob_sgen_idx);
  P.ob_shunt_p.assign(std::max(nf, 1), 0.0); P.ob_shunt_q.assign(std::max(nf, 1), 0.0);
  for (int i = 0; i < net_o.n_shunt; ++i) if (fidx[net_o.shunt_bus[i]] >= 0) {
    P.ob_shunt_p[fidx[net_o.shunt_bus[i]]] += net_o.shunt_p_mw[i]; P.ob_shunt_q[fidx[net_o.shunt_bus[i]]] += net_o.shunt_q_mvar[i]; }
  return MAPDN_OK;
}

// the plan of the electrical nodes of `net` (a net without fusion, or the merged net); zone frames / get_state / the sgens' buses
// follow the ORIGINAL net `net_o`
static int build_plan_nodes(const mapdn_netspec& net, const mapdn_netspec& net_o, const mapdn_env_config& cfg, Plan& P, std::string& err) {
  const int nb = net.n_bus;
  const int nbo = net_o.n_bus;
  if (nb < 2) { err = "netspec: need at least 2 buses"; return MAPDN_E_INVALID; }
  if (net.ext_grid_bus < 0 || net.ext_grid_bus >= nb) { err = "netspec: ext_grid_bus out of range"; return MAPDN_E_INVALID; }
  if (!(net.sn_mva > 0)) { err = "netspec: sn_mva must be > 0"; return MAPDN_E_INVALID; }
  auto bad_bus = [&](int b) { return b < 0 || b >= nb; };
  for (int l = 0; l < net.n_line; ++l)
    if (bad_bus(net.line_from_bus[l]) || bad_bus(net.line_to_bus[l]) || net.line_from_bus[l] == net.line_to_bus[l]) {
      err = "netspec: line " + std::to_string(l) + " has invalid endpoints"; return MAPDN_E_INVALID; }
  for (int k = 0; k < net.n_branch_pu; ++k)
    if (bad_bus(net.br_from_bus[k]) || bad_bus(net.br_to_bus[k]) || net.br_from_bus[k] == net.br_to_bus[k]) {
      err = "netspec: branch " + std::to_string(k) + " has invalid endpoints"; return MAPDN_E_INVALID; }
  for (int i = 0; i < net.n_load; ++i) if (bad_bus(net.load_bus[i])) { err = "netspec: load bus out of range"; return MAPDN_E_INVALID; }
  for (int i = 0; i < net.n_sgen; ++i) {
    if (bad_bus(net.sgen_bus[i])) { err = "netspec: sgen bus out of range"; return MAPDN_E_INVALID; }
    // reference: `.loc[sgen_bus]` on the zone frame raises KeyError otherwise (voltage_control_env.py:239)
    if (net_o.bus_zone[net_o.sgen_bus[i]] != net_o.sgen_zone[i]) {
      err = "netspec: sgen " + std::to_string(i) + " sits on a bus outside its own zone (reference get_obs would raise KeyError)";
      return MAPDN_E_INVALID; }
    if (net.sgen_zone[i] <= 0) { err = "netspec: sgen zone must be a non-main zone"; return MAPDN_E_INVALID; }
  }
  for (int i = 0; i < net.n_shunt; ++i) if (bad_bus(net.shunt_bus[i])) { err = "netspec: shunt bus out of range"; return MAPDN_E_INVALID; }
  if (net.n_sgen < 1) { err = "netspec: need at least one sgen (agent)"; return MAPDN_E_INVALID; }

  P.nb = nb; P.n = nb - 1; P.nl = net.n_load; P.ns = net.n_sgen; P.n_line = net.n_line;
  P.root_bus = net.ext_grid_bus; P.vroot = net.ext_grid_vm_pu; P.sn_mva = net.sn_mva;
  // runpp(tolerance_mva = 1e-8) on a per-unit mismatch: ||F||inf < tolerance_mva / sn_mva as restated from pandapower 2.7.0 (UNPINNED,
  // see mapdn_env_config.tolerance_is_pu: 1 takes tolerance_mva as the per-unit bound without the division)
  if (cfg.tolerance_mva < 0.0) { err = "tolerance_mva must be > 0 (0 = the runpp default 1e-8)"; return MAPDN_E_INVALID; }
  P.tol = (cfg.tolerance_mva > 0.0 ? cfg.tolerance_mva : 1e-8) / (cfg.tolerance_is_pu ? 1.0 : net.sn_mva);

  // ---- Ybus (dense on the host; nb <= a few hundred) -------------------------------------------
  P.ybus.assign((size_t)nb * nb, cplx(0, 0));
  auto Y = [&](int i, int j) -> cplx& { return P.ybus[(size_t)i * nb + j]; };
  std::vector<PiBranch> line_pi(net.n_line);
  auto stamp = [&](const PiBranch& b) {
    Y(b.f, b.f) += b.yff; Y(b.f, b.t) += b.yft; Y(b.t, b.f) += b.ytf; Y(b.t, b.t) += b.ytt;
  };
  std::vector<std::vector<int>> adj(nb);
  auto link = [&](int f, int t) {
    if (std::find(adj[f].begin(), adj[f].end(), t) == adj[f].end()) { adj[f].push_back(t); adj[t].push_back(f); }
  };
  for (int l = 0; l < net.n_line; ++l) {
    if (!net.line_in_service[l]) continue;
    line_pi[l] = pi_from_line(net, l);
    stamp(line_pi[l]); link(line_pi[l].f, line_pi[l].t);
  }
  for (int k = 0; k < net.n_branch_pu; ++k) { PiBranch b = pi_from_pu(net, k); stamp(b); link(b.f, b.t); }
  P.shunt_p.assign(nb, 0.0); P.shunt_q.assign(nb, 0.0);
  std::vector<double> sh_p_bus(nb, 0.0), sh_q_bus(nb, 0.0);
  for (int i = 0; i < net.n_shunt; ++i) {   // pandapower shunt: GS = p_mw, BS = -q_mvar
    Y(net.shunt_bus[i], net.shunt_bus[i]) += cplx(net.shunt_p_mw[i], -net.shunt_q_mvar[i]) / net.sn_mva;
    sh_p_bus[net.shunt_bus[i]] += net.shunt_p_mw[i]; sh_q_bus[net.shunt_bus[i]] += net.shunt_q_mvar[i];
  }

  // ---- radial check + reverse-preorder elimination order ----------------------------------------
  size_t n_edges = 0;
  for (int i = 0; i < nb; ++i) { std::sort(adj[i].begin(), adj[i].end()); n_edges += adj[i].size(); }
  n_edges /= 2;
  std::vector<int> parent_bus(nb, -1), preorder;
  preorder.reserve(nb);
  {
    std::vector<char> seen(nb, 0);
    std::vector<int> stack{P.root_bus};
    seen[P.root_bus] = 1;
    while (!stack.empty()) {
      int u = stack.back(); stack.pop_back();
      preorder.push_back(u);
      // push in reverse so the smallest-index neighbour is visited first (deterministic)
      for (auto it = adj[u].rbegin(); it != adj[u].rend(); ++it)
        if (!seen[*it]) { seen[*it] = 1; parent_bus[*it] = u; stack.push_back(*it); }
    }
  }
  if ((int)preorder.size() != nb) { err = "topology: network is not connected to the ext_grid bus (pandapower would drop unsupplied buses; not supported)"; return MAPDN_E_TOPOLOGY; }
  P.radial = (n_edges == (size_t)nb - 1);

  // ---- elimination forest.  The slack has a known voltage: it is not eliminated, it only adds the
  // constant term V_k conj(Y_k,slack V_slack) to the S of its neighbours.  Removing it leaves one tree
  // per slack neighbour; each is rooted at its CENTER (minimum eccentricity), so the critical path of
  // a sweep is the tree radius instead of the feeder depth.  Positions = reverse DFS pre-order
  // (children before parents, the largest subtree adjacent to its parent).
  const int slack = P.root_bus;
  std::vector<int> el_parent(nb, -1);            // elimination parent (bus id), -1 for elimination roots / slack
  std::vector<int> el_order; el_order.reserve(nb);
  if (!P.radial) {
    // Meshed net (closed tie switches, parallel feeders): no fill-free elimination order exists.  The general-topology
    // kernel (dense.hip) factorises the full Jacobian, for which any bus order will do: descending bus index here,
    // i.e. ascending positions after the reversal below.
    for (int b = nb - 1; b >= 0; --b) if (b != slack) el_order.push_back(b);
  } else {
    std::vector<int> dist(nb), from(nb);
    auto bfs = [&](int src, std::vector<int>& comp) {      // BFS inside (tree - slack); returns farthest node
      comp.clear();
      std::fill(dist.begin(), dist.end(), -1);
      std::vector<int> q{src}; dist[src] = 0; from[src] = -1;
      for (size_t i = 0; i < q.size(); ++i) {
        int u = q[i]; comp.push_back(u);
        for (int w : adj[u]) if (w != slack && dist[w] < 0) { dist[w] = dist[u] + 1; from[w] = u; q.push_back(w); }
      }
      int far = src;
      for (int u : comp) if (dist[u] > dist[far] || (dist[u] == dist[far] && u < far)) far = u;
      return far;
    };
    std::vector<int> comp, comp2;
    for (int s0 : adj[slack]) {                  // one component per slack neighbour
      const int a = bfs(s0, comp);
      const int b = bfs(a, comp2);               // a..b is a diameter path
      int c = b;
      for (int i = 0; i < dist[b] / 2; ++i) c = from[c];     // its middle node = the tree center
      // root the component at c
      std::vector<std::vector<int>> children(nb);
      std::vector<int> pre, stack{c};
      std::vector<char> seen(nb, 0);
      seen[c] = 1; el_parent[c] = -1;
      while (!stack.empty()) {
        int u = stack.back(); stack.pop_back(); pre.push_back(u);
        for (int w : adj[u]) if (w != slack && !seen[w]) { seen[w] = 1; el_parent[w] = u; children[u].push_back(w); stack.push_back(w); }
      }
      std::vector<int> sub(nb, 1);
      for (auto it = pre.rbegin(); it != pre.rend(); ++it) if (el_parent[*it] >= 0) sub[el_parent[*it]] += sub[*it];
      for (int u : pre) std::stable_sort(children[u].begin(), children[u].end(), [&](int x, int y) { return sub[x] > sub[y]; });
      stack.assign(1, c);
      while (!stack.empty()) {                   // true pre-order: parent, then the largest child's whole subtree, ...
        int u = stack.back(); stack.pop_back(); el_order.push_back(u);
        for (auto it = children[u].rbegin(); it != children[u].rend(); ++it) stack.push_back(*it);
      }
    }
  }
  if ((int)el_order.size() != nb - 1) { err = "internal: elimination forest does not cover the network"; return MAPDN_E_INVALID; }
  P.bus_of_pos.assign(nb, -1); P.pos_of_bus.assign(nb, -1);
  for (int i = 0; i < nb - 1; ++i) {             // reverse pre-order -> positions 0..n-1 (children first)
    const int bus = el_order[nb - 2 - i];
    P.bus_of_pos[i] = bus; P.pos_of_bus[bus] = i;
  }
  P.bus_of_pos[P.n] = slack; P.pos_of_bus[slack] = P.n;
  P.par.assign(P.n, 0); P.flags.assign(P.n, 0u); P.yc.assign((size_t)P.n * 8, 0.0);
  for (int k = 0; P.radial && k < P.n; ++k) {
    const int bus = P.bus_of_pos[k], pb = el_parent[bus];
    const int p = pb >= 0 ? P.pos_of_bus[pb] : P.n;          // elimination roots point at the slack position (Y = 0)
    if (p <= k) { err = "internal: elimination order violated"; return MAPDN_E_INVALID; }
    P.par[k] = p;
    const cplx ykk = Y(bus, bus), ykp = pb >= 0 ? Y(bus, pb) : cplx(0, 0), ypk = pb >= 0 ? Y(pb, bus) : cplx(0, 0);
    const cplx cks = Y(bus, slack) * P.vroot;                // Y_k,slack V_slack (0 unless k neighbours the slack)
    double* c = &P.yc[(size_t)k * 8];
    c[0] = ykk.real(); c[1] = ykk.imag(); c[2] = ykp.real(); c[3] = ykp.imag(); c[4] = ypk.real(); c[5] = ypk.imag();
    c[6] = cks.real(); c[7] = cks.imag();
    if (p == P.n) P.flags[k] |= F_PARENT_ROOT;
    else if (p == k + 1) P.flags[k] |= F_PARENT_NEXT;
  }
  P.yrr[0] = Y(slack, slack).real(); P.yrr[1] = Y(slack, slack).imag();
  P.root_children.clear(); P.root_y.clear();     // the slack's neighbours and Y[slack, k] (slack injection in res_bus)
  for (int w : adj[slack]) { P.root_children.push_back(P.pos_of_bus[w]); P.root_y.push_back(Y(slack, w).real()); P.root_y.push_back(Y(slack, w).imag()); }

  // ---- Ybus rows by position (CSR; the slack row is not needed: its injection comes from root_children / root_y)
  P.gy_ptr.assign(1, 0); P.gy_col.clear(); P.gy_val.clear();
  for (int k = 0; k < P.n; ++k) {
    const int bus = P.bus_of_pos[k];
    for (int j = 0; j <= P.n; ++j) {
      const int bj = P.bus_of_pos[j];
      const cplx y = Y(bus, bj);
      const bool linked = bj == bus || std::find(adj[bus].begin(), adj[bus].end(), bj) != adj[bus].end();
      if (!linked) continue;                     // structural non-zeros only (a zero-valued link stays in the pattern)
      P.gy_col.push_back(j); P.gy_val.push_back(y.real()); P.gy_val.push_back(y.imag());
    }
    P.gy_ptr.push_back((int32_t)P.gy_col.size());
  }

  // ---- res_line flows ---------------------------------------------------------------------------
  P.lines.resize(net.n_line);
  for (int l = 0; l < net.n_line; ++l) {
    LineFlow& L = P.lines[l];
    if (!net.line_in_service[l]) { L = LineFlow{}; L.fpos = L.tpos = P.n; continue; }   // zero admittances at the slack: pl == 0
    const PiBranch& b = line_pi[l];
    L.fpos = P.pos_of_bus[b.f]; L.tpos = P.pos_of_bus[b.t];
    L.c[0] = b.yff.real(); L.c[1] = b.ytt.real(); L.c[2] = b.yft.real() + b.ytf.real(); L.c[3] = b.yft.imag() - b.ytf.imag();
  }

  // ---- element CSR by position ------------------------------------------------------------------
  auto csr = [&](int n_el, const int32_t* el_bus, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
    ptr.assign(nb + 1, 0); idx.assign(n_el, 0);
    for (int i = 0; i < n_el; ++i) ptr[P.pos_of_bus[el_bus[i]] + 1]++;
    for (int k = 0; k < nb; ++k) ptr[k + 1] += ptr[k];
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int i = 0; i < n_el; ++i) idx[fill[P.pos_of_bus[el_bus[i]]]++] = i;   // ascending element index
  };
  csr(net.n_load, net.load_bus, P.load_ptr, P.load_idx);
  csr(net.n_sgen, net.sgen_bus, P.sgen_ptr, P.sgen_idx);
  for (int b = 0; b < nb; ++b) { P.shunt_p[P.pos_of_bus[b]] = sh_p_bus[b]; P.shunt_q[P.pos_of_bus[b]] = sh_q_bus[b]; }
  P.sgen_bus.assign(net_o.sgen_bus, net_o.sgen_bus + net_o.n_sgen);      // ORIGINAL bus ids (the add-back rows of get_obs)
  P.load_scale.assign(net.n_load, 1.0); P.sgen_scale.assign(net.n_sgen, 1.0);
  if (net.load_scaling) P.load_scale.assign(net.load_scaling, net.load_scaling + net.n_load);
  if (net.sgen_scaling) P.sgen_scale.assign(net.sgen_scaling, net.sgen_scaling + net.n_sgen);

  // ---- get_obs (distributed mode) — voltage_control_env.py:245-274 --------------------------------
  const int ss = cfg.state_space;
  P.n_agents = net.n_sgen;
  std::vector<std::vector<int>> zone_rows(net.n_sgen);
  size_t max_len = 0; P.max_zone = 0;
  auto obs_len = [&](size_t z) {
    return ((ss & MAPDN_SS_DEMAND) ? 2 * z : 0) + ((ss & MAPDN_SS_PV) ? 1 : 0) + ((ss & MAPDN_SS_REACTIVE) ? 1 : 0) +
           ((ss & MAPDN_SS_VM_PU) ? z : 0) + ((ss & MAPDN_SS_VA_DEGREE) ? z : 0);
  };
  for (int i = 0; i < net.n_sgen; ++i) {
    for (int b = 0; b < nbo; ++b) if (net_o.bus_zone[b] == net_o.sgen_zone[i]) zone_rows[i].push_back(b);  // ascending bus index (:536)
    max_len = std::max(max_len, obs_len(zone_rows[i].size()));
    P.max_zone = std::max<int32_t>(P.max_zone, (int32_t)zone_rows[i].size());
  }
  P.obs_size = (int32_t)max_len;
  if (P.obs_size == 0) { err = "config: empty state_space"; return MAPDN_E_INVALID; }
  P.obs_kind.assign((size_t)P.n_agents * P.obs_size, G_ZERO);
  P.obs_idx.assign((size_t)P.n_agents * P.obs_size, 0);
  for (int i = 0; i < net.n_sgen; ++i) {
    size_t c = (size_t)i * P.obs_size;
    auto put = [&](int32_t kind, int32_t idx) { P.obs_kind[c] = kind; P.obs_idx[c] = idx; ++c; };
    const auto& rows = zone_rows[i];
    if (ss & MAPDN_SS_DEMAND) { for (int b : rows) put(G_P_ADDBACK, b); for (int b : rows) put(G_Q_ADDBACK, b); }  // :254-257
    if (ss & MAPDN_SS_PV) put(G_SGEN_P, i);                                                                        // :258-259
    if (ss & MAPDN_SS_REACTIVE) put(G_SGEN_Q, i);                                                                  // :260-261
    if (ss & MAPDN_SS_VM_PU) for (int b : rows) put(G_VM, b);                                                      // :262-263
    if (ss & MAPDN_SS_VA_DEGREE) for (int b : rows) put(G_VA_RAD, b);                                              // :264-266
  }
  // ---- get_state — voltage_control_env.py:213-230 -------------------------------------------------
  P.state_kind.clear(); P.state_idx.clear();
  auto sput = [&](int32_t kind, int32_t idx) { P.state_kind.push_back(kind); P.state_idx.push_back(idx); };
  if (ss & MAPDN_SS_DEMAND) { for (int b = 0; b < nbo; ++b) sput(G_P, b); for (int b = 0; b < nbo; ++b) sput(G_Q, b); }
  if (ss & MAPDN_SS_PV) for (int j = 0; j < net.n_sgen; ++j) sput(G_SGEN_P, j);
  if (ss & MAPDN_SS_REACTIVE) for (int j = 0; j < net.n_sgen; ++j) sput(G_SGEN_Q, j);
  if (ss & MAPDN_SS_VM_PU) for (int b = 0; b < nbo; ++b) sput(G_VM, b);
  if (ss & MAPDN_SS_VA_DEGREE) for (int b = 0; b < nbo; ++b) sput(G_VA_DEG, b);
  P.state_size = (int32_t)P.state_kind.size();
  return MAPDN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// general sparse path: symbolic factorisation + program (see plan.hpp)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Symbolic {
  std::vector<int> order;                       // elimination order
  std::vector<std::vector<int>> nbr;            // nbr[k]: neighbours of pivot k at its elimination (ascending)
  std::map<std::pair<int, int>, int> slot;      // off-diagonal block (i, j) -> slot
  std::vector<int> fill;                        // slots that are fill only
  int n_blocks = 0;
};
Symbolic symbolic(const Plan& P) {
  const int n = P.n;
  Symbolic Y;
  std::vector<std::vector<char>> adj(n, std::vector<char>(n, 0));
  int next = 2 * n;
  for (int i = 0; i < n; ++i)
    for (int q = P.gy_ptr[i]; q < P.gy_ptr[i + 1]; ++q) {
      const int j = P.gy_col[q];
      if (j < n && j != i) { adj[i][j] = 1; if (!Y.slot.count({i, j})) Y.slot[{i, j}] = next++; }
    }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (adj[i][j] && !adj[j][i]) { adj[j][i] = 1; if (!Y.slot.count({j, i})) { Y.slot[{j, i}] = next; Y.fill.push_back(next++); } }
  std::vector<char> gone(n, 0);
  Y.nbr.assign(n, {});
  for (int step = 0; step < n; ++step) {        // minimum degree, ties to the smallest position
    int best = -1, bd = 1 << 30;
    for (int k = 0; k < n; ++k) if (!gone[k]) {
      int dg = 0;
      for (int j = 0; j < n; ++j) dg += (!gone[j] && adj[k][j]);
      if (dg < bd) { bd = dg; best = k; }
    }
    const int k = best;
    gone[k] = 1; Y.order.push_back(k);
    for (int j = 0; j < n; ++j) if (!gone[j] && adj[k][j]) Y.nbr[k].push_back(j);
    for (int i : Y.nbr[k]) for (int j : Y.nbr[k]) if (i != j && !adj[i][j]) {      // fill
      adj[i][j] = 1;
      Y.slot[{i, j}] = next; Y.fill.push_back(next++);
    }
  }
  Y.n_blocks = next + 1;                        // + one scratch slot (NOPs, padding)
  return Y;
}
}  // namespace

void sparse_symbolic(const Plan& P, SparseProg& G) {
  const Symbolic Y = symbolic(P);
  G.n_blocks = Y.n_blocks; G.n_fill = (int32_t)Y.fill.size();
  G.order.assign(Y.order.begin(), Y.order.end());
}

void sparse_program(const Plan& P, int S, SparseProg& G) {
  const int n = P.n;
  const Symbolic Y = symbolic(P);
  G = SparseProg{};
  G.S = S; G.n_blocks = Y.n_blocks; G.n_fill = (int32_t)Y.fill.size();
  G.order.assign(Y.order.begin(), Y.order.end());
  const uint32_t scratch = (uint32_t)(Y.n_blocks - 1);
  G.fill_slots.assign(Y.fill.begin(), Y.fill.end());
  while (G.fill_slots.size() % S) G.fill_slots.push_back((int32_t)scratch);
  // ---- assembly rows: node i -> sub-lane i % S
  G.rows_per_sub = (n + S - 1) / S;
  G.max_nnz = 1;
  for (int i = 0; i < n; ++i) G.max_nnz = std::max<int32_t>(G.max_nnz, P.gy_ptr[i + 1] - P.gy_ptr[i]);
  G.rows.assign((size_t)S * G.rows_per_sub, SpRow{(uint32_t)(n + 1), 0u, 0u, 0u});
  G.nz.assign((size_t)S * G.rows_per_sub * G.max_nnz, SpNz{(uint32_t)(n + 1), -1, {0.0, 0.0}, {0u, 0u}});
  for (int i = 0; i < n; ++i) {
    const size_t r = (size_t)(i % S) * G.rows_per_sub + i / S;
    G.rows[r] = SpRow{(uint32_t)i, (uint32_t)(P.gy_ptr[i + 1] - P.gy_ptr[i]), (uint32_t)i, 1u};
    for (int q = P.gy_ptr[i]; q < P.gy_ptr[i + 1]; ++q) {
      const int j = P.gy_col[q];
      SpNz& z = G.nz[r * G.max_nnz + (q - P.gy_ptr[i])];
      z.col = (uint32_t)j; z.y[0] = P.gy_val[2 * q]; z.y[1] = P.gy_val[2 * q + 1];
      z.slot = (j < n && j != i) ? Y.slot.at({i, j}) : -1;
    }
  }
  // ---- operations in elimination order, then list scheduling into phases of <= S independent ops
  std::vector<SpOp> seq;
  auto off = [&](int i, int j) { return (uint32_t)Y.slot.at({i, j}); };
  auto diag = [&](int i) { return (uint32_t)i; };
  auto rhs = [&](int i) { return (uint32_t)(n + i); };
  for (int k : Y.order) {
    seq.push_back(SpOp{1u, diag(k), diag(k), diag(k)});                                   // D_k^-1 in place
    for (int i : Y.nbr[k]) seq.push_back(SpOp{2u, off(i, k), off(i, k), diag(k)});        // L_ik = A_ik D_k^-1
    for (int i : Y.nbr[k]) {
      for (int j : Y.nbr[k]) seq.push_back(SpOp{3u, i == j ? diag(i) : off(i, j), off(i, k), off(k, j)});   // A_ij -= L_ik A_kj
      seq.push_back(SpOp{3u, rhs(i), off(i, k), rhs(k)});                                 // b_i -= L_ik b_k
    }
  }
  for (auto it = Y.order.rbegin(); it != Y.order.rend(); ++it) {
    const int k = *it;
    for (int j : Y.nbr[k]) seq.push_back(SpOp{3u, rhs(k), off(k, j), rhs(j)});            // b_k -= U_kj x_j
    seq.push_back(SpOp{2u, rhs(k), diag(k), rhs(k)});                                     // x_k = D_k^-1 b_k
  }
  std::vector<int> lw((size_t)Y.n_blocks, -1), lr((size_t)Y.n_blocks, -1);               // last phase that wrote / read a block
  std::vector<std::vector<SpOp>> phases;
  for (const SpOp& o : seq) {
    // reads (a, b, and c of an UPD) need the last write to be in an EARLIER phase; the write of c may share a phase with
    // earlier reads of c (all lanes read before any lane writes within a phase) but not with another write of c
    int p = std::max(lw[o.a], lw[o.b]) + 1;
    p = std::max(p, lw[o.c] + 1);
    p = std::max(p, lr[o.c]);
    while (p < (int)phases.size() && (int)phases[p].size() >= S) ++p;
    // (a later phase is always admissible: dependencies only bound p from below)
    if (p >= (int)phases.size()) phases.resize(p + 1);
    phases[p].push_back(o);
    lr[o.a] = std::max(lr[o.a], p); lr[o.b] = std::max(lr[o.b], p);
    if (o.type == 3u) lr[o.c] = std::max(lr[o.c], p);
    lw[o.c] = p;
  }
  for (const auto& kv : Y.slot) { G.slots_ij.push_back(kv.first.first); G.slots_ij.push_back(kv.first.second); G.slots_ij.push_back(kv.second); }
  G.n_phases = (int32_t)phases.size();
  G.ops.assign((size_t)G.n_phases * S, SpOp{0u, scratch, scratch, scratch});
  for (int p = 0; p < G.n_phases; ++p) {
    for (size_t q = 0; q < phases[p].size(); ++q) G.ops[(size_t)p * S + q] = phases[p][q];
    // wave-uniform hints (the same in every record of a phase): bit 8 = the phase holds an INV, bit 9 = it holds an UPD —
    // the kernel skips the 2x2 inverse / the loads of C for phases that have neither
    uint32_t hint = 0;
    for (const SpOp& o : phases[p]) hint |= o.type == 1u ? 256u : (o.type == 3u ? 512u : 0u);
    for (int q = 0; q < S; ++q) G.ops[(size_t)p * S + q].type |= hint;
  }
}

void build_schedule(const Plan& P, int W, Schedule& S, int min_cslots, int Sw, int min_rows) {
  const int n = P.n;
  if (Sw < 1 || W % Sw) Sw = 1;
  S.W = W; S.S = Sw; S.steps.clear(); S.clist.clear();
  std::vector<std::vector<int>> children(n + 1);
  for (int k = 0; k < n; ++k) children[P.par[k]].push_back(k);     // ascending k
  std::vector<int> depth(n + 1, 0);
  for (int k = n - 1; k >= 0; --k) depth[k] = depth[P.par[k]] + 1;  // parents have larger positions
  auto chain_child = [&](int k) { return (k > 0 && (P.flags[k - 1] & F_PARENT_NEXT)) ? k - 1 : -1; };
  std::vector<int> pending(n, 0), row_of(n, -1), wave_of(n, -1);
  for (int k = 0; k < n; ++k) pending[k] = (int)children[k].size();
  std::vector<int> ready;
  for (int k = 0; k < n; ++k) if (!pending[k]) ready.push_back(k);
  std::vector<std::vector<int>> rows;   // rows[r][w] = node or -1
  int scheduled = 0;
  if (W == 1) {   // one worker: the plan's own order keeps the feeder chains contiguous
    for (int k = 0; k < n; ++k) { rows.push_back({k}); row_of[k] = k; wave_of[k] = 0; }
    scheduled = n;
  }
  while (scheduled < n) {
    const int r = (int)rows.size();
    // highest level first; among equals prefer nodes that continue a chain from the previous row
    auto continues = [&](int k) { int c = chain_child(k); return c >= 0 && row_of[c] == r - 1; };
    std::stable_sort(ready.begin(), ready.end(), [&](int a, int b) {
      if (depth[a] != depth[b]) return depth[a] > depth[b];
      bool ca = continues(a), cb = continues(b);
      if (ca != cb) return ca;
      return a < b;
    });
    const int take = std::min<int>(W, (int)ready.size());
    std::vector<int> row(W, -1), chosen(ready.begin(), ready.begin() + take);
    ready.erase(ready.begin(), ready.begin() + take);
    std::vector<int> rest;
    for (int k : chosen) {            // chain affinity first
      int c = chain_child(k);
      if (c >= 0 && row_of[c] == r - 1 && row[wave_of[c]] < 0) row[wave_of[c]] = k; else rest.push_back(k);
    }
    // a wave whose previous-row node's parent is NOT placed on it may take any remaining node
    int w = 0;
    for (int k : rest) { while (row[w] >= 0) ++w; row[w] = k; }
    for (int ww = 0; ww < W; ++ww) if (row[ww] >= 0) { row_of[row[ww]] = r; wave_of[row[ww]] = ww; ++scheduled; }
    rows.push_back(row);
    for (int ww = 0; ww < W; ++ww) {
      int k = row[ww];
      if (k < 0) continue;
      int p = P.par[k];
      if (p < n && --pending[p] == 0) ready.push_back(p);
    }
  }
  while ((int)rows.size() < min_rows) rows.push_back(std::vector<int>(W, -1));   // idle rows (the kernel's peeled rows always exist)
  const int R = (int)rows.size();
  S.R = R;
  S.steps.assign((size_t)W * R, StepRec{});
  std::vector<char> carry_out(n, 0);
  for (int k = 0; k < n; ++k) {
    int p = P.par[k];
    if (p < n && chain_child(p) == k && row_of[p] == row_of[k] + 1 && wave_of[p] == wave_of[k]) carry_out[k] = 1;
  }
  // ---- LDS slot allocation by interval colouring.
  // contribution slot of child c: written in forward row row_of[c], read in row row_of[par]; it may
  // be rewritten only in a row AFTER the read (rows are separated by barriers, a row is not).
  std::vector<int> oslot(n, -1), xslot(n, -1);
  {
    std::vector<std::vector<int>> writers(R), release(R + 1);
    for (int k = 0; k < n; ++k) if (P.par[k] < n && !carry_out[k]) writers[row_of[k]].push_back(k);
    std::vector<int> freelist; int next = 0;
    for (int r = 0; r < R; ++r) {
      for (int sl : release[r]) freelist.push_back(sl);
      std::sort(freelist.begin(), freelist.end(), std::greater<int>());
      for (int k : writers[r]) {
        int sl; if (!freelist.empty()) { sl = freelist.back(); freelist.pop_back(); } else sl = next++;
        oslot[k] = sl;
        release[row_of[P.par[k]] + 1].push_back(sl);
      }
    }
    S.n_cslots = std::max(std::max(next, 1), min_cslots - 2);   // two more (ZERO, TRASH) are appended below
  }
  // x slot of parent p: written in backward row row_of[p], read by its non-carried children at
  // rows < row_of[p]; reusable by writers at rows strictly below the lowest reader row.
  {
    std::vector<int> low(n, -1);   // lowest reader row
    for (int k = 0; k < n; ++k) { int p = P.par[k]; if (p < n && !carry_out[k]) low[p] = (low[p] < 0) ? row_of[k] : std::min(low[p], row_of[k]); }
    std::vector<std::vector<int>> writers(R), release(R + 1);
    for (int k = 0; k < n; ++k) if (low[k] >= 0) writers[row_of[k]].push_back(k);
    std::vector<int> freelist; int next = 0;
    for (int r = R - 1; r >= 0; --r) {
      for (int sl : release[r]) freelist.push_back(sl);      // released for writers at row r: readers all at rows > r
      std::sort(freelist.begin(), freelist.end(), std::greater<int>());
      for (int k : writers[r]) {
        int sl; if (!freelist.empty()) { sl = freelist.back(); freelist.pop_back(); } else sl = next++;
        xslot[k] = sl;
        if (low[k] - 1 >= 0) release[low[k] - 1].push_back(sl);
      }
    }
    S.n_xslots = std::max(next, 1);
  }
  // zero + trash slots
  const uint32_t c_zero = (uint32_t)S.n_cslots, c_trash = c_zero + 1; S.n_cslots += 2;
  const uint32_t x_zero = (uint32_t)S.n_xslots, x_trash = x_zero + 1; S.n_xslots += 2;
  S.step_of_node.assign(n + 1, -1);
  for (int w = 0; w < W; ++w)
    for (int r = 0; r < R; ++r) {
      StepRec& T = S.steps[(size_t)w * R + r];
      const int k = rows[r][w];
      T = StepRec{};
      T.kp = (uint32_t)(n + 1) | ((uint32_t)n << 16);   // idle: trash node, slack parent
      T.slots = c_trash | (x_trash << 10) | (x_zero << 20);
      T.chs = c_zero | (c_zero << 10) | (c_zero << 20);
      // An idle step works on the trash node as if it hung off the slack by a line of admittance 1 - 1j with no load:
      // mismatch exactly 0, a regular 2x2 pivot, Newton step exactly 0 — every value it leaves in the carry registers
      // is finite (the kernel masks carries by a 0/1 factor, and 0 * NaN would poison the next live step)
      T.ykk[0] = 1.0; T.ykk[1] = -1.0; T.ykp[0] = -1.0; T.ykp[1] = 1.0; T.ypk[0] = -1.0; T.ypk[1] = 1.0;
      if (k < 0) continue;
      S.step_of_node[k] = w * R + r;
      const double* c = &P.yc[(size_t)k * 8];
      T.ykk[0] = c[0]; T.ykk[1] = c[1]; T.ykp[0] = c[2]; T.ykp[1] = c[3]; T.ypk[0] = c[4]; T.ypk[1] = c[5];
      T.cks[0] = c[6]; T.cks[1] = c[7];
      const int p = P.par[k];
      T.kp = (uint32_t)k | ((uint32_t)p << 16);
      uint32_t f = S_LIVE, os = c_trash, xsl = x_trash, pxs = x_zero;
      if (p == n) f |= S_PARENT_ROOT;
      else if (carry_out[k]) f |= S_CARRY_OUT;
      else { f |= S_SCRATCH_OUT; os = (uint32_t)oslot[k]; pxs = (uint32_t)xslot[p]; }
      std::vector<int> kids;
      const int cc = chain_child(k);
      if (cc >= 0) { if (carry_out[cc]) f |= S_CARRY_IN; else kids.push_back(oslot[cc]); }
      for (int ch : children[k]) if (ch != cc) kids.push_back(oslot[ch]);
      uint32_t ch3[3] = {c_zero, c_zero, c_zero};
      for (size_t j = 0; j < kids.size() && j < 3; ++j) ch3[j] = (uint32_t)kids[j];
      const uint32_t cptr = (uint32_t)S.clist.size();
      for (size_t j = 3; j < kids.size(); ++j) S.clist.push_back(kids[j]);
      if (xslot[k] >= 0) { f |= S_X_OUT; xsl = (uint32_t)xslot[k]; }
      T.slots = os | (xsl << 10) | (pxs << 20) | (((cptr >> 8) & 3u) << 30);
      T.chs = ch3[0] | (ch3[1] << 10) | (ch3[2] << 20) | (((cptr >> 10) & 3u) << 30);
      T.flags = f | ((uint32_t)std::min<size_t>(kids.size(), 255) << 16) | ((cptr & 255u) << 24);
    }
  // wave-uniform hints: OR / max over the Sw workers that share a wavefront, per row
  for (int w0 = 0; w0 < W; w0 += Sw)
    for (int r = 0; r < R; ++r) {
      uint32_t gmax = 0, any = 0;
      for (int w = w0; w < w0 + Sw; ++w) {
        const StepRec& T = S.steps[(size_t)w * R + r];
        if (!(T.flags & S_LIVE)) continue;
        gmax = std::max(gmax, std::min<uint32_t>((T.flags >> 16) & 255u, 3u));
        if (T.flags & S_SCRATCH_OUT) any |= SU_W_ANY;
        if (T.flags & S_X_OUT) any |= SU_XW_ANY;
        if ((T.flags & S_SCRATCH_OUT)) any |= SU_XR_ANY;       // a non-carried child of a real parent reads the parent's x slot
        if (T.cks[0] != 0.0 || T.cks[1] != 0.0) any |= SU_SLACK_ANY;
      }
      for (int w = w0; w < w0 + Sw; ++w) S.steps[(size_t)w * R + r].flags |= (gmax << SU_GMAX_SHIFT) | any;
    }
  if (S.clist.empty()) S.clist.push_back(0);
  // ---- canonical child lists (independent of W: the same order the records above encode)
  S.mm_ptr.assign(1, 0); S.mm_child.clear();
  for (int k = 0; k < n; ++k) {
    const int cc = chain_child(k);
    if (cc >= 0) S.mm_child.push_back(cc);
    for (int ch : children[k]) if (ch != cc) S.mm_child.push_back(ch);
    S.mm_ptr.push_back((int32_t)S.mm_child.size());
  }
  if (S.mm_child.empty()) S.mm_child.push_back(0);
  S.mm_np = (n + W - 1) / W;
  S.mm_recs.assign((size_t)W * S.mm_np, StepRec{});
  for (int w = 0; w < W; ++w)
    for (int j = 0; j < S.mm_np; ++j) {
      StepRec& T = S.mm_recs[(size_t)w * S.mm_np + j];
      const int kx = w + j * W;
      const bool live = kx < n;
      const int k = live ? kx : n - 1;               // (a repeated node rewrites the same value; its verdict is masked)
      const int c_lo = S.mm_ptr[k], nch = live ? S.mm_ptr[k + 1] - c_lo : 0;
      auto child = [&](int q) { return (uint32_t)(q < nch ? S.mm_child[c_lo + q] : n + 1); };   // absent: the trash node
      T.flags = (live ? 1u : 0u) | ((uint32_t)std::min(nch, 255) << 8);
      T.slots = child(0) | (child(1) << 16);
      T.chs = child(2);
      T.kp = (uint32_t)k | ((uint32_t)P.par[k] << 16);
      const double* c = &P.yc[(size_t)k * 8];
      T.ykk[0] = c[0]; T.ykk[1] = c[1]; T.ykp[0] = c[2]; T.ykp[1] = c[3]; T.ypk[0] = c[4]; T.ypk[1] = c[5]; T.cks[0] = c[6]; T.cks[1] = c[7];
    }
  // ---- flat-start factorisation (same formulas as k_nr_tree's forward step with V == vroot everywhere)
  {
    const double v = P.vroot, v2 = v * v;
    std::vector<double> aS((size_t)n * 2, 0.0), aD((size_t)n * 4, 0.0), node((size_t)n * FLAT_N, 0.0);
    for (int k = 0; k < n; ++k) {                 // children have smaller positions than their parents
      const double* c = &P.yc[(size_t)k * 8];
      const double gkk = c[0], bkk = c[1], gkp = c[2], bkp = c[3], gpk = c[4], bpk = c[5];
      const double akp_r = v2 * gkp, akp_i = -v2 * bkp, apk_r = v2 * gpk, apk_i = -v2 * bpk;
      const double akk_r = v2 * gkk, akk_i = -v2 * bkk, aks_r = v * c[6], aks_i = -v * c[7];
      const double sr = (akk_r + aks_r) + akp_r + aS[2 * k], si = (akk_i + aks_i) + akp_i + aS[2 * k + 1];
      const double D0 = -(si - akk_i) - aD[4 * k], D1 = (sr + akk_r) - aD[4 * k + 1];
      const double D2 = (sr - akk_r) - aD[4 * k + 2], D3 = (si + akk_i) - aD[4 * k + 3];
      const double idet = 1.0 / (D0 * D3 - D1 * D2);
      const double I0 = D3 * idet, I1 = -D1 * idet, I2 = -D2 * idet, I3 = D0 * idet;
      const double G0 = I0 * akp_i - I1 * akp_r, G1 = I0 * akp_r + I1 * akp_i;
      const double G2 = I2 * akp_i - I3 * akp_r, G3 = I2 * akp_r + I3 * akp_i;
      double* o = &node[(size_t)k * FLAT_N];
      o[FL_SR] = sr; o[FL_SI] = si; o[FL_I0] = I0; o[FL_I1] = I1; o[FL_I2] = I2; o[FL_I3] = I3;
      o[FL_APR] = apk_r; o[FL_API] = apk_i; o[FL_G0] = G0; o[FL_G1] = G1; o[FL_G2] = G2; o[FL_G3] = G3;
      const int p = P.par[k];
      if (p < n) {
        aS[2 * p] += apk_r; aS[2 * p + 1] += apk_i;
        aD[4 * p] += apk_i * G0 + apk_r * G2; aD[4 * p + 1] += apk_i * G1 + apk_r * G3;
        aD[4 * p + 2] += apk_i * G2 - apk_r * G0; aD[4 * p + 3] += apk_i * G3 - apk_r * G1;
      }
    }
    S.flat.assign((size_t)W * R * FLAT_N, 0.0);   // idle steps: all zero (h = t = 0 go to the trash slots)
    for (int w = 0; w < W; ++w)
      for (int r = 0; r < R; ++r) {
        const int k = rows[r][w];
        if (k >= 0) std::copy_n(&node[(size_t)k * FLAT_N], (size_t)FLAT_N, &S.flat[((size_t)w * R + r) * FLAT_N]);
      }
  }
}

}  // namespace mapdn
