"""TEST INFRASTRUCTURE — CPU restatement of ``pandapower.runpp(net)`` (all defaults) in numpy/scipy.

PARITY UNPINNED: the arithmetic of this path lives in the third-party package
``pandapower==2.7.0`` (reference pin: /root/reference/environment.yml:133; call sites
/root/reference/environments/var_voltage_control/voltage_control_env.py:124,165,557).  pandapower
is neither vendored in the reference nor installable here (no network), and the reference holds
no tests / golden vectors, so this file restates pandapower's *published* algorithm
(pd2ppc -> makeYbus -> pypower newtonpf -> pfsoln -> result tables) from recollection of the
upstream 2.7.0 sources.  It is anchored instead on
  * the public IEEE 33-bus Baran-Wu known answer (tests/test_oracle.py),
  * a solver-independent residual certificate (``residual_inf``), and
  * an independent backward/forward-sweep solver (oracle/sweep.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (mapdn_amd/) never does.

Every function names the pandapower module it restates; the MAPDN call site is
voltage_control_env.py:557 (``pp.runpp(self.powergrid)`` with no kwargs), hence the options:
algorithm='nr', init='auto' (flat: all buses start at the mean ext_grid/gen vm_pu set-point, 0 rad),
max_iteration='auto' (10 for nr), tolerance_mva=1e-8, calculate_voltage_angles='auto' (False below
70 kV), enforce_q_lims=False, voltage_depend_loads=True (all loads constant power here).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

MAX_ITER = 10            # pandapower runpp: max_iteration="auto" -> 10 for "nr"
TOLERANCE_MVA = 1e-8     # pandapower runpp: tolerance_mva default


class LoadflowNotConverged(Exception):
    """pandapower.powerflow.LoadflowNotConverged (subclass of ppException, caught at env.py:559)"""


# ------------------------------------------------------------------------------------------------
# pd2ppc: build_branch._calc_line_parameter  +  generic branches
# ------------------------------------------------------------------------------------------------
def build_branches(net):
    """Return (f, t, r, x, b_complex, tap_complex, is_line) in per unit on base ``net.sn_mva``.

    pandapower/build_branch.py::_calc_line_parameter: baseR = vn_kv(from_bus)^2 / sn_mva;
    r = r_ohm_per_km*length/baseR/parallel; x likewise;
    b = 2*pi*f_hz*c_nf_per_km*1e-9*baseR*length*parallel; g = g_us_per_km*1e-6*baseR*length*parallel;
    ppc BR_B = b - 1j*g.  Out-of-service lines are dropped.
    """
    on = net.line_in_service.astype(bool)
    f = net.line_from_bus[on].astype(np.int64)
    t = net.line_to_bus[on].astype(np.int64)
    length = net.line_length_km[on]
    par = net.line_parallel[on].astype(np.float64)
    base_r = np.square(net.bus_vn_kv[f]) / net.sn_mva
    r = net.line_r_ohm_per_km[on] * length / base_r / par
    x = net.line_x_ohm_per_km[on] * length / base_r / par
    b = 2 * net.f_hz * np.pi * net.line_c_nf_per_km[on] * 1e-9 * base_r * length * par
    g = net.line_g_us_per_km[on] * 1e-6 * base_r * length * par
    bc = b - 1j * g
    tap = np.ones(f.shape[0], dtype=np.complex128)
    is_line = np.ones(f.shape[0], dtype=bool)
    if net.n_branch_pu:
        ratio = np.where(net.br_ratio == 0.0, 1.0, net.br_ratio)
        f = np.concatenate([f, net.br_from_bus.astype(np.int64)])
        t = np.concatenate([t, net.br_to_bus.astype(np.int64)])
        r = np.concatenate([r, net.br_r_pu])
        x = np.concatenate([x, net.br_x_pu])
        bc = np.concatenate([bc, net.br_b_pu - 1j * net.br_g_pu])          # BR_B = b - 1j*g (trafo: iron losses)
        tap = np.concatenate([tap, ratio * np.exp(1j * np.pi / 180.0 * net.br_shift_deg)])
        is_line = np.concatenate([is_line, np.zeros(net.n_branch_pu, dtype=bool)])
    return f, t, r, x, bc, tap, is_line


def make_ybus(net):
    """pandapower/pypower/makeYbus.py: returns (Ybus, Yf, Yt) as scipy CSR complex128.

    Ys = 1/(r+jx); Bc = BR_B; Ytt = Ys + 1j*Bc/2; Yff = Ytt/(tap*conj(tap)); Yft = -Ys/conj(tap);
    Ytf = -Ys/tap; Ysh = (GS + 1j*BS)/baseMVA with pandapower shunts GS = p_mw, BS = -q_mvar.
    """
    nb = net.n_bus
    f, t, r, x, bc, tap, _ = build_branches(net)
    nl = f.shape[0]
    ys = 1.0 / (r + 1j * x)
    ytt = ys + 1j * bc / 2
    yff = ytt / (tap * np.conj(tap))
    yft = -ys / np.conj(tap)
    ytf = -ys / tap
    ysh = np.zeros(nb, dtype=np.complex128)
    if net.shunt_bus.shape[0]:
        np.add.at(ysh, net.shunt_bus, (net.shunt_p_mw - 1j * net.shunt_q_mvar) / net.sn_mva)
    i = np.arange(nl)
    yf = sp.csr_matrix((np.r_[yff, yft], (np.r_[i, i], np.r_[f, t])), (nl, nb))
    yt = sp.csr_matrix((np.r_[ytf, ytt], (np.r_[i, i], np.r_[f, t])), (nl, nb))
    cf = sp.csr_matrix((np.ones(nl), (i, f)), (nl, nb))
    ct = sp.csr_matrix((np.ones(nl), (i, t)), (nl, nb))
    ybus = cf.T @ yf + ct.T @ yt + sp.diags(ysh, 0, (nb, nb), format="csr")
    ybus = sp.csr_matrix(ybus)
    ybus.sum_duplicates()
    ybus.sort_indices()
    return ybus, yf, yt


def bus_demand(net, p_load, q_load, p_sgen, q_sgen):
    """pandapower/build_bus.py::_calc_pq_elements_and_add_on_ppc: PD/QD per bus in MW/MVAr,
    loads positive, sgens negative, each times its scaling * in_service."""
    pd_ = np.zeros(net.n_bus)
    qd = np.zeros(net.n_bus)
    np.add.at(pd_, net.load_bus, np.asarray(p_load) * net.load_scaling)
    np.add.at(qd, net.load_bus, np.asarray(q_load) * net.load_scaling)
    np.add.at(pd_, net.sgen_bus, -np.asarray(p_sgen) * net.sgen_scaling)
    np.add.at(qd, net.sgen_bus, -np.asarray(q_sgen) * net.sgen_scaling)
    return pd_, qd


def make_sbus(net, pd_, qd):
    """pandapower/pypower/makeSbus.py with no PV/ref generation inside Sbus: -(PD + jQD)/baseMVA."""
    return -(pd_ + 1j * qd) / net.sn_mva


# ------------------------------------------------------------------------------------------------
# pypower/dSbus_dV.py + pf/create_jacobian.py
# ------------------------------------------------------------------------------------------------
def dSbus_dV(ybus, v):
    ib = ybus @ v
    n = v.shape[0]
    diag_v = sp.diags(v, 0, (n, n), format="csr")
    diag_ib = sp.diags(ib, 0, (n, n), format="csr")
    diag_vn = sp.diags(v / np.abs(v), 0, (n, n), format="csr")
    ds_dvm = diag_v @ np.conj(ybus @ diag_vn) + np.conj(diag_ib) @ diag_vn
    ds_dva = 1j * diag_v @ np.conj(diag_ib - ybus @ diag_v)
    return sp.csr_matrix(ds_dvm), sp.csr_matrix(ds_dva)


def jacobian(ybus, v, pvpq, pq):
    ds_dvm, ds_dva = dSbus_dV(ybus, v)
    j11 = ds_dva[pvpq][:, pvpq].real
    j12 = ds_dvm[pvpq][:, pq].real
    j21 = ds_dva[pq][:, pvpq].imag
    j22 = ds_dvm[pq][:, pq].imag
    return sp.bmat([[j11, j12], [j21, j22]], format="csc")


def _fx(ybus, v, sbus, pvpq, pq):
    mis = v * np.conj(ybus @ v) - sbus
    return np.r_[mis[pvpq].real, mis[pq].imag]


def newtonpf(ybus, sbus, v0, ref, pv, pq, tol, max_it=MAX_ITER):
    """pandapower/pypower/newtonpf.py (polar NR, full Jacobian, SuperLU spsolve)."""
    pvpq = np.r_[pv, pq].astype(np.int64)
    pq = np.asarray(pq, np.int64)
    npvpq = pvpq.shape[0]
    v = v0.astype(np.complex128).copy()
    va = np.angle(v)
    vm = np.abs(v)
    f = _fx(ybus, v, sbus, pvpq, pq)
    converged = np.linalg.norm(f, np.inf) < tol
    i = 0
    while (not converged) and i < max_it:
        i += 1
        jac = jacobian(ybus, v, pvpq, pq)
        dx = -1.0 * spla.spsolve(jac, f)
        va[pvpq] = va[pvpq] + dx[:npvpq]
        vm[pq] = vm[pq] + dx[npvpq:]
        v = vm * np.exp(1j * va)
        vm = np.abs(v)
        va = np.angle(v)
        f = _fx(ybus, v, sbus, pvpq, pq)
        converged = np.linalg.norm(f, np.inf) < tol
    return v, bool(converged), i


# ------------------------------------------------------------------------------------------------
# runpp
# ------------------------------------------------------------------------------------------------
class PPResult(dict):
    __getattr__ = dict.__getitem__


_YCACHE = {}


def _cached_ybus(net):
    key = id(net)
    hit = _YCACHE.get(key)
    if hit is None or hit[0] is not net:
        hit = (net,) + make_ybus(net) + (build_branches(net),)
        _YCACHE.clear()
        _YCACHE[key] = hit
    return hit[1], hit[2], hit[3], hit[4]


def dc_angles(net, pd_):
    """pandapower/pf/run_dc_pf.py::_run_dc_pf + pypower makeBdc / dcpf — the angles runpp starts from when init="auto" resolves to
    init_va_degree="dc" (calculate_voltage_angles=True: a line at a bus above 70 kV).  [PP-recalled.]
      b = 1 / x / tap_ratio per in-service branch;  Bbus = (Cf - Ct)' diag(b) (Cf - Ct);  Pfinj = b * (-shift_rad);
      Pbusinj = (Cf - Ct)' Pfinj;  Pbus = Re(Sbus) - Pbusinj - GS / baseMVA;  Va[pvpq] = Bbus[pvpq, pvpq]^-1 (Pbus[pvpq] - Bbus[pvpq, ref] Va0[ref])
    with Va0[ref] = 0 (ext_grid.va_degree is taken as given: 0 in every MAPDN net)."""
    f, t, _, x, _, tap, _ = build_branches(net)
    nb, nl = net.n_bus, f.shape[0]
    ratio = np.abs(tap)
    shift = np.angle(tap)
    b = 1.0 / x / ratio
    i = np.arange(nl)
    cft = sp.csr_matrix((np.r_[np.ones(nl), -np.ones(nl)], (np.r_[i, i], np.r_[f, t])), (nl, nb))
    bbus = sp.csr_matrix(cft.T @ sp.diags(b) @ cft)
    pbusinj = cft.T @ (b * -shift)
    gs = np.zeros(nb)
    if net.shunt_bus.shape[0]:
        np.add.at(gs, net.shunt_bus, net.shunt_p_mw)
    pbus = -pd_ / net.sn_mva - pbusinj - gs / net.sn_mva
    ref = int(net.ext_grid_bus)
    pvpq = np.setdiff1d(np.arange(nb), [ref])
    va = np.zeros(nb)
    va[pvpq] = spla.spsolve(sp.csc_matrix(bbus[pvpq][:, pvpq]), pbus[pvpq])
    return va


def runpp_restated(net, p_load, q_load, p_sgen, q_sgen, raise_on_fail=False, cache=True, tolerance_mva=TOLERANCE_MVA,
                   tolerance_is_pu=False, init="flat"):
    """``pp.runpp(net)`` for a net whose load/sgen columns hold the given MW / MVAr values.

    Returns res_bus (vm_pu, va_degree, p_mw, q_mvar sorted by bus index), res_line.pl_mw,
    converged flag and iteration count.  Ybus is cached per NetSpec object (pandapower rebuilds it
    on every call; the values are identical because the topology never changes inside MAPDN).

    Stopping rule: ||F||inf < tolerance_mva / sn_mva with F in per unit — as recalled from pandapower 2.7.0 (UNPINNED: pandapower is
    not installable here; tests/test_pandapower_pin.py::test_tolerance_rule_on_sn_mva_not_one decides it wherever pandapower is).
    `tolerance_is_pu=True` is the other reading (||F||inf < tolerance_mva, no division) — mapdn_env_config.tolerance_is_pu.

    `init`: "flat" (runpp's init="auto" below 70 kV) or "dc" (what init="auto" resolves to when calculate_voltage_angles is on: angles
    from a DC power flow, magnitudes flat).  The product solvers only start flat; mapdn_amd.data.from_pandapower refuses HV nets.
    """
    if init not in ("flat", "dc"):
        raise ValueError("init must be 'flat' or 'dc'")
    if getattr(net, "has_fused_buses", False):
        if init != "flat":
            raise NotImplementedError("init='dc' on a net with fused buses")
        return _runpp_fused(net, p_load, q_load, p_sgen, q_sgen, raise_on_fail, cache, tolerance_mva, tolerance_is_pu)
    if cache:
        ybus, yf, yt, br = _cached_ybus(net)
    else:
        ybus, yf, yt = make_ybus(net)
        br = build_branches(net)
    f, t, _, _, _, _, is_line = br
    nb = net.n_bus
    ref = np.array([net.ext_grid_bus], np.int64)
    pq = np.setdiff1d(np.arange(nb), ref)
    pv = np.zeros(0, np.int64)
    pd_, qd = bus_demand(net, p_load, q_load, p_sgen, q_sgen)
    sbus = make_sbus(net, pd_, qd)
    # init="auto" -> flat start at mean vm set-point of voltage-controlled elements (one ext_grid)
    v0 = np.full(nb, net.ext_grid_vm_pu, dtype=np.complex128)
    if init == "dc":
        v0 = v0 * np.exp(1j * dc_angles(net, pd_))
    tol = tolerance_mva / (1.0 if tolerance_is_pu else net.sn_mva)
    v, converged, it = newtonpf(ybus, sbus, v0, ref, pv, pq, tol)
    if not converged and raise_on_fail:
        raise LoadflowNotConverged(f"Power Flow nr did not converge after {MAX_ITER} iterations!")
    # pfsoln + results_bus / results_branch
    vm = np.abs(v)
    va_deg = np.angle(v) * 180.0 / np.pi
    s_inj = v * np.conj(ybus @ v) * net.sn_mva          # network injection per bus, MVA
    p_bus = pd_.copy()
    q_bus = qd.copy()
    # ext_grid (generator sign) = S_inj_ref + demand_ref; res_bus is consumer sign => -S_inj_ref
    p_bus[ref] = -s_inj[ref].real
    q_bus[ref] = -s_inj[ref].imag
    if net.shunt_bus.shape[0]:
        np.add.at(p_bus, net.shunt_bus, net.shunt_p_mw * vm[net.shunt_bus] ** 2)
        np.add.at(q_bus, net.shunt_bus, net.shunt_q_mvar * vm[net.shunt_bus] ** 2)
    sf = v[f] * np.conj(yf @ v) * net.sn_mva
    st = v[t] * np.conj(yt @ v) * net.sn_mva
    pl = (sf.real + st.real)[is_line]
    # res_line is indexed like net.line; out-of-service lines report 0
    pl_full = np.zeros(net.n_line)
    pl_full[net.line_in_service.astype(bool)] = pl
    return PPResult(vm_pu=vm, va_degree=va_deg, p_mw=p_bus, q_mvar=q_bus, pl_mw=pl_full,
                    converged=converged, iterations=it, V=v, Sbus=sbus)


# ---- bus fusion (closed bus-bus switches) ------------------------------------------------------------------------------
# pd2ppc gives all buses joined by closed bus-bus switches ONE ppc bus (build_bus._build_bus_ppc: `bus_lookup` maps every member of a
# group to the same row) and runs the power flow on the merged buses; the result tables go back per pandapower bus: vm_pu / va_degree
# of a bus = those of its ppc bus, p_mw / q_mvar = the bus's OWN elements (results_bus._get_p_q_results sums loads / sgens / shunts
# per pandapower bus; the ext_grid's injection lands on the ext_grid's own bus).  [PP-recalled, like the rest of this file.]
_REDUCED = {}


def reduced_net(net):
    """(NetSpec on the merged buses, eid [n_bus]: electrical bus of every original bus).  Merged bus ids = the representatives in
    ascending order; zones of the reduced net are placeholders (they have no electrical meaning)."""
    key = id(net)
    hit = _REDUCED.get(key)
    if hit is not None and hit[0] is net:
        return hit[1], hit[2]
    import dataclasses
    alias = np.asarray(net.bus_alias)
    reps = np.flatnonzero(alias == np.arange(net.n_bus))
    comp = np.full(net.n_bus, -1); comp[reps] = np.arange(reps.shape[0])
    eid = comp[alias]
    if np.any(eid[net.line_from_bus[net.line_in_service.astype(bool)]] == eid[net.line_to_bus[net.line_in_service.astype(bool)]]) or \
            np.any(eid[net.br_from_bus] == eid[net.br_to_bus]):
        raise NotImplementedError("a branch between two buses of one fused group is not supported")
    lf, lt = eid[net.line_from_bus], eid[net.line_to_bus]
    dead = lf == lt                                   # (only out-of-service lines can get here): keep them out of service with valid ends
    lt = np.where(dead, (lf + 1) % reps.shape[0], lt)
    red = dataclasses.replace(
        net, name=net.name + "_merged", bus_vn_kv=net.bus_vn_kv[reps], bus_zone=net.bus_zone[reps], line_from_bus=lf, line_to_bus=lt,
        br_from_bus=eid[net.br_from_bus], br_to_bus=eid[net.br_to_bus], load_bus=eid[net.load_bus], sgen_bus=eid[net.sgen_bus],
        shunt_bus=eid[net.shunt_bus], ext_grid_bus=int(eid[net.ext_grid_bus]), bus_alias=np.arange(reps.shape[0]))
    _REDUCED[key] = (net, red, eid)
    return red, eid


def _runpp_fused(net, p_load, q_load, p_sgen, q_sgen, raise_on_fail, cache, tolerance_mva, tolerance_is_pu):
    red, eid = reduced_net(net)
    r = runpp_restated(red, p_load, q_load, p_sgen, q_sgen, raise_on_fail, cache, tolerance_mva, tolerance_is_pu)
    vm, va = r.vm_pu[eid], r.va_degree[eid]
    pd_, qd = bus_demand(net, p_load, q_load, p_sgen, q_sgen)          # per ORIGINAL bus: own loads - own sgens
    p_bus, q_bus = pd_.copy(), qd.copy()
    if net.shunt_bus.shape[0]:
        np.add.at(p_bus, net.shunt_bus, net.shunt_p_mw * vm[net.shunt_bus] ** 2)
        np.add.at(q_bus, net.shunt_bus, net.shunt_q_mvar * vm[net.shunt_bus] ** 2)
    # the ext_grid's own bus: its elements minus the ext_grid's injection.  The merged slack row of the reduced result is
    # (everything on the group) - ext_grid, so ext_grid = (sum over the group of the per-bus values above) - that row
    e0, slack_e = int(net.ext_grid_bus), int(eid[net.ext_grid_bus])
    grp = eid == slack_e
    p_ext, q_ext = p_bus[grp].sum() - r.p_mw[slack_e], q_bus[grp].sum() - r.q_mvar[slack_e]
    p_bus[e0] -= p_ext; q_bus[e0] -= q_ext
    return PPResult(vm_pu=vm, va_degree=va, p_mw=p_bus, q_mvar=q_bus, pl_mw=r.pl_mw, converged=r.converged, iterations=r.iterations,
                    V=r.V[eid], Sbus=r.Sbus[eid])


# ---- the tolerance edge -------------------------------------------------------------------------------------------------
# Evaluating F in another summation order (the GPU's tree sweeps vs scipy's CSR products) moves ||F||inf by rounding: ~2e-12 p.u.
# on the MAPDN feeders.  An iterate whose norm lands within that noise of the tolerance is stopped by one implementation and
# continued by the other; both results are converged power flows (the extra Newton step moves the voltages by ~1e-10).
# The agreement rule of the parity tests (INTEGRATION.md "Iteration counts at the tolerance edge"):
EDGE_REL, EDGE_ABS = 1e-3, 2e-12


def edge_band(tol):
    return EDGE_REL * tol + EDGE_ABS


def iterate_norms(net, p_load, q_load, p_sgen, q_sgen, n_it=MAX_ITER + 2):
    """||F||inf of the flat start and of the first n_it Newton iterates (newtonpf's arithmetic without its stopping rule)"""
    ybus = _cached_ybus(net)[0]
    nb = net.n_bus
    pq = np.setdiff1d(np.arange(nb), [net.ext_grid_bus])
    sbus = make_sbus(net, *bus_demand(net, p_load, q_load, p_sgen, q_sgen))
    v = np.full(nb, net.ext_grid_vm_pu, dtype=np.complex128); va = np.angle(v); vm = np.abs(v)
    out = []
    for _ in range(n_it + 1):
        f = _fx(ybus, v, sbus, pq, pq)
        out.append(float(np.linalg.norm(f, np.inf)))
        if not np.isfinite(out[-1]) or out[-1] > 1e6:
            out += [np.inf] * (n_it + 1 - len(out)); break
        dx = -spla.spsolve(jacobian(ybus, v, pq, pq), f)
        va[pq] += dx[:len(pq)]; vm[pq] += dx[len(pq):]
        v = vm * np.exp(1j * va); vm = np.abs(v); va = np.angle(v)
    return np.array(out)


def iterations_agree(it_a, conv_a, it_b, conv_b, norms, tol, max_it=MAX_ITER):
    """The parity rule for (iteration count, convergence flag) pairs of two implementations of newtonpf on the same inputs, given
    the oracle's per-iterate norms: equal — or they differ by ONE Newton step (or, at iteration max_it, in the flag only) and the
    iterate at which one side stopped has | ||F||inf - tol | <= edge_band(tol)."""
    if it_a == it_b and bool(conv_a) == bool(conv_b):
        return True
    if it_a == it_b == max_it:                                  # the flag alone: the 10th iterate at the tolerance
        return abs(norms[max_it] - tol) <= edge_band(tol)
    if abs(it_a - it_b) == 1 and bool(conv_a) and bool(conv_b):
        k = min(it_a, it_b)
        return abs(norms[k] - tol) <= edge_band(tol)
    return False


def residual_inf(net, v, p_load, q_load, p_sgen, q_sgen):
    """Solver-independent certificate: ||V conj(Ybus V) - Sbus||_inf over non-slack buses (p.u.)."""
    ybus, _, _, _ = _cached_ybus(net)
    pd_, qd = bus_demand(net, p_load, q_load, p_sgen, q_sgen)
    mis = v * np.conj(ybus @ v) - make_sbus(net, pd_, qd)
    mis = np.delete(mis, net.ext_grid_bus)
    return float(max(np.abs(mis.real).max(), np.abs(mis.imag).max()))
