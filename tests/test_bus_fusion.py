"""Bus fusion — closed bus-bus switches (net.switch, et == "b"), which pandapower's pd2ppc merges into one ppc bus (VERDICT r3 #9:
`from_pandapower` refused them).  Fused buses are one electrical node but stay rows of everything the env reads."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from mapdn_amd import _lib
from mapdn_amd.netspec import add_fused_buses, add_lines, make_case
from oracle.pp_restated import reduced_net, runpp_restated

SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}


def fused_case(case):
    """the case's feeder with four extra buses fused onto existing ones — one of them onto the ext_grid's bus, one onto a PV bus —
    and loads / an sgen / a shunt moved onto the new buses"""
    net, prof = make_case(case)
    slack = int(net.ext_grid_bus)
    pvbus = int(net.sgen_bus[1])
    lbus = int(net.load_bus[4])
    other = int(net.load_bus[10])
    net = dataclasses.replace(net, shunt_bus=np.array([lbus], np.int32), shunt_p_mw=np.array([0.01]), shunt_q_mvar=np.array([-0.05]))
    # a load onto the slack's twin needs a load on the slack: add one by moving load 0 there first
    lb = net.load_bus.copy(); lb[0] = slack
    net = dataclasses.replace(net, load_bus=lb)
    return add_fused_buses(net, [slack, pvbus, lbus, other], move_loads=[(0, 0), (4, 2), (10, 3)], move_sgens=[(1, 1)], move_shunts=[(0, 2)]), prof


def inputs(net, prof, B, seed, scale):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, prof.n_rows, B)
    pv = prof.pv[rows]
    qs = rng.uniform(-scale, scale, (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    return prof.load_p[rows], prof.load_q[rows], pv, qs


@pytest.mark.parametrize("case", ["case33", "case141"])
def test_oracle_fusion_semantics(case):
    """the oracle's restatement of the fusion rule: (1) the fused net solves exactly like the net with the elements back on the
    representatives (one electrical node); (2) every member of a group reports the group's voltage; p_mw / q_mvar are the bus's OWN
    elements and add up to the node's; the ext_grid's injection lands on the ext_grid's own bus; (3) independent of the
    restatement: replacing the fusion by very short lines gives the same voltages in the limit"""
    net, prof = fused_case(case)
    base, _ = make_case(case)
    nb = base.n_bus
    pl, ql, pv, qs = [x[0] for x in inputs(net, prof, 1, 1, SCALE[case])]
    r = runpp_restated(net, pl, ql, pv, qs)
    red, eid = reduced_net(net)
    assert red.n_bus == nb and r.converged and r.vm_pu.shape == (nb + 4,)
    rr = runpp_restated(red, pl, ql, pv, qs)
    assert np.array_equal(r.vm_pu, rr.vm_pu[eid]) and np.array_equal(r.va_degree, rr.va_degree[eid])
    for g in np.unique(eid):
        m = eid == g
        if g != eid[net.ext_grid_bus]:
            assert abs(r.p_mw[m].sum() - rr.p_mw[g]) < 1e-12 and abs(r.q_mvar[m].sum() - rr.q_mvar[g]) < 1e-12
    # own elements: the twin of the PV bus carries sgen 1 and nothing else
    twin = nb + 1
    assert abs(r.p_mw[twin] + pv[1]) < 1e-12 and abs(r.q_mvar[twin] + qs[1]) < 1e-12
    # slack group: the ext_grid's bus reports -(everything it feeds) + its own (now empty) elements; its twin the moved load
    assert abs(r.p_mw[nb + 0] - pl[0]) < 1e-12
    assert abs((r.p_mw[net.ext_grid_bus] + r.p_mw[nb + 0]) - rr.p_mw[eid[net.ext_grid_bus]]) < 1e-12
    # (3) short lines instead of the fusion: the voltages approach the fused ones linearly in the line impedance (10 and 1 milliohm;
    # shorter still and the Newton iteration no longer reaches 1e-8 MVA — why fusion is an index map, not a tiny impedance)
    k = 4
    reps = net.bus_alias[nb:]
    unfused = dataclasses.replace(net, bus_alias=np.arange(nb + k))
    diff = []
    for ohm in (1e-2, 1e-3):
        rl = runpp_restated(add_lines(unfused, [int(x) for x in reps], list(range(nb, nb + k)), ohm, ohm), pl, ql, pv, qs)
        assert rl.converged
        diff.append(np.abs(rl.vm_pu - r.vm_pu).max())
    assert diff[1] < 2e-5 and diff[1] < 0.2 * diff[0], diff


@pytest.mark.parametrize("case", ["case33", "case322"])
def test_plan_accepts_fused_buses_and_keeps_the_env_tables(case):
    """(CPU, host-only handle) dims, obs / state layout and index tables keep the ORIGINAL buses; a line inside a fused group is refused"""
    lib = _lib.load()
    net, prof = fused_case(case)
    base, _ = make_case(case)
    cn, keep = _lib.make_cnetspec(net)
    cc = _lib.make_cconfig(dict(episode_limit=240, action_scale=0.8, action_bias=0.0))
    h = C.c_void_p()
    assert lib.mapdn_create(C.byref(cn), C.byref(cc), 4, -1, C.byref(h)) == 0, lib.mapdn_last_error(None)
    d = _lib.CDims()
    lib.mapdn_dims(h, C.byref(d))
    assert d.n_bus == base.n_bus + 4 and d.is_radial == 1
    assert d.state_size == 4 * (base.n_bus + 4) + 2 * net.n_sgen
    # the twin of the PV bus sits in that sgen's zone: one more bus in its zone frame
    z = int(net.sgen_zone[1])
    assert d.max_zone_size >= int((net.bus_zone == z).sum())
    kind = np.zeros(d.n_agents * d.obs_size, np.int32); idx = np.zeros_like(kind)
    lib.mapdn_get_obs_index(h, kind.ctypes.data_as(_lib._pi), idx.ctypes.data_as(_lib._pi))
    assert (base.n_bus + 1) in idx[(kind == 5)].tolist()                 # the PV bus's twin is a vm_pu column of that agent
    lib.mapdn_destroy(h)
    bad = dataclasses.replace(net, line_to_bus=np.where(np.arange(net.n_line) == 0, base.n_bus + 0, net.line_to_bus).astype(np.int32),
                              line_from_bus=np.where(np.arange(net.n_line) == 0, int(net.ext_grid_bus), net.line_from_bus).astype(np.int32))
    cn, keep = _lib.make_cnetspec(bad)
    assert lib.mapdn_create(C.byref(cn), C.byref(cc), 4, -1, C.byref(h)) == -1 and b"fused group" in lib.mapdn_last_error(None)


def test_from_pandapower_converts_closed_bus_bus_switches():
    """a pandapower-shaped table set (the stand-in package's tables) with closed and open bus-bus switches -> NetSpec.bus_alias"""
    import pandas as pd
    from mapdn_amd.data import InertNet, from_pandapower
    base, _ = make_case("case33")
    nb = base.n_bus
    net = InertNet()
    net["bus"] = pd.DataFrame({"vn_kv": np.concatenate([base.bus_vn_kv, base.bus_vn_kv[[0, 7, 9]]]),
                               "zone": [("main" if z == 0 else f"zone{z}") for z in np.concatenate([base.bus_zone, base.bus_zone[[0, 7, 9]]])],
                               "in_service": True})
    net["line"] = pd.DataFrame({"from_bus": base.line_from_bus, "to_bus": base.line_to_bus, "r_ohm_per_km": base.line_r_ohm_per_km,
                                "x_ohm_per_km": base.line_x_ohm_per_km, "c_nf_per_km": base.line_c_nf_per_km, "length_km": base.line_length_km,
                                "parallel": base.line_parallel, "in_service": base.line_in_service.astype(bool)})
    net["load"] = pd.DataFrame({"bus": base.load_bus, "p_mw": 0.1, "q_mvar": 0.02})
    net["sgen"] = pd.DataFrame({"bus": base.sgen_bus, "p_mw": 0.5, "q_mvar": 0.0, "name": [f"zone{z}" for z in base.sgen_zone]})
    net["ext_grid"] = pd.DataFrame({"bus": [0], "vm_pu": [1.0], "in_service": [True]})
    net["switch"] = pd.DataFrame({"bus": [nb, 7, 9, 3], "element": [0, nb + 1, nb + 2, 4], "et": ["b", "b", "b", "b"], "closed": [True, True, True, False]})
    net["sn_mva"] = 1.0; net["f_hz"] = 50.0; net["name"] = "fused33"
    spec = from_pandapower(net)
    want = np.arange(nb + 3); want[nb] = 0; want[nb + 1] = 7; want[nb + 2] = 9
    assert np.array_equal(spec.bus_alias, want) and spec.has_fused_buses
    r = runpp_restated(spec, np.full(spec.n_load, 0.1), np.full(spec.n_load, 0.02), np.full(spec.n_sgen, 0.5), np.zeros(spec.n_sgen))
    assert r.converged and r.vm_pu[nb + 1] == r.vm_pu[7] and r.vm_pu[nb] == 1.0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case,solver", [("case33", "tree"), ("case141", "tree"), ("case322", "tree"), ("case33", "sparse"), ("case33", "dense"),
                                         ("case141", "sparse")])
def test_fused_net_solve_and_episode_match_the_oracle(case, solver):
    """pure power flow and noisy env steps (obs, state, reward, info, res_bus tables, an unsolvable step) on a net with fused buses —
    incl. the ext_grid's bus and a PV bus — through all three solvers"""
    import torch
    from mapdn_amd.env import VoltageControlBatch
    from oracle.env_restated import INFO_KEYS, VoltageControlOracle
    net, prof = fused_case(case)
    a = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=5)
    B = 12
    env = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=torch.float64, tuning=dict(nr_solver=solver if solver != "tree" else 0))
    assert env.n_bus == net.n_bus and env.geometry()["solver"] == dict(tree=0, sparse=1, dense=2)[solver]
    # Ybus is exported over the electrical NODES (merged buses) — the Python accessor sizes its buffer accordingly
    from oracle.pp_restated import make_ybus, reduced_net
    red, _ = reduced_net(net)
    yb = env.ybus_dense()
    assert yb.shape == (red.n_bus, red.n_bus) and env.geometry()["n_nodes"] == red.n_bus < net.n_bus
    assert np.abs(yb - make_ybus(red)[0].toarray()).max() < 1e-9
    pl, ql, pv, qs = inputs(net, prof, B, 2, SCALE[case])
    vm, va, it, cv = [x.cpu().numpy() for x in env.solve(pl, ql, pv, qs)]
    assert cv.all() and vm.shape == (B, net.n_bus)
    for e in range(B):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert r.iterations == it[e] and np.abs(vm[e] - r.vm_pu).max() < 1e-9 and np.abs(va[e] - r.va_degree).max() < 1e-7
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(B)]
    obs, state = env.reset()
    for e, o in enumerate(oracles):
        oo, os_ = o.reset()
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9 and np.abs(os_ - state[e].cpu().numpy()).max() < 1e-7
    rng = np.random.default_rng(0)
    for t in range(5):
        act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
        if t == 2:
            act[3] = 60.0
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        res = {k: v.cpu().numpy() for k, v in env.results().items()}
        ob, st = env.get_obs().cpu().numpy(), env.get_state().cpu().numpy()
        for e, o in enumerate(oracles):
            if o.steps >= o.episode_limit or getattr(o, "_dead", False):
                continue
            ro, to, io = o.step(act[e])
            o._dead = to
            assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item()), (t, e)
            assert max(abs(io[k] - info[e, c].item()) for c, k in enumerate(INFO_KEYS)) < 1e-9
            assert np.abs(np.array(o.get_obs()) - ob[e]).max() < 1e-9 and np.abs(o.get_state() - st[e]).max() < 1e-7
            assert np.abs(res["vm_pu"][e] - o.res.vm_pu).max() < 1e-9 and np.abs(res["p_mw"][e] - o.res.p_mw).max() < 1e-9
            assert np.abs(res["q_mvar"][e] - o.res.q_mvar).max() < 1e-9
    assert bool(term[3])
    env.close()
