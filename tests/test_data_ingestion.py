"""pandapower tables -> NetSpec (mapdn_amd.data.from_pandapower; reference loader voltage_control_env.py:400-405):
transformers (T -> pi, tap changer), switches, load / sgen scaling and in_service, shunts, loud refusals.
pandapower itself is absent, so the transformer conversion is pinned on (a) the nameplate definitions
(no-load losses / current, short-circuit voltage), (b) an explicit three-node T circuit solved by the oracle: the
wye-delta conversion is an exact circuit identity, so both must give the same terminal voltages."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

import importlib.util                                    # noqa: E402
# the stub's table container (plain attribute dict, no arithmetic), loaded by path so that `import pandapower` keeps
# failing for tests/test_pandapower_pin.py's importorskip
_spec = importlib.util.spec_from_file_location(
    "_pp_stub_auxiliary", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "pp_stub", "pandapower", "auxiliary.py"))
_aux = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_aux)
pandapowerNet = _aux.pandapowerNet

from mapdn_amd.data import from_pandapower, trafo_to_pi  # noqa: E402
from mapdn_amd.netspec import NetSpec                    # noqa: E402
from oracle.pp_restated import make_ybus, runpp_restated  # noqa: E402

TRAFO = dict(hv_bus=0, lv_bus=1, sn_mva=25.0, vn_hv_kv=110.0, vn_lv_kv=20.0, vk_percent=12.0, vkr_percent=0.41,
             pfe_kw=14.0, i0_percent=0.07, shift_degree=150.0, tap_side="hv", tap_neutral=0, tap_min=-9, tap_max=9,
             tap_step_percent=1.5, tap_step_degree=0.0, tap_pos=-2, tap_phase_shifter=False, parallel=1, df=1.0, in_service=True)


def substation_net(trafo=TRAFO, sn_mva=10.0):
    """110 kV grid bus -> 25 MVA 110/20 kV transformer -> small 20 kV feeder with two PV zones"""
    net = pandapowerNet()
    net["name"] = "substation"; net["sn_mva"] = sn_mva; net["f_hz"] = 50.0
    net["bus"] = pd.DataFrame({"name": [f"b{i}" for i in range(6)], "vn_kv": [110.0, 20.0, 20.0, 20.0, 20.0, 20.0], "type": "b",
                               "zone": ["main", "main", "zone1", "zone1", "zone2", "zone2"], "in_service": True})
    net["line"] = pd.DataFrame({"from_bus": [1, 2, 1, 4], "to_bus": [2, 3, 4, 5], "length_km": [2.0, 1.5, 3.0, 1.0],
                                "r_ohm_per_km": [0.2, 0.3, 0.25, 0.4], "x_ohm_per_km": [0.12, 0.1, 0.11, 0.1],
                                "c_nf_per_km": [250.0, 200.0, 240.0, 210.0], "g_us_per_km": 0.0, "max_i_ka": 0.4, "df": 1.0,
                                "parallel": 1, "type": "cs", "in_service": True})
    net["load"] = pd.DataFrame({"name": None, "bus": [2, 3, 4, 5, 5], "p_mw": [1.0, 0.8, 1.2, 0.5, 0.3], "q_mvar": [0.3, 0.2, 0.4, 0.1, 0.1],
                                "const_z_percent": 0.0, "const_i_percent": 0.0, "sn_mva": np.nan, "scaling": [1.0, 1.0, 0.9, 1.0, 1.0],
                                "in_service": [True, True, True, True, False], "type": "wye"})
    net["sgen"] = pd.DataFrame({"name": ["zone1", "zone2"], "bus": [3, 5], "p_mw": [1.5, 0.7], "q_mvar": [0.2, -0.1], "sn_mva": np.nan,
                                "scaling": [1.0, 0.5], "in_service": True, "type": "PV", "current_source": True})
    net["ext_grid"] = pd.DataFrame({"name": [None], "bus": [0], "vm_pu": [1.02], "va_degree": [0.0], "in_service": [True]})
    net["trafo"] = pd.DataFrame([trafo])
    net["shunt"] = pd.DataFrame({"bus": [4], "p_mw": [0.0], "q_mvar": [-0.25], "vn_kv": [20.0], "step": [2], "max_step": [3], "in_service": [True]})
    net["switch"] = pd.DataFrame({"bus": [4], "element": [3], "et": ["l"], "type": ["LBS"], "closed": [True]})
    return net


def pre_pi(t, vn_lv_bus, net_sn):
    """the T-model values of trafo_to_pi before the wye-delta step (same restated formulas)"""
    tapped_hv = t["vn_hv_kv"] * (1 + (t["tap_pos"] - t["tap_neutral"]) * t["tap_step_percent"] / 100) if t["tap_side"] == "hv" else t["vn_hv_kv"]
    tapped_lv = t["vn_lv_kv"] * (1 + (t["tap_pos"] - t["tap_neutral"]) * t["tap_step_percent"] / 100) if t["tap_side"] == "lv" else t["vn_lv_kv"]
    tap_lv = (tapped_lv / vn_lv_bus) ** 2 * net_sn
    z = t["vk_percent"] / 100 / t["sn_mva"] * tap_lv
    r = t["vkr_percent"] / 100 / t["sn_mva"] * tap_lv
    x = np.sqrt(z * z - r * r)
    base_r = vn_lv_bus ** 2 / net_sn
    pfe = t["pfe_kw"] * 1e-3
    b_real = pfe / t["vn_lv_kv"] ** 2 * base_r
    b_img = np.sqrt(max((t["i0_percent"] / 100 * t["sn_mva"]) ** 2 - pfe ** 2, 0.0)) * base_r / t["vn_lv_kv"] ** 2
    y = (-1j * b_real - b_img) / (tapped_lv / t["vn_lv_kv"]) ** 2
    return r, x, y, tapped_hv, tapped_lv


def test_transformer_nameplate_definitions():
    """no-load: P = pfe_kw, |S| = i0% * sn (at rated lv voltage); short circuit: |z| = vk% on the transformer's own base"""
    t = dict(TRAFO, tap_pos=0)
    sn = 10.0
    r, x, y, _, _ = pre_pi(t, 20.0, sn)
    ym = 1j * y                                            # magnetising admittance (makeYbus: Ytt = Ys + 1j*BR_B/2, both halves)
    s0 = np.conj(ym) * sn                                  # MVA drawn at 1 p.u.
    assert abs(s0.real - t["pfe_kw"] * 1e-3) < 1e-12
    assert abs(abs(s0) - t["i0_percent"] / 100 * t["sn_mva"]) < 1e-12 and s0.imag > 0      # inductive
    assert abs(abs(r + 1j * x) * t["sn_mva"] / sn - t["vk_percent"] / 100) < 1e-15
    assert abs(r * t["sn_mva"] / sn - t["vkr_percent"] / 100) < 1e-15


@pytest.mark.parametrize("variant", ["hv_tap", "lv_tap", "offnominal_bus_kv", "parallel2", "no_magnetising"])
def test_trafo_pi_equals_the_explicit_t_circuit(variant):
    t = dict(TRAFO)
    vn_lv_bus = 20.0
    if variant == "lv_tap":
        t.update(tap_side="lv", tap_pos=3)
    if variant == "offnominal_bus_kv":
        t.update(vn_lv_kv=21.0, vn_hv_kv=115.0)
    if variant == "parallel2":
        t.update(parallel=2)
    if variant == "no_magnetising":
        t.update(pfe_kw=0.0, i0_percent=0.0)
    pnet = substation_net(t)
    a = from_pandapower(pnet)
    assert a.n_branch_pu == 1 and a.br_from_bus[0] == 0 and a.br_to_bus[0] == 1
    assert a.br_shift_deg[0] == 0.0                        # no line touches a bus above 70 kV: angles are not calculated
    assert abs(a.br_ratio[0] - pre_pi(t, vn_lv_bus, 10.0)[3] / 110.0 / (pre_pi(t, vn_lv_bus, 10.0)[4] / 20.0)) < 1e-15
    # explicit T: hv --[za, ideal ratio at the hv side]-- star --[zb]-- lv, magnetising shunt at the star point
    r, x, y, _, _ = pre_pi(t, vn_lv_bus, 10.0)
    par = t["parallel"]
    r, x, y = r / par, x / par, y * par
    star = a.n_bus
    ym = 1j * y
    b = NetSpec(name="T", bus_vn_kv=np.append(a.bus_vn_kv, 20.0), bus_zone=np.append(a.bus_zone, 0),
                line_from_bus=a.line_from_bus, line_to_bus=a.line_to_bus, line_r_ohm_per_km=a.line_r_ohm_per_km,
                line_x_ohm_per_km=a.line_x_ohm_per_km, line_c_nf_per_km=a.line_c_nf_per_km, line_g_us_per_km=a.line_g_us_per_km,
                line_length_km=a.line_length_km, line_parallel=a.line_parallel, line_in_service=a.line_in_service,
                load_bus=a.load_bus, sgen_bus=a.sgen_bus, sgen_zone=a.sgen_zone, ext_grid_bus=0, ext_grid_vm_pu=a.ext_grid_vm_pu,
                sn_mva=a.sn_mva, f_hz=50.0, load_scaling=a.load_scaling, sgen_scaling=a.sgen_scaling,
                br_from_bus=[0, star], br_to_bus=[star, 1], br_r_pu=[r / 2, r / 2], br_x_pu=[x / 2, x / 2], br_b_pu=[0, 0],
                br_ratio=[a.br_ratio[0], 1.0], br_shift_deg=[0, 0],
                shunt_bus=np.append(a.shunt_bus, star), shunt_p_mw=np.append(a.shunt_p_mw, ym.real * a.sn_mva),
                shunt_q_mvar=np.append(a.shunt_q_mvar, -ym.imag * a.sn_mva))
    pl, ql = pnet.load["p_mw"].to_numpy(), pnet.load["q_mvar"].to_numpy()
    ps, qs = pnet.sgen["p_mw"].to_numpy(), pnet.sgen["q_mvar"].to_numpy()
    ra, rb = runpp_restated(a, pl, ql, ps, qs), runpp_restated(b, pl, ql, ps, qs)
    assert ra.converged and rb.converged
    assert np.abs(ra.V - rb.V[:a.n_bus]).max() < 1e-9       # (each Newton solve stops at 1e-8 MVA mismatch)
    # the slack supplies the same power either way (losses of the T circuit == losses of its pi equivalent)
    assert abs(ra.p_mw[0] - rb.p_mw[0]) < 1e-7 and abs(ra.q_mvar[0] - rb.q_mvar[0]) < 1e-7      # (both solves stop at 1e-8 MVA)
    if variant != "no_magnetising":
        assert a.br_g_pu[0] > 0 and a.br_b_pu[0] < 0        # iron losses, inductive magnetising current


def hv_line_net():
    """substation_net plus a second 110 kV bus fed over a 110 kV line: runpp's calculate_voltage_angles='auto' turns on"""
    pnet = substation_net()
    pnet["bus"] = pd.concat([pnet["bus"], pd.DataFrame({"name": ["b6"], "vn_kv": [110.0], "type": "b", "zone": ["main"], "in_service": True})],
                            ignore_index=True)
    pnet["line"] = pd.concat([pnet["line"], pd.DataFrame({"from_bus": [6], "to_bus": [0], "length_km": [30.0], "r_ohm_per_km": [0.06],
                                                          "x_ohm_per_km": [0.4], "c_nf_per_km": [9.0], "g_us_per_km": 0.0, "max_i_ka": 0.6, "df": 1.0,
                                                          "parallel": 1, "type": "ol", "in_service": True})], ignore_index=True)
    pnet["ext_grid"] = pd.DataFrame({"name": [None], "bus": [6], "vm_pu": [1.02], "va_degree": [0.0], "in_service": [True]})
    return pnet


def test_hv_nets_are_refused_because_runpp_would_start_from_a_dc_power_flow():
    """VERDICT r4 missing 3: a line at a bus above 70 kV makes runpp's defaults (voltage_control_env.py:557) use calculate_voltage_angles
    = True AND init_va_degree = 'dc'.  The solvers here start flat -> refused by name; hv_init='flat' converts (phase shift applied), and
    the oracle's two starts (init='flat' / 'dc': oracle.pp_restated.dc_angles) reach the same voltages, the DC start in no more iterations."""
    pnet = hv_line_net()
    with pytest.raises(NotImplementedError, match="DC power flow"):
        from_pandapower(pnet)
    with pytest.raises(ValueError):
        from_pandapower(pnet, hv_init="dc")
    a = from_pandapower(pnet, hv_init="flat")
    assert a.br_shift_deg[0] == 150.0
    pl, ql = pnet.load["p_mw"].to_numpy(), pnet.load["q_mvar"].to_numpy()
    ps, qs = pnet.sgen["p_mw"].to_numpy(), pnet.sgen["q_mvar"].to_numpy()
    from oracle.pp_restated import dc_angles, bus_demand
    th = dc_angles(a, bus_demand(a, pl, ql, ps, qs)[0])
    assert abs(np.degrees(th[1]) + 150.0) < 5.0 and th[6] == 0.0   # the DC start carries the 150 degree vector group ...
    flat, dc = runpp_restated(a, pl, ql, ps, qs, init="flat"), runpp_restated(a, pl, ql, ps, qs, init="dc")
    assert dc.converged and dc.iterations <= 4
    assert not flat.converged                                      # ... and a flat start 150 degrees away from it does not even converge:
    # this is why the refusal is not a formality (the product would report an unsolvable step, -200, where pandapower solves)
    flat_net = hv_line_net(); flat_net["trafo"].loc[0, "shift_degree"] = 0.0
    b = from_pandapower(flat_net, hv_init="flat")
    f0, d0 = runpp_restated(b, pl, ql, ps, qs, init="flat"), runpp_restated(b, pl, ql, ps, qs, init="dc")
    assert f0.converged and d0.converged and np.abs(f0.V - d0.V).max() < 1e-9 and d0.iterations <= f0.iterations
    assert np.abs(np.abs(d0.V) - np.abs(dc.V)).max() < 1e-9        # the phase shift rotates the lv side, magnitudes are the same


def test_bus_bus_switch_with_impedance_and_mixed_voltage_groups_are_refused():
    """a closed bus-bus switch with z_ohm > 0 is an impedance branch in pandapower 2.x, not a fused bus; fused
    buses of different vn_kv have no single per-unit base.  Both were converted silently before."""
    pnet = substation_net()
    pnet["switch"] = pd.DataFrame({"bus": [2, 4], "element": [3, 3], "et": ["b", "l"], "type": ["CB", "LBS"], "closed": [True, True], "z_ohm": [0.0, 0.0]})
    pnet["line"] = pnet["line"].drop(index=1).reset_index(drop=True)           # (2-3 is now a switch, not a line)
    pnet["switch"].loc[1, "element"] = 2
    ok = from_pandapower(pnet)
    assert ok.has_fused_buses and ok.bus_alias[3] == 2
    bad = substation_net(); bad["line"] = pnet["line"]
    bad["switch"] = pnet["switch"].copy(); bad["switch"].loc[0, "z_ohm"] = 0.05
    with pytest.raises(NotImplementedError, match="z_ohm"):
        from_pandapower(bad)
    bad["switch"].loc[0, "z_ohm"] = 0.0
    bad["bus"].loc[3, "vn_kv"] = 10.0
    with pytest.raises(NotImplementedError, match="vn_kv"):
        from_pandapower(bad)


def test_scaling_in_service_shunt_steps_and_switches():
    pnet = substation_net()
    a = from_pandapower(pnet)
    assert np.array_equal(a.load_scaling, [1.0, 1.0, 0.9, 1.0, 0.0])          # scaling * in_service
    assert np.array_equal(a.sgen_scaling, [1.0, 0.5])
    assert a.shunt_q_mvar[0] == -0.5 and a.shunt_p_mw[0] == 0.0               # q_mvar * step
    assert a.line_in_service.all()
    # An OPEN line switch: pandapower (neglect_open_switch_branches=False) keeps the line energised from its closed end, so it
    # still draws its charging current.  Dropping the line is exact only without shunt terms or when both ends are open;
    # anything else is refused.
    pnet["switch"].loc[0, "closed"] = False
    with pytest.raises(NotImplementedError, match="open switch at one end"):
        from_pandapower(pnet)
    both = substation_net()
    f3, t3 = int(both.line["from_bus"].iloc[3]), int(both.line["to_bus"].iloc[3])
    both["switch"] = pd.DataFrame({"bus": [f3, t3], "element": [3, 3], "et": ["l", "l"], "type": ["LBS", "LBS"], "closed": [False, False]})
    assert list(from_pandapower(both).line_in_service) == [1, 1, 1, 0]           # open at both ends: out
    bare = substation_net()
    bare["switch"].loc[0, "closed"] = False
    bare.line.loc[bare.line.index[3], ["c_nf_per_km"]] = 0.0
    if "g_us_per_km" in bare.line:
        bare.line.loc[bare.line.index[3], ["g_us_per_km"]] = 0.0
    b = from_pandapower(bare)
    assert list(b.line_in_service) == [1, 1, 1, 0]                               # no shunt terms: nothing flows, out
    # scaled elements enter the power flow scaled: compare with hand-scaled inputs on an unscaled copy
    pl, ql = pnet.load["p_mw"].to_numpy(), pnet.load["q_mvar"].to_numpy()
    ps, qs = pnet.sgen["p_mw"].to_numpy(), pnet.sgen["q_mvar"].to_numpy()
    plain = a.copy(); plain.load_scaling[:] = 1.0; plain.sgen_scaling[:] = 1.0
    r1 = runpp_restated(a, pl, ql, ps, qs)
    r2 = runpp_restated(plain, pl * a.load_scaling, ql * a.load_scaling, ps * a.sgen_scaling, qs * a.sgen_scaling)
    assert np.abs(r1.V - r2.V).max() < 1e-14
    assert abs(r1.p_mw[5] - (0.5 * 1.0 - 0.7 * 0.5)) < 1e-12                   # res_bus: scaled element powers


def test_what_cannot_be_represented_is_refused():
    for mutate, msg in [
        (lambda n: n["load"].__setitem__("const_z_percent", 30.0), "const_z_percent"),
        (lambda n: n["load"].__setitem__("const_i_percent", 10.0), "const_i_percent"),
        (lambda n: n.__setitem__("gen", pd.DataFrame({"bus": [3], "p_mw": [1.0], "vm_pu": [1.0], "in_service": [True]})), "net.gen"),
        (lambda n: n["trafo"].__setitem__("tap_step_degree", 2.0), "tap_step_degree"),
        (lambda n: n["trafo"].__setitem__("tap_phase_shifter", True), "tap_phase_shifter"),
        (lambda n: n["bus"].__setitem__("in_service", [True] * 5 + [False]), "out-of-service buses"),
    ]:
        pnet = substation_net()
        mutate(pnet)
        with pytest.raises(NotImplementedError, match=msg):
            from_pandapower(pnet)


def test_closed_bus_bus_switch_becomes_a_bus_alias():
    """round 4: bus fusion is converted (NetSpec.bus_alias), no longer refused; an open bus-bus switch changes nothing"""
    pnet = substation_net()
    pnet["switch"] = pd.DataFrame({"bus": [2, 4], "element": [3, 5], "et": ["b", "b"], "type": ["CB", "CB"], "closed": [True, False]})
    a = from_pandapower(pnet)
    want = np.arange(a.n_bus); want[3] = 2
    assert np.array_equal(a.bus_alias, want)
    assert not from_pandapower(substation_net()).has_fused_buses


def test_host_plan_accepts_the_converted_net_and_builds_the_same_ybus():
    import ctypes as C
    from mapdn_amd import _lib
    a = from_pandapower(substation_net())
    lib = _lib.load()
    cnet, keep = _lib.make_cnetspec(a)
    ccfg = _lib.make_cconfig(dict(episode_limit=240, action_scale=0.8, action_bias=0.0))
    h = C.c_void_p()
    assert lib.mapdn_create(C.byref(cnet), C.byref(ccfg), 4, -1, C.byref(h)) == 0, lib.mapdn_last_error(None)
    out = np.zeros((a.n_bus, a.n_bus, 2))
    assert lib.mapdn_get_ybus_dense(h, _lib._p(out, _lib._pd)) == 0
    yo = make_ybus(a)[0].toarray()
    assert np.abs(out[..., 0] + 1j * out[..., 1] - yo).max() <= 1e-12 * np.abs(yo).max()
    lib.mapdn_destroy(h)


@pytest.mark.gpu
def test_env_on_the_converted_substation_net_matches_the_oracle_env():
    """trafo pi branch with iron losses + scaled / out-of-service elements through the whole GPU step"""
    import torch
    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.netspec import Profiles
    from oracle.env_restated import INFO_KEYS, VoltageControlOracle
    net = from_pandapower(substation_net())
    rng = np.random.default_rng(0)
    T = 1500
    prof = Profiles(pv=rng.uniform(0, 1.5, (T, 2)), load_p=rng.uniform(0.1, 1.2, (T, 5)), load_q=rng.uniform(0.0, 0.4, (T, 5)), time_delta_min=3)
    a = dict(episode_limit=240, action_scale=0.8, action_bias=0.0, voltage_barrier_type="bowl", seed=2)
    B = 6
    env = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(B)]
    obs, state = env.reset()
    for e, o in enumerate(oracles):
        oo, os_ = o.reset()
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9 and np.abs(os_ - state[e].cpu().numpy()).max() < 1e-7
    for t in range(4):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        res = env.results(); obs = env.get_obs()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item())
            assert max(abs(io[k] - info[e, c].item()) for c, k in enumerate(INFO_KEYS)) < 1e-9
            assert np.abs(res["vm_pu"][e].cpu().numpy() - o.res.vm_pu).max() < 1e-9
            assert np.abs(res["p_mw"][e].cpu().numpy() - o.res.p_mw).max() < 1e-9
            assert np.abs(np.array(o.get_obs()) - obs[e].cpu().numpy()).max() < 1e-9
    env.close()


def test_non_consecutive_bus_indices_are_mapped_to_sorted_positions():
    """pandapower bus indices need not be 0..n-1 (pd2ppc maps them through a lookup; the env reads results via sort_index)"""
    ref = from_pandapower(substation_net())
    pnet = substation_net()
    new = np.array([40, 3, 17, 25, 8, 99])                      # old bus i -> new label new[i]
    order = np.argsort(new)                                      # sorted label order = position order
    pnet["bus"].index = new
    for tab, cols in (("line", ("from_bus", "to_bus")), ("load", ("bus",)), ("sgen", ("bus",)), ("ext_grid", ("bus",)),
                      ("trafo", ("hv_bus", "lv_bus")), ("shunt", ("bus",)), ("switch", ("bus",))):
        for c in cols:
            pnet[tab][c] = new[pnet[tab][c].to_numpy()]
    got = from_pandapower(pnet)
    pos = np.empty(6, np.int64); pos[order] = np.arange(6)       # old bus i sits at position pos[i]
    assert np.array_equal(got.line_from_bus, pos[ref.line_from_bus]) and np.array_equal(got.line_to_bus, pos[ref.line_to_bus])
    assert np.array_equal(got.load_bus, pos[ref.load_bus]) and np.array_equal(got.sgen_bus, pos[ref.sgen_bus])
    assert got.ext_grid_bus == pos[ref.ext_grid_bus] and np.array_equal(got.br_from_bus, pos[ref.br_from_bus])
    assert np.array_equal(got.bus_vn_kv, ref.bus_vn_kv[order]) and np.array_equal(got.bus_zone, ref.bus_zone[order])
    pl, ql = pnet.load["p_mw"].to_numpy(), pnet.load["q_mvar"].to_numpy()
    ps, qs = pnet.sgen["p_mw"].to_numpy(), pnet.sgen["q_mvar"].to_numpy()
    ra, rb = runpp_restated(ref, pl, ql, ps, qs), runpp_restated(got, pl, ql, ps, qs)
    assert np.abs(ra.vm_pu[order] - rb.vm_pu).max() < 1e-12      # same physics, buses reported in sorted-label order


# ------------------------------------------------------------------------------------------------ model.p without pandapower
def _same_netspec(a, b):
    import dataclasses
    for f in dataclasses.fields(a):
        x, y = getattr(a, f.name), getattr(b, f.name)
        assert np.array_equal(np.asarray(x), np.asarray(y)), f.name


def _write_pp2_pickle(net, path):
    """what pandapower 2.x `to_pickle` writes (io_utils.to_dict_with_coord_transform): a plain dict, every table as
    {"DF": DataFrame.to_dict("split"), "dtypes": {column: numpy dtype}}; protocol 2"""
    import pickle
    save = {}
    for k, v in net.items():
        save[k] = {"DF": v.to_dict("split"), "dtypes": {c: dt for c, dt in zip(v.columns, v.dtypes)}} if isinstance(v, pd.DataFrame) else v
    save["version"] = "2.7.0"
    save["std_types"] = {"line": {}, "trafo": {}, "trafo3w": {}}
    save["_options"] = {"calculate_voltage_angles": "auto"}
    with open(path, "wb") as f:
        pickle.dump(save, f, protocol=2)


def test_model_p_reads_without_pandapower_dict_of_tables_form(tmp_path):
    """the file layout of pandapower 2.x to_pickle (the reference's model.p, voltage_control_env.py:400-405), read as data"""
    from mapdn_amd.data import read_pandapower_pickle
    pnet = substation_net()
    p = str(tmp_path / "model.p")
    _write_pp2_pickle(pnet, p)
    got = read_pandapower_pickle(p)
    assert got.sn_mva == 10.0 and list(got.bus["zone"]) == list(pnet.bus["zone"]) and got.line["from_bus"].dtype == pnet.line["from_bus"].dtype
    _same_netspec(from_pandapower(got), from_pandapower(pnet))


def test_model_p_reads_without_pandapower_object_form(tmp_path):
    """a net pickled as an object: class path pandapower.auxiliary.pandapowerNet (written in a subprocess under the stand-in
    package, so that `import pandapower` keeps failing here), tables = real pandas DataFrames.  The unpickler maps the class
    to an inert attribute dict; no pandapower module is imported."""
    import subprocess
    from mapdn_amd.data import InertNet, read_pandapower_pickle, save_netspec
    from mapdn_amd.netspec import make_case
    net, _ = make_case("case33")
    npz, p = str(tmp_path / "netspec.npz"), str(tmp_path / "model.p")
    save_netspec(net, npz)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, pickle; sys.path[:0] = [%r, %r]; import pandapower as pp; net = pp.from_pickle(%r);"
            "assert type(net).__module__ == 'pandapower.auxiliary';"
            "net.pop('_branch_pu'); pickle.dump(net, open(%r, 'wb'), protocol=4)" % (root, os.path.join(root, "oracle", "pp_stub"), p, p))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert b"pandapower.auxiliary" in open(p, "rb").read()[:4000] or b"pandapower.auxiliary" in open(p, "rb").read()
    assert "pandapower" not in sys.modules
    got = read_pandapower_pickle(p)
    assert isinstance(got, InertNet) and "pandapower" not in sys.modules
    a = from_pandapower(got)
    for k in ("bus_vn_kv", "bus_zone", "line_from_bus", "line_to_bus", "line_r_ohm_per_km", "line_x_ohm_per_km", "line_c_nf_per_km",
              "line_length_km", "load_bus", "sgen_bus", "sgen_zone"):
        assert np.array_equal(getattr(a, k), getattr(net, k)), k
    assert a.ext_grid_bus == net.ext_grid_bus and a.sn_mva == net.sn_mva


@pytest.mark.parametrize("form", ["to_pickle_proto2", "object_proto3", "object_proto2"])
def test_model_p_as_the_pinned_stack_writes_it(form, tmp_path):
    """the real model.p (voltage_control_env.py:403-404) was written by Python 3.7 / numpy 1.19.5 / pandas
    1.1.3 / pandapower 2.7.0 (environment.yml:66,125-134), the pickles of the other tests by today's pandas.  tests/pinned_stack_pickle.py
    ASSEMBLES the bytes that stack would emit — opcode by opcode, old module paths (pandas.core.indexes.numeric.Int64Index,
    numpy.core.multiarray._reconstruct), old state layouts (BlockManager's "0.14.1" dict, NDFrame's _mgr / _typ / attrs), bytes as
    _codecs.encode under protocol 2 — for both on-disk forms, with an EMPTY table and a RangeIndex among them.  The restricted
    unpickler must read them into the same NetSpec, every global on its allow-list, without importing a module that no longer exists."""
    from mapdn_amd import data
    from mapdn_amd.data import read_pandapower_pickle
    from tests import pinned_stack_pickle as psp
    pnet = substation_net()
    items = dict(pnet)
    items["ward"] = pd.DataFrame({"bus": np.zeros(0, dtype=np.int64), "ps_mw": np.zeros(0), "in_service": np.zeros(0, dtype=bool), "name": np.zeros(0, dtype=object)})
    items.update(version="2.7.0", std_types={"line": {"NAYY 4x50 SE": {"r_ohm_per_km": 0.642, "type": "cs"}}, "trafo": {}, "trafo3w": {}},
                 _options={"calculate_voltage_angles": "auto"}, converged=False, user_pf_options={})
    if form == "to_pickle_proto2":
        blob, want = psp.to_pickle_form(items), psp.GLOBALS_A
    elif form == "object_proto3":
        blob, want = psp.object_form(items, range_index_tables=("ext_grid",), proto=3), psp.GLOBALS_B
    else:
        blob = psp.object_form(items, range_index_tables=("ext_grid",), proto=2)
        want = (psp.GLOBALS_B - {("builtins", "slice")}) | psp.GLOBALS_B2
    assert blob[:2] == bytes([0x80, 3 if form == "object_proto3" else 2])
    used = psp.globals_in(blob)
    assert used == want, (used ^ want)
    allowed = data._NUMPY_OK | data._MISC_OK | data._PANDAS_OK | data._PANDAS_LEGACY | {("builtins", b) for b in data._BUILTINS_OK}
    assert all(g in allowed or g[0].startswith("pandapower") for g in used), [g for g in used if g not in allowed]
    p = str(tmp_path / "model.p")
    open(p, "wb").write(blob)
    got = read_pandapower_pickle(p)
    assert "pandapower" not in sys.modules and "pandas.core.indexes.numeric" not in sys.modules
    assert got.sn_mva == 10.0 and got.version == "2.7.0" and len(got.ward) == 0 and list(got.ward.columns) == ["bus", "ps_mw", "in_service", "name"]
    for name in ("bus", "line", "load", "sgen", "ext_grid", "trafo", "shunt", "switch"):
        a, b = got[name], pnet[name]
        assert list(a.columns) == list(b.columns) and list(a.index) == list(b.index), name
        for c in b.columns:
            assert a[c].dtype == b[c].dtype, (name, c, a[c].dtype, b[c].dtype)
            assert all((x == y) or (x != x and y != y) or (x is None and y is None) for x, y in zip(a[c].tolist(), b[c].tolist())), (name, c)
    _same_netspec(from_pandapower(got), from_pandapower(pnet))


def test_model_p_with_code_in_it_is_refused(tmp_path):
    """a pickle is a program: anything that is not a numpy / pandas / builtin data class is refused, pandapower classes are
    replaced by inert stand-ins (their code never runs)"""
    import pickle
    from mapdn_amd.data import read_pandapower_pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    p = str(tmp_path / "model.p")
    with open(p, "wb") as f:
        pickle.dump({"bus": Evil(), "sn_mva": 1.0}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        read_pandapower_pickle(p)
    with open(p, "wb") as f:                                    # builtins.eval must not resolve either
        f.write(b"cbuiltins\neval\n(S'1+1'\ntR.")
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        read_pandapower_pickle(p)
    with open(p, "wb") as f:                                    # a pre-2.0 net (kW / kVA columns) needs convert_format: refused
        pickle.dump({"bus": {"DF": {"index": [0], "columns": ["vn_kv"], "data": [[20.0]]}, "dtypes": {}}, "sn_kva": 1000.0}, f, protocol=2)
    with pytest.raises(NotImplementedError, match="convert_format"):
        read_pandapower_pickle(p)


def test_model_p_gadgets_inside_pandas_are_refused(tmp_path):
    """a module-prefix allow-list lets a hand-assembled pickle reach functions that pandas modules merely
    RE-EXPORT (import helpers, file handles, functools.partial) and chain them into os.system.  The allow-list is now explicit
    (module, name) pairs of data classes and reconstructors: every link of that chain must be refused, and a frame of every
    column kind must still load."""
    import pickle
    import pandas as pd
    from mapdn_amd.data import _restricted_unpickler, read_pandapower_pickle
    gadgets = [
        b"cpandas._libs.tslibs.timezones\nimport_optional_dependency\n(S'os'\ntR.",   # returns the os module
        b"cpandas.core.indexes.extension\n_inherit_from_data\n(S'system'\nS'x'\ntR.",  # builds method(self) -> getattr(self._data, ...)
        b"cpandas.core.frame\nget_handle\n(S'/tmp/mapdn_should_not_exist'\nS'w'\ntR.",  # truncates a file
        b"cpandas.core.frame\nfunctools\n.",                                           # a module re-exported as an attribute
        b"cpandas._libs.lib\nmap_infer\n.",
        b"cpandas.core.generic\npickle\n.",
        b"cpandas.core.series\n_coerce_method\n.",
        b"cpandas.io.pickle\nread_pickle\n.",
        b"cfunctools\npartial\n.",
        b"cos\nsystem\n.",
    ]
    p = str(tmp_path / "model.p")
    for g in gadgets:
        with open(p, "wb") as f:
            f.write(g)
        with pytest.raises(pickle.UnpicklingError, match="refused"):
            read_pandapower_pickle(p)
    assert not os.path.exists("/tmp/mapdn_should_not_exist")
    # ... while real tables of every column kind still load, in every protocol
    df = pd.DataFrame({"a": [1.0, 2.0], "b": [1, 2], "c": [True, False], "d": ["x", None], "e": pd.Categorical(["u", "v"]),
                       "f": pd.to_datetime(["2020-01-01", "2020-01-02"]), "g": pd.array([1, None], dtype="Int64"),
                       "h": pd.array(["s", None], dtype="string"), "i": pd.array([True, None], dtype="boolean"),
                       "j": pd.array([1.5, None], dtype="Float64")})
    for obj in (df, df.set_index(["a", "b"]), df.set_index("d"), df["a"], pd.Timestamp("2020-01-01")):
        for proto in (2, 4, 5):
            with open(p, "wb") as f:
                pickle.dump({"t": obj}, f, protocol=proto)
            with open(p, "rb") as f:
                got = _restricted_unpickler(f).load()["t"]
            assert got.equals(obj) if hasattr(obj, "equals") else got == obj


def test_load_scenario_opens_a_directory_with_model_p(tmp_path):
    """the reference's data directory layout: model.p + three CSVs, no netspec.npz, no pandapower"""
    from mapdn_amd.data import load_scenario, save_profiles_csv
    from mapdn_amd.netspec import make_case
    _, prof = make_case("case33", days=3)
    pnet = substation_net()
    d = str(tmp_path)
    _write_pp2_pickle(pnet, os.path.join(d, "model.p"))
    from mapdn_amd.netspec import Profiles
    small = Profiles(pv=prof.pv[:, :2], load_p=prof.load_p[:, :5], load_q=prof.load_q[:, :5], time_delta_min=3)
    save_profiles_csv(small, d)
    net, pr = load_scenario(d)
    assert net.n_bus == 6 and net.n_sgen == 2 and pr.pv.shape == small.pv.shape and net.n_branch_pu == 1
