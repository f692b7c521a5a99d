"""The pin the oracle is missing here: wherever pandapower is installed, build the
NetSpec test feeders as pandapower nets, run the REAL `pp.runpp`, and compare the restated oracle — bus
voltages to 1e-9 p.u., res_line losses, slack injection and the iteration count.  pandapower cannot be
installed in the build container (no network), so this module is skipped there; it documents exactly
how `oracle/pp_restated.py` is to be pinned the moment the reference stack is available."""
import numpy as np
import pytest

pp = pytest.importorskip("pandapower")
if not hasattr(pp, "create_empty_network"):      # oracle/pp_stub on sys.path is not pandapower
    pytest.skip("the pandapower on sys.path is the test stub", allow_module_level=True)

from mapdn_amd.data import from_pandapower          # noqa: E402
from mapdn_amd.netspec import make_case             # noqa: E402
from oracle.pp_restated import runpp_restated       # noqa: E402


def to_pandapower(net, p_load, q_load, p_sgen, q_sgen):
    n = pp.create_empty_network(sn_mva=float(net.sn_mva), f_hz=float(net.f_hz))
    for b in range(net.n_bus):
        z = int(net.bus_zone[b])
        pp.create_bus(n, vn_kv=float(net.bus_vn_kv[b]), zone="main" if z == 0 else f"zone{z}", index=b)
    for i in range(net.line_from_bus.shape[0]):
        pp.create_line_from_parameters(
            n, int(net.line_from_bus[i]), int(net.line_to_bus[i]), length_km=float(net.line_length_km[i]),
            r_ohm_per_km=float(net.line_r_ohm_per_km[i]), x_ohm_per_km=float(net.line_x_ohm_per_km[i]),
            c_nf_per_km=float(net.line_c_nf_per_km[i]), g_us_per_km=float(net.line_g_us_per_km[i]), max_i_ka=10.0,
            parallel=int(net.line_parallel[i]), in_service=bool(net.line_in_service[i]))
    for j, b in enumerate(net.load_bus):
        pp.create_load(n, int(b), p_mw=float(p_load[j]), q_mvar=float(q_load[j]))
    for j, b in enumerate(net.sgen_bus):
        pp.create_sgen(n, int(b), p_mw=float(p_sgen[j]), q_mvar=float(q_sgen[j]), name=f"zone{int(net.sgen_zone[j])}")
    for j, b in enumerate(net.shunt_bus):
        pp.create_shunt(n, int(b), q_mvar=float(net.shunt_q_mvar[j]), p_mw=float(net.shunt_p_mw[j]))
    pp.create_ext_grid(n, int(net.ext_grid_bus), vm_pu=float(net.ext_grid_vm_pu))
    return n


@pytest.mark.parametrize("case", ["case33", "case141", "case322"])
def test_oracle_matches_real_pandapower(case):
    net, prof = make_case(case)
    if net.br_from_bus.shape[0]:
        pytest.skip("per-unit pi branches have no pandapower element in this converter")
    rng = np.random.default_rng(0)
    for _ in range(5):
        row = int(rng.integers(0, prof.n_rows))
        pv = prof.pv[row]
        q = rng.uniform(-0.8, 0.8, net.n_sgen) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
        n = to_pandapower(net, prof.load_p[row], prof.load_q[row], pv, q)
        pp.runpp(n)                                              # all defaults, as voltage_control_env.py:557
        r = runpp_restated(net, prof.load_p[row], prof.load_q[row], pv, q)
        rb = n.res_bus.sort_index()
        assert np.abs(rb.vm_pu.to_numpy() - r.vm_pu).max() < 1e-9
        assert np.abs(rb.va_degree.to_numpy() - r.va_degree).max() < 1e-7
        assert np.abs(rb.p_mw.to_numpy() - r.p_mw).max() < 1e-8 and np.abs(rb.q_mvar.to_numpy() - r.q_mvar).max() < 1e-8
        assert np.abs(n.res_line.sort_index().pl_mw.to_numpy() - r.pl_mw).max() < 1e-8
        assert int(n._ppc["iterations"]) == r.iterations
        back = from_pandapower(n)                                # and the converter round-trips the topology
        assert np.array_equal(back.line_from_bus, net.line_from_bus) and np.array_equal(back.sgen_zone, net.sgen_zone)


def test_transformer_net_matches_real_pandapower():
    """from_pandapower's restated trafo -> pi conversion (T model, tap changer, iron losses) and the element scaling /
    in_service handling against the real pd2ppc + runpp on a 110/20 kV substation feeder"""
    n = pp.create_empty_network(sn_mva=10.0)
    hv = pp.create_bus(n, vn_kv=110.0, zone="main")
    b = [pp.create_bus(n, vn_kv=20.0, zone=z) for z in ("main", "zone1", "zone1", "zone2", "zone2")]
    pp.create_transformer_from_parameters(n, hv, b[0], sn_mva=25.0, vn_hv_kv=110.0, vn_lv_kv=20.0, vk_percent=12.0, vkr_percent=0.41,
                                          pfe_kw=14.0, i0_percent=0.07, shift_degree=150.0, tap_side="hv", tap_neutral=0, tap_min=-9,
                                          tap_max=9, tap_step_percent=1.5, tap_pos=-2)
    for f, t, l, r, x, c in ((0, 1, 2.0, 0.2, 0.12, 250.0), (1, 2, 1.5, 0.3, 0.1, 200.0), (0, 3, 3.0, 0.25, 0.11, 240.0), (3, 4, 1.0, 0.4, 0.1, 210.0)):
        pp.create_line_from_parameters(n, b[f], b[t], length_km=l, r_ohm_per_km=r, x_ohm_per_km=x, c_nf_per_km=c, max_i_ka=0.4)
    for bus, p, q, sc, on in ((1, 1.0, 0.3, 1.0, True), (2, 0.8, 0.2, 1.0, True), (3, 1.2, 0.4, 0.9, True), (4, 0.5, 0.1, 1.0, True), (4, 0.3, 0.1, 1.0, False)):
        pp.create_load(n, b[bus], p_mw=p, q_mvar=q, scaling=sc, in_service=on)
    pp.create_sgen(n, b[2], p_mw=1.5, q_mvar=0.2, name="zone1")
    pp.create_sgen(n, b[4], p_mw=0.7, q_mvar=-0.1, name="zone2", scaling=0.5)
    pp.create_shunt(n, b[3], q_mvar=-0.25, p_mw=0.0, step=2, max_step=3)
    pp.create_ext_grid(n, hv, vm_pu=1.02)
    pp.runpp(n)
    net = from_pandapower(n)
    r = runpp_restated(net, n.load.p_mw.to_numpy(), n.load.q_mvar.to_numpy(), n.sgen.p_mw.to_numpy(), n.sgen.q_mvar.to_numpy())
    rb = n.res_bus.sort_index()
    assert np.abs(rb.vm_pu.to_numpy() - r.vm_pu).max() < 1e-9
    assert np.abs(rb.p_mw.to_numpy() - r.p_mw).max() < 1e-8 and np.abs(rb.q_mvar.to_numpy() - r.q_mvar).max() < 1e-8
    assert np.abs(n.res_line.sort_index().pl_mw.to_numpy() - r.pl_mw).max() < 1e-8
    assert int(n._ppc["iterations"]) == r.iterations


def test_tolerance_rule_on_sn_mva_not_one():
    """THE assertion that decides mapdn_env_config.tolerance_is_pu (missing item 1): how pandapower 2.7.0 turns
    runpp's tolerance_mva into the stopping rule of newtonpf.  On a net with sn_mva = 100 the two readings — ||F||inf <
    tolerance_mva / sn_mva (the restatement's default) or ||F||inf < tolerance_mva — stop at different iterations for a suitably
    coarse tolerance, while the converged voltages agree; the iteration count pandapower reports picks the reading.  If the
    SECOND assertion below is the one that holds, set tolerance_is_pu = 1 (oracle: runpp_restated(..., tolerance_is_pu=True);
    library: mapdn_env_config.tolerance_is_pu / tuning=dict(tolerance_is_pu=1)) — a one-line change on either side."""
    net, prof = make_case("case141")
    net.sn_mva = 100.0
    row = 1234
    pv = prof.pv[row]
    q = 0.3 * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    decided = None
    for tol in (1e-3, 1e-4, 1e-5, 1e-6, 1e-8):
        n = to_pandapower(net, prof.load_p[row], prof.load_q[row], pv, q)
        pp.runpp(n, tolerance_mva=tol)
        it_pp = int(n._ppc["iterations"])
        a = runpp_restated(net, prof.load_p[row], prof.load_q[row], pv, q, tolerance_mva=tol, tolerance_is_pu=False).iterations
        b = runpp_restated(net, prof.load_p[row], prof.load_q[row], pv, q, tolerance_mva=tol, tolerance_is_pu=True).iterations
        if a != b:
            assert it_pp in (a, b), (tol, it_pp, a, b)
            verdict = "divide" if it_pp == a else "as_is"
            assert decided in (None, verdict), "pandapower's iteration counts fit neither reading consistently"
            decided = verdict
    assert decided is not None, "no tolerance in the sweep separates the two readings on this net"
    assert decided == "divide", ("pandapower uses ||F||inf < tolerance_mva WITHOUT dividing by sn_mva: set tolerance_is_pu = 1 "
                                 "(mapdn_env_config / oracle.runpp_restated) — see this test's docstring")


def test_hv_net_dc_initialisation():
    """VERDICT r4 missing item 3: on a net with a line at a bus above 70 kV runpp's defaults switch calculate_voltage_angles on and
    start Newton from a DC power flow's angles (init='auto' -> init_va_degree='dc').  The oracle's init='dc' restates that start;
    the product solvers start flat and from_pandapower refuses such nets unless hv_init='flat' is passed (with this net's 150 degree
    vector group a flat start does not converge at all: tests/test_data_ingestion.py).  Decides: the voltages and the iteration
    count pandapower reports equal the oracle's with init='dc'."""
    n = pp.create_empty_network(sn_mva=10.0)
    hv = [pp.create_bus(n, vn_kv=110.0, zone="main") for _ in range(2)]
    b = [pp.create_bus(n, vn_kv=20.0, zone=z) for z in ("main", "zone1", "zone1")]
    pp.create_line_from_parameters(n, hv[0], hv[1], length_km=30.0, r_ohm_per_km=0.06, x_ohm_per_km=0.4, c_nf_per_km=9.0, max_i_ka=0.6)
    pp.create_transformer_from_parameters(n, hv[1], b[0], sn_mva=25.0, vn_hv_kv=110.0, vn_lv_kv=20.0, vk_percent=12.0, vkr_percent=0.41,
                                          pfe_kw=14.0, i0_percent=0.07, shift_degree=150.0)
    pp.create_line_from_parameters(n, b[0], b[1], length_km=2.0, r_ohm_per_km=0.2, x_ohm_per_km=0.12, c_nf_per_km=250.0, max_i_ka=0.4)
    pp.create_line_from_parameters(n, b[1], b[2], length_km=1.5, r_ohm_per_km=0.3, x_ohm_per_km=0.1, c_nf_per_km=200.0, max_i_ka=0.4)
    pp.create_load(n, b[1], p_mw=4.0, q_mvar=1.0)
    pp.create_load(n, b[2], p_mw=3.0, q_mvar=0.8)
    pp.create_sgen(n, b[2], p_mw=1.5, q_mvar=0.2, name="zone1")
    pp.create_ext_grid(n, hv[0], vm_pu=1.02)
    pp.runpp(n)
    with pytest.raises(NotImplementedError, match="DC power flow"):
        from_pandapower(n)
    net = from_pandapower(n, hv_init="flat")
    args = (n.load.p_mw.to_numpy(), n.load.q_mvar.to_numpy(), n.sgen.p_mw.to_numpy(), n.sgen.q_mvar.to_numpy())
    dc = runpp_restated(net, *args, init="dc")
    rb = n.res_bus.sort_index()
    assert np.abs(rb.vm_pu.to_numpy() - dc.vm_pu).max() < 1e-9 and np.abs(rb.va_degree.to_numpy() - dc.va_degree).max() < 1e-7
    assert dc.converged and int(n._ppc["iterations"]) == dc.iterations
