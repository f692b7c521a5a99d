"""Power-flow pins whose expected values do NOT pass through oracle/pp_restated.py: published results
of the Baran-Wu 33-bus feeder after reconfiguration and in meshed operation, the Baran-Wu 69-bus feeder, and two- / three-bus
nets with line charging solved in closed form in SI units (tests/golden/literature_cases.py).  CPU: the oracle
(`runpp_restated`, i.e. the restatement of `pp.runpp`, voltage_control_env.py:557) against them; GPU (-m gpu): every HIP
solver that accepts the net (k_nr_tree, k_nr_sparse, k_nr_dense), through the C ABI, against them directly."""
import numpy as np
import pytest

from oracle.pp_restated import runpp_restated
from tests.golden.literature_cases import (BW33_PUBLISHED, BW69_PUBLISHED, bw33_variant, bw69, flat_profiles, three_bus,
                                           two_bus)

CF_TOL = 1e-9        # closed forms: p.u. / MW


def _check_published_33(kind, vm, pl_mw):
    loss, vmin, bus = BW33_PUBLISHED[kind]
    assert abs(pl_mw.sum() * 1e3 - loss) < 0.006, (kind, pl_mw.sum() * 1e3)          # published to 0.01 kW
    if vmin is not None:
        assert abs(vm.min() - vmin) < 5.1e-5 and int(vm.argmin()) + 1 == bus, (kind, vm.min(), vm.argmin() + 1)


def _check_published_69(vm, pl_mw, q_slack_mvar, q):
    P = BW69_PUBLISHED
    lo, hi = P["loss_kw"]
    assert lo <= pl_mw.sum() * 1e3 <= hi, pl_mw.sum() * 1e3
    q_loss = (-q_slack_mvar - q.sum()) * 1e3                                           # slack export - load = series losses (c = 0)
    assert P["loss_kvar"][0] <= q_loss <= P["loss_kvar"][1], q_loss
    assert abs(vm.min() - P["v_min"]) < 5.1e-5 and int(vm.argmin()) + 1 == P["v_min_bus"]


def _check_closed_form(exp, vm, va_deg, pl, p_bus, q_bus, tol=CF_TOL):
    v = vm * np.exp(1j * np.deg2rad(va_deg))
    assert np.abs(v - exp["V"]).max() < tol, np.abs(v - exp["V"]).max()               # complex voltages: magnitude AND angle
    assert np.abs(pl - exp["pl_mw"]).max() < tol
    assert abs(p_bus[0] - exp["p_slack_mw"]) < tol and abs(q_bus[0] - exp["q_slack_mvar"]) < tol
    if "p_bus2_mw" in exp:                                                             # res_bus at the shunt bus: p, q * vm^2
        assert abs(p_bus[1] - exp["p_bus2_mw"]) < tol and abs(q_bus[1] - exp["q_bus2_mvar"]) < tol


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("kind", ["base", "reconfigured", "meshed"])
def test_oracle_reproduces_published_33bus_results(kind):
    net, p, q = bw33_variant(kind)
    z = np.zeros(net.n_sgen)
    r = runpp_restated(net, p, q, z, z, cache=False)
    assert r.converged
    _check_published_33(kind, r.vm_pu, r.pl_mw)
    if kind == "reconfigured":                       # the opened switches carry nothing, the closed ties do
        assert (r.pl_mw[[6, 8, 13, 31, 36]] == 0).all() and (r.pl_mw[[32, 33, 34, 35]] > 0).all()


def test_oracle_reproduces_published_69bus_results():
    net, p, q = bw69()
    assert abs(p.sum() * 1e3 - BW69_PUBLISHED["p_total_kw"]) < 1e-6 and abs(q.sum() * 1e3 - BW69_PUBLISHED["q_total_kvar"]) < 1e-6
    r = runpp_restated(net, p, q, np.zeros(1), np.zeros(1), cache=False)
    assert r.converged and r.iterations == 4
    _check_published_69(r.vm_pu, r.pl_mw, r.q_mvar[0], q)


@pytest.mark.parametrize("make", [two_bus, three_bus])
def test_oracle_matches_closed_form_with_line_charging(make):
    """pins the per-unit rules (c_nf_per_km, g_us_per_km, parallel, length, shunt sign, non-unit slack) and the ANGLES"""
    net, p, q, exp = make()
    r = runpp_restated(net, p, q, np.zeros(1), np.zeros(1), cache=False)
    assert r.converged
    _check_closed_form(exp, r.vm_pu, r.va_degree, r.pl_mw, r.p_mw, r.q_mvar)      # (NR stops at 1e-8 MVA mismatch)
    assert abs(np.angle(exp["V"][-1])) > 1e-3 and abs(abs(exp["V"][-1]) - abs(exp["V"][0])) > 1e-3   # a non-trivial operating point


def test_closed_form_is_a_solution_of_the_circuit_equations():
    """the closed form itself: Kirchhoff at the load bus in SI units, residual at rounding level"""
    from tests.golden.literature_cases import TWO_BUS, _line_si
    _, _, _, exp = two_bus()
    c = TWO_BUS; L = c["line"]
    z, y = _line_si(L["r"], L["x"], L["c"], L["g"], L["length"], L["parallel"], c["f_hz"])
    v1, v2 = exp["V"] * c["vn_kv"]
    s = c["p_mw"] + 1j * c["q_mvar"]
    assert abs((v1 - v2) / z - v2 * y / 2 - np.conj(s / v2)) < 1e-12


# ------------------------------------------------------------------------------------------------ GPU: the HIP solvers
def _gpu_results(net, p, q, solver, monkeypatch):
    import torch
    from mapdn_amd.env import VoltageControlBatch
    for v in ("MAPDN_NR_SPARSE", "MAPDN_NR_DENSE"):
        monkeypatch.delenv(v, raising=False)
    if solver != "default":
        monkeypatch.setenv({"sparse": "MAPDN_NR_SPARSE", "dense": "MAPDN_NR_DENSE"}[solver], "1")
    prof = flat_profiles(net, p, q)
    env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=0.8, action_bias=0.0, voltage_barrier_type="l1"),
                              n_envs=3, device="cuda:0", obs_dtype=torch.float64)
    env.manual_reset(0, 0, 0)                        # solves the first profile row: the given loads, PV 0, q 0
    assert env.stats()["reset_failures"] == 0
    res = {k: v.cpu().numpy() for k, v in env.results().items()}
    for k, v in res.items():
        assert (v == v[0]).all(), k                  # the three envs agree bit for bit
    z = np.zeros((3, net.n_sgen))
    vm, va, it, cv = env.solve(np.tile(p, (3, 1)), np.tile(q, (3, 1)), z, z)
    assert cv.cpu().numpy().all()
    assert np.abs(vm.cpu().numpy()[0] - res["vm_pu"][0]).max() < 1e-13
    env.close()
    return {k: v[0] for k, v in res.items()}, int(it[0].item())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,solver", [("reconfigured", "default"), ("reconfigured", "sparse"), ("reconfigured", "dense"),
                                         ("meshed", "default"), ("meshed", "dense"), ("base", "sparse"), ("base", "dense")])
def test_hip_solvers_reproduce_published_33bus_results(kind, solver, monkeypatch):
    net, p, q = bw33_variant(kind)
    res, it = _gpu_results(net, p, q, solver, monkeypatch)
    _check_published_33(kind, res["vm_pu"], res["pl_mw"])


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["default", "sparse"])
def test_hip_solvers_reproduce_published_69bus_results(solver, monkeypatch):
    net, p, q = bw69()
    res, it = _gpu_results(net, p, q, solver, monkeypatch)
    assert it == 4
    _check_published_69(res["vm_pu"], res["pl_mw"], res["q_mvar"][0], q)


@pytest.mark.gpu
@pytest.mark.parametrize("make", [two_bus, three_bus])
@pytest.mark.parametrize("solver", ["default", "sparse", "dense"])
def test_hip_solvers_match_closed_form_with_line_charging(make, solver, monkeypatch):
    net, p, q, exp = make()
    res, it = _gpu_results(net, p, q, solver, monkeypatch)
    _check_closed_form(exp, res["vm_pu"], res["va_degree"], res["pl_mw"], res["p_mw"], res["q_mvar"])
