"""GPU (-m gpu) parity tests: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Tolerances: bus voltages <= 1e-9 p.u. (north_star bar: 1e-6), rewards/info
<= 1e-9, float64 obs/state <= 1e-9, integer artefacts (iterations, start rows, flags) exact."""
import numpy as np
import pytest
import torch

from mapdn_amd.env import VoltageControl, VoltageControlBatch
from mapdn_amd.netspec import make_case
from oracle import philox
from oracle.env_restated import INFO_KEYS, VoltageControlOracle
from oracle.pp_restated import residual_inf, runpp_restated
from tests.edge_rule import same_newton_count

pytestmark = pytest.mark.gpu

V_TOL = 1e-9
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}     # train.py:34-42


def args_for(case, **kw):
    a = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)
    a.update(kw)
    return a


def make(case, B, tuning=None, **kw):
    net, prof = make_case(case)
    env = VoltageControlBatch(net, prof, args_for(case, **kw), n_envs=B, device="cuda:0", obs_dtype=torch.float64, tuning=tuning)
    return net, prof, env


def test_native_library_is_loaded():
    import os
    from mapdn_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libmapdn_hip.so" in maps and os.path.exists(_lib.LIB_PATH)


@pytest.mark.parametrize("case", ["case33", "case141", "case322"])
def test_solve_only_matches_oracle(case):
    """pp.runpp parity on explicit load/PV inputs; B deliberately not a multiple of 64"""
    B = 70
    net, prof, env = make(case, B)
    rng = np.random.default_rng(3)
    rows = rng.integers(0, prof.n_rows, B)
    act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
    smax = prof.s_max()
    pl, ql, pv = prof.load_p[rows], prof.load_q[rows], prof.pv[rows]
    qs = act * np.sqrt(smax ** 2 - pv ** 2)
    vm, va, it, cv = env.solve(pl, ql, pv, qs)
    vm, va, it, cv = vm.cpu().numpy(), va.cpu().numpy(), it.cpu().numpy(), cv.cpu().numpy()
    assert cv.all()
    worst = 0.0
    for e in range(B):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert r.converged and same_newton_count(net, (pl[e], ql[e], pv[e], qs[e]), it[e], cv[e], r)   # exact, or one step apart AT the tolerance
        worst = max(worst, np.abs(vm[e] - r.vm_pu).max(), np.abs(va[e] - r.va_degree).max() * np.pi / 180)
        # solver-independent certificate on the GPU answer
        v = vm[e] * np.exp(1j * va[e] * np.pi / 180)
        assert residual_inf(net, v, pl[e], ql[e], pv[e], qs[e]) < 1e-8 / net.sn_mva
    assert worst < V_TOL, worst
    env.close()


def test_ieee33_published_profile_on_gpu():
    """Known-answer pin that does not pass through the oracle: the HIP solver on the public Baran-Wu
    33-bus base case against the voltage profile tabulated in the literature (4 decimals), the
    202.68 kW / 135.14 kVAr losses and Vmin 0.9131 p.u. at bus 18."""
    from mapdn_amd.netspec import case33bw_base
    _, p, q = case33bw_base()                      # the literature loads; make_case adds 6 PVs to the same feeder
    net, prof, env = make("case33", 3)
    z = np.zeros((3, net.n_sgen))
    vm, va, it, cv = env.solve(np.tile(p, (3, 1)), np.tile(q, (3, 1)), z, z)
    vm = vm.cpu().numpy()
    published = np.array([1.0000, 0.9970, 0.9829, 0.9755, 0.9681, 0.9497, 0.9462, 0.9413, 0.9351, 0.9292, 0.9284,
                          0.9269, 0.9208, 0.9185, 0.9171, 0.9157, 0.9137, 0.9131, 0.9965, 0.9929, 0.9922, 0.9916,
                          0.9794, 0.9727, 0.9694, 0.9477, 0.9452, 0.9337, 0.9255, 0.9220, 0.9178, 0.9169, 0.9166])
    assert cv.cpu().numpy().all() and (it.cpu().numpy() == 4).all()
    for e in range(3):
        assert np.abs(vm[e] - published).max() <= 5.0e-5 + 1e-12
        assert int(vm[e].argmin()) + 1 == 18
    env.close()


def test_solve_nonconvergence_flag():
    """LoadflowNotConverged after 10 iterations (pandapower max_iteration='auto'), per env, while the
    other lanes of the same wavefront converge normally"""
    case, B = "case33", 64
    net, prof, env = make(case, B)
    row = 250
    pl = np.tile(prof.load_p[row], (B, 1)); ql = np.tile(prof.load_q[row], (B, 1))
    pv = np.tile(prof.pv[row], (B, 1)); qs = np.zeros((B, net.n_sgen))
    bad = [3, 17, 63]
    pl[bad] *= 40.0                                    # far beyond the feeder's loadability
    vm, va, it, cv = env.solve(pl, ql, pv, qs)
    it, cv = it.cpu().numpy(), cv.cpu().numpy()
    for e in range(B):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert same_newton_count(net, (pl[e], ql[e], pv[e], qs[e]), it[e], cv[e], r)
    assert not cv[bad].any() and (it[bad] == 10).all() and cv[[0, 1, 62]].all()
    env.close()


@pytest.mark.parametrize("case,barrier,tuning", [
    ("case33", "bowl", None), ("case141", "l1", None), ("case322", "courant_beltrami", None), ("case33", "l2", None), ("case33", "bump", None),
    # round 5: 4 envs per workgroup — the reward / commit epilogue on 64 resp. 32 workers, the per-env reductions by DPP row rotations
    ("case141", "bowl", dict(nr_waves=4, nr_lanes=4)), ("case322", "bowl", dict(nr_waves=2, nr_lanes=4)), ("case33", "l1", dict(nr_waves=1, nr_lanes=4))])
def test_episode_parity_with_noise(case, barrier, tuning):
    """reset -> steps with keyed noise: reward, terminated, 11 info values, obs, state, res_* tables"""
    B, T = 4, 8
    net, prof, env = make(case, B, tuning=tuning, voltage_barrier_type=barrier)
    if tuning:
        assert env.geometry()["lanes"] == 4
    a = args_for(case, voltage_barrier_type=barrier)
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(B)]
    obs, state = env.reset()
    starts = env.start_rows().cpu().numpy()
    for e, o in enumerate(oracles):
        oo, os_ = o.reset()
        assert o._episode_start == starts[e]                       # sampled start time: exact
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9
        assert np.abs(os_ - state[e].cpu().numpy()).max() < 1e-9
    rng = np.random.default_rng(5)
    for t in range(T):
        act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        obs, state, res = env.get_obs(), env.get_state(), env.results()
        lp, lq = env.loads()
        r, term, info = r.cpu().numpy(), term.cpu().numpy(), info.cpu().numpy()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e]) < 1e-9 and bool(term[e]) == to
            for c, k in enumerate(INFO_KEYS):
                assert abs(io[k] - info[e, c]) < 1e-9, (k, io[k], info[e, c])
            assert np.abs(res["vm_pu"][e].cpu().numpy() - o.res.vm_pu).max() < V_TOL
            assert np.abs(res["va_degree"][e].cpu().numpy() - o.res.va_degree).max() < 1e-7
            assert np.abs(res["p_mw"][e].cpu().numpy() - o.res.p_mw).max() < 1e-9
            assert np.abs(res["q_mvar"][e].cpu().numpy() - o.res.q_mvar).max() < 1e-9
            assert np.abs(res["pl_mw"][e].cpu().numpy() - o.res.pl_mw).max() < 1e-9
            assert np.abs(res["sgen_p"][e].cpu().numpy() - o.sgen_p).max() < 1e-12      # next row + noise
            assert np.abs(lp[e].cpu().numpy() - o.load_p).max() < 1e-12
            assert np.abs(lq[e].cpu().numpy() - o.load_q).max() < 1e-12
            assert np.abs(np.array(o.get_obs()) - obs[e].cpu().numpy()).max() < 1e-9
            assert np.abs(o.get_state() - state[e].cpu().numpy()).max() < 1e-7          # va in degrees
    env.close()


def test_manual_reset_and_termination():
    """manual_reset (no noise), profile off-by-one (Appendix B.1), terminate at steps >= episode_limit"""
    case, B = "case33", 3
    net, prof, env = make(case, B, episode_limit=5)
    o = VoltageControlOracle(net, prof, args_for(case, episode_limit=5), do_reset=False)
    env.manual_reset(3, 11, 7)
    o.manual_reset(3, 11, 7)
    assert (env.start_rows().cpu().numpy() == prof.start_row(3, 11, 7)).all()
    # reset_action draws differ per env id -> only env 0 equals the env_id=0 oracle; pv is noise-free for all
    assert np.abs(env.results()["sgen_p"].cpu().numpy() - prof.pv[prof.start_row(3, 11, 7) + 1]).max() == 0.0
    terms = []
    for t in range(4):
        z = torch.zeros(B, net.n_sgen, dtype=torch.float64, device="cuda:0")
        r, term, info = env.step(z, add_noise=False)
        ro, to, io = o.step(np.zeros(net.n_sgen), add_noise=False)
        assert abs(r[0].item() - ro) < 1e-9 and bool(term[0].item()) == to
        terms.append(bool(term[0].item()))
        want_row = prof.start_row(3, 11, 7) + max(1, t + 1)
        assert np.abs(env.results()["sgen_p"][0].cpu().numpy() - prof.pv[want_row]).max() == 0.0
    assert terms == [False, False, False, True]
    # frozen after termination: reward 0, terminated stays 1
    r, term, info = env.step(torch.zeros(B, net.n_sgen, dtype=torch.float64, device="cuda:0"))
    assert (r == 0).all() and term.all() and (info == 0).all()
    env.close()


def test_unsolvable_step_branch():
    """voltage_control_env.py:188-196 per env: rollback, reward-200, destroy=1, episode ends"""
    case, B = "case33", 6
    net, prof, env = make(case, B)
    a = args_for(case)
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(B)]
    env.reset()
    for o in oracles:
        o.reset()
    before = env.results()
    act = np.zeros((B, net.n_sgen))
    act[2] = -60.0
    act[4] = 55.0
    r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
    after = env.results()
    r, term, info = r.cpu().numpy(), term.cpu().numpy(), info.cpu().numpy()
    for e, o in enumerate(oracles):
        ro, to, io = o.step(act[e])
        assert abs(ro - r[e]) < 1e-9 and bool(term[e]) == to
        for c, k in enumerate(INFO_KEYS):
            assert abs(io[k] - info[e, c]) < 1e-9, (e, k)
    assert list(term) == [False, False, True, False, True, False]
    assert info[2, 10] == 1.0 and info[2, 3] == 0.0 and r[2] < -200
    for k in ("vm_pu", "p_mw", "q_mvar", "pl_mw", "sgen_q"):          # rolled back
        assert torch.equal(before[k][2], after[k][2]) and torch.equal(before[k][4], after[k][4])
    assert not torch.equal(before["vm_pu"][0], after["vm_pu"][0])
    # and the profile still advances on the restored net (:199)
    assert np.abs(after["sgen_p"][2].cpu().numpy() - oracles[2].sgen_p).max() < 1e-12
    env.close()


def test_f32_outputs_and_action_dtypes():
    case, B = "case141", 8
    net, prof, env = make(case, B)
    env.reset()
    act = torch.rand(B, net.n_sgen, device="cuda:0") * 1.2 - 0.6
    o64 = env.get_obs(torch.float64).clone()
    o32 = env.get_obs(torch.float32)
    assert o32.dtype == torch.float32 and o32.shape == (B, net.n_sgen, net.obs_size())
    assert torch.equal(o32, o64.float())
    s32 = env.get_state(torch.float32)
    assert torch.equal(s32, env.get_state(torch.float64).float())
    # f32 actions are promoted exactly like `q = constraint * f32_action` in numpy
    net2, prof2, env2 = make(case, B)
    env2.reset()
    r1, _, _ = env.step(act)
    r2, _, _ = env2.step(act.double())
    assert torch.equal(r1, r2)
    env.close(); env2.close()


def test_full_size_case141_properties():
    """BASELINE config: case141 at 4096 envs — size-independent properties + sampled oracle check"""
    case, B = "case141", 4096
    net, prof, env = make(case, B)
    obs0, _ = env.reset()
    assert env.stats()["reset_failures"] == 0
    # 64 of the 4096 envs (every 64th: one per workgroup row of lanes) replayed on the oracle, noise included
    watch = list(range(5, B, 64))
    assert len(watch) == 64
    oracles = {e: VoltageControlOracle(net, prof, args_for(case), env_id=e, do_reset=False) for e in watch}
    for e, o in oracles.items():
        oo, _ = o.reset()
        assert np.abs(np.array(oo) - obs0[e].cpu().numpy()).max() < 1e-9
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
    for t in range(3):
        act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * 0.6
        r, term, info = env.step(act)
        obs = env.get_obs().cpu().numpy(); vmw = env.results(("vm_pu",))["vm_pu"].cpu().numpy()
        acpu, rcpu, icpu = act.cpu().numpy(), r.cpu().numpy(), info.cpu().numpy()
        for e, o in oracles.items():
            ro, to, io = o.step(acpu[e])
            assert abs(ro - rcpu[e]) < 1e-9 and not to
            assert max(abs(io[k] - icpu[e, c]) for c, k in enumerate(INFO_KEYS)) < 1e-9
            assert np.abs(vmw[e] - o.res.vm_pu).max() < V_TOL
            assert np.abs(np.array(o.get_obs()) - obs[e]).max() < 1e-9
    st = env.stats()
    assert st["max_nr_iters"] <= 6 and 2.5 < st["mean_nr_iters"] < 5.5
    assert torch.isfinite(r).all() and not term.any() and (info[:, 10] == 0).all()
    res = env.results()
    lp, lq = env.loads()
    vm = res["vm_pu"].cpu().numpy(); va = res["va_degree"].cpu().numpy()
    assert (vm[:, 0] == 1.0).all() and (va[:, 0] == 0.0).all()
    # info consistency: average_voltage == mean(vm), total_line_loss == sum(pl)
    assert np.abs(info[:, 5].cpu().numpy() - vm.mean(1)).max() < 1e-12
    assert np.abs(info[:, 8].cpu().numpy() - res["pl_mw"].sum(1).cpu().numpy()).max() < 1e-10
    # power balance: slack injection == sum(bus demand) + line losses   (c_nf > 0: charging in pl too)
    p = res["p_mw"].cpu().numpy()
    assert np.abs(p.sum(1) + res["pl_mw"].sum(1).cpu().numpy()).max() < 1e-6
    env.close()
    # determinism + independence of batch composition: env 4000 alone (same global id) gives the same result
    net, prof, big = make(case, 4096)
    big.reset()
    small = VoltageControlBatch(make_case(case)[0], make_case(case)[1], args_for(case), n_envs=1, device="cuda:0",
                                env_id_offset=4000, obs_dtype=torch.float64)
    small.reset()
    assert torch.equal(big.get_obs()[4000], small.get_obs()[0])
    a = torch.full((4096, 22), 0.3, dtype=torch.float64, device="cuda:0")
    rb, _, _ = big.step(a)
    rs, _, _ = small.step(a[:1])
    assert torch.equal(rb[4000:4001], rs)
    big.close(); small.close()


def test_history_stacking():
    case, B = "case33", 2
    net, prof, env = make(case, B, history=3)
    o = VoltageControlOracle(net, prof, args_for(case, history=3), do_reset=False)
    obs, _ = env.reset()
    oo, _ = o.reset()
    assert obs.shape == (B, net.n_sgen, 3 * net.obs_size()) and env.get_obs_size() == 3 * net.obs_size()
    assert np.abs(np.array(oo) - obs[0].cpu().numpy()).max() < 1e-9
    for t in range(4):
        act = np.full((B, net.n_sgen), 0.1 * t)
        env.step(torch.as_tensor(act, device="cuda:0"))
        o.step(act[0])
        assert np.abs(np.array(o.get_obs()) - env.get_obs()[0].cpu().numpy()).max() < 1e-9
    env.close()


def test_dropin_b1_adapter():
    """the reference's minimal loop (code_examples.py:36-62) against the drop-in class"""
    cfg = dict(voltage_barrier_type="l1", voltage_weight=1.0, q_weight=0.1, line_weight=None, dq_dv_weight=None,
               history=1, pv_scale=1.0, demand_scale=1.0, state_space=["pv", "demand", "reactive", "vm_pu", "va_degree"],
               v_upper=1.05, v_lower=0.95, data_path="./environments/var_voltage_control/data/case33_3min_final",
               episode_limit=240, action_scale=0.8, action_bias=0.0, mode="distributed", reset_action=True, seed=0)
    env = VoltageControl(cfg)
    n_agents, n_actions = env.get_num_of_agents(), env.get_total_actions()
    assert (n_agents, n_actions) == (6, 1)
    info_env = env.get_env_info()
    assert info_env == {"state_shape": 144, "obs_shape": 50, "n_actions": 1, "n_agents": 6, "episode_limit": 240}
    o = VoltageControlOracle(*make_case("case33"), dict(cfg), env_id=0, do_reset=False)
    state, global_state = env.reset()
    o.draw = 1                      # the constructor's reset consumed draw 0 (voltage_control_env.py:85)
    so, go = o.reset()
    assert isinstance(state, list) and len(state) == 6 and state[0].shape == (50,) and state[0].dtype == np.float64
    assert global_state.shape == (144,)
    assert np.abs(np.array(so) - np.array(state)).max() < 1e-9
    np.random.seed(0)
    for t in range(5):
        obs = env.get_obs(); st = env.get_state()
        actions = []
        for agent_id in range(n_agents):
            avail = env.get_avail_agent_actions(agent_id)
            ind = np.nonzero(avail)[0]
            actions.append(np.random.normal(0, 0.5, n_actions)[ind])
        actions = np.concatenate(actions, axis=0)
        reward, done, info = env.step(actions)
        ro, to, io = o.step(actions)
        assert isinstance(reward, float) and isinstance(done, bool) and isinstance(info, dict)
        assert abs(reward - ro) < 1e-9 and done == to and set(info) == set(io)
    assert env.get_avail_actions().shape == (1, 6, 1)
    assert env._get_res_bus_v().shape == (33,) and env._get_res_line_loss().shape == (32,)
    env.close()


# ---- committed golden fixtures (tests/golden/, oracle outputs; see make_golden.py) -----------------
import os as _os
_G = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["case33", "case141", "case322"])
def test_golden_solves(case):
    g = np.load(_os.path.join(_G, f"solve_{case}.npz"))
    B = g["vm_pu"].shape[0]
    net, prof, env = make(case, B)
    vm, va, it, cv = env.solve(g["p_load"], g["q_load"], g["p_sgen"], g["q_sgen"])
    assert cv.all() and np.array_equal(it.cpu().numpy(), g["iterations"])
    assert np.abs(vm.cpu().numpy() - g["vm_pu"]).max() < V_TOL
    assert np.abs(va.cpu().numpy() - g["va_degree"]).max() < 1e-7
    env.close()


@pytest.mark.parametrize("case", ["case33", "case141"])
def test_golden_episode(case):
    g = np.load(_os.path.join(_G, f"episode_{case}.npz"))
    T, B = g["reward"].shape
    net, prof, env = make(case, B, voltage_barrier_type=str(g["barrier"]), seed=int(g["seed"]))
    obs, state = env.reset()
    assert np.array_equal(env.start_rows().cpu().numpy(), g["start_rows"])
    assert np.abs(obs.cpu().numpy() - g["obs"][0]).max() < 1e-9
    for t in range(T):
        r, term, info = env.step(torch.as_tensor(g["actions"][t], device="cuda:0"))
        assert np.abs(r.cpu().numpy() - g["reward"][t]).max() < 1e-9
        assert np.abs(info.cpu().numpy() - g["info"][t]).max() < 1e-9
        assert np.abs(env.get_obs().cpu().numpy() - g["obs"][t + 1]).max() < 1e-9
        assert np.abs(env.get_state().cpu().numpy() - g["state"][t + 1]).max() < 1e-7
    env.close()


# ---- generic network features: shunts, per-unit branches (tap ratio / phase shift), parallel and
# ---- out-of-service lines, c_nf / g_us line charging, non-unit ext_grid set-point
def _featured_net():
    from mapdn_amd.netspec import NetSpec
    net, prof = make_case("case33")
    n = net.copy()
    # the feeder head (bus 0-1) becomes a transformer-like per-unit branch with off-nominal tap + phase shift
    keep = np.ones(n.n_line, bool); keep[0] = False
    def cut(a): return a[keep]
    kw = dict(name="case33x", bus_vn_kv=n.bus_vn_kv, bus_zone=n.bus_zone,
              line_from_bus=cut(n.line_from_bus), line_to_bus=cut(n.line_to_bus),
              line_r_ohm_per_km=cut(n.line_r_ohm_per_km), line_x_ohm_per_km=cut(n.line_x_ohm_per_km),
              line_c_nf_per_km=np.full(31, 120.0), line_g_us_per_km=np.full(31, 2.0),
              line_length_km=cut(n.line_length_km), line_parallel=cut(n.line_parallel), line_in_service=cut(n.line_in_service),
              load_bus=n.load_bus, sgen_bus=n.sgen_bus, sgen_zone=n.sgen_zone,
              ext_grid_bus=0, ext_grid_vm_pu=1.02, sn_mva=5.0, f_hz=50.0,
              br_from_bus=[0], br_to_bus=[1], br_r_pu=[0.002], br_x_pu=[0.02], br_b_pu=[0.01], br_ratio=[0.98], br_shift_deg=[1.5],
              br_g_pu=[0.004],                         # iron-loss conductance of the branch (BR_B = b - 1j*g)
              shunt_bus=[5, 17], shunt_p_mw=[0.01, 0.0], shunt_q_mvar=[-0.3, 0.15])
    x = NetSpec(**kw)
    # element scaling / in_service (pd2ppc: PD = sum p * scaling): a load out of service, one scaled, a scaled sgen —
    # runpp and res_sgen (q_loss) see the scaled values, obs / the q clip the raw table values
    x.load_scaling[4] = 0.0; x.load_scaling[9] = 1.3; x.sgen_scaling[2] = 0.8
    # a doubled line (parallel = 2), an extra out-of-service tie line, and a line entered twice (two rows, same bus pair)
    x.line_parallel[3] = 2
    for k, v in (("line_from_bus", 20), ("line_to_bus", 7)):
        setattr(x, k, np.append(getattr(x, k), v).astype(np.int32))
    for k, v in (("line_r_ohm_per_km", 2.0), ("line_x_ohm_per_km", 2.0), ("line_c_nf_per_km", 0.0), ("line_g_us_per_km", 0.0), ("line_length_km", 1.0)):
        setattr(x, k, np.append(getattr(x, k), v))
    x.line_parallel = np.append(x.line_parallel, 1).astype(np.int32)
    x.line_in_service = np.append(x.line_in_service, 0).astype(np.uint8)
    dup = 10                                           # duplicate row of an in-service line: Ybus sums both
    for k in ("line_from_bus", "line_to_bus", "line_parallel"):
        setattr(x, k, np.append(getattr(x, k), getattr(x, k)[dup]).astype(np.int32))
    for k in ("line_r_ohm_per_km", "line_x_ohm_per_km", "line_c_nf_per_km", "line_g_us_per_km", "line_length_km"):
        setattr(x, k, np.append(getattr(x, k), getattr(x, k)[dup]))
    x.line_in_service = np.append(x.line_in_service, 1).astype(np.uint8)
    return x, prof


def test_generic_network_features():
    net, prof = _featured_net()
    B = 21                                             # odd batch size: padded lanes must stay inert
    a = args_for("case33")
    env = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    from oracle.pp_restated import make_ybus
    y = env.ybus_dense(); yo = make_ybus(net)[0].toarray()
    assert np.abs(y - yo).max() < 1e-12 * np.abs(yo).max()
    rng = np.random.default_rng(9)
    rows = rng.integers(0, prof.n_rows, B)
    pl, ql, pv = prof.load_p[rows], prof.load_q[rows], prof.pv[rows]
    qs = rng.uniform(-0.8, 0.8, (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    vm, va, it, cv = env.solve(pl, ql, pv, qs)
    vm, va, it = vm.cpu().numpy(), va.cpu().numpy(), it.cpu().numpy()
    assert cv.all()
    for e in range(B):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert r.converged and same_newton_count(net, (pl[e], ql[e], pv[e], qs[e]), it[e], True, r)
        assert np.abs(vm[e] - r.vm_pu).max() < V_TOL and np.abs(va[e] - r.va_degree).max() < 1e-7
    assert (vm[:, 0] == 1.02).all()
    # env level: res_bus incl. shunt terms, slack injection through the tapped branch, res_line incl. the
    # out-of-service (0) and duplicated rows, rewards
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(3)]
    env.reset()
    for o in oracles:
        o.reset()
    for t in range(3):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        res = env.results()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9
            assert np.abs(res["p_mw"][e].cpu().numpy() - o.res.p_mw).max() < 1e-9
            assert np.abs(res["q_mvar"][e].cpu().numpy() - o.res.q_mvar).max() < 1e-9
            assert np.abs(res["pl_mw"][e].cpu().numpy() - o.res.pl_mw).max() < 1e-9
            assert res["pl_mw"][e, 31].item() == 0.0                       # out-of-service row
            assert abs(io["total_line_loss"] - info[e, 8].item()) < 1e-9
            assert abs(io["q_loss"] - info[e, 9].item()) < 1e-12
            assert np.abs(np.array(o.get_obs()) - env.get_obs()[e].cpu().numpy()).max() < 1e-9
    env.close()


@pytest.mark.parametrize("B", [1, 15, 16, 17, 100])
def test_batch_sizes(B):
    """batches that are not multiples of the workgroup size: padded envs are inert, results unchanged"""
    net, prof, env = make("case33", B)
    ref_net, ref_prof, ref = make("case33", 128)
    env.reset(); ref.reset()
    assert torch.equal(env.get_obs(), ref.get_obs()[:B])
    a = torch.linspace(-0.7, 0.7, 128 * net.n_sgen, dtype=torch.float64, device="cuda:0").reshape(128, net.n_sgen)
    r1, t1, i1 = env.step(a[:B]); r2, t2, i2 = ref.step(a)
    assert torch.equal(r1, r2[:B]) and torch.equal(i1, i2[:B]) and torch.equal(env.get_obs(), ref.get_obs()[:B])
    env.close(); ref.close()


def test_full_episode_with_reset_boundary():
    """one complete 240-step episode + the reset into the next one, large batch, two envs checked
    against the oracle at every step (catches drift, termination and reset bookkeeping errors)"""
    case, B = "case141", 1024
    net, prof, env = make(case, B, voltage_barrier_type="bowl")
    a = args_for(case)
    watch = [0, 777]
    oracles = {e: VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in watch}
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
    for episode in range(2):
        obs, _ = env.reset()
        for e, o in oracles.items():
            oo, _ = o.reset()
            assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9
        for t in range(239):
            act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * 0.6
            r, term, info = env.step(act)
            if t % 17 == 0 or t >= 236:
                obs = env.get_obs()
            for e, o in oracles.items():
                ro, to, io = o.step(act[e].cpu().numpy())
                assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item()), (episode, t, e)
                if t % 17 == 0 or t >= 236:
                    assert np.abs(np.array(o.get_obs()) - obs[e].cpu().numpy()).max() < 1e-9
            assert bool(term.all().item()) == (t == 238)
        ret = env.episode_returns()
        for e, o in oracles.items():
            assert abs(ret[e].item() - o.sum_rewards) < 1e-7
    assert env.stats()["reset_failures"] == 0
    env.close()


def test_batched_rollout_driver():
    """device-resident train_process — translate_action, window, stat means"""
    from mapdn_amd.rollout import BatchedRollout, translate_action
    x = torch.tensor([-2.0, -1.0, 0.0, 0.5, 3.0], device="cuda:0")
    assert torch.allclose(translate_action(x, 0.6, 0.1), torch.tensor([-0.5, -0.5, 0.1, 0.4, 0.7], device="cuda:0"))
    case, B, T = "case33", 32, 12
    net, prof, env = make(case, B, episode_limit=240)
    env.obs_dtype = torch.float32

    def policy(obs, hid):                      # deterministic toy policy in [-1, 1] space
        return torch.tanh(obs[..., :4].sum(-1) * 3.0), hid
    ro = BatchedRollout(env, policy, max_steps=T)
    win, stat = ro.run()
    assert win.steps == T and win.state.shape == (T, B, net.n_sgen, net.obs_size())
    assert torch.equal(win.last_step[T - 1], torch.ones(B, dtype=torch.bool, device="cuda:0")) and not win.done.any()
    assert torch.equal(win.next_state[:-1], win.state[1:])
    # replay the same episode on the oracle for env 0 with the recorded actions
    o = VoltageControlOracle(net, prof, args_for(case), env_id=0, do_reset=False)
    o.reset()
    tot = 0.0
    for t in range(T):
        act = translate_action(win.action[t, 0, :, 0], SCALE[case], 0.0).double().cpu().numpy()
        r, term, info = o.step(act)
        assert abs(r - win.reward[t, 0, 0].item()) < 1e-5        # f32 window
        tot += r
    assert set(stat) == {"mean_train_" + k for k in INFO_KEYS} | {"mean_train_reward"}
    assert abs(stat["mean_train_reward"] - win.reward[:, :, 0].double().mean().item()) < 1e-6
    env.close()


@pytest.mark.parametrize("kw", [dict(state_space=["vm_pu", "pv", "reactive"]), dict(line_weight=0.5, q_weight=None),
                                dict(line_weight=2.0), dict(reset_action=False), dict(v_lower=0.97, v_upper=1.03, voltage_weight=3.0, q_weight=0.3)])
def test_constructor_options(kw):
    """env kwargs that change the reward / obs composition (var_voltage_control.yaml:3-20)"""
    case, B = "case33", 3
    net, prof, env = make(case, B, **kw)
    a = args_for(case, **kw)
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(B)]
    obs, state = env.reset()
    for e, o in enumerate(oracles):
        oo, os_ = o.reset()
        assert obs[e].shape == np.array(oo).shape and np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9
        assert np.abs(os_ - state[e].cpu().numpy()).max() < 1e-7
    if kw.get("reset_action") is False:
        assert (env.results()["sgen_q"] == 0).all()
    rng = np.random.default_rng(2)
    for t in range(3):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        obs = env.get_obs()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9
            assert np.abs(np.array(o.get_obs()) - obs[e].cpu().numpy()).max() < 1e-9
            assert max(abs(io[k] - info[e, c].item()) for c, k in enumerate(INFO_KEYS)) < 1e-9
    env.close()


def test_reset_keeps_time_when_asked():
    """reset(reset_time=False) re-uses the episode start (voltage_control_env.py:110-113)"""
    net, prof, env = make("case33", 5)
    env.reset()
    s0 = env.start_rows().clone()
    env.step(torch.zeros(5, net.n_sgen, device="cuda:0"))
    env.reset(reset_time=False)
    assert torch.equal(env.start_rows(), s0)
    env.reset()
    assert not torch.equal(env.start_rows(), s0)
    env.close()


def test_unsolvable_reset_is_reported_not_hidden():
    """reference reset loops forever on unsolvable starts (voltage_control_env.py:108); the batched env
    retries max_reset_tries times, then leaves the env terminated and counts it"""
    from mapdn_amd.netspec import Profiles
    net, prof = make_case("case33")
    heavy = Profiles(pv=prof.pv, load_p=prof.load_p * 40.0, load_q=prof.load_q * 40.0, time_delta_min=3, days=prof.days)
    env = VoltageControlBatch(net, heavy, args_for("case33"), n_envs=10, device="cuda:0", max_reset_tries=2)
    env.reset()
    assert env.stats()["reset_failures"] == 10
    r, term, info = env.step(torch.zeros(10, net.n_sgen, device="cuda:0"))
    assert (r == 0).all() and term.all()
    env.close()
    # two handles sharing one kernel instantiation with different LDS needs keep working side by side
    a = VoltageControlBatch(*make_case("case33"), args_for("case33"), n_envs=4, device="cuda:0")
    b = VoltageControlBatch(*_featured_net(), args_for("case33"), n_envs=4, device="cuda:0")
    a.reset(); b.reset(); a.step(torch.zeros(4, 6, device="cuda:0")); b.step(torch.zeros(4, 6, device="cuda:0"))
    assert torch.isfinite(a.get_obs()).all() and torch.isfinite(b.get_obs()).all()
    a.close(); b.close()


def _random_radial_net(seed):
    """random tree, slack at a random bus (=> several elimination components), random lines / loads / sgens"""
    from mapdn_amd.netspec import NetSpec, Profiles
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(6, 150)) if seed < 100 else int(rng.integers(3, 8))     # seeds >= 100: tiny nets
    parent = np.array([-1] + [int(rng.integers(max(0, i - int(rng.integers(1, 6))), i)) for i in range(1, nb)])
    perm = rng.permutation(nb)                                   # scramble labels: slack lands anywhere in the tree
    f = perm[parent[1:]]; t = perm[np.arange(1, nb)]
    slack = int(perm[int(rng.integers(0, nb))])
    n_line = nb - 1
    nz = int(rng.integers(1, 5))
    zone = rng.integers(1, nz + 1, nb).astype(np.int32); zone[slack] = 0
    ns = int(rng.integers(1, 9))
    cand = np.array([b for b in range(nb) if b != slack])
    sgen_bus = rng.choice(cand, size=ns, replace=True).astype(np.int32)
    nl = int(rng.integers(1, 2 * nb))
    load_bus = rng.integers(0, nb, nl).astype(np.int32)          # loads may sit on the slack bus too
    vn = float(rng.choice([0.4, 11.0, 20.0]))
    sn = float(rng.choice([0.5, 1.0, 10.0]))
    zb = vn * vn / sn
    net = NetSpec(name=f"rand{seed}", bus_vn_kv=np.full(nb, vn), bus_zone=zone, line_from_bus=f, line_to_bus=t,
                  line_r_ohm_per_km=rng.uniform(0.05, 0.5, n_line) * zb * 0.02, line_x_ohm_per_km=rng.uniform(0.02, 0.4, n_line) * zb * 0.02,
                  line_c_nf_per_km=rng.uniform(0, 50, n_line), line_g_us_per_km=rng.uniform(0, 1, n_line),
                  line_length_km=rng.uniform(0.2, 1.5, n_line), line_parallel=rng.integers(1, 3, n_line).astype(np.int32),
                  line_in_service=np.ones(n_line, np.uint8), load_bus=load_bus, sgen_bus=sgen_bus, sgen_zone=zone[sgen_bus],
                  ext_grid_bus=slack, ext_grid_vm_pu=float(rng.uniform(0.98, 1.03)), sn_mva=sn, f_hz=50.0)
    T = 1500
    pv = rng.uniform(0, 0.3 * sn / ns, (T, ns)); lp = rng.uniform(0, 0.4 * sn / nl, (T, nl)); lq = lp * rng.uniform(0.1, 0.5, (T, nl))
    return net, Profiles(pv=pv, load_p=lp, load_q=lq, time_delta_min=3)


@pytest.mark.parametrize("seed", list(range(12)) + [101, 102, 103])
def test_random_topologies(seed):
    net, prof = _random_radial_net(seed)
    a = dict(episode_limit=240, action_scale=0.8, action_bias=0.0, voltage_barrier_type="l2", seed=seed)
    B = 9
    env = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    from oracle.pp_restated import make_ybus
    y = env.ybus_dense(); yo = make_ybus(net)[0].toarray()
    assert np.abs(y - yo).max() <= 1e-12 * np.abs(yo).max()
    oracles = [VoltageControlOracle(net, prof, a, env_id=e, do_reset=False) for e in range(2)]
    obs, state = env.reset()
    for e, o in enumerate(oracles):
        oo, os_ = o.reset()
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9 and np.abs(os_ - state[e].cpu().numpy()).max() < 1e-7
    rng = np.random.default_rng(seed)
    for t in range(3):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        res = env.results(); obs = env.get_obs()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item())
            assert np.abs(res["vm_pu"][e].cpu().numpy() - o.res.vm_pu).max() < V_TOL
            assert np.abs(res["p_mw"][e].cpu().numpy() - o.res.p_mw).max() < 1e-9
            assert np.abs(res["pl_mw"][e].cpu().numpy() - o.res.pl_mw).max() < 1e-9
            assert np.abs(np.array(o.get_obs()) - obs[e].cpu().numpy()).max() < 1e-9
    env.close()


def test_end_to_end_ddpg_training_on_device(tmp_path):
    """rollout through the HIP env + GPU replay + MADDPG/IDDPG updates: runs, learns state, checkpoints
    in the reference's model.pt layout"""
    from mapdn_amd.learner import PGTrainer, make_alg_args
    net, prof = make_case("case33")
    for alg in ("maddpg", "iddpg"):
        torch.manual_seed(0); np.random.seed(0)
        env = VoltageControlBatch(net, prof, args_for("case33", episode_limit=24), n_envs=64, device="cuda:0", copy=True)
        args = make_alg_args(env.n_agents, env.obs_size, 1, 0.8, 0.0, max_steps=24, batch_size=128,
                             replay_buffer_size=64 * 16, behaviour_update_freq=8, target_update_freq=8,
                             value_update_epochs=2, num_eval_episodes=64)
        tr = PGTrainer(args, alg, env)
        before = {k: v.clone() for k, v in tr.behaviour_net.state_dict().items()}
        stat = {}
        tr.run(stat, 0)
        assert tr.steps == 24 and len(tr.replay_buffer) == 64 * 16
        assert np.isfinite([stat["mean_train_reward"], stat["mean_test_reward"], stat["mean_train_value_loss"]]).all()
        assert 0.0 <= stat["mean_train_totally_controllable_ratio"] <= 1.0
        after = tr.behaviour_net.state_dict()
        assert any(not torch.equal(before[k], after[k]) for k in before if k.startswith("policy_dicts"))
        b = tr.replay_buffer.get_batch(128)
        assert b["state"].is_cuda and b["state"].shape == (128, env.n_agents, env.obs_size) and bool(b["valid"].all())
        tr.save(tmp_path / f"{alg}.pt")
        sd = torch.load(tmp_path / f"{alg}.pt")["model_state_dict"]
        assert "policy_dicts.0.rnn.weight_ih" in sd and "target_net.value_dicts.0.fc3.bias" in sd
        env.close()


@pytest.mark.parametrize("check_dx", [1e30, 1e-300])
def test_convergence_prediction_is_only_a_shortcut(check_dx):
    """The NR kernel runs a forward sweep mismatch-only when it predicts convergence and redoes it in full
    when the prediction was wrong.  Forcing the prediction to 'always' (every sweep after the first is tried
    mismatch-only and redone) and to 'never' must give the same iterations and bit-identical voltages."""
    case, B = "case141", 96
    net, prof = make_case(case)
    rng = np.random.default_rng(5)
    rows = rng.integers(0, prof.n_rows, B)
    act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
    pl, ql, pv = prof.load_p[rows], prof.load_q[rows], prof.pv[rows]
    qs = act * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    pl[7] *= 30.0                                    # one env that never converges
    ref_env = VoltageControlBatch(net, prof, args_for(case), n_envs=B, device="cuda:0")
    vm0, va0, it0, cv0 = [x.cpu().numpy() for x in ref_env.solve(pl, ql, pv, qs)]
    ref_env.close()
    # always / never, for the second predictor too (mapdn_env_config.nr_check_dx / nr_check_quad; 0 would mean "default")
    tuning = dict(nr_check_dx=check_dx, nr_check_quad=1e-300 if check_dx == 1e30 else float("inf"))
    env = VoltageControlBatch(net, prof, args_for(case), n_envs=B, device="cuda:0", tuning=tuning)
    vm, va, it, cv = [x.cpu().numpy() for x in env.solve(pl, ql, pv, qs)]
    env.close()
    assert np.array_equal(it, it0) and np.array_equal(cv, cv0) and not cv[7] and it[7] == 10
    ok = cv0.astype(bool)
    assert np.array_equal(vm[ok], vm0[ok]) and np.array_equal(va[ok], va0[ok])
    for e in (0, 1, 50):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert same_newton_count(net, (pl[e], ql[e], pv[e], qs[e]), it[e], cv[e], r) and np.abs(vm[e] - r.vm_pu).max() < V_TOL


def test_tester_records_match_the_oracle_env(tmp_path):
    """PGTester.run (utilities/tester.py:19-63): same record keys / shapes as the reference, values equal to the
    oracle env driven by the same greedy policy; batch_run aggregates the 11 info keys."""
    import pickle
    from mapdn_amd.learner import DDPGNet, make_alg_args
    from mapdn_amd.tester import PGTester
    torch.manual_seed(3)
    case = "case33"
    net, prof, env = make(case, 4, episode_limit=12)
    args = make_alg_args(env.n_agents, env.obs_size, 1, SCALE[case], 0.0, max_steps=12)
    pol = DDPGNet(args, "iddpg").to("cuda:0")
    tester = PGTester(args, pol, env)
    rec = tester.run(2, 11, 5)
    assert set(rec) == {"pv_active", "pv_reactive", "bus_active", "bus_reactive", "bus_voltage", "line_loss"}
    # reset leaves steps == 1 (voltage_control_env.py:100), so an episode_limit of 12 ends after 11 steps (:204)
    assert all(len(v) == 12 for v in rec.values()) and rec["bus_voltage"][0].shape == (net.n_bus,)
    # oracle env, same start, no noise, the actions the GPU run took (recomputed from the recorded sgen q is
    # not possible before the clip, so drive the oracle with the same policy on its own observations)
    o = VoltageControlOracle(net, prof, args_for(case, episode_limit=12), env_id=0, do_reset=False)   # same draw counter as the fresh GPU env
    obs_list, _ = o.manual_reset(2, 11, 5)
    hid = pol.init_hidden(1)
    assert np.abs(o.res.vm_pu - rec["bus_voltage"][0]).max() < V_TOL
    for t in range(11):
        ob = torch.tensor(np.array(obs_list), dtype=torch.float32, device="cuda:0").unsqueeze(0)
        with torch.no_grad():
            a, _, _, _, hid = pol.get_actions(ob, "test", False, torch.ones(1, env.n_agents, 1, device="cuda:0"), False, hid)
        actual = (a.squeeze().clamp(-1, 1) * SCALE[case]).cpu().numpy().astype(np.float64)
        o.step(actual, add_noise=False)
        obs_list = o.get_obs()
        assert np.abs(o.res.vm_pu - rec["bus_voltage"][t + 1]).max() < 1e-6          # f32 policy on f32 obs: tiny action differences
        assert np.abs(o.res.pl_mw - rec["line_loss"][t + 1]).max() < 1e-6
    PGTester.save_record(rec, tmp_path / "rec.pickle")
    assert set(pickle.load(open(tmp_path / "rec.pickle", "rb"))) == set(rec)
    stat = tester.batch_run(8)
    assert set(stat) == {"mean_test_" + k for k in INFO_KEYS} and all(len(v) == 2 for v in stat.values())
    assert 0.0 <= stat["mean_test_totally_controllable_ratio"][0] <= 1.0
    env.close()


def test_start_rows_outside_the_table_are_refused():
    """a late / negative start must not become an out-of-bounds profile read.  The Python class raises
    (the reference dies with IndexError on the empty slice, :446-447,473-475); through the raw C ABI the kernels leave
    such envs terminated and count them as reset failures."""
    from mapdn_amd import _lib
    net, prof, env = make("case33", 5)
    T = prof.n_rows
    for bad in (-1, T - 240, T + 5):
        with pytest.raises(IndexError):
            env.reset(start_rows=torch.tensor([100, 200, bad, 300, 400]))
    with pytest.raises(ValueError):
        env.manual_reset(1, 24, 0)
    with pytest.raises(ValueError):
        env.manual_reset(1, 3, 20)
    # raw C ABI: no host check in the way
    sr = torch.tensor([100, T - 100, 300, -7, 2 ** 40], dtype=torch.int64, device="cuda:0")
    _lib.check(env._lib.mapdn_reset(env._h, sr.data_ptr(), 0, 2, env._stream()), env._h)
    env._was_reset = True
    assert env.stats()["reset_failures"] == 3
    act = torch.zeros(5, net.n_sgen, dtype=torch.float64, device="cuda:0")
    r, term, info = env.step(act, add_noise=False)
    assert term.cpu().tolist() == [False, True, False, True, True] and r[1].item() == 0.0 and r[3].item() == 0.0
    o = VoltageControlOracle(net, prof, args_for("case33", reset_action=True), env_id=0, do_reset=False)
    assert np.isfinite(env.get_obs().cpu().numpy()).all()
    env.close()


def test_history_frames_do_not_allocate_and_match_oracle():
    net, prof, env = make("case33", 3, history=3, voltage_barrier_type="l1")
    o = VoltageControlOracle(net, prof, args_for("case33", history=3, voltage_barrier_type="l1"), env_id=1, do_reset=False)
    obs, _ = env.manual_reset(2, 10, 1)
    oo, _ = o.manual_reset(2, 10, 1)
    assert np.abs(np.array(oo) - obs[1].cpu().numpy()).max() < 1e-9
    ptrs = set()
    rng = np.random.default_rng(5)
    for t in range(6):
        a = rng.uniform(-0.8, 0.8, net.n_sgen)
        env.step(torch.as_tensor(np.tile(a, (3, 1)), device="cuda:0"), add_noise=False)
        o.step(a, add_noise=False)
        ob = env.get_obs()
        ptrs.add(ob.data_ptr())
        assert np.abs(np.array(o.get_obs()) - ob[1].cpu().numpy()).max() < 1e-9
    assert len(ptrs) <= 2                         # two preallocated stacked frames, no per-call allocation
    env.close()


def test_auto_reset_runs_consecutive_episodes_like_the_reference_loop():
    """with auto_reset an env that terminates (here: env 1 early, on an unsolvable step; all of them at
    the episode limit) starts its next episode on the following call and reports `terminated` exactly once — the cadence of
    the reference's train loop (models/model.py:204-262: reset() right after `done`).  Replayed on the oracle env."""
    case, B, limit = "case33", 3, 12
    kw = dict(episode_limit=limit, auto_reset=True)
    net, prof, env = make(case, B, **kw)
    oracles = [VoltageControlOracle(net, prof, args_for(case, episode_limit=limit), env_id=e, do_reset=False) for e in range(B)]
    obs, _ = env.reset()
    for e, o in enumerate(oracles):
        oo, _ = o.reset()
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9
    rng = np.random.default_rng(11)
    pending = [False] * B                     # oracle env e terminated in the previous call
    n_term = [0] * B
    for t in range(2 * limit + 6):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        if t == 4:
            act[1] = 60.0                     # env 1: unsolvable power flow -> terminates early (:188-196)
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        obs = env.get_obs().cpu().numpy()
        mask = env.auto_reset_mask().cpu().numpy()
        for e, o in enumerate(oracles):
            if pending[e]:                    # this call is the env's reset()
                assert mask[e] and r[e].item() == 0.0 and not term[e].item() and (info[e] == 0).all()
                oo, _ = o.reset()
                assert np.abs(np.array(oo) - obs[e]).max() < 1e-9, (t, e)
                pending[e] = False
                continue
            assert not mask[e]
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item()), (t, e)
            assert max(abs(io[k] - info[e, c].item()) for c, k in enumerate(INFO_KEYS)) < 1e-9
            assert np.abs(np.array(o.get_obs()) - obs[e]).max() < 1e-9, (t, e)
            if to:
                pending[e] = True; n_term[e] += 1
    assert n_term == [2, 3, 2] and env.stats()["reset_failures"] == 0     # env 1: early end + two full episodes of 11 steps
    env.close()
    # without the flag nothing changes: a terminated env stays frozen
    net, prof, env = make(case, 2, episode_limit=4)
    env.reset()
    for t in range(6):
        r, term, info = env.step(torch.zeros(2, net.n_sgen, dtype=torch.float64, device="cuda:0"))
    assert term.all() and (r == 0).all() and not env.auto_reset_mask().any()
    env.close()


@pytest.mark.parametrize("case,obs_dtype", [("case33", torch.float64), ("case141", torch.float32), ("case322", torch.float32)])
def test_fused_step_obs_equals_step_then_get_obs(case, obs_dtype):
    """mapdn_step_obs (step + get_obs as one C call, what VoltageControlBatch.step() issues) against mapdn_step followed
    by mapdn_get_obs on twin envs: bit-identical obs / reward / info / results over noisy steps, including an unsolvable
    step (rollback: the obs of the previous state with the new PV)"""
    B = 77
    net, prof = make_case(case)
    a = args_for(case, seed=5)
    fused = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=obs_dtype)
    plain = VoltageControlBatch(net, prof, a, n_envs=B, device="cuda:0", obs_dtype=obs_dtype)
    plain.fused_step = False
    o1, s1 = fused.reset(); o2, s2 = plain.reset()
    assert torch.equal(o1, o2) and torch.equal(s1, s2)
    rng = np.random.default_rng(1)
    for t in range(6):
        act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
        if t == 3:
            act[5] = 400.0                                       # absurd q: this env's power flow diverges
        ta = torch.as_tensor(act, device="cuda:0")
        r1, d1, i1 = fused.step(ta); r2, d2, i2 = plain.step(ta)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1, i2)
        assert torch.equal(fused.get_obs(), plain.get_obs())
        assert torch.equal(fused.get_state(), plain.get_state())
        for k, v in fused.results().items():
            assert torch.equal(v, plain.results()[k]), k
        lp1, lq1 = fused.loads(); lp2, lq2 = plain.loads()
        assert torch.equal(lp1, lp2) and torch.equal(lq1, lq2)
        if t == 3:
            assert bool(d1[5]) and i1[5, 10] == 1.0              # destroy
    # a second get_obs with another dtype still launches its own gather
    assert torch.allclose(fused.get_obs(torch.float64).float(), fused.get_obs(torch.float32), rtol=1e-6, atol=1e-6)
    fused.close(); plain.close()


@pytest.mark.parametrize("case,geoms", [
    ("case141", [(4, 16, 0), (2, 16, 1), (2, 16, 2), (1, 8, 1), (4, 8, 0), (1, 32, 1), (2, 8, 2), (1, 16, 2), (4, 16, 1),
                 (4, 4, 0), (2, 4, 2), (4, 4, 1), (1, 4, 2)]),
    ("case33", [(1, 16, 0), (2, 16, 0), (1, 8, 0), (4, 16, 0), (2, 8, 1), (4, 4, 0), (1, 4, 0)]),
    ("case322", [(4, 8, 0), (2, 8, 1), (4, 16, 1), (4, 16, 2), (2, 16, 1), (4, 4, 0), (2, 4, 0), (4, 4, 1)]),
])
def test_every_nr_launch_geometry_gives_the_same_bits(case, geoms):
    """mapdn_env_config.nr_waves / nr_lanes / nr_lean (0 auto, 1 lean, 2 fat): every (waves, envs per workgroup, LDS residency)
    variant of k_nr_tree — the specialised default instantiations, the generic ones, both per-env reduction paths (row swaps at
    16 envs per workgroup, LDS elsewhere) — sums the children in the same canonical order, so voltages, angles and iteration
    counts are bit-identical.  The handles differ per `tuning`, not per process: the reference handle stays alive throughout."""
    B = 200
    net, prof = make_case(case)
    rng = np.random.default_rng(11)
    rows = rng.integers(0, prof.n_rows, B)
    pv = prof.pv[rows]
    qs = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    ins = (prof.load_p[rows], prof.load_q[rows], pv, qs)
    ref, ref_env, seen = None, None, set()
    for w, l, lean in geoms:
        try:
            env = VoltageControlBatch(net, prof, args_for(case), n_envs=B, device="cuda:0", tuning=dict(nr_waves=w, nr_lanes=l, nr_lean=lean))
        except Exception as exc:                              # a geometry whose LDS need exceeds a CU is refused, not wrong
            assert "LDS" in str(exc), exc
            continue
        g = env.geometry()
        assert (g["waves"], g["lanes"]) == (w, l) and (lean == 0 or g["lean"] == (lean == 1))
        seen.add((g["waves"], g["lanes"], g["h_lds"], g["g_lds"], g["rec_lds"], g["flat_lds"]))
        out = [t.cpu().numpy() for t in env.solve(*ins)]
        assert out[3].all()
        if ref is None:
            ref, ref_env = out, env                           # two live handles with different geometries from here on
        else:
            assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2]), (w, l, lean)
            again = [t.cpu().numpy() for t in ref_env.solve(*ins)]      # ... and the first handle is unaffected by the others
            assert all(np.array_equal(a, b) for a, b in zip(again, ref))
            env.close()
    ref_env.close()
    assert len(seen) >= 3


def test_two_handles_with_different_tuning_in_one_process(monkeypatch):
    """the launch knobs are fields of mapdn_env_config, not process-global getenv reads.  Two handles on the same
    net with different geometry, injection kernel and mismatch-evaluation form step side by side — interleaved calls — and
    stay bit-identical; an environment variable still overrides the field (tools), read once at mapdn_create."""
    case, B = "case141", 96
    net, prof, a = make(case, B, tuning=dict(nr_waves=4, nr_lanes=16), episode_limit=6, auto_reset=True)
    _, _, b = make(case, B, tuning=dict(nr_waves=2, nr_lanes=16, nr_lean=1, inject_full=1, nr_mm_pass=2), episode_limit=6, auto_reset=True)
    ga, gb = a.geometry(), b.geometry()
    assert (ga["waves"], ga["lean"], ga["h_lds"], ga["mm_pass"]) == (4, 0, 1, 1) and (gb["waves"], gb["lean"], gb["h_lds"], gb["mm_pass"]) == (2, 1, 0, 0)
    oa, _ = a.reset(); ob, _ = b.reset()
    assert torch.equal(oa, ob)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    for t in range(9):
        act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * SCALE[case]
        ra, ta, ia = a.step(act); rb, tb, ib = b.step(act)
        assert torch.equal(ta, tb) and torch.equal(a.get_obs(), b.get_obs())
        assert torch.allclose(ra, rb, rtol=0, atol=1e-12) and torch.allclose(ia, ib, rtol=0, atol=1e-12)
    a.close(); b.close()
    monkeypatch.setenv("MAPDN_NR_WAVES", "2"); monkeypatch.setenv("MAPDN_NR_LANES", "16"); monkeypatch.setenv("MAPDN_NR_LEAN", "0")
    _, _, c = make(case, B, tuning=dict(nr_waves=4, nr_lanes=16))
    gc = c.geometry()
    assert (gc["waves"], gc["lanes"], gc["lean"]) == (2, 16, 0)
    c.close()


@pytest.mark.parametrize("case", ["case33", "case141", "case322"])
def test_pv_bus_injection_equals_all_bus_injection(case):
    """step() / reset() inject through k_inject_sgen — PV buses only, k_advance keeps the Sbus entries of all other buses and
    the load part of the PV buses current — while mapdn_env_config.inject_full = 1 keeps the all-bus k_inject of round 2.  Same expressions,
    same order: bit-identical over noisy episodes with an unsolvable step, per-env auto-reset boundaries (an auto-resetting env
    refreshes all its loads inside k_inject_sgen) and a mapdn_solve_only call in between (which leaves Sbus stale)."""
    B, limit = 70, 5
    kw = dict(episode_limit=limit, auto_reset=True)
    net, prof, fast = make(case, B, **kw)
    _, _, full = make(case, B, tuning=dict(inject_full=1), **kw)
    of, _ = fast.reset(); ou, _ = full.reset()
    assert torch.equal(of, ou)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(23)
    rng = np.random.default_rng(2)
    n_resets = 0
    for t in range(14):
        act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * SCALE[case]
        if t == 3:
            act[11] = 60.0
        if t == 6:                                   # a solve on explicit inputs in between: Sbus no longer reflects cur_pl / cur_ql
            rows = rng.integers(0, prof.n_rows, B)
            qs = rng.uniform(-0.3, 0.3, (B, net.n_sgen)) * prof.pv[rows]
            a = fast.solve(prof.load_p[rows], prof.load_q[rows], prof.pv[rows], qs)
            b = full.solve(prof.load_p[rows], prof.load_q[rows], prof.pv[rows], qs)
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        ra, ta, ia = fast.step(act); rb, tb, ib = full.step(act)
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(ia, ib), t
        assert torch.equal(fast.get_obs(), full.get_obs()) and torch.equal(fast.get_state(), full.get_state()), t
        ma = fast.auto_reset_mask(); assert torch.equal(ma, full.auto_reset_mask())
        n_resets += int(ma.sum().item())
        fa, fb = fast.results(), full.results()
        assert all(torch.equal(fa[k], fb[k]) for k in fa), t
        la, lb = fast.loads(), full.loads()
        assert torch.equal(la[0], lb[0]) and torch.equal(la[1], lb[1])
    assert n_resets >= 2 * B
    fast.close(); full.close()


@pytest.mark.parametrize("case,B,geom", [("case33", 70, None), ("case141", 70, None), ("case322", 70, None), ("case141", 200, (2, 16, 1)),
                                         ("case322", 130, (4, 16, 2)), ("case141", 1, None)])
@pytest.mark.parametrize("adt", [torch.float64, torch.float32])
def test_fused_injection_equals_the_injection_launch(case, B, geom, adt):
    """step() of a handle without auto_reset performs the PV-bus injection (_clip_reactive_power + the Sbus entries of the buses
    with sgens + the step bookkeeping) in the PROLOGUE of k_nr_tree (mapdn_env_config.fuse_inject, default) instead of as a launch
    of its own (fuse_inject = 2: k_inject_sgen).  Same expressions in the same order: rewards, flags, info, observations, state
    and the result tables are bit-identical over noisy episodes with an unsolvable step, frozen envs after it, a whole-batch
    reset() and a mapdn_solve_only call in between (after which one all-bus injection runs as a launch in both)."""
    t0 = dict(nr_waves=geom[0], nr_lanes=geom[1], nr_lean=geom[2]) if geom else {}
    net, prof, fused = make(case, B, tuning=dict(fuse_inject=1, **t0), episode_limit=7)
    _, _, plain = make(case, B, tuning=dict(fuse_inject=2, **t0), episode_limit=7)
    of, _ = fused.reset(); ou, _ = plain.reset()
    assert torch.equal(of, ou)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(29)
    rng = np.random.default_rng(4)
    for t in range(13):
        act = ((torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * SCALE[case]).to(adt)
        if t == 2:
            act[B // 2] = 60.0                        # unsolvable: that env terminates and stays frozen until the reset below
        if t == 4:                                    # a solve on explicit inputs in between: Sbus no longer reflects cur_pl / cur_ql
            rows = rng.integers(0, prof.n_rows, B)
            qs = rng.uniform(-0.3, 0.3, (B, net.n_sgen)) * prof.pv[rows]
            a = fused.solve(prof.load_p[rows], prof.load_q[rows], prof.pv[rows], qs)
            b = plain.solve(prof.load_p[rows], prof.load_q[rows], prof.pv[rows], qs)
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        if t == 7:                                    # every env hit the episode limit at call 6
            of, sf = fused.reset(); ou, su = plain.reset()
            assert torch.equal(of, ou) and torch.equal(sf, su)
        ra, ta, ia = fused.step(act); rb, tb, ib = plain.step(act)
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(ia, ib), t
        if t == 2:
            assert bool(ta[B // 2]) and ia[B // 2, 10] == 1.0
        assert torch.equal(fused.get_obs(), plain.get_obs()) and torch.equal(fused.get_state(), plain.get_state()), t
        fa, fb = fused.results(), plain.results()
        assert all(torch.equal(fa[k], fb[k]) for k in fa), t
        la, lb = fused.loads(), plain.loads()
        assert torch.equal(la[0], lb[0]) and torch.equal(la[1], lb[1])
    fused.close(); plain.close()
    # a handle with auto_reset keeps the injection launch (its restarting envs refresh all their loads there); asking for the
    # fused form on it is refused
    with pytest.raises(Exception, match="auto_reset"):
        make(case, B, tuning=dict(fuse_inject=1), auto_reset=True)


@pytest.mark.parametrize("case", ["case33", "case141", "case322"])
def test_mismatch_pass_equals_mismatch_sweep(case):
    """The predicted-final mismatch evaluation runs as a barrier-free pass over all nodes (k_nr_tree::mismatch_pass, the default
    when the h array is LDS-resident) instead of a mismatch-only tree sweep (nr_mm_pass = 2): same expressions summed in the
    same canonical order, so iterations, flags and voltages are bit-identical — also with the predictor forced to 'always',
    where every verdict after the first comes from the pass and a wrong prediction redoes the sweep in full."""
    B = 96
    net, prof = make_case(case)
    rng = np.random.default_rng(8)
    rows = rng.integers(0, prof.n_rows, B)
    act = rng.uniform(-SCALE[case], SCALE[case], (B, net.n_sgen))
    pl, ql, pv = prof.load_p[rows].copy(), prof.load_q[rows], prof.pv[rows]
    qs = act * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    pl[11] *= 30.0                                   # one env that never converges
    out = {}
    for tag, mm, always in (("sweep", 2, False), ("pass", 1, False), ("sweep_always", 2, True), ("pass_always", 1, True)):
        tuning = dict(nr_mm_pass=mm)                       # mapdn_env_config.nr_mm_pass: 1 pass, 2 tree sweep
        if always:
            tuning.update(nr_check_dx=1e30, nr_check_quad=1e-300)
        env = VoltageControlBatch(net, prof, args_for(case), n_envs=B, device="cuda:0", tuning=tuning)
        assert env.geometry()["mm_pass"] == (1 if mm == 1 else 0)
        out[tag] = [x.cpu().numpy() for x in env.solve(pl, ql, pv, qs)]
        env.close()
    vm0, va0, it0, cv0 = out["sweep"]
    assert not cv0[11] and it0[11] == 10 and cv0[[0, 1, 50]].all()
    ok = cv0.astype(bool)
    for tag in ("pass", "sweep_always", "pass_always"):
        vm, va, it, cv = out[tag]
        assert np.array_equal(it, it0) and np.array_equal(cv, cv0), tag
        assert np.array_equal(vm[ok], vm0[ok]) and np.array_equal(va[ok], va0[ok]), tag


@pytest.mark.parametrize("tuning", [dict(), dict(tolerance_is_pu=1), dict(tolerance_mva=1e-3), dict(tolerance_mva=1e-3, tolerance_is_pu=1)])
def test_tolerance_options_follow_the_oracle_on_a_net_with_sn_mva_100(tuning):
    """mapdn_env_config.tolerance_mva / tolerance_is_pu (the two readings of how pandapower forms newtonpf's stopping rule from
    runpp's tolerance_mva; they coincide on sn_mva = 1 nets): iteration counts and voltages follow oracle.runpp_restated with the
    same options on a 141-bus net rescaled to sn_mva = 100, where the readings stop at different iterations"""
    B = 64
    net, prof = make_case("case141")
    net.sn_mva = 100.0
    rng = np.random.default_rng(3)
    rows = rng.integers(0, prof.n_rows, B)
    pv = prof.pv[rows]
    qs = rng.uniform(-0.6, 0.6, (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    env = VoltageControlBatch(net, prof, args_for("case141"), n_envs=B, device="cuda:0", tuning=tuning)
    vm, va, it, cv = [x.cpu().numpy() for x in env.solve(prof.load_p[rows], prof.load_q[rows], pv, qs)]
    env.close()
    assert cv.all()
    its = set()
    for e in range(0, B, 4):
        r = runpp_restated(net, prof.load_p[rows[e]], prof.load_q[rows[e]], pv[e], qs[e], cache=False,
                           tolerance_mva=tuning.get("tolerance_mva", 1e-8), tolerance_is_pu=bool(tuning.get("tolerance_is_pu", 0)))
        assert same_newton_count(net, (prof.load_p[rows[e]], prof.load_q[rows[e]], pv[e], qs[e]), it[e], cv[e], r,
                                 tuning.get("tolerance_mva", 1e-8), bool(tuning.get("tolerance_is_pu", 0))), (e, r.iterations, it[e])
        assert np.abs(vm[e] - r.vm_pu).max() < (1e-9 if "tolerance_mva" not in tuning else 1e-6)
        its.add(int(it[e]))
    assert its


@pytest.mark.parametrize("case,B", [("case141", 300), ("case322", 70)])
def test_xcd_aligned_env_order_changes_nothing(case, B):
    """mapdn_env_config.xcd_map = 1: the wide kernels walk the envs in XCD-aligned order (a permutation of which thread serves
    which env) — bit-identical steps, observations, state and result tables, batch sizes that are not multiples of anything"""
    net, prof, a = make(case, B, tuning=dict(xcd_map=1), episode_limit=6, auto_reset=True)
    _, _, b = make(case, B, episode_limit=6, auto_reset=True)
    oa, sa = a.reset(); ob, sb = b.reset()
    assert torch.equal(oa, ob) and torch.equal(sa, sb)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(3)
    for t in range(8):
        act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * SCALE[case]
        ra, ta, ia = a.step(act); rb, tb, ib = b.step(act)
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(ia, ib)
        assert torch.equal(a.get_obs(), b.get_obs()) and torch.equal(a.get_state(), b.get_state())
        fa, fb = a.results(), b.results()
        assert all(torch.equal(fa[k], fb[k]) for k in fa)
    a.close(); b.close()


def test_step_composition_switches_are_validated_for_every_solver():
    """fuse_inject = 1 / overlap_advance / xcd_map exist on the tree solver only and used to be ignored silently on the
    general solvers (their set-up returned before the check); overlap_advance was also a silent no-op beside the fused prologue."""
    from mapdn_amd._lib import MapdnError
    net, prof = make_case("case33")
    for bad in (dict(nr_solver="sparse", fuse_inject=1), dict(nr_solver="dense", overlap_advance=1), dict(nr_solver="sparse", xcd_map=1)):
        with pytest.raises(MapdnError, match="tree solver only"):
            VoltageControlBatch(net, prof, args_for("case33"), n_envs=8, device="cuda:0", tuning=bad)
    with pytest.raises(MapdnError, match="fuse_inject = 2"):
        VoltageControlBatch(net, prof, args_for("case33"), n_envs=8, device="cuda:0", tuning=dict(overlap_advance=1))
    a = VoltageControlBatch(net, prof, args_for("case33"), n_envs=8, device="cuda:0", obs_dtype=torch.float64, tuning=dict(overlap_advance=1, fuse_inject=2))
    b = VoltageControlBatch(net, prof, args_for("case33"), n_envs=8, device="cuda:0", obs_dtype=torch.float64)
    oa, _ = a.reset(); ob, _ = b.reset()
    act = torch.full((8, net.n_sgen), 0.3, device="cuda:0", dtype=torch.float64)
    for _ in range(3):
        ra, ta, ia = a.step(act); rb, tb, ib = b.step(act)
        assert torch.equal(ra, rb) and torch.equal(a.get_obs(), b.get_obs())
    a.close(); b.close()
