#!/usr/bin/env python
"""Generate tests/golden/env_ref_*.npz by running the REFERENCE's own class, unmodified:

    /root/reference/environments/var_voltage_control/voltage_control_env.py :: VoltageControl

pandapower (un-vendored third-party dependency, not installable offline) is replaced by the stand-in package
oracle/pp_stub/pandapower: `runpp` -> oracle/pp_restated.py writing real pandas res_* tables, `from_pickle` ->
pandas tables of the synthetic net.  Every line of the reference's env logic — reset / manual_reset / step /
_calc_reward / get_obs (incl. the pandas chained-assignment add-back :238-244) / get_state / the tester getters /
voltage_barrier/*.py — executes as written.  Also dumped: the five reference barrier functions on a grid and
utilities/util.py::translate_action.

Run here (needs /root/reference):  python tests/golden/make_env_golden.py
The fixtures are committed; tests/test_env_reference_pin.py holds oracle/env_restated.py to them (1e-12) and the
HIP path to them (1e-9, -m gpu).  Nothing on the GPU box reads /root/reference.
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "pp_stub"), REF]

from mapdn_amd.data import save_netspec, save_profiles_csv          # noqa: E402
from mapdn_amd.netspec import make_case                             # noqa: E402
from tests.golden.env_scenarios import DIGITS, SCENARIOS, actions_for, scenario_data  # noqa: E402

warnings.simplefilter("ignore")      # pandas ChainedAssignment FutureWarnings of the reference's :239-244
from environments.var_voltage_control.voltage_control_env import VoltageControl   # noqa: E402  (the reference class)
from environments.var_voltage_control.voltage_barrier.voltage_barrier_registry import Voltage_Barrier  # noqa: E402

INFO_KEYS = ("percentage_of_v_out_of_control", "percentage_of_lower_than_lower_v", "percentage_of_higher_than_upper_v",
             "totally_controllable_ratio", "average_voltage_deviation", "average_voltage", "max_voltage_drop_deviation",
             "max_voltage_rise_deviation", "total_line_loss", "q_loss", "destroy")


def snapshot(env, with_obs=True):
    """get_obs() advances the obs history when history > 1 (:303-315), so it is called exactly once per step here"""
    return dict(**(dict(obs=np.array(env.get_obs())) if with_obs else {}), state=env.get_state(), v=env._get_res_bus_v(), p=env._get_res_bus_active(),
                q=env._get_res_bus_reactive(), pl=env._get_res_line_loss(), sgen_p=env._get_sgen_active(),
                sgen_q=env._get_sgen_reactive(), avail=np.array(env.get_avail_actions(), dtype=np.float64))


def run_scenario(name):
    case, over, (day, hour, interval), n_steps, unsolv, noisy = SCENARIOS[name]
    net, prof_q, args, _, _, _ = scenario_data(name, scaled=False)               # unscaled: the reference scales (:415)
    d = tempfile.mkdtemp(prefix="mapdn_ref_")
    save_netspec(net, os.path.join(d, "netspec.npz"))
    save_profiles_csv(prof_q, d, float_format=f"%.{DIGITS}g")
    args["data_path"] = d
    env = VoltageControl(args)                       # the reference constructor: loaders, stds, s_max, first reset()
    out = {}
    if noisy:
        # replay of numpy's global MT19937 stream in the reference's call order (reset: :111-122, 498-508, 337)
        rs = np.random.RandomState(args["seed"])
        ns, nl = net.n_sgen, net.n_load
        h, dd, iv = rs.choice(24), rs.choice(env.pv_data.index[-1].__sub__(env.pv_data.index[0]).days - 1), rs.choice(20)
        draws = dict(init_time=np.array([dd, h, iv]), init_n_pv=rs.randn(ns), init_n_lp=rs.randn(nl), init_n_lq=rs.randn(nl),
                     init_u=rs.uniform(env.action_space.low, env.action_space.high, ns))
        assert (env._episode_start_day, env._episode_start_hour, env._episode_start_interval) == (dd, h, iv)
        for k, v in snapshot(env).items():
            out[f"init_{k}"] = v                      # state right after the constructor's own random reset()
        out.update({f"draw_{k}": v for k, v in draws.items()})
    obs, state = env.manual_reset(day, hour, interval)
    if noisy:
        out["draw_reset_u"] = rs.uniform(env.action_space.low, env.action_space.high, net.n_sgen)
    out["reset_obs"], out["reset_state"] = np.array(obs), state
    for k, v in snapshot(env, with_obs=False).items():
        out[f"reset_{k}"] = v
    acts = actions_for(name, net.n_sgen, n_steps, unsolv, args["action_scale"], args["action_bias"])
    rec = {k: [] for k in ("reward", "terminated", "info", "obs", "state", "v", "p", "q", "pl", "sgen_p", "sgen_q", "steps", "sum_rewards")}
    noise = {k: [] for k in ("n_pv", "n_lp", "n_lq")}
    for t in range(n_steps):
        r, term, info = env.step(acts[t], add_noise=noisy)
        if noisy:
            noise["n_pv"].append(rs.randn(net.n_sgen)); noise["n_lp"].append(rs.randn(net.n_load)); noise["n_lq"].append(rs.randn(net.n_load))
        s = snapshot(env)
        rec["reward"].append(r); rec["terminated"].append(float(term)); rec["info"].append([float(info[k]) for k in INFO_KEYS])
        for k in ("obs", "state", "v", "p", "q", "pl", "sgen_p", "sgen_q"):
            rec[k].append(s[k])
        rec["steps"].append(env.steps); rec["sum_rewards"].append(env.sum_rewards)
    if noisy:   # the replayed stream must be where the reference's global stream is
        assert np.random.get_state()[1][:8].tolist() == rs.get_state()[1][:8].tolist() and np.random.get_state()[2] == rs.get_state()[2]
        out.update({f"draw_{k}": np.array(v) for k, v in noise.items()})
    out["actions"] = acts
    out.update({f"step_{k}": np.array(v) for k, v in rec.items()})
    out["meta_sizes"] = np.array([env.n_agents, env.n_actions, env.obs_size, env.state_size, env.episode_limit])
    out["meta_action_space"] = np.array([env.action_space.low, env.action_space.high])
    out["meta_stds_smax"] = np.concatenate([env.pv_std, env.active_demand_std, env.reactive_demand_std, env.s_max])
    return out


def main():
    only = sys.argv[1:]
    for name in (only or SCENARIOS):
        out = run_scenario(name)
        path = os.path.join(HERE, f"env_ref_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays -> {os.path.basename(path)} ({os.path.getsize(path)} B); "
              f"rewards {np.round(out['step_reward'], 4).tolist()} term {out['step_terminated'].tolist()}")
    if only:
        return
    # ---- pure functions of the reference ---------------------------------------------------------
    grid = np.concatenate([np.linspace(0.5, 1.5, 201), np.array([0.95, 1.05, 1.0, 0.9499999, 1.0500001, 2.0, 2.5, 3.0, -0.5, 1.0 - 1e-9])])
    fun = {f"barrier_{k}": np.asarray(f(grid), dtype=np.float64) for k, f in Voltage_Barrier.items()}
    import torch as th
    from collections import namedtuple
    from utilities.util import translate_action          # the reference function (utilities/util.py:123-135)
    A = namedtuple("A", "continuous action_bias action_scale")
    raw = th.tensor(np.linspace(-1.7, 1.7, 35).reshape(1, 5, 7))
    ta = {}
    for i, (b, s) in enumerate([(0.0, 0.8), (0.0, 0.6), (0.25, 0.5)]):
        a, cp = translate_action(A(True, b, s), raw, None)
        ta[f"translate_{i}_args"] = np.array([b, s]); ta[f"translate_{i}_out"] = np.asarray(cp, dtype=np.float64)
        ta[f"translate_{i}_raw"] = a.numpy()
    np.savez_compressed(os.path.join(HERE, "env_ref_functions.npz"), grid=grid, translate_in=raw.numpy(), **fun, **ta)
    print("functions: barriers on", grid.shape[0], "points, translate_action x3")


if __name__ == "__main__":
    main()
