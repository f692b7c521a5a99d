"""TEST INFRASTRUCTURE — one-env CPU restatement of ``VoltageControl``
(/root/reference/environments/var_voltage_control/voltage_control_env.py), pandas replaced by numpy,
``pp.runpp`` replaced by oracle/pp_restated.py, the global MT19937 stream replaced by the keyed
Philox of oracle/philox.py (same mapping as the HIP path).  Distributed mode only: the reference's
decentralised mode dies with ``KeyError: 'sgen0'`` at voltage_control_env.py:239 (clusters are keyed
``zone{i}`` there), so there is no behaviour to restate.

PARITY UNPINNED for the power flow (see pp_restated.py header); the env logic is restated from
in-repo reference source and each method cites the lines it follows.
"""
from __future__ import annotations

import numpy as np

from . import philox
from .pp_restated import runpp_restated

INFO_KEYS = (  # order == column order of the product's info tensor
    "percentage_of_v_out_of_control", "percentage_of_lower_than_lower_v",
    "percentage_of_higher_than_upper_v", "totally_controllable_ratio",
    "average_voltage_deviation", "average_voltage", "max_voltage_drop_deviation",
    "max_voltage_rise_deviation", "total_line_loss", "q_loss", "destroy",
)

DEFAULT_ARGS = dict(  # args/env_args/var_voltage_control.yaml:3-20
    voltage_barrier_type="l1", voltage_weight=1.0, q_weight=0.1, line_weight=None, dq_dv_weight=None,
    history=1, pv_scale=1.0, demand_scale=1.0,
    state_space=["pv", "demand", "reactive", "vm_pu", "va_degree"],
    v_upper=1.05, v_lower=0.95, episode_limit=240, action_scale=0.8, action_bias=0.0,
    mode="distributed", reset_action=True, seed=0,
)


# ---- voltage_barrier/*.py ----------------------------------------------------------------------
def barrier_l1(v, v_ref=1.0):                    # l1.py:5-8
    return np.abs(v - v_ref)


def barrier_l2(v, v_ref=1.0):                    # l2.py:5-8
    return 2 * np.square(v - v_ref)


def barrier_courant_beltrami(v, v_lower=0.95, v_upper=1.05):   # courant_beltrami.py:5-8
    return np.square(np.maximum(0, v - v_upper)) + np.square(np.maximum(0, v_lower - v))


def barrier_bowl(v, v_ref=1.0, scale=.1):        # bowl.py:5-13
    normal = 1 / np.sqrt(2 * np.pi * scale ** 2) * np.exp(-0.5 * np.square(v - v_ref) / scale ** 2)
    return np.where(np.abs(v - v_ref) > 0.05, 2 * np.abs(v - v_ref) - 0.095, -0.01 * normal + 0.04)


def barrier_bump(v):                             # bump.py:5-13 (raw v, no v_ref)
    out = np.zeros_like(v)
    a = np.abs(v) < 1
    b = (~a) & (v > 1) & (v < 3)
    with np.errstate(all="ignore"):
        out[a] = np.exp(-1 / (1 - v[a] ** 4))
        out[b] = np.exp(-1 / (1 - (v[b] - 2) ** 4))
    return out


BARRIERS = dict(l1=barrier_l1, l2=barrier_l2, bowl=barrier_bowl, bump=barrier_bump,
                courant_beltrami=barrier_courant_beltrami)          # voltage_barrier_registry.py:9-15
BARRIER_IDS = dict(l1=0, l2=1, courant_beltrami=2, bowl=3, bump=4)


class VoltageControlOracle:
    """Single env.  `env_id` is the *global* env index used to key the RNG."""

    def __init__(self, net, prof, args=None, env_id=0, do_reset=True):
        a = dict(DEFAULT_ARGS)
        a.update(args or {})
        self.args = a
        self.net, self.prof = net, prof
        self.env_id = int(env_id)
        self._runpp_override = None                     # see _take_action
        self.seed = int(a["seed"])
        self.draw = 0
        self.episode_limit = a["episode_limit"]
        self.voltage_barrier = BARRIERS[a["voltage_barrier_type"]]
        self.voltage_weight, self.q_weight, self.line_weight = a["voltage_weight"], a["q_weight"], a["line_weight"]
        self.v_upper, self.v_lower = a["v_upper"], a["v_lower"]
        self.history = a["history"]
        self.state_space = a["state_space"]
        self.pv_std, self.active_demand_std, self.reactive_demand_std = prof.stds()        # :70-72
        self.s_max = prof.s_max(1.2)                                                         # :515-520
        self.low = -a["action_scale"] + a["action_bias"]                                     # :76
        self.high = a["action_scale"] + a["action_bias"]
        assert a["mode"] == "distributed"
        self.n_actions = 1
        self.n_agents = net.n_sgen                                                           # :80-81
        self.time_delta = prof.time_delta_min
        # "powergrid" state
        self.sgen_p = np.zeros(net.n_sgen)
        self.sgen_q = np.zeros(net.n_sgen)           # base net q_mvar (model.p value unknown -> 0)
        self.load_p = np.zeros(net.n_load)
        self.load_q = np.zeros(net.n_load)
        self.res = None
        self.steps = 1
        self.sum_rewards = 0.0
        if do_reset:
            obs, state = self.reset()
            self.obs_size = obs[0].shape[0]                                                  # :87
            self.state_size = state.shape[0]

    # ---- reset (:96-135) / manual_reset (:137-176) ----------------------------------------------
    def _begin_episode(self):
        self.steps = 1
        self.sum_rewards = 0.0
        if self.history > 1:
            self.obs_history = {i: [] for i in range(self.n_agents)}
        self.sgen_q = np.zeros(self.net.n_sgen)      # deepcopy(base_powergrid), :106

    def reset(self, start=None, add_noise=True):
        """start=None samples (hour, day, interval) as :111-113; else (day, hour, interval)."""
        self._begin_episode()
        solvable = False
        while not solvable:
            d = self.draw
            self.draw += 1
            if start is None:
                hour, day, interval = philox.start_time(
                    self.seed, self.env_id, d, self.prof.n_start_days(self.episode_limit),
                    self.prof.intervals_per_hour)
            else:
                day, hour, interval = start
            self._episode_start = self.prof.start_row(day, hour, interval)                   # :445
            self._set_demand_and_pv(add_noise=add_noise, draw=d)                             # :118
            if self.args["reset_action"]:                                                    # :120-122
                u = philox.uniforms(self.seed, self.env_id, d, philox.STREAM_ACTION, self.net.n_sgen)
                act = self.low + (self.high - self.low) * u                                  # :337
                self.sgen_q = self._clip_reactive_power(act, self.sgen_p)
            res = runpp_restated(self.net, self.load_p, self.load_q, self.sgen_p, self.sgen_q)
            solvable = res.converged
            if solvable:
                self.res = res
                self.res_sgen_q = self.sgen_q * self.net.sgen_scaling     # res_sgen.q_mvar = q_mvar * scaling
        return self.get_obs(), self.get_state()

    def manual_reset(self, day, hour, interval):
        return self.reset(start=(day, hour, interval), add_noise=False)                      # :159

    # ---- step (:178-211) --------------------------------------------------------------------------
    def step(self, actions, add_noise=True):
        last = (self.sgen_q.copy(), self.res, self.res_sgen_q.copy())                        # :181
        solvable = self._take_action(actions)                                                # :184
        if solvable:
            reward, info = self._calc_reward()
        else:
            q_loss = np.mean(np.abs(self.sgen_q))                                            # :189
            self.sgen_q, self.res, self.res_sgen_q = last                                    # :190
            reward, info = self._calc_reward()
            reward -= 200.
            info["destroy"] = 1.
            info["totally_controllable_ratio"] = 0.
            info["q_loss"] = q_loss
        d = self.draw
        self.draw += 1
        self._set_demand_and_pv(add_noise=add_noise, draw=d)                                 # :199
        self.steps += 1
        self.sum_rewards += reward
        terminated = bool(self.steps >= self.episode_limit or not solvable)                  # :204
        return reward, terminated, info

    def _take_action(self, actions):                                                         # :548-566
        self.sgen_q = self._clip_reactive_power(np.asarray(actions, dtype=np.float64), self.sgen_p)
        # `_runpp_override`: a result computed outside for exactly these inputs (bench.py's batched-numpy CPU baseline solves the
        # power flows of many envs as one vectorised Newton iteration and hands every env its share); consumed once
        res, self._runpp_override = self._runpp_override, None
        if res is None:
            res = runpp_restated(self.net, self.load_p, self.load_q, self.sgen_p, self.sgen_q)
        if res.converged:
            self.res = res
            self.res_sgen_q = self.sgen_q * self.net.sgen_scaling     # res_sgen.q_mvar = q_mvar * scaling
        return res.converged

    def _clip_reactive_power(self, reactive_actions, active_power):                          # :568-572
        return np.sqrt(self.s_max ** 2 - active_power ** 2) * reactive_actions

    def _set_demand_and_pv(self, add_noise=True, draw=0):                                    # :491-513
        t = self._episode_start + self.steps            # row `steps` of the episode window, :473-475
        pv = self.prof.pv[t].copy()
        active = self.prof.load_p[t].copy()
        reactive = self.prof.load_q[t].copy()
        if add_noise:
            n = philox.normals
            pv += self.pv_std * np.abs(n(self.seed, self.env_id, draw, philox.STREAM_PV, pv.shape[0]))
            active += self.active_demand_std * np.abs(
                n(self.seed, self.env_id, draw, philox.STREAM_LOAD_P, active.shape[0]))
            reactive += self.reactive_demand_std * np.abs(
                n(self.seed, self.env_id, draw, philox.STREAM_LOAD_Q, reactive.shape[0]))
        self.sgen_p, self.load_p, self.load_q = pv, active, reactive

    # ---- reward (:574-623) --------------------------------------------------------------------------
    def _calc_reward(self):
        info = {}
        v = self.res.vm_pu
        n = v.shape[0]
        out = (np.sum(v < self.v_lower) + np.sum(v > self.v_upper)) / n
        info["percentage_of_v_out_of_control"] = out
        info["percentage_of_lower_than_lower_v"] = np.sum(v < self.v_lower) / n
        info["percentage_of_higher_than_upper_v"] = np.sum(v > self.v_upper) / n
        info["totally_controllable_ratio"] = 0. if out > 1e-3 else 1.
        v_ref = 0.5 * (self.v_lower + self.v_upper)
        info["average_voltage_deviation"] = np.mean(np.abs(v - v_ref))
        info["average_voltage"] = np.mean(v)
        info["max_voltage_drop_deviation"] = np.max((v < self.v_lower) * (self.v_lower - v))
        info["max_voltage_rise_deviation"] = np.max((v > self.v_upper) * (v - self.v_upper))
        line_loss = np.sum(self.res.pl_mw)
        avg_line_loss = np.mean(self.res.pl_mw)
        info["total_line_loss"] = line_loss
        q_loss = np.mean(np.abs(self.res_sgen_q))
        info["q_loss"] = q_loss
        v_loss = np.mean(self.voltage_barrier(v)) * self.voltage_weight
        if self.line_weight is not None:
            loss = avg_line_loss * self.line_weight + v_loss
        elif self.q_weight is not None:
            loss = q_loss * self.q_weight + v_loss
        else:
            raise NotImplementedError("Please at least give one weight, either q_weight or line_weight.")
        info["destroy"] = 0.0
        return -loss, info

    # ---- observations (:213-316) ------------------------------------------------------------------
    def get_state(self):
        r = self.res
        state = []
        if "demand" in self.state_space:
            state += list(r.p_mw) + list(r.q_mvar)
        if "pv" in self.state_space:
            state += list(self.sgen_p)
        if "reactive" in self.state_space:
            state += list(self.sgen_q)
        if "vm_pu" in self.state_space:
            state += list(r.vm_pu)
        if "va_degree" in self.state_space:
            state += list(r.va_degree)
        return np.array(state)

    def get_obs(self):
        net, r = self.net, self.res
        obs_list = []
        for i in range(net.n_sgen):
            zone = net.sgen_zone[i]
            rows = net.zone_buses(int(zone))                      # :536 ascending bus index
            p = r.p_mw[rows].copy()
            q = r.q_mvar[rows].copy()
            for j in range(net.n_sgen):                           # :238-244 effective add-back
                if net.sgen_zone[j] == zone:
                    k = np.nonzero(rows == net.sgen_bus[j])[0]
                    assert k.shape[0] == 1, "sgen bus must lie in its own zone (pandas .loc KeyError otherwise)"
                    p[k[0]] += self.sgen_p[j]
                    q[k[0]] += self.sgen_q[j]
            obs = []
            if "demand" in self.state_space:
                obs += list(p) + list(q)
            if "pv" in self.state_space:
                obs.append(self.sgen_p[i])
            if "reactive" in self.state_space:
                obs.append(self.sgen_q[i])
            if "vm_pu" in self.state_space:
                obs += list(r.vm_pu[rows])
            if "va_degree" in self.state_space:
                obs += list(r.va_degree[rows] * np.pi / 180)
            obs_list.append(np.array(obs))
        m = max(o.shape[0] for o in obs_list)
        agents_obs = [np.concatenate([o, np.zeros(m - o.shape[0])]) for o in obs_list]      # :270-274
        if self.history > 1:                                                                 # :303-315
            out = []
            for i, obs in enumerate(agents_obs):
                if len(self.obs_history[i]) >= self.history - 1:
                    obs_ = np.concatenate(self.obs_history[i][-self.history + 1:] + [obs], axis=0)
                else:
                    zeros = [np.zeros_like(obs)] * (self.history - len(self.obs_history[i]) - 1)
                    obs_ = np.concatenate(zeros + self.obs_history[i] + [obs], axis=0)
                out.append(obs_.copy())
                self.obs_history[i].append(obs.copy())
            agents_obs = out
        return agents_obs

    def get_avail_actions(self):                                                             # :345-351
        return np.expand_dims(np.array([[1]] * self.n_agents), axis=0)

    # tester getters (:625-647)
    def _get_res_bus_v(self):
        return self.res.vm_pu.copy()

    def _get_res_bus_active(self):
        return self.res.p_mw.copy()

    def _get_res_bus_reactive(self):
        return self.res.q_mvar.copy()

    def _get_res_line_loss(self):
        return self.res.pl_mw.copy()

    def _get_sgen_active(self):
        return self.sgen_p.copy()

    def _get_sgen_reactive(self):
        return self.sgen_q.copy()
