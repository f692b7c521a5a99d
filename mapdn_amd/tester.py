"""Deterministic evaluation of a trained policy on the batched env — the counterpart of `PGTester`
(utilities/tester.py:8-106) and of the record files test.py pickles (test.py:105-114).

`run(day, hour, quarter)`: every env of the batch is reset to the same instant without noise
(`manual_reset`, voltage_control_env.py:137-176), the policy acts greedily (`status='test'`), and the
per-step tester getters (voltage_control_env.py:625-647) are recorded.  The returned dict has the
reference's keys — `pv_active, pv_reactive, bus_active, bus_reactive, bus_voltage, line_loss`, each a
list (initial state + one entry per step) of float64 numpy vectors of env `env_index` — so the plotting
scripts that read `test_record_*.pickle` keep working.  `batch_run(num_episodes)`: B episodes at a time
from random starts, returning `{'mean_test_<info key>': (mean, 2·std)}` over all steps of all episodes
(tester.py:65-99).  The whole rollout stays on the device; one host copy per step in `run`, none in
`batch_run`.
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from ._lib import INFO_KEYS
from .rollout import translate_action

RECORD_KEYS = {"pv_active": "sgen_p", "pv_reactive": "sgen_q", "bus_active": "p_mw", "bus_reactive": "q_mvar",
               "bus_voltage": "vm_pu", "line_loss": "pl_mw"}            # tester.py:26-39 -> VoltageControlBatch.results()


class PGTester:
    def __init__(self, args, behaviour_net, env):
        self.args, self.env = args, env
        self.behaviour_net = behaviour_net.to(env.device).eval()        # tester.py:11

    def _snapshot(self, record, env_index):
        res = self.env.results()
        for k, src in RECORD_KEYS.items():
            record[k].append(res[src][env_index].cpu().numpy())

    @torch.no_grad()
    def _act(self, obs, last_hid, avail):
        action, _, _, _, hid = self.behaviour_net.get_actions(obs, "test", False, avail, False, last_hid)
        return translate_action(action.squeeze(-1), self.args.action_scale, self.args.action_bias), hid

    def run(self, day, hour, quarter, env_index: int = 0):
        env, net = self.env, self.behaviour_net
        obs, _ = env.manual_reset(day, hour, quarter)
        obs = obs.float().clone()
        last_hid = net.init_hidden(env.n_envs)
        avail = env.get_avail_actions()
        record = {k: [] for k in RECORD_KEYS}
        self._snapshot(record, env_index)
        for t in range(self.args.max_steps):
            actual, hid = self._act(obs, last_hid, avail)
            _, done, _ = env.step(actual, add_noise=False)
            self._snapshot(record, env_index)
            obs, last_hid = env.get_obs().float().clone(), hid
            if bool(done[env_index]) or t == self.args.max_steps - 1:
                break
        return record

    def batch_run(self, num_episodes: int = 100):
        env, net = self.env, self.behaviour_net
        rounds = max(1, -(-int(num_episodes) // env.n_envs))
        samples = []                                                     # [steps, B, 11] per round, masked
        for _ in range(rounds):
            obs, _ = env.reset()
            obs = obs.float().clone()
            last_hid = net.init_hidden(env.n_envs)
            avail = env.get_avail_actions()
            alive = torch.ones(env.n_envs, dtype=torch.bool, device=env.device)
            infos, masks = [], []
            for t in range(self.args.max_steps):
                actual, hid = self._act(obs, last_hid, avail)
                _, done, info = env.step(actual, add_noise=False)
                infos.append(info.clone()); masks.append(alive.clone())
                alive = alive & ~done.bool()
                obs, last_hid = env.get_obs().float().clone(), hid
            samples.append(torch.stack(infos)[torch.stack(masks)])       # [n_valid, 11]
        allv = torch.cat(samples).double()
        mean, std = allv.mean(0).tolist(), allv.std(0, unbiased=False).tolist()       # np.mean / np.std (tester.py:93-94)
        return {"mean_test_" + k: (m, 2.0 * s) for k, m, s in zip(INFO_KEYS, mean, std)}

    @staticmethod
    def save_record(record, path):
        with open(path, "wb") as f:                                      # test.py:107-108
            pickle.dump(record, f, pickle.HIGHEST_PROTOCOL)

    @staticmethod
    def print_info(stat):
        print("\n".join(["Test Results:"] + [f"{k}: mean: {v[0]:2.4f}, \t2std: {v[1]:2.4f}" for k, v in stat.items()]))
