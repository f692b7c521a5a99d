#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the CPU oracle.

The reference itself (pandapower 2.7.0 + the MAPDN data) cannot run here, so these vectors are
outputs of the *restated* oracle (oracle/pp_restated.py, oracle/env_restated.py) on seeded inputs;
their purpose is (a) to pin the oracle against silent drift and (b) to let the GPU path be compared
with committed numbers, not only with a live oracle.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mapdn_amd.netspec import make_case  # noqa: E402
from oracle.env_restated import INFO_KEYS, VoltageControlOracle  # noqa: E402
from oracle.pp_restated import runpp_restated  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}


def solve_vectors(case, n=24, seed=11):
    net, prof = make_case(case)
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, prof.n_rows, n)
    act = rng.uniform(-SCALE[case], SCALE[case], (n, net.n_sgen))
    pl, ql, pv = prof.load_p[rows], prof.load_q[rows], prof.pv[rows]
    qs = act * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    out = dict(p_load=pl, q_load=ql, p_sgen=pv, q_sgen=qs, vm_pu=[], va_degree=[], iterations=[], pl_mw=[], p_slack=[])
    for e in range(n):
        r = runpp_restated(net, pl[e], ql[e], pv[e], qs[e])
        assert r.converged
        out["vm_pu"].append(r.vm_pu); out["va_degree"].append(r.va_degree)
        out["iterations"].append(r.iterations); out["pl_mw"].append(r.pl_mw); out["p_slack"].append(r.p_mw[0])
    return {k: np.asarray(v) for k, v in out.items()}


def episode_vectors(case, n_envs=3, T=10, barrier="bowl", seed=0):
    net, prof = make_case(case)
    args = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type=barrier, seed=seed)
    rng = np.random.default_rng(21)
    acts = rng.uniform(-SCALE[case], SCALE[case], (T, n_envs, net.n_sgen))
    rew = np.zeros((T, n_envs)); info = np.zeros((T, n_envs, len(INFO_KEYS)))
    obs = np.zeros((T + 1, n_envs, net.n_sgen, net.obs_size())); state = np.zeros((T + 1, n_envs, net.state_size()))
    start = np.zeros(n_envs, np.int64)
    for e in range(n_envs):
        o = VoltageControlOracle(net, prof, args, env_id=e, do_reset=False)
        ob, st = o.reset()
        start[e] = o._episode_start
        obs[0, e], state[0, e] = np.array(ob), st
        for t in range(T):
            r, term, inf = o.step(acts[t, e])
            rew[t, e] = r
            info[t, e] = [inf[k] for k in INFO_KEYS]
            obs[t + 1, e], state[t + 1, e] = np.array(o.get_obs()), o.get_state()
    return dict(actions=acts, reward=rew, info=info, obs=obs, state=state, start_rows=start,
                barrier=np.array(barrier), seed=np.array(seed))


if __name__ == "__main__":
    for case in ("case33", "case141", "case322"):
        np.savez_compressed(os.path.join(HERE, f"solve_{case}.npz"), **solve_vectors(case))
    np.savez_compressed(os.path.join(HERE, "episode_case33.npz"), **episode_vectors("case33"))
    np.savez_compressed(os.path.join(HERE, "episode_case141.npz"), **episode_vectors("case141", n_envs=2, T=6, barrier="l1"))
    print("golden fixtures written to", HERE)
