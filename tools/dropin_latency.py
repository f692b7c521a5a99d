#!/usr/bin/env python
"""Latency of the B = 1 drop-in adapter (`mapdn_amd.env.VoltageControl`): the reference's calling
pattern — numpy action in, Python float / bool / dict / list-of-numpy out, one env."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControl
from mapdn_amd.netspec import make_case

for case, scale in (("case33", 0.8), ("case141", 0.6), ("case322", 0.8)):
    net, prof = make_case(case)
    env = VoltageControl(dict(net=net, profiles=prof, episode_limit=240, action_scale=scale, action_bias=0.0,
                              voltage_barrier_type="bowl", mode="distributed", seed=0))
    env.reset()
    rng = np.random.default_rng(0)
    for _ in range(20):
        env.step(rng.uniform(-scale, scale, env.n_agents)); env.get_obs()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        r, term, info = env.step(rng.uniform(-scale, scale, env.n_agents)); obs = env.get_obs(); n += 1
        if term:
            env.reset()
    dt = time.perf_counter() - t0
    print(f"{case}: B=1 drop-in adapter {n / dt:.0f} step()+get_obs() per s ({dt / n * 1e6:.0f} us each); "
          f"types {type(r).__name__}, {type(term).__name__}, {type(info).__name__}, list[{type(obs[0]).__name__} {obs[0].dtype}]", flush=True)
    env.close()
