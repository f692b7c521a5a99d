#!/usr/bin/env python
"""Run only the power-flow solve (mapdn_solve_only) N times on fixed inputs — for profiling k_nr_*."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="case141"); ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
net, prof = make_case(a.case)
scale = {"case33": 0.8, "case141": 0.6, "case322": 0.8, "case141_deep": 0.6}[a.case]
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=scale, action_bias=0.0), n_envs=a.envs, device="cuda:0")
rng = np.random.default_rng(0)
rows = rng.integers(0, prof.n_rows, a.envs)
pv = prof.pv[rows]
qs = rng.uniform(-scale, scale, (a.envs, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
ins = [torch.as_tensor(x, device="cuda:0") for x in (prof.load_p[rows], prof.load_q[rows], pv, qs)]
for _ in range(3):
    env.solve(*ins)
torch.cuda.synchronize()
env.nr_timing(True)
t0 = time.perf_counter()
for _ in range(a.iters):
    vm, va, it, cv = env.solve(*ins)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = env.nr_time_ms()
print(f"{a.case} B={a.envs} W={os.environ.get('MAPDN_NR_WAVES','auto')} L={os.environ.get('MAPDN_NR_LANES','auto')}: "
      f"nr kernel {ms/n*1e3:.1f} us avg over {n}; wall/solve {dt/a.iters*1e3:.3f} ms; iters mean {it.float().mean().item():.2f} max {it.max().item()} conv {cv.all().item()}")
