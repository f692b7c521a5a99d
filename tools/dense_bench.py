#!/usr/bin/env python
"""General-topology solver (k_nr_dense) on the meshed 33-bus feeder (Baran-Wu tie lines closed): env-steps/s of
step()+get_obs(), NR kernel time, and the MFMA share of the dense LU.  Profile with
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -- python tools/dense_bench.py"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import case33_meshed, make_case

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--ties", type=int, default=5)
a = ap.parse_args()
base, prof = make_case("case33")
net = case33_meshed(base, a.ties) if a.ties else base          # --ties 0 + MAPDN_NR_DENSE=1: the dense solver on the radial feeder
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=0.8, action_bias=0.0, voltage_barrier_type="bowl"),
                          n_envs=a.envs, device="cuda:0")
acts = torch.empty(64, a.envs, env.n_sgen, device="cuda:0").uniform_(-0.8, 0.8)
env.reset()
for t in range(10):
    env.step(acts[t]); env.get_obs()
torch.cuda.synchronize()
env.nr_timing(True)
t0 = time.perf_counter()
for t in range(a.steps):
    env.step(acts[t % 64]); env.get_obs()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = env.nr_time_ms()
st = env.stats()
N = (2 * (net.n_bus - 1) + 15) // 16 * 16
NP = N // 16
tiles = sum((NP - k - 1) ** 2 for k in range(NP))
it = st["mean_nr_iters"]
lu_flops = 2.0 / 3.0 * N ** 3 + 2.0 * N * N
mfma_flops = tiles * 4 * 2 * 16 * 16 * 4
print(json.dumps({"net": net.name, "radial": env.is_radial, "envs": a.envs, "env_steps_per_s": a.envs * a.steps / dt,
                  "ms_per_step": dt / a.steps * 1e3, "nr_kernel_us": ms / n * 1e3, "mean_nr_iters": it, "N": N,
                  "lu_flops_per_iteration": lu_flops, "mfma_flops_per_iteration": mfma_flops, "mfma_instructions_per_iteration": tiles * 4,
                  "dense_tflops": a.envs * it * lu_flops / (ms / n * 1e-3) / 1e12,
                  "mfma_tflops": a.envs * it * mfma_flops / (ms / n * 1e-3) / 1e12}))
env.close()
