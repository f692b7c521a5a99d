#!/usr/bin/env python
"""General-topology solvers on meshed nets: env-steps/s of step()+get_obs() and the solver kernel's time.
  default          : k_nr_sparse (host-compiled block elimination program) — case33 with the Baran-Wu ties closed, or
                     case141 / case322 with a few tie lines added (--case, --ties)
  MAPDN_NR_DENSE=1 : k_nr_dense (dense LU with f64 MFMA; Jacobian in LDS up to 65 buses, in global memory beyond) + the MFMA
                     share of its LU; profile with
      rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -- python tools/general_bench.py
  --ties 0 with MAPDN_NR_SPARSE=1 / MAPDN_NR_DENSE=1: a general solver forced onto the radial feeder (vs the tree kernel)"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import add_lines, case33_meshed, make_case

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--ties", type=int, default=5)
ap.add_argument("--case", default="case33")
a = ap.parse_args()
base, prof = make_case(a.case)
TIES = {"case141": ([5, 40, 77, 100, 20], [77, 120, 130, 12, 66]), "case322": ([5, 40, 177, 200, 300], [77, 120, 30, 12, 150])}
if a.ties == 0:
    net = base                                                   # + MAPDN_NR_SPARSE=1 / MAPDN_NR_DENSE=1: a general solver on the radial feeder
elif a.case == "case33":
    net = case33_meshed(base, a.ties)
else:
    f, t = TIES[a.case]
    net = add_lines(base, f[:a.ties], t[:a.ties], 0.3, 0.2)
    net.name = f"{a.case}_meshed{a.ties}"
scale = {"case33": 0.8, "case141": 0.6, "case322": 0.8}[a.case]
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=scale, action_bias=0.0, voltage_barrier_type="bowl"),
                          n_envs=a.envs, device="cuda:0")
acts = torch.empty(64, a.envs, env.n_sgen, device="cuda:0").uniform_(-scale, scale)
env.reset()
for t in range(min(10, max(2, a.steps // 2))):
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
dense = os.environ.get("MAPDN_NR_DENSE", "0") == "1"
solver = "k_nr_dense" if dense else ("k_nr_tree" if env.is_radial and os.environ.get("MAPDN_NR_SPARSE", "0") != "1" else "k_nr_sparse")
if not dense:
    print(json.dumps({"net": net.name, "solver": solver, "radial": env.is_radial, "envs": a.envs, "env_steps_per_s": a.envs * a.steps / dt,
                      "ms_per_step": dt / a.steps * 1e3, "nr_kernel_us": ms / n * 1e3, "mean_nr_iters": it}))
    env.close()
    sys.exit(0)
print(json.dumps({"net": net.name, "solver": solver, "radial": env.is_radial, "envs": a.envs, "env_steps_per_s": a.envs * a.steps / dt,
                  "ms_per_step": dt / a.steps * 1e3, "nr_kernel_us": ms / n * 1e3, "mean_nr_iters": it, "N": N,
                  "lu_flops_per_iteration": lu_flops, "mfma_flops_per_iteration": mfma_flops, "mfma_instructions_per_iteration": tiles * 4,
                  "dense_tflops": a.envs * it * lu_flops / (ms / n * 1e-3) / 1e12,
                  "mfma_tflops": a.envs * it * mfma_flops / (ms / n * 1e-3) / 1e12}))
env.close()
