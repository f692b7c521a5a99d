#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched VoltageControl hot path on N MI355X of one node.

One "step" = one pass of the hot path over the whole env batch of every rank:
    actions -> q clip -> Sbus -> Newton-Raphson power flow -> results/reward/info -> next profile row
    + noise -> get_obs                      (reference: step() + get_obs(), models/model.py:216,219)
Synthetic case141 (141 buses / 84 loads / 22 PV agents), 4096 envs per GPU (BASELINE.json configs[2],
the configuration `metric` is quoted on), float64 arithmetic.  Episodes are 240 steps
(var_voltage_control.yaml:16): whenever the batch terminates it is reset inside the timed region
(the reset's own power flow is extra work that is NOT counted as env-steps).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}     # reference train.py:34-42
HBM_PEAK_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP64_PEAK_TFLOPS = 78.6                                       # MI355X f64 vector peak


def algorithmic_bytes_per_env_step(env):
    """compulsory per-env traffic of one step()+get_obs(); shared constants excluded.
    in : p_load, q_load [nl], p_pv, action [ns]                  f64
    out: vm, va [nb] f64; obs [n_agents, obs_size] f32; reward f64 + terminated u8 + info[11] f64"""
    nl, ns, nb = env.n_load, env.n_sgen, env.n_bus
    return 8 * (2 * nl + 2 * ns) + 8 * 2 * nb + 4 * env.n_agents * env.obs_size + (8 + 1 + 8 * 11)


def measured_traffic(case, envs):
    """HBM bytes per k_nr_wtree launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py;
    FETCH_SIZE and WRITE_SIZE need separate passes, so bench.py cannot measure them live)."""
    path = os.path.join(ROOT, "profiles", f"r01_traffic_{case}_b{envs}.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path))["traffic_bytes_per_launch"]


def cpu_baseline(case, seconds=12.0):
    """Restated pandapower-equivalent CPU path (oracle/, numpy+scipy, 1 env, 1 core): step()+get_obs().
    pandapower 2.7.0 itself is not installable offline, hence kind='port'."""
    from mapdn_amd.netspec import make_case
    from oracle.env_restated import VoltageControlOracle
    net, prof = make_case(case)
    args = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)
    env = VoltageControlOracle(net, prof, args, env_id=0)
    rng = np.random.default_rng(0)
    n = 0
    for _ in range(5):
        env.step(rng.uniform(-SCALE[case], SCALE[case], net.n_sgen)); env.get_obs()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, term, _ = env.step(rng.uniform(-SCALE[case], SCALE[case], net.n_sgen))
        env.get_obs()
        n += 1
        if term:
            env.reset()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} sequential step()+get_obs() of one {case} env in {dt:.1f} s (numpy/scipy restatement "
                      f"of pandapower runpp + env logic; host has {len(os.sched_getaffinity(0))} cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--case", default="case141")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm); 'gloo' only to "
                                                        "exercise the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    dist = None
    if a.backend != "nccl":                       # test mode: ranks may share a GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.netspec import make_case
    from mapdn_amd.sharding import gather_rollout

    net, prof = make_case(a.case)
    B = a.envs
    args = dict(episode_limit=240, action_scale=SCALE[a.case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)
    env = VoltageControlBatch(net, prof, args, n_envs=B, device=dev, env_id_offset=rank * B)   # weak scaling
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    scale = SCALE[a.case]
    # fresh random actions for every step, drawn on the device BEFORE the timed region so that the
    # (excluded) policy costs nothing inside it; a ring of `n_act` distinct action tensors
    n_act = min(a.steps + a.warmup + 60, 256)
    acts = torch.empty(n_act, B, env.n_sgen, dtype=torch.float32, device=dev).uniform_(-scale, scale, generator=gen)
    steps_in_ep = [0]
    step_no = [0]

    def one_step():
        act = acts[step_no[0] % n_act]
        step_no[0] += 1
        r, term, info = env.step(act)
        env.get_obs()
        steps_in_ep[0] += 1
        if steps_in_ep[0] >= env.episode_limit - 1:           # all envs terminate together (:204)
            env.reset()
            steps_in_ep[0] = 0

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    env.reset()
    for _ in range(a.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    if dist is not None:                                      # end-of-rollout RCCL gather
        ret = env.episode_returns()
        allret = gather_rollout(ret if a.backend == "nccl" else ret.cpu())
        assert allret.shape[0] == world * B
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = env.stats()

    # ---- dominant kernel (k_nr_wtree) duration, HIP events on its launch stream, separate short pass
    env.nr_timing(True)
    for _ in range(min(a.steps, 60)):
        one_step()
    torch.cuda.synchronize(dev)
    nr_ms, nr_launches = env.nr_time_ms()
    env.nr_timing(False)

    if rank == 0:
        n_gpus = world
        value = n_gpus * B * a.steps / dt
        bytes_step = algorithmic_bytes_per_env_step(env)
        nr_avg_s = nr_ms / max(nr_launches, 1) * 1e-3
        achieved = bytes_step * B / nr_avg_s / 1e9
        out = {
            "metric": "env-steps/sec (whole node), case141 batch=4096, at 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.case} ({env.n_bus}-bus, {env.n_agents} agents), {B} parallel envs per GPU, "
                                   f"bowl voltage barrier, step()+get_obs(), 240-step episodes with in-region resets",
                       "envs_per_gpu": B, "global_envs": n_gpus * B, "obs_size": env.obs_size,
                       "parallelism": f"env-batch sharded x{n_gpus}, no data-path collective"},
            "nr_iterations": {"mean": stats["mean_nr_iters"], "max": stats["max_nr_iters"]},
            "roofline": {"bound": "hbm", "kernel": "k_nr_wtree", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(a.case, B),
                         "algorithmic_bytes_per_launch": bytes_step * B,
                         "algorithmic_bytes_per_env_step": bytes_step, "envs_per_launch": B,
                         "kernel_avg_ms": nr_avg_s * 1e3, "kernel_launches_timed": nr_launches},
        }
        # compute-side view: ~184 * nb f64 flops per NR iteration (SpMV, mismatch, Jacobian,
        # block-tree solve, update) x (iterations + 1 mismatch evaluation), against the f64 vector peak
        flops_step = 184.0 * env.n_bus * (stats["mean_nr_iters"] + 1.0)
        out["compute"] = {"algorithmic_flops_per_env_step": flops_step, "achieved_tflops": flops_step * B / nr_avg_s / 1e12,
                          "peak_tflops": FP64_PEAK_TFLOPS, "frac": flops_step * B / nr_avg_s / 1e12 / FP64_PEAK_TFLOPS,
                          "note": "f64 vector (no MFMA: the radial Jacobian is eliminated without fill, there is no dense contraction)"}
        if not a.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.case, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
