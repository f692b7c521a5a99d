#!/usr/bin/env python
"""End-to-end MADDPG / IDDPG training on the batched GPU env (BASELINE.json configs[4]): rollout of B
envs per GPU through the HIP hot path, GPU-resident replay, DDPG updates, all on the device.

    python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/train_ddpg.py --case case322 --envs 8192          # 65536 envs, DP learner over RCCL

Mirrors train.py of the reference (env args from args/env_args/var_voltage_control.yaml, per-scenario
action scale train.py:34-42, `model.pt` checkpoint train.py:119).

Update intensity.  The reference (one env) runs 10 value + 1 policy update of batch 32 every 60 env-steps
(models/model.py:39-52, args/default.yaml): 11 * 32 / 60 = 5.87 sampled transitions per env-step.  With B envs one
batched step inserts B transitions, so
  --intensity reference (default): the same 11 updates per 60 batched steps on batches of 32 * B transitions — a
        contiguous replay window of 32 consecutive steps of every env, i.e. per env exactly the reference's window —
        = 5.87 sampled transitions per env-step, the reference's ratio;
  --intensity light: batches of --batch-size transitions (round 2's setting: 11 * 4096 / (60 * B) per env-step);
  --updates-per-env-step X: batches of 32 * B, update epochs scaled so that X transitions are sampled per env-step.
Prints one JSON line per episode on rank 0 (and appends it to --log).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="case322")
    ap.add_argument("--alg", default="maddpg", choices=["maddpg", "iddpg"])
    ap.add_argument("--envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--max-steps", type=int, default=240)
    ap.add_argument("--batch-size", type=int, default=4096, help="transitions per update with --intensity light")
    ap.add_argument("--intensity", default="reference", choices=["reference", "light"])
    ap.add_argument("--updates-per-env-step", type=float, default=None, help="sampled transitions per env-step (reference: 5.87)")
    ap.add_argument("--log", default=None, help="append the JSON lines to this file as well")
    ap.add_argument("--replay-steps", type=int, default=64, help="replay capacity in batched steps (x envs transitions)")
    ap.add_argument("--update-freq", type=int, default=60)
    ap.add_argument("--voltage-barrier", default="bowl")
    ap.add_argument("--save", default=None)
    ap.add_argument("--phases", action="store_true", help="per-phase device time of every episode in the JSON line (CUDA events around "
                                                         "replay insertion / sampling / value update / policy update / target update; rollout = the rest)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.learner import PGTrainer, make_alg_args
    from mapdn_amd.netspec import make_case

    torch.manual_seed(1 + rank); np.random.seed(1 + rank)
    net, prof = make_case(a.case)
    env_args = dict(episode_limit=a.max_steps, action_scale=SCALE[a.case], action_bias=0.0,
                    voltage_barrier_type=a.voltage_barrier, seed=0)
    env = VoltageControlBatch(net, prof, env_args, n_envs=a.envs, device=dev, env_id_offset=rank * a.envs, copy=True)
    ref_ratio = 11 * 32 / 60.0
    v_ep, p_ep = 10, 1
    if a.intensity == "light" and a.updates_per_env_step is None:
        batch = a.batch_size
    else:
        batch = 32 * a.envs                                       # 32 consecutive steps of every env
        if a.updates_per_env_step is not None:
            total = a.updates_per_env_step * a.update_freq * a.envs / batch
            p_ep = max(1, round(total / 11)); v_ep = max(1, round(total - p_ep))
    ratio = (v_ep + p_ep) * batch / (a.update_freq * a.envs)
    args = make_alg_args(env.n_agents, env.obs_size, env.n_actions, SCALE[a.case], 0.0, max_steps=a.max_steps,
                         batch_size=batch, replay_buffer_size=a.envs * max(a.replay_steps, 2 * batch // a.envs),
                         behaviour_update_freq=a.update_freq, target_update_freq=2 * a.update_freq, num_eval_episodes=a.envs,
                         value_update_epochs=v_ep, policy_update_epochs=p_ep)
    trainer = PGTrainer(args, a.alg, env, device=dev)
    trainer.profile_phases = a.phases
    for ep in range(a.episodes):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        stat = {}
        trainer.train_process(stat)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if rank == 0:
            line = {"episode": ep, "alg": a.alg, "case": a.case, "n_gpus": world, "envs_per_gpu": a.envs,
                    "intensity": a.intensity, "batch_size": batch, "value_epochs": v_ep, "policy_epochs": p_ep,
                    "sampled_transitions_per_env_step": ratio, "reference_ratio": ref_ratio,
                    "env_steps_per_s": world * a.envs * a.max_steps / dt, "seconds": dt,
                    "replay_transitions": len(trainer.replay_buffer),
                    "hbm_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
            if a.phases:
                ph = trainer.phase_seconds()
                ph["rollout_and_host"] = dt - sum(ph.values())
                line["phase_seconds"] = {k: round(v, 4) for k, v in ph.items()}
                line["phase_share"] = {k: round(v / dt, 4) for k, v in ph.items()}
            line.update({k: v for k, v in stat.items() if k in (
                "mean_train_reward", "mean_train_value_loss", "mean_train_policy_loss", "mean_train_totally_controllable_ratio",
                "mean_train_q_loss")})
            print(json.dumps(line), flush=True)
            if a.log:
                with open(a.log, "a") as f:
                    f.write(json.dumps(line) + "\n")
    if a.save and rank == 0:
        trainer.save(a.save)
    env.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
