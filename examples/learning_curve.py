#!/usr/bin/env python
"""Seeded IDDPG / MADDPG training on the batched GPU env with periodic checkpoints — the evidence run that the learner
learns.  Checkpoints (`model.pt` layout, train.py:119) are evaluated afterwards on the CPU ORACLE env
(tools/eval_checkpoints_on_oracle.py), i.e. outside the product.

    python examples/learning_curve.py --case case33 --alg iddpg --envs 256 --episodes 300 --out gpurun_out/curve

Update schedule = the reference's per env (10 value + 1 policy update per 60 steps on a 32-step replay window of every env:
5.87 sampled transitions per env-step, models/model.py:39-52).  Writes <out>/train.jsonl (one line per episode) and
<out>/ckpt_<episode>.pt (episode 0 = the untrained network).
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
    ap.add_argument("--case", default="case33")
    ap.add_argument("--alg", default="iddpg", choices=["maddpg", "iddpg"])
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--episodes", type=int, default=300)
    ap.add_argument("--ckpt-every", type=int, default=25)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--voltage-barrier", default="bowl")
    ap.add_argument("--out", default="gpurun_out/curve")
    a = ap.parse_args()
    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.learner import PGTrainer, make_alg_args
    from mapdn_amd.netspec import make_case

    os.makedirs(a.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(a.seed); np.random.seed(a.seed)
    net, prof = make_case(a.case)
    env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=SCALE[a.case], action_bias=0.0,
                                               voltage_barrier_type=a.voltage_barrier, seed=a.seed), n_envs=a.envs, device=dev, copy=True)
    batch = 32 * a.envs
    args = make_alg_args(env.n_agents, env.obs_size, env.n_actions, SCALE[a.case], 0.0, max_steps=240, batch_size=batch,
                         replay_buffer_size=a.envs * 160, num_eval_episodes=a.envs)
    tr = PGTrainer(args, a.alg, env, device=dev)
    meta = dict(case=a.case, alg=a.alg, envs=a.envs, seed=a.seed, barrier=a.voltage_barrier, batch_size=batch,
                sampled_transitions_per_env_step=11 * batch / (60 * a.envs))
    json.dump(meta, open(os.path.join(a.out, "meta.json"), "w"))
    tr.save(os.path.join(a.out, "ckpt_0000.pt"))
    log = open(os.path.join(a.out, "train.jsonl"), "w")
    t_all = time.perf_counter()
    for ep in range(1, a.episodes + 1):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        stat = {}
        tr.train_process(stat)
        torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
        line = {"episode": ep, "seconds": dt, "env_steps_per_s": a.envs * 240 / dt}
        line.update({k: float(v) for k, v in stat.items() if k in (
            "mean_train_reward", "mean_train_totally_controllable_ratio", "mean_train_q_loss", "mean_train_value_loss",
            "mean_train_policy_loss", "mean_train_percentage_of_v_out_of_control")})
        log.write(json.dumps(line) + "\n"); log.flush()
        if ep % a.ckpt_every == 0 or ep == a.episodes:
            tr.save(os.path.join(a.out, f"ckpt_{ep:04d}.pt"))
            print(json.dumps(line), flush=True)
    print(f"done: {a.episodes} episodes x {a.envs} envs in {time.perf_counter() - t_all:.1f} s")
    env.close()


if __name__ == "__main__":
    main()
