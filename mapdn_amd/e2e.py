"""End-to-end MADDPG / IDDPG loop on the batched GPU env (BASELINE.json configs[4]; SURVEY 8(f) rows 1-3 wired together): rollout of B envs
per GPU through the HIP hot path, GPU-resident replay, DDPG updates.  One function, used by examples/train_ddpg.py (the CLI) and by
bench.py's `e2e` block (so that a driver run records the loop the env feeds, not only the env).

Update intensity.  The reference (one env) runs 10 value + 1 policy update of batch 32 every 60 env-steps (models/model.py:39-52,
args/default.yaml): 11 * 32 / 60 = 5.87 sampled transitions per env-step.  With B envs one batched step inserts B transitions, so
  intensity "reference": the same 11 updates per 60 batched steps on batches of 32 * B transitions — a contiguous replay window of 32
        consecutive steps of every env, i.e. per env exactly the reference's window — = 5.87 sampled transitions per env-step;
  intensity "light": batches of `batch_size` transitions (round 2's setting: 11 * 4096 / (60 * B) per env-step);
  updates_per_env_step = X: batches of 32 * B, update epochs scaled so that X transitions are sampled per env-step.
"""
from __future__ import annotations

import time

REFERENCE_RATIO = 11 * 32 / 60.0
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}


def run(case="case322", envs=8192, alg="maddpg", episodes=3, max_steps=240, intensity="reference", batch_size=4096,
        updates_per_env_step=None, replay_steps=64, update_freq=60, voltage_barrier="bowl", phases=True, device=None, rank=0, world=1,
        save=None, on_line=None, check_replicas=False):
    """Runs `episodes` training episodes and returns one dict per episode (rank 0 semantics: env_steps_per_s is the whole job's).
    `on_line(dict)` is called after every episode (the CLI prints / logs there)."""
    import numpy as np
    import torch
    from .env import VoltageControlBatch
    from .learner import PGTrainer, make_alg_args
    from .netspec import make_case
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(1 + rank); np.random.seed(1 + rank)
    net, prof = make_case(case)
    env_args = dict(episode_limit=max_steps, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type=voltage_barrier, seed=0)
    env = VoltageControlBatch(net, prof, env_args, n_envs=envs, device=dev, env_id_offset=rank * envs, copy=True)
    v_ep, p_ep = 10, 1
    if intensity == "light" and updates_per_env_step is None:
        batch = batch_size
    else:
        batch = 32 * envs                                       # 32 consecutive steps of every env
        if updates_per_env_step is not None:
            total = updates_per_env_step * update_freq * envs / batch
            p_ep = max(1, round(total / 11)); v_ep = max(1, round(total - p_ep))
    ratio = (v_ep + p_ep) * batch / (update_freq * envs)
    args = make_alg_args(env.n_agents, env.obs_size, env.n_actions, SCALE[case], 0.0, max_steps=max_steps,
                         batch_size=batch, replay_buffer_size=envs * max(replay_steps, 2 * batch // envs),
                         behaviour_update_freq=update_freq, target_update_freq=2 * update_freq, num_eval_episodes=envs,
                         value_update_epochs=v_ep, policy_update_epochs=p_ep)
    trainer = PGTrainer(args, alg, env, device=dev)
    trainer.profile_phases = bool(phases)
    lines = []
    try:
        for ep in range(episodes):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            stat = {}
            trainer.train_process(stat)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            line = {"episode": ep, "alg": alg, "case": case, "n_gpus": world, "envs_per_gpu": envs,
                    "intensity": intensity, "batch_size": batch, "value_epochs": v_ep, "policy_epochs": p_ep,
                    "sampled_transitions_per_env_step": ratio, "reference_ratio": REFERENCE_RATIO,
                    "env_steps_per_s": world * envs * max_steps / dt, "seconds": dt,
                    "replay_transitions": len(trainer.replay_buffer),
                    "hbm_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
            if phases:
                ph = trainer.phase_seconds()
                ph["rollout_and_host"] = dt - sum(ph.values())
                line["phase_seconds"] = {k: round(v, 4) for k, v in ph.items()}
                line["phase_share"] = {k: round(v / dt, 4) for k, v in ph.items()}
            if check_replicas or world > 1:                     # data-parallel ranks must hold bit-identical replicas after every episode
                line["replicas_identical"] = bool(trainer.replicas_identical())
                line["learner_collectives"] = dict(trainer.collectives)
                if not line["replicas_identical"]:
                    raise RuntimeError(f"rank {rank}: the data-parallel replicas diverged in episode {ep}")
            line.update({k: v for k, v in stat.items() if k in (
                "mean_train_reward", "mean_train_value_loss", "mean_train_policy_loss", "mean_train_totally_controllable_ratio",
                "mean_train_q_loss")})
            lines.append(line)
            if on_line is not None:
                on_line(line)
        if save and rank == 0:
            trainer.save(save)
    finally:
        env.close()
    return lines
