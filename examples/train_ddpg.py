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

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm); 'gloo' only to exercise the N-rank "
                                                      "data-parallel learner with ranks sharing GPUs (pre-flight on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed at world size 1 as well: the learner's broadcast, flat "
                                                              "gradient all-reduce and early-exit all-reduce run through the backend on one rank")
    ap.add_argument("--phases", action="store_true", help="per-phase device time of every episode in the JSON line (CUDA events around "
                                                         "replay insertion / sampling / value update / policy update / target update; rollout = the rest)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.backend != "nccl":                                   # test mode: ranks may share a GPU
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                   # --force-dist without a launcher: a one-rank rendezvous on the loopback
            import socket
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    from mapdn_amd import e2e

    def emit(line):
        if use_dist:
            line["dist"] = {"backend": a.backend, "world_size": world, "rccl_loaded": "librccl" in open("/proc/self/maps").read()}
        if rank == 0:
            print(json.dumps(line), flush=True)
            if a.log:
                with open(a.log, "a") as f:
                    f.write(json.dumps(line) + "\n")
    e2e.run(case=a.case, envs=a.envs, alg=a.alg, episodes=a.episodes, max_steps=a.max_steps, intensity=a.intensity, batch_size=a.batch_size,
            updates_per_env_step=a.updates_per_env_step, replay_steps=a.replay_steps, update_freq=a.update_freq,
            voltage_barrier=a.voltage_barrier, phases=a.phases, device=dev, rank=rank, world=world, save=a.save, on_line=emit,
            check_replicas=use_dist)
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
