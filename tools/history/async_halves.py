#!/usr/bin/env python
"""Throughput of ONE GPU's env batch stepped as P independent sub-batches on P streams (each sub-batch is synchronous with its own
actions / obs; the sub-batches do not wait for each other — the double-buffered rollout pattern): the solver launch of one
sub-batch runs beside the wide launches of another.  Prints env-steps/s for P = 1, 2, 4.  usage: async_halves.py [case] [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case

SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8, "case141_deep": 0.6}


def run(case, B, P, steps=480, warmup=24, repeats=5):
    dev = torch.device("cuda:0")
    net, prof = make_case(case)
    args = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)
    b = B // P
    envs = [VoltageControlBatch(net, prof, args, n_envs=b, device=dev, env_id_offset=p * b) for p in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    acts = torch.empty(256, B, net.n_sgen, dtype=torch.float32, device=dev).uniform_(-SCALE[case], SCALE[case], generator=gen)
    torch.cuda.synchronize()
    n_ep = [0] * P
    for p in range(P):
        with torch.cuda.stream(streams[p]):
            envs[p].reset()

    def one_round(i):
        for p in range(P):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[i % 256, p * b:(p + 1) * b])
                envs[p].get_obs()
                n_ep[p] += 1
                if n_ep[p] >= envs[p].episode_limit - 1:
                    envs[p].reset(); n_ep[p] = 0
    for i in range(warmup):
        one_round(i)
    times = []
    for r in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one_round(i)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[len(times) // 2]
    for e in envs:
        e.close()
    return B * steps / dt, dt / steps * 1e6


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "case141"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    for P in (1, 2, 4):
        v, us = run(case, B, P)
        print(f"{case} x {B}: {P} sub-batch(es) of {B // P} envs on {P} stream(s): {v / 1e6:.2f} M env-steps/s, {us:.1f} us per round of {B} env-steps", flush=True)
