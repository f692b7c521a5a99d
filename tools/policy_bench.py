#!/usr/bin/env python
"""Rollout-side policy forward: one fused HIP launch (mapdn_policy_forward) vs the PyTorch modules, rows = envs x agents."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.learner import DDPGNet, make_alg_args
for n_agents, obs_dim, B in ((22, 58, 4096), (38, 82, 8192)):
    args = make_alg_args(n_agents, obs_dim, 1, action_scale=0.8, action_bias=0.0)
    net = DDPGNet(args, "maddpg").to("cuda:0")
    obs = torch.randn(B, n_agents, obs_dim, device="cuda:0"); hid = torch.randn(B, n_agents, 64, device="cuda:0")
    for fused in ("1", "0"):
        os.environ["MAPDN_FUSED_POLICY"] = fused
        with torch.no_grad():
            for _ in range(3):
                net.policy(obs, hid)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                m, _, h = net.policy(obs, hid)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        rows = B * n_agents
        flops = 2.0 * rows * (64 * obs_dim + 2 * 3 * 64 * 64 + 64)
        print(f"agents {n_agents} obs {obs_dim} envs {B}: rows {rows}  fused={fused}  {dt*1e6:8.1f} us  {flops/dt/1e12:6.2f} TFLOP/s fp32")
