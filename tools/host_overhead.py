#!/usr/bin/env python
"""Host-side cost of one step()+get_obs() (ctypes + torch plumbing): enqueue-only timing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case
net, prof = make_case("case141")
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=0.6, action_bias=0.0), n_envs=int(sys.argv[1]) if len(sys.argv) > 1 else 4096, device="cuda:0")
env.reset()
a = torch.zeros(env.n_envs, env.n_sgen, device="cuda:0")
for _ in range(50): env.step(a); env.get_obs()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for i in range(N):
    env.step(a); env.get_obs()
    if i % 200 == 199: env.reset()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={env.n_envs}: host enqueue {(t1-t0)/N*1e6:.1f} us/step, total {(t2-t0)/N*1e6:.1f} us/step")

# ---- breakdown: raw ctypes calls with cached arguments
from mapdn_amd import _lib
lib = env._lib; h = env._h; st = env._stream()
ap = a.data_ptr(); rp = env._reward.data_ptr(); tp = env._term.data_ptr(); ip = env._info.data_ptr()
ob = env.get_obs(); op = ob.data_ptr()
torch.cuda.synchronize()
env.reset()
t0 = time.perf_counter()
for i in range(200):
    lib.mapdn_step(h, ap, 0, 1, rp, tp, ip, st)
t1 = time.perf_counter()
for i in range(200):
    lib.mapdn_get_obs(h, op, 0, st)
t2 = time.perf_counter()
torch.cuda.synchronize()
print(f"raw ctypes: mapdn_step {(t1-t0)/200*1e6:.1f} us, mapdn_get_obs {(t2-t1)/200*1e6:.1f} us")
env.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    env.step(a)
t1 = time.perf_counter()
for i in range(200):
    env.get_obs()
t2 = time.perf_counter()
torch.cuda.synchronize()
print(f"python methods: step {(t1-t0)/200*1e6:.1f} us, get_obs {(t2-t1)/200*1e6:.1f} us")
