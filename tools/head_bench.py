#!/usr/bin/env python
"""Micro-benchmark of the learner's critic trunk at the end-to-end configuration's size (case322: 32 steps x 8192 envs x 38 agents = 10 M
rows of 64): the one-launch head (csrc/critic.hip) against the round-5 route (LayerNorm kernel + GEMM + relu-dot kernel), forward
and forward + backward, read rows and formed rows.  Prints ms per call and the f32-MFMA fraction of the head (peak 157.3 TFLOP/s)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.learner import MLPCritic, critic_head, layernorm_act_bc, make_alg_args     # noqa: E402

dev = torch.device("cuda:0")
nb, n = int(os.environ.get("NB", 262144)), int(os.environ.get("N", 38))
rows = nb * n
torch.manual_seed(0)
cr = MLPCritic(7, 1, make_alg_args(3, 5, 1)).to(dev)
base = torch.randn(nb, 64, device=dev, requires_grad=True)
pern = torch.randn(n, 64, device=dev, requires_grad=True)
dv = torch.randn(rows, 1, device=dev)


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def head_fwd():
    with torch.no_grad():
        return critic_head(cr, base, pern)


def head_fb():
    v = critic_head(cr, base, pern)
    v.backward(dv)


def old_fwd():
    with torch.no_grad():
        return cr.head(layernorm_act_bc(cr.layernorm, cr.act, base, pern))[0]


def old_fb():
    v = cr.head(layernorm_act_bc(cr.layernorm, cr.act, base, pern))[0]
    v.backward(dv)


gf = rows * 64 * 64 * 2 / 1e9
t = timeit(head_fwd); print(f"formed rows {rows}: head fwd        {t:7.3f} ms  ({gf / t:6.1f} TFLOP/s = {gf / t / 157.3:.2%} of f32 MFMA peak)")
t2 = timeit(head_fb); print(f"formed rows {rows}: head fwd + bwd  {t2:7.3f} ms  (bwd {t2 - t:.3f} ms: 3 products — pre recomputed, dxn, dW2 — {3 * gf / (t2 - t):6.1f} TFLOP/s = {3 * gf / (t2 - t) / 157.3:.2%})")
os.environ["MAPDN_FUSED_HEAD"] = "0"
t = timeit(old_fwd); print(f"formed rows {rows}: r05 route fwd   {t:7.3f} ms")
t2 = timeit(old_fb); print(f"formed rows {rows}: r05 route f + b {t2:7.3f} ms")
os.environ["MAPDN_FUSED_HEAD"] = "1"
if rows * 64 * 4 * 6 < 60e9:
    x = torch.randn(rows, 64, device=dev, requires_grad=True)
    t = timeit(lambda: critic_head(cr, x.detach())); print(f"read rows {rows}: head fwd          {t:7.3f} ms")
    def fb():
        critic_head(cr, x).backward(dv)
    t2 = timeit(fb); print(f"read rows {rows}: head fwd + bwd    {t2:7.3f} ms")
    os.environ["MAPDN_FUSED_HEAD"] = "0"
    def fb0():
        cr.trunk(x)[0].backward(dv)
    t2 = timeit(fb0); print(f"read rows {rows}: r05 route f + b   {t2:7.3f} ms")
