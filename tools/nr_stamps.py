#!/usr/bin/env python
"""Cycle breakdown of one k_nr_tree workgroup (debug build: MAPDN_EXTRA_FLAGS=-DMAPDN_NR_STAMPS python -m mapdn_amd.build --force).
Prints per-phase cycles (s_memtime, ~100 MHz-independent shader clock) of workgroup 0 / wave 0."""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd import _lib
from mapdn_amd.env import VoltageControlBatch
from mapdn_amd.netspec import make_case
ap = argparse.ArgumentParser(); ap.add_argument("--case", default="case141"); ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--rows", action="store_true")
ap.add_argument("--step", action="store_true", help="stamp a step() launch (with the fused reward / commit epilogue) instead of solve-only")
a = ap.parse_args()
net, prof = make_case(a.case)
scale = {"case33": 0.8, "case141": 0.6, "case322": 0.8, "case141_deep": 0.6}[a.case]
env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=scale, action_bias=0.0, voltage_barrier_type="bowl"), n_envs=a.envs, device="cuda:0")
rng = np.random.default_rng(0)
rows = rng.integers(0, prof.n_rows, a.envs)
pv = prof.pv[rows]
qs = rng.uniform(-scale, scale, (a.envs, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
ins = [torch.as_tensor(x, device="cuda:0") for x in (prof.load_p[rows], prof.load_q[rows], pv, qs)]
if a.step:
    env.reset()
    act = torch.as_tensor(rng.uniform(-scale, scale, (a.envs, net.n_sgen)), device="cuda:0")
    for _ in range(3):
        env.step(act)                        # MODE_STEP: solve + epilogue
    torch.cuda.synchronize()
    print("step mode", env.stats())
else:
    for _ in range(3):
        vm, va, it, cv = env.solve(*ins)         # MODE_SOLVE: every env active, fixed inputs
    torch.cuda.synchronize()
    print("iters mean", it.float().mean().item(), "conv", cv.float().mean().item())
lib = _lib.load()
out = (ctypes.c_ulonglong * 4096)()
assert lib.mapdn_debug_stamps(out, 4096) == 0
n = out[0]
st = [(out[i] >> 48, out[i] & 0xffffffffffff) for i in range(1, n + 1)]
if n == 0:
    n = max(i for i in range(1, 4096) if out[i])
names = {1: "start", 2: "init done", 10: "fwd begin", 11: "fwd verdict", 12: "bwd begin", 30: "update pass", 31: "mismatch pass: A_pk written", 20: "solve end", 21: "epilogue body done", 22: "epilogue: buses done", 23: "epilogue: lines done", 24: "epilogue: combine + outputs done",
         100: "row full", 101: "row light", 102: "row flat", 110: "row bwd(flat G)", 111: "row bwd", 120: "mismatch pass turn"}
if any(sid >= 200 for sid, _ in st):       # fine mode: dump the sequence of one full fwd sweep and one bwd sweep
    seq = [(sid, c) for sid, c in st]
    out_lines = []
    for i in range(1, len(seq)):
        out_lines.append((seq[i][0], seq[i][1] - seq[i - 1][1]))
    import collections
    agg = collections.defaultdict(list)
    for sid, d in out_lines:
        agg[sid].append(d)
    for sid in sorted(agg):
        v = np.array(agg[sid]); print(f"stamp {sid}: n {len(v)} mean delta from previous stamp {v.mean():8.1f} min {v.min()} max {v.max()}")
    sys.exit(0)
prev = st[0][1]; t0 = prev
rows = []
for sid, c in st:
    if sid >= 100:
        rows.append((sid, c)); continue
    if rows:
        d = np.diff([prev] + [c2 for _, c2 in rows])
        print(f"    {names[rows[0][0]]:18s} x{len(rows):3d}: total {rows[-1][1]-prev:7d}  per row mean {d.mean():7.1f} min {d.min()} max {d.max()}" + (f"  {d.tolist()}" if a.rows else ""))
        prev = rows[-1][1]; rows = []
    print(f"{names.get(sid, sid):20s} +{c - prev:7d}  (t = {c - t0})")
    prev = c
print("W L:", os.environ.get("MAPDN_NR_WAVES"), os.environ.get("MAPDN_NR_LANES"))
