#!/usr/bin/env python
"""Evaluate policy checkpoints of examples/learning_curve.py on the CPU ORACLE env (oracle/env_restated.py) — the learner's
evidence is measured outside the product: the policy runs in PyTorch on the CPU, the env is the restated reference env.

    python tools/eval_checkpoints_on_oracle.py gpurun_out/curve --episodes 16 --out profiles/e2e/r03_learning_curve.json

Greedy policy (`status="test"`, models/model.py:265-302), the SAME seeded start times and noise for every checkpoint, plus a
uniform-random policy on the same episodes as the baseline.  Reports per checkpoint the mean step reward, the
`totally_controllable_ratio` and the share of out-of-control buses.
"""
import argparse
import glob
import json
import multiprocessing as mp
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}


def run_episode(job):
    kind, path, meta, ep = job
    from mapdn_amd.learner import DDPGNet, make_alg_args
    from mapdn_amd.netspec import make_case
    from mapdn_amd.rollout import translate_action
    from oracle.env_restated import INFO_KEYS, VoltageControlOracle
    torch.set_num_threads(1)
    case = meta["case"]
    net, prof = make_case(case)
    env = VoltageControlOracle(net, prof, dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0,
                                               voltage_barrier_type=meta["barrier"], seed=12345), env_id=ep, do_reset=False)
    obs, _ = env.reset()
    n, o = env.n_agents, len(obs[0])
    rng = np.random.default_rng(ep)
    pol = None
    if kind == "ckpt":
        args = make_alg_args(n, o, 1, SCALE[case], 0.0)
        pol = DDPGNet(args, meta["alg"], DDPGNet(args, meta["alg"]))      # behaviour net + its target, as PGTrainer builds it
        pol.load_state_dict(torch.load(path, map_location="cpu")["model_state_dict"])
        pol.eval()
        hid = pol.init_hidden(1)
        avail = torch.ones(1, n, 1)
    tot = dict(reward=0.0, tcr=0.0, out=0.0, q_loss=0.0, steps=0)
    for t in range(239):
        if pol is None:
            act = rng.uniform(-SCALE[case], SCALE[case], n)
        else:
            with torch.no_grad():
                x = torch.as_tensor(np.array(obs), dtype=torch.float32).unsqueeze(0)
                action, _, _, _, hid = pol.get_actions(x, "test", False, avail, False, hid)
                act = translate_action(action.squeeze(-1), SCALE[case], 0.0)[0].double().numpy()
        r, term, info = env.step(act)
        obs = env.get_obs()
        tot["reward"] += r; tot["tcr"] += info["totally_controllable_ratio"]; tot["out"] += info["percentage_of_v_out_of_control"]
        tot["q_loss"] += info["q_loss"]; tot["steps"] += 1
        if term:
            break
    return kind, path, {k: (v / tot["steps"] if k != "steps" else v) for k, v in tot.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--episodes", type=int, default=16)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    meta = json.load(open(os.path.join(a.dir, "meta.json")))
    ckpts = sorted(glob.glob(os.path.join(a.dir, "ckpt_*.pt")))
    jobs = [("random", "random", meta, e) for e in range(a.episodes)]
    jobs += [("ckpt", c, meta, e) for c in ckpts for e in range(a.episodes)]
    with mp.get_context("fork").Pool(a.procs) as pool:
        res = pool.map(run_episode, jobs, chunksize=1)
    rows = {}
    for kind, path, r in res:
        rows.setdefault(path, []).append(r)
    out = {"meta": meta, "eval": {"env": "oracle/env_restated.py (CPU restatement of the reference env)", "episodes_per_point": a.episodes,
                                  "policy": "greedy (status='test'); 'random' = uniform actions on the same episodes"}, "points": []}
    for path in ["random"] + ckpts:
        rs = rows[path]
        ep = -1 if path == "random" else int(os.path.basename(path)[5:9])
        out["points"].append({"train_episodes": ep, "name": os.path.basename(path),
                              "mean_test_reward": float(np.mean([r["reward"] for r in rs])),
                              "totally_controllable_ratio": float(np.mean([r["tcr"] for r in rs])),
                              "percentage_of_v_out_of_control": float(np.mean([r["out"] for r in rs])),
                              "q_loss": float(np.mean([r["q_loss"] for r in rs])),
                              "mean_episode_steps": float(np.mean([r["steps"] for r in rs]))})
        p = out["points"][-1]
        print(f"{p['name']:>14s}: reward {p['mean_test_reward']:+.4f}  controllable {p['totally_controllable_ratio']:.3f}  "
              f"v_out {p['percentage_of_v_out_of_control']:.4f}  q_loss {p['q_loss']:.4f}  steps {p['mean_episode_steps']:.0f}")
    train = os.path.join(a.dir, "train.jsonl")
    if os.path.exists(train):
        out["train_log"] = [json.loads(l) for l in open(train)]
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
