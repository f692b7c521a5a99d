"""The data-parallel learner on the GPU box (utilities/trainer.py:73-98 scaled out as SURVEY 8(e) prescribes: replay
sharded and GPU-resident, gradients averaged by ONE flat all-reduce per update):
  * RCCL first contact at world size 1 — examples/train_ddpg.py --force-dist: init_process_group("nccl", device_id=...), the broadcast of
    the state_dict, the flat gradient all-reduce, the reward-statistics all-reduce and the collective early-exit flag all run through
    librccl on device tensors;
  * the 8-rank path pre-flighted on one GPU (gloo standing in for RCCL, ranks sharing cuda:0): after every episode the eight replicas'
    state_dicts are bit-identical (PGTrainer.replicas_identical gathers a fingerprint) although every rank rolled out different envs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "examples", "train_ddpg.py")
SMALL = ["--case", "case33", "--envs", "64", "--episodes", "2", "--max-steps", "24", "--update-freq", "8", "--replay-steps", "16",
         "--intensity", "light", "--batch-size", "256"]


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["maddpg", "iddpg"])
def test_learner_rccl_first_contact_at_world_size_one(alg):
    r = subprocess.run([sys.executable, CLI, "--force-dist", "--alg", alg] + SMALL, capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _lines(r.stdout)
    assert len(lines) == 2
    last = lines[-1]
    assert last["dist"] == {"backend": "nccl", "world_size": 1, "rccl_loaded": True}
    assert last["replicas_identical"] is True
    c = last["learner_collectives"]
    # 2 episodes x 24 steps, an update round at steps 8, 16, 24, 32, 40 (10 value + 1 policy epochs each): every epoch is one gradient
    # all-reduce and one reward-statistics all-reduce; the early-exit flag is agreed on once per 16 steps
    assert c["broadcast"] > 20 and c["all_reduce_grads"] >= 44 and c["all_reduce_reward_stats"] == c["all_reduce_grads"] and c["all_reduce_flag"] >= 2
    assert last["mean_train_value_loss"] == last["mean_train_value_loss"]                       # finite, not NaN


@pytest.mark.gpu
def test_eight_rank_learner_preflight_keeps_replicas_identical():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()          # a free port, not a fixed one
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           CLI, "--backend", "gloo", "--alg", "maddpg"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=_clean_env())
    assert r.returncode == 0, r.stderr[-4000:]
    lines = _lines(r.stdout)
    assert len(lines) == 2, r.stdout                             # rank 0 prints; every rank ran the identity check (a divergence raises)
    for ln in lines:
        assert ln["n_gpus"] == 8 and ln["replicas_identical"] is True and ln["dist"]["world_size"] == 8 and ln["dist"]["backend"] == "gloo"
    assert lines[-1]["learner_collectives"]["all_reduce_grads"] >= 44
