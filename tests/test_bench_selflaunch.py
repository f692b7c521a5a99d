"""`python bench.py --gpus N` must launch its own N ranks (the driver calls it exactly like the N = 1 case)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_self_launch_builds_a_torchrun_command(monkeypatch):
    bench = _bench_module()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)                  # an 8-GPU node
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_fails_fast_when_the_node_has_fewer_gpus_than_ranks(monkeypatch):
    """`--gpus N` under nccl (one rank per GPU over RCCL) on a node with fewer GPUs says so in one sentence before
    any rank starts; gloo (ranks may share a GPU: the pre-flight tests) is still launched"""
    import torch
    bench = _bench_module()
    called = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: called.append(cmd) or 0)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "exposes 1 GPU" in str(e.value.code) and "--backend gloo" in str(e.value.code) and not called
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--backend", "gloo"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(called) == 1


@pytest.mark.gpu
def test_bench_gpus2_without_a_launcher():
    """N = 2 on a one-GPU box: both ranks on cuda:0, gloo for the collective (the nccl path differs only in the backend
    string); the single JSON line must describe the whole job."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--envs", "64",
                        "--steps", "4", "--warmup", "2", "--no-traffic", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_envs"] == 128 and j["config"]["envs_per_gpu"] == 64
    assert j["steps"] == 4 and j["warmup"] == 2 and j["value"] > 0 and "cpu_baseline" not in j
    assert abs(j["value"] - 128 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_bench_gpus8_preflight_on_one_gpu(tmp_path):
    """Pre-flight of the driver's first real SCALE run: `python bench.py --gpus 8` exactly as the driver calls
    it — eight ranks, here all on cuda:0 with gloo standing in for RCCL (the nccl path differs in the backend string and in
    the device the gathered tensor lives on) — at the headline shard size, 4096 envs per rank.  One JSON line that describes the
    whole job, the NR kernel's time on every rank, the CPU baseline from rank 0; and rank r's episode returns are bit-identical to
    a ONE-rank run that covers the same global env ids (--env-id-offset r x 4096): results do not depend on the number of
    GPUs."""
    import numpy as np
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump8 = str(tmp_path / "ret8.npy")
    common = ["--envs", "4096", "--steps", "6", "--warmup", "2", "--repeats", "3", "--no-traffic", "--no-other-shapes"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--cpu-seconds", "2",
                        "--dump-returns", dump8] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_envs"] == 8 * 4096 and j["config"]["envs_per_gpu"] == 4096 and j["scaling"] == "weak"
    assert j["metric"].startswith("env-steps/sec (whole node), case141 batch=4096")
    per_rank = j["roofline"]["kernel_avg_ms_per_rank"]
    assert len(per_rank) == 8 and all(t > 0 for t in per_rank) and abs(j["roofline"]["kernel_avg_ms"] - max(per_rank)) < 1e-12
    assert abs(j["value"] - 8 * 4096 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0 and "rank 0 of 8" in j["cpu_baseline"]["note"]
    ret8 = np.load(dump8)
    assert ret8.shape == (8 * 4096,) and np.isfinite(ret8).all()
    for rank in (0, 5):
        dump1 = str(tmp_path / f"ret1_{rank}.npy")
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--env-id-offset", str(rank * 4096), "--no-cpu-baseline",
                             "--dump-returns", dump1] + common, capture_output=True, text=True, timeout=600, env=env)
        assert r1.returncode == 0, r1.stderr[-3000:]
        j1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
        assert j1["repeats"] == j["repeats"] == 3 and j1["config"]["global_envs"] == 4096
        assert np.array_equal(np.load(dump1), ret8[rank * 4096:(rank + 1) * 4096]), rank      # same ids -> same keyed noise, start rows, actions


@pytest.mark.gpu
def test_bench_single_gpu_line_has_live_traffic_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "256", "--steps", "20", "--warmup", "4",
                        "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= len(os.sched_getaffinity(0)) and c["value"] >= c["single_core_value"] > 0
    assert c["all_cores_batched_numpy"]["value"] > 0 and c["all_cores_one_env_per_process"]["processes"] == c["cores"]
    assert j["repeats"] >= 3 and j["timed_seconds_total"] >= 0.5 and j["steps"] == 20
    assert j["ms_per_step_repeats"]["min"] <= j["ms_per_step"] <= j["ms_per_step_repeats"]["max"]
    assert "resets_in_timed_region" in j["config"]
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and "latency" in ro["limited_by"] and abs(ro["frac"] - ro["achieved"] / 8000.0) < 1e-12
    assert ro["traffic_detail"] is not None and (ro["traffic_detail"].get("bytes") or ro["traffic_detail"].get("error"))


@pytest.mark.gpu
def test_bench_force_dist_rccl_first_contact():
    """the nccl (RCCL) code path of bench.py — init_process_group("nccl", device_id=...), all_gather_into_tensor on a
    DEVICE tensor, all_reduce(MAX) of the block time, the barriers of fence() — on ONE rank, so that an RCCL initialisation / IPC failure
    shows up here and not in the driver's first real SCALE run.  librccl must be mapped into the process."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--envs", "256", "--steps", "6", "--warmup", "2",
                        "--no-traffic", "--no-cpu-baseline", "--no-other-shapes"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    d = j["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_at_one_rank"] and d["rccl_loaded"] is True
    assert d["gathered_rows_last_block"] == 256 and j["n_gpus"] == 1 and j["value"] > 0
