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
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_gpus2_without_a_launcher():
    """N = 2 on a one-GPU box: both ranks on cuda:0, gloo for the collective (the nccl path differs only in the backend
    string); the single JSON line must describe the whole job."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--envs", "64",
                        "--steps", "4", "--warmup", "2", "--no-traffic"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_envs"] == 128 and j["config"]["envs_per_gpu"] == 64
    assert j["steps"] == 4 and j["warmup"] == 2 and j["value"] > 0 and "cpu_baseline" not in j
    assert abs(j["value"] - 128 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6


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
