"""The committed bench record (profiles/r01_final_bench_case141.json, produced by bench.py on an MI355X) carries
every key of the driver's contract with consistent values; bench.py's algorithmic-byte formula matches
SURVEY.md 8(d)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_record_follows_the_contract():
    j = json.load(open(os.path.join(ROOT, "profiles", "r01_final_bench_case141.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "env-steps/s" and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["envs_per_gpu"] == 4096
    # value = global envs * steps / time
    assert abs(j["value"] - j["n_gpus"] * j["config"]["envs_per_gpu"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["algorithmic_bytes_per_launch"] == r["algorithmic_bytes_per_env_step"] * r["envs_per_launch"]
    assert r["traffic"] is None or r["traffic"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and c["sample"]


def test_algorithmic_bytes_formula():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class E:        # case141 sizes
        n_load, n_sgen, n_bus, n_agents, obs_size = 84, 22, 141, 22, 58
    assert bench.algorithmic_bytes_per_env_step(E) == 8 * (2 * 84 + 2 * 22) + 16 * 141 + 4 * 22 * 58 + 97 == 9153
