#!/bin/bash
# round 6: data-parallel learner on the GPU box (RCCL at world size 1, 8-rank gloo pre-flight) + the full GPU suite + a default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_learner_dist_gpu.py -x -q -m gpu > gpurun_out/r06_q3_dist.txt 2>&1; echo "dist tests rc=$?"; tail -12 gpurun_out/r06_q3_dist.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_q3_gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -6 gpurun_out/r06_q3_gpu_suite.txt
timeout 900 python bench.py > gpurun_out/r06_q3_bench.json 2> gpurun_out/r06_q3_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06_q3_bench.json") if l.startswith("{")][-1])
r = j["roofline"]
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", r["frac"], "frac_kernel_bytes", r["frac_kernel_bytes"], "frac_step", r["frac_step"], "kernel_ms", r["kernel_avg_ms"], "launches", r["kernel_launches_timed"], "traffic", r["traffic"])
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds_per_episode", "phase_seconds", "error")})
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
for s in j.get("other_shapes", []):
    print(s["workload"][:40], s["value"], s["roofline"]["frac"], s["roofline"]["kernel_avg_ms"], s["roofline"]["traffic"])
PY
