#!/usr/bin/env python
"""BASELINE.json configs[0] ("case33_3min_final, IDDPG, 1 env, pandapower CPU runpp — reference plumbing, no GPU") as far as it can be
timed in the build container: the REFERENCE's own `VoltageControl` class (/root/reference/environments/var_voltage_control/
voltage_control_env.py, unmodified: its deepcopy of the net :181, pandas writes :553, sort_index / chained .loc obs assembly :232-316)
stepping the synthetic 33-bus feeder, with `pp.runpp` provided by oracle/pp_stub (= oracle/pp_restated.py + pandas result tables),
because pandapower 2.7.0 itself is not installable offline.  One env, one core, 240-step episodes, step() + get_obs() per step as
models/model.py:216-219 does.  Printed next to the restated oracle env (oracle/env_restated.py, no pandas in the loop) on the same
inputs: the difference is what the reference's pandas / deepcopy plumbing costs.  A reported baseline, never a target.

    python tools/reference_class_cpu.py [--episodes 3] > profiles/r06_reference_class_cpu_case33.txt      (needs /root/reference)
"""
import argparse
import os
import sys
import tempfile
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "pp_stub"), REF]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=3)
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "environments")):
        raise SystemExit("the reference checkout (/root/reference) only exists in the build container")
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})            # one core
    warnings.simplefilter("ignore")
    from mapdn_amd.data import save_netspec, save_profiles_csv
    from mapdn_amd.netspec import make_case
    from tests.golden.env_scenarios import BASE_ARGS, DIGITS, quantized_profiles
    from environments.var_voltage_control.voltage_control_env import VoltageControl
    from oracle.env_restated import VoltageControlOracle
    net, prof = make_case("case33")
    d = tempfile.mkdtemp(prefix="mapdn_refcpu_")
    save_netspec(net, os.path.join(d, "netspec.npz"))
    pq = quantized_profiles(prof)
    save_profiles_csv(pq, d, float_format=f"%.{DIGITS}g")
    args = dict(BASE_ARGS, voltage_barrier_type="bowl", data_path=d, seed=0)
    env = VoltageControl(args)
    oracle = VoltageControlOracle(net, pq, dict(BASE_ARGS, voltage_barrier_type="bowl", seed=0))
    rng = np.random.default_rng(0)
    rows = []
    for name, e in (("reference class (voltage_control_env.py) + restated runpp (oracle/pp_stub)", env), ("restated oracle env (oracle/env_restated.py)", oracle)):
        per_ep = []
        for ep in range(a.episodes):
            e.reset()
            n, t0 = 0, time.perf_counter()
            for _ in range(240):
                r, term, info = e.step(rng.uniform(-0.8, 0.8, net.n_sgen))
                e.get_obs()
                n += 1
                if term:
                    break
            per_ep.append(n / (time.perf_counter() - t0))
        rows.append((name, per_ep))
    print("# BASELINE configs[0] plumbing on this container's CPU: case33 (33 buses, 6 agents), ONE env on ONE core, 240-step episodes,")
    print("# step() + get_obs() per step (models/model.py:216-219); pandapower 2.7.0 is not installable offline: `runpp` is oracle/pp_restated.py")
    print(f"# python {sys.version.split()[0]}, numpy {np.__version__}, pandas {__import__('pandas').__version__}; cpu: "
          + next((ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')), '?'))
    for name, per_ep in rows:
        print(f"{name:88s} {np.median(per_ep):8.1f} env-steps/s   (per episode: {', '.join(f'{v:.1f}' for v in per_ep)})")
    print(f"# ratio oracle / reference-class plumbing: {np.median(rows[1][1]) / np.median(rows[0][1]):.1f} x  — the reference's per-step deepcopy(net), "
          "pandas row writes and per-agent sort_index / chained .loc observation assembly")


if __name__ == "__main__":
    main()
