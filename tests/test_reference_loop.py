"""The reference's own loops, replayed against the drop-in.

tests/golden/make_loop_golden.py ran the reference's Model.train_process / Model.evaluation (models/model.py:197-302) and
PGTester.run (utilities/tester.py:19-63) on the reference VoltageControl through a recording proxy.  Here every recorded env
call is issued, in order and with the recorded arguments (float32 action vectors as the loops produce them), against
mapdn_amd.env.VoltageControl: the drop-in must serve the same sequence with the same return types / dtypes / shapes, and — for
the tester episode, which is deterministic — the same values."""
import json
import os

import numpy as np
import pytest

from mapdn_amd.netspec import make_case
from tests.golden.env_scenarios import DIGITS, quantized_profiles
from tests.golden.loop_protocol import summarize

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture(case):
    meta = json.load(open(os.path.join(HERE, "golden", f"loop_ref_{case}.json")))
    vals = np.load(os.path.join(HERE, "golden", f"loop_ref_{case}.npz"))
    return meta, vals


def _flatten(prefix, value, store):
    scalar = lambda v: isinstance(v, (bool, int, float, np.bool_, np.integer, np.floating))   # noqa: E731
    if value is None:
        return
    if isinstance(value, dict):
        store[prefix + "/dict"] = np.array([float(value[k]) for k in sorted(value)], dtype=np.float64)
    elif isinstance(value, (tuple, list)) and not all(scalar(v) for v in value):
        for i, v in enumerate(value):
            _flatten(f"{prefix}/{i}", v, store)
    else:
        store[prefix] = np.asarray(value, dtype=np.float64)


@pytest.mark.parametrize("case", ["case33", "case141"])
def test_fixture_covers_the_loops_env_surface(case):
    """(CPU) the recorded sequences are what the reference's loops issue: reset / get_avail_actions / step / get_obs per training
    step, manual_reset + the six tester getters per tester step — and every method the loops use exists on the drop-in class"""
    from mapdn_amd.env import VoltageControl
    meta, vals = _fixture(case)
    names = [c["m"] for c in meta["calls"]]
    ph = meta["phases"]
    train, evalu, tester = names[:ph["train_process"]], names[ph["train_process"]:ph["evaluation"]], names[ph["evaluation"]:]
    assert train[0] == "reset" and train.count("step") == meta["max_steps"] and train.count("get_obs") == meta["max_steps"]
    assert evalu[0] == "reset" and evalu.count("step") == meta["max_steps"]
    assert tester[0] == "manual_reset" and tester.count("step") == meta["max_steps"]
    assert tester.count("_get_res_bus_v") == meta["max_steps"] + 1
    for m in set(names):
        assert callable(getattr(VoltageControl, m)), m
    step = meta["calls"][names.index("step")]
    assert step["args"][0][:2] == ["ndarray", "float"] and step["ret"][0] == "tuple" and step["ret"][1] == 3
    assert meta["info_keys_sorted"] == sorted(("percentage_of_v_out_of_control", "percentage_of_lower_than_lower_v",
                                               "percentage_of_higher_than_upper_v", "totally_controllable_ratio", "average_voltage_deviation",
                                               "average_voltage", "max_voltage_drop_deviation", "max_voltage_rise_deviation",
                                               "total_line_loss", "q_loss", "destroy"))


@pytest.mark.parametrize("case", ["case33", "case141"])
def test_oracle_env_reproduces_the_recorded_tester_episode(case):
    """(CPU) the deterministic part of the recording — PGTester.run: manual_reset, step(add_noise=False), get_obs, the getters —
    replayed on oracle/env_restated.py: the checker the GPU tests lean on agrees with the reference's own loop run to 1e-12"""
    from oracle.env_restated import VoltageControlOracle
    meta, vals = _fixture(case)
    net, prof = make_case(case)
    env = VoltageControlOracle(net, quantized_profiles(prof), dict(meta["env_args"]), env_id=0, do_reset=False)
    worst, n_steps = 0.0, 0
    for i, c in enumerate(meta["calls"]):
        if i < meta["phases"]["evaluation"]:
            continue
        if c["m"] == "manual_reset":
            obs, state = env.manual_reset(*[int(vals[f"{i}/a{j}"]) for j in range(3)])
            worst = max(worst, float(np.abs(state - vals[f"{i}/r/1"]).max()), max(float(np.abs(np.array(obs[k]) - vals[f"{i}/r/0/{k}"]).max()) for k in range(len(obs))))
        elif c["m"] == "step":
            r, t, info = env.step(np.asarray(vals[f"{i}/a0"], dtype=np.float32), add_noise=False)
            assert t == bool(vals[f"{i}/r/1"])
            worst = max(worst, abs(r - float(vals[f"{i}/r/0"])), float(np.abs(np.array([info[k] for k in sorted(info)]) - vals[f"{i}/r/2/dict"]).max()))
            n_steps += 1
        elif c["m"] == "get_obs":
            o = env.get_obs()
            worst = max(worst, max(float(np.abs(np.array(o[k]) - vals[f"{i}/r/{k}"]).max()) for k in range(len(o))))
        elif c["m"] == "_get_res_bus_v":
            worst = max(worst, float(np.abs(env.res.vm_pu - vals[f"{i}/r"]).max()))
    assert n_steps == meta["max_steps"] and worst < 1e-12, worst


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["case33", "case141"])
def test_drop_in_serves_the_reference_loops_call_sequence(case, tmp_path):
    from mapdn_amd.data import save_netspec, save_profiles_csv
    from mapdn_amd.env import VoltageControl
    meta, vals = _fixture(case)
    net, prof = make_case(case)
    d = str(tmp_path)
    save_netspec(net, os.path.join(d, "netspec.npz"))
    save_profiles_csv(quantized_profiles(prof), d, float_format=f"%.{DIGITS}g")
    args = dict(meta["env_args"], data_path=d)
    env = VoltageControl(args)                                   # the drop-in, constructed from the reference's own env_args
    sz = meta["sizes"]
    assert (env.n_agents, env.n_actions, env.obs_size, env.state_size, env.episode_limit) == \
        (sz["n_agents"], sz["n_actions"], sz["obs_size"], sz["state_size"], sz["episode_limit"])
    first_tester_call = meta["phases"]["evaluation"]
    worst = 0.0
    for i, c in enumerate(meta["calls"]):
        a = []
        for j, summ in enumerate(c["args"]):
            v = vals[f"{i}/a{j}"]
            if summ == "int":
                v = int(v)
            elif isinstance(summ, list) and summ[0] == "ndarray":
                v = np.asarray(v, dtype=np.float32)              # translate_action hands float32 (utilities/util.py:123-132)
            a.append(v)
        kw = {k: bool(vals[f"{i}/k_{k}"]) if s == "bool" else vals[f"{i}/k_{k}"] for k, s in c["kwargs"].items()}
        ret = getattr(env, c["m"])(*a, **kw)
        assert summarize(ret) == c["ret"], (i, c["m"], summarize(ret), c["ret"])
        if i >= first_tester_call:                               # deterministic episode: values too
            got = {}
            _flatten(f"{i}/r", ret, got)
            for k, g in got.items():
                ref = vals[k]
                assert g.shape == ref.shape, (i, c["m"], k)
                err = float(np.abs(g - ref).max()) if g.size else 0.0
                worst = max(worst, err)
                assert err < 1e-9, (i, c["m"], k, err)
    assert worst < 1e-9
    env.close()
