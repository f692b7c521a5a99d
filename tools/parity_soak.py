#!/usr/bin/env python
"""Parity soak: a full-size batch on the GPU (default launch geometry, auto_reset) against the CPU oracle on MANY envs and steps —
more than the test suite replays (64 envs x 9 calls per configuration) — in parallel oracle processes.

  python tools/parity_soak.py --case case141 --envs 4096 --watch 1024 --steps 48

Every watched env is replayed by its own VoltageControlOracle (same global env id => same Philox draws): reward, terminated, the
11 info values, the obs row and res_bus.vm_pu of every call are compared; the per-env auto-reset that follows a terminated step
(episode limit = 12, and one env per 97 gets an absurd action at call 5: unsolvable step) is replayed as the reference loop's
reset().  Prints one summary line per case; exit status 1 when a flag differs, voltages / obs differ by more than 1e-9 or reward / info by
more than 1e-8.  When the largest voltage difference is above rounding (> 1e-12) the power flow of that env-step is probed: the oracle's
mismatch norm per iterate and the GPU's iteration count on the same inputs."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}
LIMIT = 12


TIES = {"case141": ([5, 40, 77, 100, 20], [77, 120, 130, 12, 66]), "case322": ([5, 40, 177, 200, 300], [77, 120, 30, 12, 150])}   # as tools/general_bench.py


def _args(case, barrier="bowl"):
    return dict(episode_limit=LIMIT, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type=barrier, seed=3)


def _net(case, ties):
    """the case's feeder, or (ties > 0) the same net with tie lines closed: meshed -> the general sparse solver"""
    from mapdn_amd.netspec import add_lines, case33_meshed, make_case
    net, prof = make_case(case)
    if ties:
        if case == "case33":
            net = case33_meshed(net, ties)
        else:
            f, t = TIES[case]
            net = add_lines(net, f[:ties], t[:ties], 0.3, 0.2)
    return net, prof


def _replay(job):
    case, envs, acts, rew, term, info, obs, vm, mask, barrier, ties = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle.env_restated import INFO_KEYS, VoltageControlOracle
    net, prof = _net(case, ties)
    worst = dict(reward=0.0, info=0.0, obs=0.0, vm=0.0)
    where = dict(vm=None, info=None)                                      # (env id, call, oracle iterations, which info key) of the largest differences
    bad_flags, n_steps, n_resets, n_fail, n_edge = 0, 0, 0, 0, 0
    for i, e in enumerate(envs):
        o = VoltageControlOracle(net, prof, _args(case, barrier), env_id=int(e), do_reset=False)
        oo, _ = o.reset()
        worst["obs"] = max(worst["obs"], float(np.abs(np.array(oo) - obs[0, i]).max()))
        pending = False
        for t in range(acts.shape[0]):
            if pending:                                                   # this call was the env's reset()
                bad_flags += int(not mask[t, i]) + int(rew[t, i] != 0.0) + int(bool(term[t, i]))
                oo, _ = o.reset()
                worst["obs"] = max(worst["obs"], float(np.abs(np.array(oo) - obs[t + 1, i]).max()))
                pending = False; n_resets += 1
                continue
            bad_flags += int(bool(mask[t, i]))
            pre = (o.load_p.copy(), o.load_q.copy(), o.sgen_p.copy(), np.asarray(o._clip_reactive_power(np.asarray(acts[t, i], dtype=np.float64), o.sgen_p)).copy())
            ro, to, io = o.step(acts[t, i])
            n_steps += 1
            bad_flags += int(to != bool(term[t, i]))
            worst["reward"] = max(worst["reward"], abs(ro - rew[t, i]))
            di = [abs(io[k] - info[t, i, c]) for c, k in enumerate(INFO_KEYS)]
            if max(di) > worst["info"]:
                worst["info"] = max(di); where["info"] = (int(e), t, int(o.res.iterations), INFO_KEYS[int(np.argmax(di))], float(io[INFO_KEYS[int(np.argmax(di))]]))
            dv = float(np.abs(vm[t, i] - o.res.vm_pu).max())
            n_edge += int(dv > 1e-12)
            if dv > worst["vm"]:
                worst["vm"] = dv; where["vm"] = (int(e), t, int(o.res.iterations), float(o.res.vm_pu.min())); where["inputs"] = pre
            worst["obs"] = max(worst["obs"], float(np.abs(np.array(o.get_obs()) - obs[t + 1, i]).max()))
            n_fail += int(io["destroy"] == 1.0) if "destroy" in io else 0
            pending = bool(to)
    return worst, bad_flags, n_steps, n_resets, n_fail, where, n_edge


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="case141"); ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--watch", type=int, default=1024); ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--barrier", default="bowl", help="voltage_barrier_type: l1 | l2 | courant_beltrami | bowl | bump")
    ap.add_argument("--ties", type=int, default=0, help="close this many tie lines (meshed net: the general sparse solver)")
    a = ap.parse_args()
    import torch
    from mapdn_amd.env import VoltageControlBatch
    net, prof = _net(a.case, a.ties)
    B, W, T = a.envs, min(a.watch, a.envs), a.steps
    env = VoltageControlBatch(net, prof, dict(_args(a.case, a.barrier), auto_reset=True), n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    watch = np.unique(np.linspace(0, B - 1, W).astype(np.int64))
    W = len(watch)
    wt = torch.as_tensor(watch, device="cuda:0")
    obs0, _ = env.reset()
    assert env.stats()["reset_failures"] == 0
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(17)
    acts = np.empty((T, W, net.n_sgen)); rew = np.empty((T, W)); term = np.empty((T, W), bool); info = np.empty((T, W, 11))
    obs = np.empty((T + 1, W) + tuple(obs0.shape[1:])); vm = np.empty((T, W, net.n_bus)); mask = np.empty((T, W), bool)
    obs[0] = obs0[wt].cpu().numpy()
    for t in range(T):
        act = (torch.rand(B, net.n_sgen, device="cuda:0", generator=gen, dtype=torch.float64) * 2 - 1) * SCALE[a.case]
        if t == 5:
            act[wt[::97]] = 60.0                                           # forced unsolvable steps
        r, tm, inf = env.step(act)
        o = env.get_obs()
        acts[t] = act[wt].cpu().numpy(); rew[t] = r[wt].cpu().numpy(); term[t] = tm[wt].cpu().numpy().astype(bool)
        info[t] = inf[wt].cpu().numpy(); obs[t + 1] = o[wt].cpu().numpy()
        vm[t] = env.results(("vm_pu",))["vm_pu"][wt].cpu().numpy(); mask[t] = env.auto_reset_mask()[wt].cpu().numpy().astype(bool)
    geo = env.geometry(); st = env.stats()
    env.close()
    procs = a.procs or min(len(os.sched_getaffinity(0)), 16)
    chunks = np.array_split(np.arange(W), procs * 4)
    jobs = [(a.case, watch[c], acts[:, c], rew[:, c], term[:, c], info[:, c], obs[:, c], vm[:, c], mask[:, c], a.barrier, a.ties) for c in chunks if len(c)]
    import multiprocessing as mp
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_replay, jobs)
    dt = time.perf_counter() - t0
    worst = {k: max(r[0][k] for r in res) for k in res[0][0]}
    flags = sum(r[1] for r in res); n_steps = sum(r[2] for r in res); n_resets = sum(r[3] for r in res); n_fail = sum(r[4] for r in res)
    wr = max(res, key=lambda r: r[0]["vm"])
    wv = wr[5]["vm"]; wi = max(res, key=lambda r: r[0]["info"])[5]["info"]
    probe = ""
    if worst["vm"] > 1e-12 and wr[5].get("inputs") is not None:          # more than rounding: which side took how many Newton steps?
        from oracle import pp_restated as ppr
        pl, ql, pv, qs = wr[5]["inputs"]
        ref = ppr.runpp_restated(net, pl, ql, pv, qs, raise_on_fail=False)
        ybus = ppr._cached_ybus(net)[0]
        nb = net.n_bus; pq = np.setdiff1d(np.arange(nb), [net.ext_grid_bus])
        v = np.full(nb, net.ext_grid_vm_pu, dtype=np.complex128); va = np.angle(v); vmag = np.abs(v)
        hist = []
        for it in range(6):                                               # the oracle's iteration, with the mismatch norm of every iterate
            f = ppr._fx(ybus, v, ref["Sbus"], pq, pq)
            hist.append(float(np.linalg.norm(f, np.inf)))
            dx = -ppr.spla.spsolve(ppr.jacobian(ybus, v, pq, pq), f)
            va[pq] += dx[:len(pq)]; vmag[pq] += dx[len(pq):]
            v = vmag * np.exp(1j * va); vmag = np.abs(v); va = np.angle(v)
        e2 = VoltageControlBatch(net, prof, _args(a.case, a.barrier), n_envs=64, device="cuda:0")
        gvm, gva, git, gcv = e2.solve(np.tile(pl, (64, 1)), np.tile(ql, (64, 1)), np.tile(pv, (64, 1)), np.tile(qs, (64, 1)))
        e2.close()
        probe = (f"; probe of that power flow: oracle iterations {ref['iterations']}, ||F||inf of its iterates {['%.6e' % h for h in hist]} (tol {1e-8 / net.sn_mva:.1e}; the noise floor of evaluating F is the level of the last entries), "
                 f"GPU solve of the same inputs: iterations {int(git[0])}, max |vm - oracle| {float(np.abs(gvm[0].cpu().numpy() - ref['vm_pu']).max()):.2e}")
    n_edge = sum(r[6] for r in res)
    # bars: voltages / obs 1e-9 (north_star: 1e-6), reward / info 1e-8 — an env-step whose mismatch norm lands within the evaluation
    # noise of the tolerance (||F|| ~ 1e-9 +- 1e-12) may take one Newton step more or fewer than the oracle: both are converged
    ok = flags == 0 and max(worst['vm'], worst['obs']) < 1e-9 and max(worst['reward'], worst['info']) < 1e-8
    print(f"{a.case}{' + %d tie lines' % a.ties if a.ties else ''} x {B} envs, barrier {a.barrier} (solver {geo['solver']}, waves {geo['waves']}, envs/workgroup {geo['lanes']}, lean {geo['lean']}): {W} envs x {T} calls replayed on the oracle "
          f"({n_steps} env-steps, {n_resets} auto-resets, {n_fail} unsolvable steps; {dt:.0f} s on {procs} processes): max |d reward| {worst['reward']:.2e}, "
          f"|d info| {worst['info']:.2e}, |d obs| {worst['obs']:.2e}, |d vm_pu| {worst['vm']:.2e}, flag mismatches {flags}, env-steps with |d vm_pu| > 1e-12 (a different number of Newton steps at the tolerance edge): {n_edge}; "
          f"GPU mean / max NR iterations {st['mean_nr_iters']:.2f} / {st['max_nr_iters']}; largest vm difference at (env, call, oracle iterations, min vm) = {wv}, "
          f"largest info difference at (env, call, oracle iterations, key, value) = {wi}  " + probe + f"  -> {'OK' if ok else 'MISMATCH'}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
