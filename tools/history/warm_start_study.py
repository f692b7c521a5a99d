#!/usr/bin/env python
"""Study for the OPT-IN warm start (mapdn_env_config.nr_init = 1; pandapower's runpp(init="results"), which the reference does
not use): would starting a step's power flow from the env's last accepted voltages change anything but the iteration count?

CPU only, on the oracle's batched Newton iteration (oracle/batched_np.py::newton, the restated pypower newtonpf).  Envs run the
reference's step dynamics on the synthetic profiles (next table row + half-normal noise, uniform random actions); a share
of the solves is STRESSED towards and beyond voltage collapse (loads scaled up, actions beyond the action range), because
the question that matters is the P7 branch (voltage_control_env.py:188-196): a warm start must never report "converged"
where the exact flat-start solve reports LoadflowNotConverged, and must never land on another root.

Rule studied (what the kernel implements): warm solve with at most WARM_ITERS Newton iterations; not converged -> the exact
flat-start solve decides (so those envs are exact by construction).  Reported per case:
  * iteration histograms flat / warm, share of solves that fall back
  * warm converged & flat NOT converged                      <- must be 0 to ship without a guard
  * warm converged & flat converged & max|V_warm - V_flat| > 1e-6    <- must be 0 (another root)
  * max |V_warm - V_flat| over the rest
  * the same counts under the GUARD min|V| >= v_guard on the warm answer (else fall back to flat)
Usage: python tools/warm_start_study.py [--solves 1e7] [--procs 8]  ->  JSON on stdout (commit under profiles/)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8}
WARM_ITERS = 3
GUARDS = (0.5, 0.6, 0.7, 0.8)


def worker(args):
    case, seed, B, steps, stress_frac, smooth = args
    from mapdn_amd.netspec import make_case
    from oracle.batched_np import BatchedRunpp
    net, prof = make_case(case)
    S = BatchedRunpp(net)
    rng = np.random.default_rng(seed)
    ns, nl, nb = net.n_sgen, net.n_load, net.n_bus
    smax = prof.s_max()
    std_pv, std_pl, std_ql = prof.pv.std(0) / 100, prof.load_p.std(0) / 100, prof.load_q.std(0) / 100
    T = prof.n_rows
    row = rng.integers(0, T - steps - 2, B)
    Vacc = np.full((B, nb), net.ext_grid_vm_pu, dtype=np.complex128)
    have = np.zeros(B, bool)
    a_prev = rng.uniform(-SCALE[case], SCALE[case], (B, ns))
    # collapse scale of this feeder (loads x k until the flat start stops converging), by bisection on a typical row
    def conv_at(k):
        r = T // 2
        pl, ql, pv = prof.load_p[r:r + 1] * k, prof.load_q[r:r + 1] * k, prof.pv[r:r + 1] * 0
        sb = -(np.array(S._demand(pl, ql, pv, pv * 0)[0]) + 1j * np.array(S._demand(pl, ql, pv, pv * 0)[1])) / net.sn_mva
        return bool(S.newton(sb)[1][0])
    lo, hi = 1.0, 2.0
    while conv_at(hi):
        lo, hi = hi, hi * 2
    for _ in range(20):
        mid = 0.5 * (lo + hi)
        if conv_at(mid):
            lo = mid
        else:
            hi = mid
    kcol = lo
    out = dict(solves=0, flat_it=np.zeros(12, np.int64), warm_it=np.zeros(12, np.int64), fallback=0, flat_nc=0,
               warm_ok_flat_nc=0, other_root=0, max_dv=0.0, stressed=0, stressed_flat_nc=0, kcol=kcol,
               guard={g: dict(warm_ok_flat_nc=0, other_root=0, accepted=0) for g in GUARDS}, vmin_warm_ok_flat_nc=[])
    for t in range(steps):
        r = row + t
        pv = prof.pv[r] + std_pv * np.abs(rng.standard_normal((B, ns)))
        pl = prof.load_p[r] + std_pl * np.abs(rng.standard_normal((B, nl)))
        ql = prof.load_q[r] + std_ql * np.abs(rng.standard_normal((B, nl)))
        if smooth > 0:                                         # a slowly varying policy output instead of i.i.d. uniform actions
            a = np.clip(a_prev + smooth * SCALE[case] * rng.standard_normal((B, ns)), -SCALE[case], SCALE[case])
            a_prev = a
        else:
            a = rng.uniform(-SCALE[case], SCALE[case], (B, ns))
        stressed = rng.random(B) < stress_frac
        kind = rng.integers(0, 3, B)
        # stress 0: loads scaled into the neighbourhood of collapse; 1: actions far beyond the range (what unsolvable steps
        # look like in the tests); 2: both, milder
        kl = np.where(stressed & (kind != 1), rng.uniform(0.6 * kcol, 1.25 * kcol, B), 1.0)
        ka = np.where(stressed & (kind != 0), rng.uniform(1.0, 80.0, B), 1.0)
        pl, ql = pl * kl[:, None], ql * kl[:, None]
        qs = np.sqrt(np.maximum(smax ** 2 - pv ** 2, 0.0)) * a * ka[:, None]
        pd_, qd = S._demand(pl, ql, pv, qs)
        sb = -(pd_ + 1j * qd) / net.sn_mva
        Vf, cf, itf = S.newton(sb)
        Vw, cw, itw = S.newton(sb, V0=Vacc, max_iter=WARM_ITERS)
        use = have                                               # envs with an accepted state to start from
        n = int(use.sum())
        out["solves"] += n
        out["stressed"] += int((use & stressed).sum()); out["stressed_flat_nc"] += int((use & stressed & ~cf).sum())
        out["flat_it"] += np.bincount(itf[use], minlength=12)[:12]
        wi = np.where(cw, itw, itw + np.where(cf, itf, 10))      # warm iterations + the fallback's
        out["warm_it"] += np.bincount(np.minimum(wi[use], 11), minlength=12)[:12]
        out["fallback"] += int((use & ~cw).sum())
        out["flat_nc"] += int((use & ~cf).sum())
        bad = use & cw & ~cf
        out["warm_ok_flat_nc"] += int(bad.sum())
        vminw = np.abs(Vw).min(1)
        if bad.any():
            out["vmin_warm_ok_flat_nc"] += [float(x) for x in vminw[bad][:50]]
        both = use & cw & cf
        dv = np.abs(np.abs(Vw) - np.abs(Vf)).max(1)
        oth = both & (dv > 1e-6)
        out["other_root"] += int(oth.sum())
        if (both & ~oth).any():
            out["max_dv"] = max(out["max_dv"], float(dv[both & ~oth].max()))
        for g in GUARDS:
            acc = use & cw & (vminw >= g)
            out["guard"][g]["accepted"] += int(acc.sum())
            out["guard"][g]["warm_ok_flat_nc"] += int((acc & ~cf).sum())
            out["guard"][g]["other_root"] += int((acc & cf & (dv > 1e-6)).sum())
        # the env's next state: accepted = the exact rule's answer (warm if converged else flat); an unsolvable step ends the
        # episode -> new start row, flat solve of its first step next time
        ok = np.where(cw & have, True, cf)
        Vnew = np.where((cw & have)[:, None], Vw, Vf)
        Vacc = np.where(ok[:, None], Vnew, Vacc)
        have = ok
        row = np.where(ok, row, rng.integers(0, T - steps - 2, B) - t)
    out["flat_it"] = out["flat_it"].tolist(); out["warm_it"] = out["warm_it"].tolist()
    return case, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solves", type=float, default=1e7)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--stress", type=float, default=0.15)
    ap.add_argument("--smooth", type=float, default=0.0, help="0: i.i.d. uniform actions (bench.py's workload); s > 0: a random walk with steps of s x the action scale")
    a = ap.parse_args()
    share = {"case33": 0.5, "case141": 0.4, "case322": 0.1}
    B, steps = 2048, 60
    jobs = []
    for case, f in share.items():
        n = int(np.ceil(a.solves * f / (B * (steps - 1))))
        jobs += [(case, 1000 * i + len(case), B, steps, a.stress, a.smooth) for i in range(n)]
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs) as pool:
        res = pool.map(worker, jobs, chunksize=1)
    agg = {}
    for case, o in res:
        g = agg.setdefault(case, None)
        if g is None:
            agg[case] = o
            continue
        for k in ("solves", "fallback", "flat_nc", "warm_ok_flat_nc", "other_root", "stressed", "stressed_flat_nc"):
            g[k] += o[k]
        g["max_dv"] = max(g["max_dv"], o["max_dv"])
        g["flat_it"] = (np.array(g["flat_it"]) + np.array(o["flat_it"])).tolist()
        g["warm_it"] = (np.array(g["warm_it"]) + np.array(o["warm_it"])).tolist()
        g["vmin_warm_ok_flat_nc"] = (g["vmin_warm_ok_flat_nc"] + o["vmin_warm_ok_flat_nc"])[:200]
        for gd in GUARDS:
            for k in g["guard"][gd]:
                g["guard"][gd][k] += o["guard"][gd][k]
    for case, g in agg.items():
        n = max(g["solves"], 1)
        fi, wi = np.array(g["flat_it"]), np.array(g["warm_it"])
        g["mean_flat_iterations"] = float((fi * np.arange(12)).sum() / n)
        g["mean_warm_iterations_incl_fallback"] = float((wi * np.arange(12)).sum() / n)
        g["guard"] = {str(k): v for k, v in g["guard"].items()}
    print(json.dumps({"warm_iters": WARM_ITERS, "stress_fraction": a.stress, "action_random_walk_step": a.smooth, "seconds": time.time() - t0, "cases": agg}, indent=1))


if __name__ == "__main__":
    main()
