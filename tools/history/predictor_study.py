#!/usr/bin/env python
"""Offline study (CPU oracle) of the NR kernel's "next sweep will find convergence" predictor.

For N envs per case it records, per Newton iteration, the mismatch norm F_k and the step size dx_k, then
evaluates candidate rules at workgroup granularity (16 envs must all agree):
  hit  : the workgroup's last forward sweep is predicted -> it runs mismatch-only (saves ~0.3-0.65 sweep)
  miss : a sweep is predicted although some env is not converged -> one wasted mismatch-only sweep
Rules: dx < eps (the kernel's, eps = 1e-7) and the quadratic extrapolation F_next ~ F_k^3 / F_{k-1}^2 < tol / safety."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.sparse.linalg import splu
from mapdn_amd.netspec import make_case
from oracle.pp_restated import make_ybus, bus_demand, make_sbus, _fx, jacobian

TOL = 1e-8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for case, scale in (("case33", 0.8), ("case141", 0.6), ("case322", 0.8)):
    net, prof = make_case(case)
    ybus = make_ybus(net)[0]
    nb = net.n_bus
    pq = np.setdiff1d(np.arange(nb), [net.ext_grid_bus]); n = len(pq)
    rng = np.random.default_rng(0)
    seqs = []
    for _ in range(N):
        row = int(rng.integers(0, prof.n_rows)); pv = prof.pv[row]
        q = rng.uniform(-scale, scale, net.n_sgen) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
        sbus = make_sbus(net, *bus_demand(net, prof.load_p[row], prof.load_q[row], pv, q))
        v = np.full(nb, net.ext_grid_vm_pu, dtype=complex)
        F, DX = [], []
        for it in range(10):
            f = _fx(ybus, v, sbus, pq, pq); F.append(np.abs(f).max())
            if F[-1] < TOL:
                break
            dx = -splu(jacobian(ybus, v, pq, pq).tocsc()).solve(f)
            va, vm = np.angle(v), np.abs(v)
            va[pq] += dx[:n]; DX.append(max(np.abs(dx[:n]).max(), np.abs(dx[n:] / vm[pq]).max())); vm[pq] += dx[n:]
            v = vm * np.exp(1j * va)
        seqs.append((F, DX))
    iters = np.array([len(s[1]) for s in seqs])
    print(f"{case}: {N} envs, iterations mean {iters.mean():.2f} max {iters.max()}, last-sweep F median {np.median([s[0][-1] for s in seqs]):.1e}")

    def evaluate(rule):
        hits = misses = groups = 0
        for g in range(0, N - 15, 16):
            grp = seqs[g:g + 16]; groups += 1
            kmax = max(len(s[1]) for s in grp)
            for k in range(1, kmax + 1):                     # sweep k+1 follows update k (1-based); sweep 1 is the flat one
                alive = [s for s in grp if len(s[1]) >= k]   # envs that took update k
                if not alive or not all(rule(s, k) for s in alive):
                    continue
                if all(len(s[1]) == k for s in alive):       # every one of them converges at the next sweep
                    hits += 1 if k == kmax else 0
                else:
                    misses += 1
        return hits / groups, misses / groups
    rules = {"dx < 1e-7": lambda s, k: s[1][k - 1] < 1e-7, "dx < 1e-6": lambda s, k: s[1][k - 1] < 1e-6,
             "dx < 1e-5": lambda s, k: s[1][k - 1] < 1e-5}
    for safety in (1.0, 3.0, 10.0):
        rules[f"F^3/Fprev^2 < tol/{safety:g}"] = (lambda sf: lambda s, k: k >= 2 and s[0][k - 1] ** 3 / s[0][k - 2] ** 2 < TOL / sf)(safety)
    for name, rule in rules.items():
        h, m = evaluate(rule)
        print(f"   {name:28s} workgroups whose last sweep is predicted {100 * h:5.1f} %   mispredicted sweeps per workgroup {m:.3f}")
