#!/usr/bin/env python
"""Tolerance-edge power flows (VERDICT r4 "pin the tolerance edge"): inputs constructed on the CPU with the oracle so that a Newton
iterate's mismatch norm ||F||inf lands AT the stopping tolerance — where the GPU's summation order and the oracle's SuperLU order can
decide differently by rounding (profiles/r04_parity_soak.txt: 1 env-step in 112 640 took one Newton step fewer on the GPU).

Two families on the 141-bus feeder (tol = 1e-8 MVA / sn_mva = 1e-9 p.u.; the noise floor of evaluating F is ~2e-12):
  * "it3": the scale of all injections (loads, PV, q) is bisected until the THIRD iterate's norm f3 = tol * (1 + delta), delta in {-5e-2, -1e-4, +1e-4, +5e-2}:
     well below / inside the noise band / well above.  Outside the band the iteration count must agree exactly (3 resp. 4); inside,
     either count is a converged power flow and the voltages agree to 1e-9.
  * "it10": the feeder loaded towards voltage collapse (Newton slows down) until the TENTH iterate's norm f10 = tol * (1 + delta):
     the convergence FLAG — the -200 / destroy branch of voltage_control_env.py:188-196 — is decided here.  Outside the band the flag
     must agree; inside, both verdicts are defensible and the voltages of a converged side agree with the other's 10th iterate to 1e-6.

Writes tests/golden/nr_edge_case141.npz (inputs, the oracle's per-iterate norms, iterations, flags, voltages).  Run from the repo root:
    python tests/golden/make_edge_golden.py
TEST INFRASTRUCTURE (uses oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mapdn_amd.netspec import make_case          # noqa: E402
from oracle import pp_restated as ppr            # noqa: E402


iterate_norms = ppr.iterate_norms


def bisect(fn, lo, hi, target, n=200):
    """lambda with fn(lambda) ~= target for fn increasing on [lo, hi] (log scale; stops at the resolution of the noise)"""
    flo, fhi = fn(lo), fn(hi)
    assert flo < target < fhi, (flo, target, fhi)
    best = None
    for _ in range(n):
        mid = 0.5 * (lo + hi)
        fm = fn(mid)
        if best is None or abs(fm - target) < abs(best[1] - target):
            best = (mid, fm)
        if fm < target:
            lo = mid
        else:
            hi = mid
        if hi - lo < 1e-15 * max(1.0, abs(hi)):
            break
    return best


def main():
    net, prof = make_case("case141")
    tol = ppr.TOLERANCE_MVA / net.sn_mva
    rng = np.random.default_rng(20260926)
    row = int(rng.integers(0, prof.n_rows))
    pv = prof.pv[row].copy()
    qs = rng.uniform(-0.6, 0.6, net.n_sgen) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    pl0, ql0 = prof.load_p[row].copy(), prof.load_q[row].copy()
    cases = []

    def f_at(k):
        return lambda lam: iterate_norms(net, pl0 * lam, ql0 * lam, pv * lam, qs * lam)[k]
    # ---- family it3: f3 = tol (1 + delta)
    for delta in (-5e-2, -1e-4, 1e-4, 5e-2):
        lam, f3 = bisect(f_at(3), 0.05, 3.0, tol * (1 + delta))
        cases.append(("it3", delta, lam))
    # ---- family it10: towards collapse until the 10th iterate sits at the tolerance
    lam_hi = 3.0
    while np.isfinite(f_at(10)(lam_hi)) and f_at(10)(lam_hi) < 1.0 and lam_hi < 200.0:
        lam_hi *= 1.25
    # the last scale at which the flat start still converges within 10 iterations, by bisection on the flag
    lo, hi = 1.0, lam_hi
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if ppr.runpp_restated(net, pl0 * mid, ql0 * mid, pv * mid, qs * mid).converged:
            lo = mid
        else:
            hi = mid
        if hi - lo < 1e-14 * hi:
            break
    lam_edge = lo
    n_edge = iterate_norms(net, pl0 * lam_edge, ql0 * lam_edge, pv * lam_edge, qs * lam_edge)
    print("collapse-side edge: lambda %.15f, norms %s" % (lam_edge, ["%.3e" % x for x in n_edge]))
    # f10(lambda) around the edge: find brackets where it is monotone, then the deltas
    span = 1e-3
    for delta in (-5e-2, -1e-4, 1e-4, 5e-2):
        try:
            lam, f10 = bisect(f_at(10), lam_edge * (1 - span), lam_edge * (1 + span), tol * (1 + delta))
        except AssertionError:
            lam, f10 = bisect(f_at(10), lam_edge * (1 - 20 * span), lam_edge * (1 + 20 * span), tol * (1 + delta))
        cases.append(("it10", delta, lam))
    # ---- family collapse10: LOADS only scaled towards the nose of the PV curve (Newton slows down near the fold) until the flat
    # start no longer converges within 10 iterations; the iterate at the tolerance there is whichever k the edge falls on
    def conv_loads(lam):
        return ppr.runpp_restated(net, pl0 * lam, ql0 * lam, pv, qs).converged
    lo, hi = 1.0, 2.0
    while conv_loads(hi):
        lo, hi = hi, hi * 1.5
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if conv_loads(mid):
            lo = mid
        else:
            hi = mid
        if hi - lo < 1e-14 * hi:
            break
    nn = iterate_norms(net, pl0 * lo, ql0 * lo, pv, qs)
    print("nose-side edge (loads only): lambda %.15f, norms %s, min vm %.4f" % (lo, ["%.3e" % x for x in nn],
          ppr.runpp_restated(net, pl0 * lo, ql0 * lo, pv, qs).vm_pu.min()))
    extra = [("nose", 0.0, lo), ("nose", 1.0, hi), ("nose", -1e-6, lo * (1 - 1e-6)), ("nose", 1e-6, hi * (1 + 1e-6))]
    fam, dl, lams = [c[0] for c in cases], np.array([c[1] for c in cases]), np.array([c[2] for c in cases])
    PL = np.stack([pl0 * l for l in lams]); QL = np.stack([ql0 * l for l in lams])
    PV = np.stack([pv * l for l in lams]); QS = np.stack([qs * l for l in lams])
    fam += [c[0] for c in extra]; dl = np.concatenate([dl, [c[1] for c in extra]]); lams = np.concatenate([lams, [c[2] for c in extra]])
    PL = np.concatenate([PL, np.stack([pl0 * c[2] for c in extra])]); QL = np.concatenate([QL, np.stack([ql0 * c[2] for c in extra])])
    PV = np.concatenate([PV, np.tile(pv, (len(extra), 1))]); QS = np.concatenate([QS, np.tile(qs, (len(extra), 1))])
    cases = cases + extra
    norms = np.stack([iterate_norms(net, PL[i], QL[i], PV[i], QS[i]) for i in range(len(cases))])
    its, conv, vm, va = [], [], [], []
    for i in range(len(cases)):
        r = ppr.runpp_restated(net, PL[i], QL[i], PV[i], QS[i])
        its.append(r.iterations); conv.append(r.converged); vm.append(r.vm_pu); va.append(r.va_degree)
        k = 3 if fam[i] == "it3" else 10
        if fam[i] == "nose":
            k = int(np.argmin(np.abs(np.log(np.maximum(norms[i], 1e-300) / tol))))
        print(f"{fam[i]} delta {dl[i]:+.0e}: lambda {lams[i]:.15f}  f{k} = {norms[i, k]:.6e} (tol {tol:.1e}, f{k}/tol - 1 = {norms[i, k] / tol - 1:+.2e})  "
              f"oracle iterations {r.iterations} converged {r.converged}  min vm {r.vm_pu.min():.4f}")
    out = os.path.join(ROOT, "tests", "golden", "nr_edge_case141.npz")
    np.savez_compressed(out, family=np.array(fam), delta=dl, scale=lams, load_p=PL, load_q=QL, pv=PV, q=QS, norms=norms, tol=tol,
                        iterations=np.array(its), converged=np.array(conv), vm_pu=np.stack(vm), va_degree=np.stack(va))
    print("wrote", out)


if __name__ == "__main__":
    main()
