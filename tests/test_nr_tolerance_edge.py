"""Newton iterates AT the stopping tolerance.  tests/golden/nr_edge_case141.npz holds power flows on the
141-bus feeder constructed with the oracle (tests/golden/make_edge_golden.py) so that the 3rd resp. 10th iterate's ||F||inf is tol x
(1 + delta): delta = +-5e-2 is far outside the rounding noise of evaluating F (~2e-12 on a tolerance of 1e-9), delta = +-1e-4 inside
it.  Rule (oracle.pp_restated.iterations_agree): outside the band iteration counts and convergence flags are EXACT;
inside it the two implementations may differ by one Newton step — or, at iteration 10, in the flag that decides the -200 / destroy
branch of voltage_control_env.py:188-196 — and every converged answer agrees to 1e-9 p.u."""
import os

import numpy as np
import pytest

from mapdn_amd.netspec import make_case
from oracle.pp_restated import MAX_ITER, edge_band, iterate_norms, iterations_agree, runpp_restated

FIX = os.path.join(os.path.dirname(__file__), "golden", "nr_edge_case141.npz")


def test_oracle_reproduces_the_edge_fixture():
    z = np.load(FIX)
    net, _ = make_case("case141")
    tol = float(z["tol"])
    assert tol == 1e-8 / net.sn_mva
    for i in range(len(z["family"])):
        ins = (z["load_p"][i], z["load_q"][i], z["pv"][i], z["q"][i])
        r = runpp_restated(net, *ins)
        assert r.iterations == int(z["iterations"][i]) and r.converged == bool(z["converged"][i])
        assert np.array_equal(r.vm_pu, z["vm_pu"][i])
        nn = iterate_norms(net, *ins)
        assert np.array_equal(nn, z["norms"][i])
        fam, d = str(z["family"][i]), float(z["delta"][i])
        if fam in ("it3", "it10"):
            k = 3 if fam == "it3" else 10
            assert abs(nn[k] / tol - 1 - d) < 0.5 * abs(d)          # the construction hit its target
            assert (abs(nn[k] - tol) <= edge_band(tol)) == (abs(d) < 1e-3)
            # the oracle's own decision follows its own norm
            if fam == "it3":
                assert r.converged and r.iterations == (3 if nn[3] < tol else 4)
            else:
                assert r.iterations == 10 and r.converged == (nn[10] < tol)


def test_agreement_rule():
    tol = 1e-9
    nn = np.array([1.0, 1e-2, 1e-4, 1.0004e-9, 2e-12, 2e-12, 0, 0, 0, 0, 0.99995e-9])
    assert iterations_agree(3, True, 3, True, nn, tol) and iterations_agree(3, True, 4, True, nn, tol) and iterations_agree(4, True, 3, True, nn, tol)
    assert not iterations_agree(2, True, 3, True, nn, tol)          # the 2nd iterate is nowhere near the tolerance
    assert not iterations_agree(3, True, 5, True, nn, tol)
    assert iterations_agree(10, True, 10, False, nn, tol)            # the flag alone, 10th iterate inside the band
    nn[10] = 0.9e-9
    assert not iterations_agree(10, True, 10, False, nn, tol)
    assert not iterations_agree(3, True, 4, False, nn, tol)
    assert MAX_ITER == 10 and abs(edge_band(1e-9) - 3e-12) < 1e-20


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", [None, dict(nr_waves=2, nr_lanes=16, nr_lean=1), dict(nr_waves=4, nr_lanes=4)])
def test_gpu_at_the_tolerance_edge(geometry):
    import torch
    from mapdn_amd.env import VoltageControlBatch
    z = np.load(FIX)
    net, prof = make_case("case141")
    tol = float(z["tol"])
    n = len(z["family"])
    B = 64                                                            # every fixture row four times, spread over workgroups
    idx = np.arange(B) % n
    env = VoltageControlBatch(net, prof, dict(episode_limit=240, action_scale=0.6, action_bias=0.0), n_envs=B, device="cuda:0",
                              obs_dtype=torch.float64, tuning=geometry)
    vm, va, it, cv = [t.cpu().numpy() for t in env.solve(z["load_p"][idx], z["load_q"][idx], z["pv"][idx], z["q"][idx])]
    env.close()
    report = []
    for e in range(B):
        i = idx[e]
        fam, d = str(z["family"][i]), float(z["delta"][i])
        o_it, o_cv = int(z["iterations"][i]), bool(z["converged"][i])
        report.append((fam, d, int(it[e]), bool(cv[e]), o_it, o_cv))
        assert iterations_agree(int(it[e]), bool(cv[e]), o_it, o_cv, z["norms"][i], tol), report[-1]
        if fam in ("it3", "it10") and abs(d) > 1e-3:                 # outside the band: exact
            assert int(it[e]) == o_it and bool(cv[e]) == o_cv, report[-1]
        if fam == "nose" and abs(d) > 0 and abs(d) < 1:              # 1e-6 either side of the nose-side edge: f10 is 16-19 % off the tolerance
            assert int(it[e]) == o_it and bool(cv[e]) == o_cv, report[-1]
        if cv[e] and o_cv:
            assert np.abs(vm[e] - z["vm_pu"][i]).max() < 1e-9, report[-1]
        if fam == "it3":
            assert cv[e]
        if e >= n:                                                    # the same inputs in another lane / workgroup: the same bits
            assert it[e] == it[i] and cv[e] == cv[i] and np.array_equal(vm[e], vm[i])
    print("\n(family, delta, gpu iterations, gpu converged, oracle iterations, oracle converged):", sorted(set(report)))
