"""Feeder shapes the three MAPDN-sized test feeders do not have, through the tree solver's host plan (CPU) and the kernel (GPU):
  star      one hub with 12 laterals (a junction with 12 children: the overflow child list of the step records, gmax = 3 rows),
  chain     a single 60-bus line (rows == nodes / 2 after rooting at the centre, every hand-off a register carry or a lone slot),
  tiny2/3   2- and 3-bus nets (fewer nodes than workers: idle steps everywhere, schedules padded to the peeled rows),
  fan       eight feeders leaving the slack bus (eight elimination roots, every one with a constant-voltage neighbour),
  bushy     a random tree with 4-6 children per junction, 90 buses.
CPU: the Hu schedule is a valid elimination on 1..64 workers and the flat-start factorisation reproduces the oracle's first Newton step.
GPU: voltages / iterations against the oracle, several launch geometries, and a short noisy episode on the star."""
import ctypes as C

import numpy as np
import pytest

from mapdn_amd import _lib
from mapdn_amd.netspec import NetSpec, synth_profiles
from oracle.pp_restated import _fx, bus_demand, jacobian, make_sbus, make_ybus, runpp_restated

ARGS = dict(episode_limit=240, action_scale=0.8, action_bias=0.0, voltage_barrier_type="bowl", seed=0)


def feeder(name, parent, n_sgen=2, seed=0, vn_kv=12.47, sn_mva=10.0, p_total=4.0):
    """NetSpec + Profiles of the tree given by parent[] (bus 0 = slack, parent[0] = -1), a load on every non-slack bus"""
    parent = np.asarray(parent)
    nb = parent.shape[0]
    rng = np.random.default_rng(seed)
    child = np.arange(1, nb)
    swap = rng.random(nb - 1) < 0.4
    f, t = np.where(swap, child, parent[1:]), np.where(swap, parent[1:], child)
    w = rng.uniform(0.5, 1.5, nb - 1)
    p_nom = p_total * w / w.sum()
    sgen_bus = rng.choice(child, size=min(n_sgen, nb - 1), replace=False).astype(np.int32)
    zone = np.zeros(nb, np.int32)
    zone[1:] = 1 + (np.arange(nb - 1) % max(1, len(sgen_bus)))
    for j, b in enumerate(sgen_bus):
        zone[b] = j + 1
    depth = np.zeros(nb, int)
    for i in range(1, nb):
        depth[i] = depth[parent[i]] + 1
    net = NetSpec(name=name, bus_vn_kv=np.full(nb, vn_kv), bus_zone=zone, line_from_bus=f, line_to_bus=t,
                  line_r_ohm_per_km=rng.uniform(0.1, 0.4, nb - 1) * 4.0 / max(4, depth.max()), line_x_ohm_per_km=rng.uniform(0.05, 0.3, nb - 1) * 4.0 / max(4, depth.max()),
                  line_c_nf_per_km=rng.uniform(5.0, 15.0, nb - 1), line_g_us_per_km=np.zeros(nb - 1), line_length_km=rng.uniform(0.2, 0.8, nb - 1),
                  line_parallel=np.ones(nb - 1, np.int32), line_in_service=np.ones(nb - 1, np.uint8), load_bus=child.astype(np.int32),
                  sgen_bus=sgen_bus, sgen_zone=(np.arange(len(sgen_bus)) + 1).astype(np.int32), ext_grid_bus=0, ext_grid_vm_pu=1.01, sn_mva=sn_mva, f_hz=50.0)
    prof = synth_profiles(p_nom, p_nom * 0.33, np.full(len(sgen_bus), 0.8 * p_total / max(1, len(sgen_bus))), days=4, seed=seed)
    return net, prof


def shapes():
    rng = np.random.default_rng(42)
    star = [-1, 0] + [1] * 12 + [2 + (i % 12) for i in range(24)]            # hub = bus 1, 12 laterals of 3 buses
    chain = [-1] + list(range(60))
    fan = [-1] + [0] * 8 + [1 + (i % 8) for i in range(24)]
    bushy = [-1, 0]
    frontier = [1]
    while len(bushy) < 90:
        p = frontier.pop(0)
        for _ in range(int(rng.integers(4, 7))):
            if len(bushy) < 90:
                frontier.append(len(bushy)); bushy.append(p)
    return {"star": feeder("star", star, 3, 1), "chain": feeder("chain", chain, 2, 2), "tiny2": feeder("tiny2", [-1, 0], 1, 3, p_total=0.5),
            "tiny3": feeder("tiny3", [-1, 0, 1], 2, 4, p_total=0.8), "fan": feeder("fan", fan, 4, 5), "bushy": feeder("bushy", bushy, 5, 6)}


SHAPES = shapes()


def _inputs(net, prof, B, seed):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, prof.n_rows, B)
    pv = prof.pv[rows]
    qs = rng.uniform(-0.8, 0.8, (B, net.n_sgen)) * np.sqrt(prof.s_max() ** 2 - pv ** 2)
    return prof.load_p[rows], prof.load_q[rows], pv, qs


@pytest.mark.parametrize("name", list(SHAPES))
def test_host_plan_of_unusual_shapes(name):
    _check_host_plan(name, *SHAPES[name])


@pytest.mark.parametrize("seed", range(24))
def test_host_plan_of_random_trees(seed):
    """24 random feeders, 2 ... 120 buses, from pure chains (attach to the newest bus) to near-stars (attach to a random old bus):
    the same host-plan checks as the named shapes"""
    _check_host_plan(f"rand{seed}", *_random_tree(seed))


def _check_host_plan(name, net, prof):
    lib = _lib.load()
    cn, keep = _lib.make_cnetspec(net)
    cc = _lib.make_cconfig(ARGS)
    h = C.c_void_p()
    assert lib.mapdn_create(C.byref(cn), C.byref(cc), 64, -1, C.byref(h)) == 0, lib.mapdn_last_error(None)
    g = _lib.nr_geometry(h)
    assert g["solver"] == 0 and g["lds_bytes"] <= 160 * 1024 and g["rows"] >= 1
    nb, n = net.n_bus, net.n_bus - 1
    par = np.zeros(n, np.int32)
    for W in (1, 4, 16):
        R = C.c_int32()
        assert lib.mapdn_get_schedule(h, W, C.byref(R), None, None) == 0
        rows = np.zeros(W * R.value, np.int32)
        assert lib.mapdn_get_schedule(h, W, C.byref(C.c_int32()), _lib._p(rows, _lib._pi), _lib._p(par, _lib._pi)) == 0
        rows = rows.reshape(W, R.value)
        assert sorted(rows[rows >= 0].tolist()) == list(range(n))
        row_of = {int(k): r for w in range(W) for r, k in enumerate(rows[w]) if k >= 0}
        assert all(par[k] == n or row_of[int(par[k])] > row_of[k] for k in range(n))
    if name == "star":
        assert np.bincount(par, minlength=n + 1)[:n].max() == 12          # the hub: 12 children in one step
    if name == "fan":
        assert (par == n).sum() == 8                                       # eight elimination roots
    # flat-start factorisation == the oracle's first Newton step
    from scipy.sparse.linalg import spsolve
    fac = np.zeros((n, 12)); bop = np.zeros(n + 1, np.int32)
    assert lib.mapdn_get_flat_factors(h, _lib._p(fac, _lib._pd), _lib._p(bop, _lib._pi)) == 0
    pl, ql, pv, qs = _inputs(net, prof, 1, 9)
    sbus = make_sbus(net, *bus_demand(net, pl[0], ql[0], pv[0], qs[0]))
    ybus = make_ybus(net)[0]
    pq = np.setdiff1d(np.arange(nb), [net.ext_grid_bus])
    v0 = np.full(nb, net.ext_grid_vm_pu, dtype=np.complex128)
    dx = -spsolve(jacobian(ybus, v0, pq, pq).tocsc(), _fx(ybus, v0, sbus, pq, pq)) if n > 1 else \
        -np.linalg.solve(jacobian(ybus, v0, pq, pq).toarray(), _fx(ybus, v0, sbus, pq, pq))
    dth = np.zeros(nb); dvm = np.zeros(nb); dth[pq] = dx[:n]; dvm[pq] = dx[n:]
    S = fac[:, 0] + 1j * fac[:, 1]
    Iinv = fac[:, 2:6].reshape(n, 2, 2); apk = fac[:, 6:8]; G = fac[:, 8:12].reshape(n, 2, 2)
    hv = np.zeros((n, 2)); acc = np.zeros((n + 1, 2)); x = np.zeros((n + 1, 2))
    for k in range(n):
        F = S[k] - sbus[bop[k]]
        hv[k] = Iinv[k] @ (np.array([F.real, F.imag]) - acc[k])
        ar, ai = apk[k]
        acc[par[k]] += (ai * hv[k, 0] + ar * hv[k, 1], ai * hv[k, 1] - ar * hv[k, 0])
    for k in range(n - 1, -1, -1):
        x[k] = hv[k] - G[k] @ (x[par[k]] if par[k] < n else np.zeros(2))
    assert np.abs(-x[:n, 0] - dth[bop[:n]]).max() < 1e-10 and np.abs(-x[:n, 1] * abs(v0[0]) - dvm[bop[:n]]).max() < 1e-10
    lib.mapdn_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SHAPES))
def test_tree_solver_on_unusual_shapes(name):
    import torch
    from mapdn_amd.env import VoltageControlBatch
    net, prof = SHAPES[name]
    B = 40
    ins = _inputs(net, prof, B, 3)
    ins[0][7] *= 60.0                                          # one env beyond the feeder's loadability
    ref = None
    for tuning in (None, dict(nr_waves=1, nr_lanes=16), dict(nr_waves=4, nr_lanes=8), dict(nr_waves=2, nr_lanes=4), dict(nr_waves=2, nr_lanes=16, nr_lean=1)):
        try:
            env = VoltageControlBatch(net, prof, ARGS, n_envs=B, device="cuda:0", obs_dtype=torch.float64, tuning=tuning)
        except Exception as exc:                               # (a geometry this shape cannot take is refused, not wrong)
            assert "LDS" in str(exc) or "compiled" in str(exc), exc
            continue
        vm, va, it, cv = [t.cpu().numpy() for t in env.solve(*ins)]
        env.close()
        if ref is None:
            ref = (vm, va, it, cv)
            for e in range(B):
                r = runpp_restated(net, ins[0][e], ins[1][e], ins[2][e], ins[3][e])
                assert bool(cv[e]) == r.converged and it[e] == r.iterations, (name, e, it[e], r.iterations)
                if r.converged:
                    assert np.abs(vm[e] - r.vm_pu).max() < 1e-9 and np.abs(va[e] - r.va_degree).max() < 1e-7
            assert cv[[0, 1, 39]].all()              # (env 7, at 60x its load, diverges on the long shapes and survives on the short ones: as the oracle says)
        else:                                                  # every geometry: the same bits
            ok = ref[3].astype(bool)
            assert np.array_equal(it, ref[2]) and np.array_equal(cv, ref[3]) and np.array_equal(vm[ok], ref[0][ok]) and np.array_equal(va[ok], ref[1][ok]), tuning


def _random_tree(seed):
    rng = np.random.default_rng(1000 + seed)
    nb = int(rng.integers(2, 121))
    chaininess = rng.random()
    parent = [-1] + [int(i - 1 if rng.random() < chaininess else rng.integers(0, i)) for i in range(1, nb)]
    return feeder(f"rand{seed}", parent, n_sgen=int(rng.integers(1, 5)), seed=seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 24, 2))
def test_tree_solver_on_random_trees(seed):
    """twelve of the random feeders of test_host_plan_of_random_trees through the solver (default geometry) against the oracle"""
    import torch
    from mapdn_amd.env import VoltageControlBatch
    from tests.edge_rule import same_newton_count
    net, prof = _random_tree(seed)
    B = 24
    ins = _inputs(net, prof, B, 11 + seed)
    env = VoltageControlBatch(net, prof, ARGS, n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    vm, va, it, cv = [t.cpu().numpy() for t in env.solve(*ins)]
    env.close()
    for e in range(B):
        r = runpp_restated(net, ins[0][e], ins[1][e], ins[2][e], ins[3][e])
        assert same_newton_count(net, tuple(x[e] for x in ins), it[e], cv[e], r), (seed, e, it[e], r.iterations)
        if r.converged and cv[e]:
            assert np.abs(vm[e] - r.vm_pu).max() < 1e-9 and np.abs(va[e] - r.va_degree).max() < 1e-7


@pytest.mark.gpu
def test_noisy_episode_on_the_star_feeder():
    """the whole step() on the star: reward / info / obs against the oracle env over a few noisy steps"""
    import torch
    from mapdn_amd.env import VoltageControlBatch
    from oracle.env_restated import INFO_KEYS, VoltageControlOracle
    net, prof = SHAPES["star"]
    B = 6
    env = VoltageControlBatch(net, prof, ARGS, n_envs=B, device="cuda:0", obs_dtype=torch.float64)
    oracles = [VoltageControlOracle(net, prof, ARGS, env_id=e, do_reset=False) for e in range(B)]
    obs, _ = env.reset()
    for e, o in enumerate(oracles):
        oo, _ = o.reset()
        assert np.abs(np.array(oo) - obs[e].cpu().numpy()).max() < 1e-9
    rng = np.random.default_rng(1)
    for t in range(5):
        act = rng.uniform(-0.8, 0.8, (B, net.n_sgen))
        r, term, info = env.step(torch.as_tensor(act, device="cuda:0"))
        ob = env.get_obs().cpu().numpy()
        for e, o in enumerate(oracles):
            ro, to, io = o.step(act[e])
            assert abs(ro - r[e].item()) < 1e-9 and to == bool(term[e].item())
            assert max(abs(io[k] - info[e, c].item()) for c, k in enumerate(INFO_KEYS)) < 1e-9
            assert np.abs(np.array(o.get_obs()) - ob[e]).max() < 1e-9
    env.close()
