"""MADDPG / IDDPG learner against fixtures produced by the reference's own code
(tests/golden/make_learner_golden.py): forward passes, both DDPG losses, and every entry of the
state_dict after two value steps, one policy step and one soft target update.  float32 throughout;
tolerances cover the different (factored) summation order of the first layers."""
import os

import numpy as np
import pytest
import torch

from mapdn_amd.learner import DDPGNet, PGTrainer, make_alg_args

HERE = os.path.dirname(__file__)
VARIANTS = {
    "maddpg_shared": dict(alg="maddpg"),
    "iddpg_shared": dict(alg="iddpg"),
    "maddpg_separate": dict(alg="maddpg", shared_params=False, agent_id=False, hid_activation="tanh"),
    "iddpg_separate_noln": dict(alg="iddpg", shared_params=False, layernorm=False, double_q=False,
                                reward_normalisation=False, normalize_advantages=True),
}
RTOL, ATOL = 2e-5, 2e-6


def _load(name, device="cpu"):
    z = np.load(os.path.join(HERE, "golden", f"learner_{name}.npz"))
    over = dict(VARIANTS[name]); alg = over.pop("alg")
    n, o = z["batch/state"].shape[1:]
    h = z["batch/hid"].shape[-1]
    args = make_alg_args(n, o, 1, hid_size=h, **over)
    trainer = PGTrainer(args, alg, env=None, device=device, data_parallel=False)
    init = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init/")}
    trainer.behaviour_net.load_state_dict(init, strict=True)          # names/shapes == reference model.pt
    batch = {k[6:]: torch.from_numpy(z[k]).float().to(device) for k in z.files if k.startswith("batch/")}
    return z, args, trainer, batch


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    assert np.allclose(a, b, rtol=rtol, atol=atol), (what, np.abs(a - b).max())


@pytest.mark.parametrize("name", list(VARIANTS))
def test_forward_and_losses_match_reference(name):
    _check_forward_and_losses(name, "cpu")


@pytest.mark.parametrize("name", list(VARIANTS))
def test_update_steps_match_reference(name):
    _check_update_steps(name, "cpu")


# the same fixtures of the reference's own learner code, on the device the learner actually runs on (fp32 on the MI355X:
# rocBLAS / MIOpen reduction orders differ from the CPU's, hence the slightly wider tolerances)
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_forward_and_losses_match_reference_on_gpu(name):
    _check_forward_and_losses(name, "cuda:0", rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_update_steps_match_reference_on_gpu(name):
    _check_update_steps(name, "cuda:0", rtol=1e-4, atol=1e-5, move_rtol=1e-2, move_atol=1e-5)


def _check_forward_and_losses(name, device, rtol=RTOL, atol=ATOL):
    import functools
    _close = functools.partial(globals()["_close"], rtol=rtol, atol=atol)
    z, args, tr, b = _load(name, device)
    net = tr.behaviour_net
    means, log_stds, hid = net.policy(b["state"], b["last_hid"])
    _close(means, z["out/means"], "means"); _close(log_stds, z["out/log_stds"], "log_stds"); _close(hid, z["out/hiddens"], "hid")
    _close(net.value(b["state"], b["action"]), z["out/value"], "value")
    a, _, lp, _, _ = net.get_actions(b["state"], "test", False, b["action_avail"], False, b["last_hid"])
    assert lp is None
    _close(a, z["out/test_action"], "test action")
    a, _, _, _, _ = net.get_actions(b["state"], "train", False, b["action_avail"], True, b["last_hid"])
    _close(a, z["out/target_mean_action"], "target policy")
    act = b["action"].clone().requires_grad_(True)
    net.value(b["state"], act).sum().backward()
    _close(act.grad, z["out/dvalue_daction"], "d value / d action (own-action gradient rule)")
    net.zero_grad()
    pl, vl, (m2, ls2) = net.get_loss(b)
    _close(pl, z["out/policy_loss"], "policy loss"); _close(vl, z["out/value_loss"], "value loss")
    only_v = net.get_loss(b, want=("value",))
    assert only_v[0] is None and only_v[2] is None


def _check_update_steps(name, device, rtol=RTOL, atol=ATOL, move_rtol=2e-3, move_atol=2e-6):
    import functools
    _close = functools.partial(globals()["_close"], rtol=rtol, atol=atol)
    z, args, tr, b = _load(name, device)
    net = tr.behaviour_net
    net.get_loss(b)                                    # the generator's loss probe also moved the BatchNorm statistics
    stat = {}
    tr.value_transition_process(stat, b)
    tr.value_transition_process(stat, b)
    tr.policy_transition_process(stat, b)
    net.update_target()
    for k in ("value_grad_norm", "value_loss", "entropy", "policy_grad_norm", "policy_loss"):
        _close(stat["mean_train_" + k], z["stat/mean_train_" + k], k)
    final = net.state_dict()
    ref_keys = sorted(k[6:] for k in z.files if k.startswith("final/"))
    assert sorted(final) == ref_keys
    for k in ref_keys:
        if k.endswith("num_batches_tracked"):
            assert int(final[k]) == int(z["final/" + k]), k
        else:
            # parameters moved by lr 1e-4 RMSprop steps: compare the MOVE, not just the value
            init = z["init/" + k]
            assert np.allclose(final[k].cpu().numpy() - init, z["final/" + k] - init, rtol=move_rtol, atol=move_atol), k


def test_valid_mask_and_argument_checks():
    z, args, tr, b = _load("maddpg_shared")
    net = tr.behaviour_net
    torch.manual_seed(0)
    p0, v0, _ = net.get_loss(b)
    b2 = dict(b); b2["valid"] = torch.ones(b["state"].shape[0], dtype=torch.bool)
    p1, v1, _ = net.get_loss(b2)
    assert torch.allclose(p0, p1) and torch.allclose(v0, v1)
    with pytest.raises(KeyError):
        make_alg_args(3, 7, nonsense=1)
    with pytest.raises(NotImplementedError):
        make_alg_args(3, 7, gaussian_policy=True)
    with pytest.raises(KeyError):
        DDPGNet(args, "coma")
    a, a_pol, lp, _, hid = net.get_actions(b["state"], "train", True, b["action_avail"], False, b["last_hid"])
    assert a.abs().max() <= 1.0 and lp.shape == a.shape and hid.shape == b["hid"].shape


class _ToyEnv:
    """stand-in with the VoltageControlBatch surface (the real one needs a GPU): reward = -|a - target|"""

    def __init__(self, B, n, o, device="cpu", episode_limit=12):
        self.n_envs, self.n_agents, self.obs_size, self.device, self.episode_limit = B, n, o, torch.device(device), episode_limit
        self.g = torch.Generator().manual_seed(0)

    def reset(self):
        self.t = 0
        self.o = torch.randn(self.n_envs, self.n_agents, self.obs_size, generator=self.g)
        return self.o, None

    def get_avail_actions(self):
        return torch.ones(self.n_envs, self.n_agents, 1)

    def get_obs(self):
        return self.o

    def step(self, a):
        self.t += 1
        r = -(a - 0.3 * self.o[..., 0]).abs().mean(1).double()
        self.o = torch.randn(self.n_envs, self.n_agents, self.obs_size, generator=self.g)
        done = torch.full((self.n_envs,), self.t >= self.episode_limit, dtype=torch.bool)
        return r, done, torch.zeros(self.n_envs, 11, dtype=torch.float64)


def test_trainer_schedule_and_checkpoint(tmp_path):
    torch.manual_seed(0); np.random.seed(0)
    env = _ToyEnv(4, 3, 5)
    args = make_alg_args(3, 5, 1, hid_size=16, max_steps=12, batch_size=8, replay_buffer_size=64,
                         behaviour_update_freq=4, target_update_freq=6, value_update_epochs=2, num_eval_episodes=4)
    tr = PGTrainer(args, "iddpg", env, device="cpu", data_parallel=False)
    before = {k: v.clone() for k, v in tr.behaviour_net.state_dict().items()}
    stat = {}
    tr.run(stat, 0)
    assert tr.steps == 12 and tr.episodes == 1 and len(tr.replay_buffer) == 48
    assert {"mean_train_reward", "mean_test_reward", "mean_train_value_loss", "mean_train_policy_loss"} <= set(stat)
    assert all(isinstance(v, float) for v in stat.values())
    after = tr.behaviour_net.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before if k.startswith("policy_dicts"))
    assert any(not torch.equal(before[k], after[k]) for k in before if k.startswith("target_net.value_dicts"))
    p = tmp_path / "model.pt"
    tr.save(p)
    tr2 = PGTrainer(args, "iddpg", env, device="cpu", data_parallel=False)
    tr2.load(p)
    for k, v in tr.behaviour_net.state_dict().items():
        assert torch.equal(v, tr2.behaviour_net.state_dict()[k])


# ---- data-parallel update: world_size-2 gloo processes vs. one process on the union batch -----------
def _dp_batch(n, o, h, bs, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    return dict(state=r(bs, n, o), action=torch.tanh(r(bs, n, 1)), reward=r(bs, 1).expand(bs, n).contiguous(),
                next_state=r(bs, n, o), done=(torch.rand(bs, 1, generator=g) < 0.2).float(), last_step=torch.zeros(bs, 1),
                action_avail=torch.ones(bs, n, 1), last_hid=0.3 * r(bs, n, h), hid=0.3 * r(bs, n, h))


def _dp_worker(rank, world, port, q, reward_norm):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different initial weights: rank 0's must win (broadcast)
    args = make_alg_args(3, 5, 1, hid_size=8, reward_normalisation=reward_norm)
    tr = PGTrainer(args, "maddpg", env=None, device="cpu")
    assert tr._dist is not None
    stat = {}
    b = _dp_batch(3, 5, 8, 6, seed=7 + rank)
    tr.value_transition_process(stat, b)
    tr.policy_transition_process(stat, b)
    assert tr.replicas_identical()
    assert tr.collectives["all_reduce_grads"] == 2 and tr.collectives["broadcast"] > 0
    assert tr.collectives["all_reduce_reward_stats"] == (2 if reward_norm else 0)
    q.put((rank, {k: v.numpy().copy() for k, v in tr.behaviour_net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reward_norm", [False, True])
def test_data_parallel_update_equals_union_batch(reward_norm):
    """two gloo ranks, one update round each on its own batch == one rank on the concatenated batch (round 6: with the reference's reward
    BatchNorm ON as well — the ranks normalise with the statistics of the union — and the whole state_dict, running statistics included,
    bit-identical across the ranks)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, reward_norm)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in got[0]:
        assert np.array_equal(got[0][k], got[1][k]), k                   # replicas stay identical
    torch.manual_seed(100)
    args = make_alg_args(3, 5, 1, hid_size=8, reward_normalisation=reward_norm)
    tr = PGTrainer(args, "maddpg", env=None, device="cpu", data_parallel=False)
    b0, b1 = _dp_batch(3, 5, 8, 6, 7), _dp_batch(3, 5, 8, 6, 8)
    union = {k: torch.cat([b0[k], b1[k]]) for k in b0}
    stat = {}
    tr.value_transition_process(stat, union)
    tr.policy_transition_process(stat, union)
    for k, v in tr.behaviour_net.state_dict().items():
        assert np.allclose(v.numpy(), got[0][k], rtol=1e-4, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("n_agents,obs_dim,agent_id,rows_b", [(6, 26, True, 37), (22, 58, True, 129), (38, 82, True, 64), (22, 58, False, 5)])
def test_fused_policy_forward_matches_pytorch(n_agents, obs_dim, agent_id, rows_b, monkeypatch):
    """mapdn_policy_forward (one HIP launch: fc1 + id column -> LayerNorm -> ReLU -> GRUCell -> fc2, weights in LDS) against
    the PyTorch modules it replaces in the rollout (agents/rnn_agent.py:5-32 via models/model.py:101-139)"""
    import torch
    from mapdn_amd.learner import DDPGNet, make_alg_args
    torch.manual_seed(0)
    args = make_alg_args(n_agents, obs_dim, 1, action_scale=0.8, action_bias=0.0, agent_id=agent_id)
    net = DDPGNet(args, "maddpg").to("cuda:0")
    with torch.no_grad():
        for p in net.policy_dicts.parameters():
            p.add_(0.3 * torch.randn_like(p))                    # non-trivial LayerNorm gains / biases
    obs = torch.randn(rows_b, n_agents, obs_dim, device="cuda:0")
    hid = torch.randn(rows_b, n_agents, 64, device="cuda:0")
    with torch.no_grad():
        assert net._fused_policy_ok(obs, hid)
        m1, ls1, h1 = net.policy(obs, hid)
        monkeypatch.setenv("MAPDN_FUSED_POLICY", "0")
        assert not net._fused_policy_ok(obs, hid)
        m0, ls0, h0 = net.policy(obs, hid)
    assert m1.shape == m0.shape and h1.shape == h0.shape and torch.equal(ls1, ls0)
    assert (h1 - h0).abs().max().item() < 2e-5 and (m1 - m0).abs().max().item() < 2e-5
    # with autograd on (training-time forward) the PyTorch modules run
    monkeypatch.setenv("MAPDN_FUSED_POLICY", "1")
    assert not net._fused_policy_ok(obs, hid)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,relu", [(70001, True), (4096, False), (1 << 20, True)])
def test_layernorm64_kernels_match_pytorch(rows, relu):
    """mapdn_layernorm64_forward / backward (LayerNorm over 64 features + fused ReLU, the learner's training-time passes)
    against torch.nn.functional.layer_norm + relu with autograd: outputs and all three gradients."""
    from mapdn_amd.learner import _LayerNorm64
    g = torch.Generator(device="cuda:0"); g.manual_seed(rows)
    x = (torch.randn(rows, 64, device="cuda:0", generator=g) * 1.7 + 0.3).requires_grad_(True)
    w = (torch.randn(64, device="cuda:0", generator=g) * 0.5 + 1.0).requires_grad_(True)
    b = (torch.randn(64, device="cuda:0", generator=g) * 0.2).requires_grad_(True)
    dy = torch.randn(rows, 64, device="cuda:0", generator=g)
    y = _LayerNorm64.apply(x, w, b, 1e-5, relu)
    y.backward(dy)
    got = [y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    x.grad = w.grad = b.grad = None
    ref = torch.nn.functional.layer_norm(x, (64,), w, b, 1e-5)
    if relu:
        ref = torch.relu(ref)
    ref.backward(dy)
    want = [ref.detach(), x.grad, w.grad, b.grad]
    assert torch.allclose(got[0], want[0], rtol=2e-5, atol=2e-5)
    # dx: an element whose pre-activation is within rounding of 0 may fall on the other side of the ReLU in the two implementations
    # (there its whole dy switches on or off); everywhere else the gradients agree
    pre = torch.nn.functional.layer_norm(x.detach(), (64,), w.detach(), b.detach(), 1e-5)
    rows_ok = ~(relu & (pre.abs() < 1e-5).any(dim=1))
    assert rows_ok.float().mean().item() > 0.99
    assert torch.allclose(got[1][rows_ok], want[1][rows_ok], rtol=1e-4, atol=1e-4)
    scale = max(1.0, rows ** 0.5)                                   # the parameter gradients are sums over the rows
    assert (got[2] - want[2]).abs().max().item() < 2e-5 * scale * 10 and (got[3] - want[3]).abs().max().item() < 2e-5 * scale * 10
    # deterministic: a second backward gives the same bits (fixed-order block reduction, no atomics)
    x.grad = w.grad = b.grad = None
    y2 = _LayerNorm64.apply(x, w, b, 1e-5, relu)
    y2.backward(dy)
    assert torch.equal(w.grad, got[2]) and torch.equal(b.grad, got[3]) and torch.equal(x.grad, got[1])


@pytest.mark.gpu
def test_e2e_entry_point_runs_the_whole_loop():
    """mapdn_amd.e2e.run — the function behind examples/train_ddpg.py and bench.py's `e2e` block (BASELINE configs[4]): rollout of B envs
    through the HIP env, GPU replay (window views), MADDPG updates at the reference's intensity; one dict per episode with the phase
    split.  Small shape: 64 envs of the 33-bus feeder, two 61-step episodes (one update round each)."""
    from mapdn_amd import e2e
    seen = []
    lines = e2e.run(case="case33", envs=64, alg="maddpg", episodes=2, max_steps=61, intensity="reference", phases=True, on_line=seen.append)
    assert len(lines) == 2 and seen == lines
    for ln in lines:
        assert ln["env_steps_per_s"] > 0 and abs(ln["sampled_transitions_per_env_step"] - 11 * 32 / 60) < 1e-12
        assert ln["batch_size"] == 32 * 64 and ln["value_epochs"] == 10 and ln["policy_epochs"] == 1
        ph = ln["phase_seconds"]
        assert set(ph) == {"replay_insert", "sample", "value_update", "policy_update", "target_update", "rollout_and_host"}
        assert ph["value_update"] > 0 and ph["policy_update"] > 0 and abs(sum(ln["phase_share"].values()) - 1.0) < 1e-3
        assert np.isfinite(ln["mean_train_value_loss"]) and np.isfinite(ln["mean_train_policy_loss"])


def test_tall_linear_weight_gradient_is_the_plain_one(monkeypatch):
    """round 5: the weight gradient of the trunk's small layers on a batch of millions of rows is computed block-wise (split-K by hand:
    the BLAS back end runs a [64, 64] product with 10 M rows of reduction on two workgroups).  Same forward, same gradients."""
    from mapdn_amd.learner import _TallLinear
    monkeypatch.setattr(_TallLinear, "BLOCK_ROWS", 64)
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 5).double()
    x = torch.randn(1000, 64, dtype=torch.float64, requires_grad=True)       # 15 full blocks + a remainder of 40 rows
    y = _TallLinear.apply(x, lin.weight, lin.bias)
    (y ** 2).sum().backward()
    got = (lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone())
    lin.zero_grad(); x.grad = None
    y2 = lin(x)
    (y2 ** 2).sum().backward()
    assert torch.equal(y, y2) and torch.allclose(got[0], lin.weight.grad, rtol=0, atol=1e-11)
    assert torch.equal(got[1], lin.bias.grad) and torch.equal(got[2], x.grad)
    assert torch.autograd.gradcheck(lambda a, w, b: _TallLinear.apply(a, w, b),
                                    (torch.randn(130, 8, dtype=torch.float64, requires_grad=True), torch.randn(3, 8, dtype=torch.float64, requires_grad=True),
                                     torch.randn(3, dtype=torch.float64, requires_grad=True)))


@pytest.mark.parametrize("alg", ["maddpg", "iddpg"])
def test_cached_targets_train_the_same_network(alg, monkeypatch):
    """round 5 / 6: the value epochs of one update round read pi(next_state) — and, round 6, the target critic's value of it — from a
    per-round cache over the whole replay ring (neither the policy nor the target net changes while the critic trains) instead of
    running both once per epoch.  Bit-identical training: two seeded runs, cache on / off, end in the same state_dict — through
    several rounds, a wrapping ring, the mirror region and a partly filled ring."""
    def run(cache):
        monkeypatch.setenv("MAPDN_CACHE_NEXT_ACTIONS", "1" if cache else "0")
        torch.manual_seed(0); np.random.seed(0)
        env = _ToyEnv(4, 3, 5)
        args = make_alg_args(3, 5, 1, hid_size=16, max_steps=12, batch_size=8, replay_buffer_size=28,
                             behaviour_update_freq=4, target_update_freq=6, value_update_epochs=4, num_eval_episodes=4)
        tr = PGTrainer(args, alg, env, device="cpu", data_parallel=False)
        seen = []
        orig = tr._cache_targets
        tr._cache_targets = lambda: (seen.append(orig()), seen[-1])[1]
        stat = {}
        for ep in range(3):
            tr.train_process(stat)
        assert "next_action_cached" not in tr.replay_buffer.store and "next_value_cached" not in tr.replay_buffer.store   # removed after every round
        return {k: v.clone() for k, v in tr.behaviour_net.state_dict().items()}, seen, stat["mean_train_value_loss"]
    a, na, la = run(True)
    b, nb, lb = run(False)
    assert len(na) == len(nb) >= 6 and all(na) and not any(nb) and la == lb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_target_cache_is_gated_on_cost():
    """a whole-ring pass pays off only when the round's value epochs sample at least the ring (value_update_epochs x
    batch_size >= ring length).  The reference's own defaults (a 5000-transition ring, 10 epochs of 32) keep the per-epoch passes."""
    torch.manual_seed(0); np.random.seed(0)
    env = _ToyEnv(4, 3, 5)
    for ring, epochs, want in ((64, 3, False), (24, 3, True), (5000, 10, False)):
        args = make_alg_args(3, 5, 1, hid_size=16, max_steps=12, batch_size=8, replay_buffer_size=ring, behaviour_update_freq=4,
                             target_update_freq=6, value_update_epochs=epochs, num_eval_episodes=4)
        tr = PGTrainer(args, "maddpg", env, device="cpu", data_parallel=False)
        got = []
        orig = tr._cache_targets
        tr._cache_targets = lambda: (got.append(orig()), got[-1])[1]
        for ep in range(3):
            tr.train_process({})
        assert len(got) >= 6 and got[-1] == want and got[-2] == want, (ring, epochs, got)     # (the ring is as full as it gets by then)
        assert got[0]                                             # a ring that still holds less than the round samples IS cached


@pytest.mark.gpu
def test_tall_batch_routes_give_the_stock_gradients(monkeypatch):
    """round 5, on the GPU at a size that takes the tall routes (b * n = 2^18 rows): the policy loss and the value loss of one batch,
    with MAPDN_TALL_LINEAR on / off — split-K weight gradients in the critic and agent trunks, the policy's first layer, the GRU cell
    through ATen's fused cell on pre-computed gates — same losses, gradients equal to f32 summation-order accuracy."""
    from mapdn_amd.learner import _gru_fused_ok
    dev = torch.device("cuda:0")
    n, o, h, bs = 32, 20, 64, 8192
    assert _gru_fused_ok(dev)
    torch.manual_seed(3)
    args = make_alg_args(n, o, 1, hid_size=h, reward_normalisation=False)
    net = DDPGNet(args, "maddpg", DDPGNet(args, "maddpg").to(dev)).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)      # noqa: E731
    batch = dict(state=r(bs, n, o), action=torch.tanh(r(bs, n, 1)), reward=r(bs, 1).expand(bs, n).contiguous(), next_state=r(bs, n, o),
                 done=(torch.rand(bs, 1, generator=g) < 0.2).float().to(dev), last_step=torch.zeros(bs, 1, device=dev),
                 action_avail=torch.ones(bs, n, 1, device=dev), last_hid=0.3 * r(bs, n, h), hid=0.3 * r(bs, n, h))
    params = [p for name, p in net.named_parameters() if p.requires_grad and not name.startswith("target_net")]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MAPDN_TALL_LINEAR", flag)
        pl, vl, _ = net.get_loss(batch)
        gp = torch.autograd.grad(pl, params, retain_graph=True, allow_unused=True)
        gv = torch.autograd.grad(vl, params, allow_unused=True)
        res[flag] = (pl.item(), vl.item(), gp, gv)
    assert abs(res["1"][0] - res["0"][0]) < 1e-6 and abs(res["1"][1] - res["0"][1]) < 1e-6
    for which in (2, 3):
        for a, b in zip(res["1"][which], res["0"][which]):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.allclose(a, b, rtol=2e-3, atol=2e-6), (which, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("nb,n", [(700, 38), (4096, 6), (33, 22)])
def test_broadcast_input_layernorm_equals_the_materialised_one(nb, n):
    """mapdn_layernorm64_bc_* (round 5): relu(LayerNorm(base[b] + per_n[i])) without the [b, n, 64] tensor — output BIT-identical to
    the materialised route through mapdn_layernorm64_forward (the row is formed by the same single f32 add), dgamma / dbeta identical,
    dbase / dper_n equal to the autograd reductions of that route."""
    from mapdn_amd.learner import _LayerNorm64, _LayerNorm64BC
    g = torch.Generator(device="cuda:0"); g.manual_seed(nb + n)
    base = (torch.randn(nb, 64, device="cuda:0", generator=g) * 1.3).requires_grad_(True)
    pern = (torch.randn(n, 64, device="cuda:0", generator=g) * 0.7).requires_grad_(True)
    w = (torch.randn(64, device="cuda:0", generator=g) * 0.5 + 1.0).requires_grad_(True)
    b = (torch.randn(64, device="cuda:0", generator=g) * 0.2).requires_grad_(True)
    dy = torch.randn(nb * n, 64, device="cuda:0", generator=g)
    y = _LayerNorm64BC.apply(base, pern, w, b, 1e-5, True)
    y.backward(dy)
    got = [y.detach().clone(), base.grad.clone(), pern.grad.clone(), w.grad.clone(), b.grad.clone()]
    base.grad = pern.grad = w.grad = b.grad = None
    x = (base.unsqueeze(1) + pern.unsqueeze(0)).reshape(nb * n, 64)
    ref = _LayerNorm64.apply(x, w, b, 1e-5, True)
    ref.backward(dy)
    assert torch.equal(got[0], ref.detach())
    assert torch.equal(got[3], w.grad) and torch.equal(got[4], b.grad)
    assert torch.allclose(got[1], base.grad, rtol=1e-5, atol=1e-5) and torch.allclose(got[2], pern.grad, rtol=1e-4, atol=1e-4 * max(1.0, nb ** 0.5))


@pytest.mark.gpu
def test_central_critic_takes_the_broadcast_route_and_gives_the_same_values(monkeypatch):
    """DDPGNet._value_central without an action-gradient path (value loss, target values): same values and parameter gradients with
    MAPDN_FUSED_LN_BC on / off"""
    dev = torch.device("cuda:0")
    n, o, h, bs = 22, 20, 64, 600
    torch.manual_seed(4)
    args = make_alg_args(n, o, 1, hid_size=h, reward_normalisation=False)
    net = DDPGNet(args, "maddpg", DDPGNet(args, "maddpg").to(dev)).to(dev)
    g = torch.Generator(device="cpu").manual_seed(6)
    obs, act = torch.randn(bs, n, o, generator=g).to(dev), torch.tanh(torch.randn(bs, n, 1, generator=g)).to(dev)
    params = [p for name, p in net.named_parameters() if name.startswith("value_dicts")]
    out = {}
    monkeypatch.setenv("MAPDN_FUSED_HEAD", "0")                  # (round 6's one-launch head would take both: tests/test_critic_head.py)
    for flag in ("1", "0"):
        monkeypatch.setenv("MAPDN_FUSED_LN_BC", flag)
        v = net.value(obs, act)
        gr = torch.autograd.grad(v.square().mean(), params, allow_unused=True)
        out[flag] = (v.detach().clone(), gr)
    assert torch.equal(out["1"][0], out["0"][0])
    for a, b in zip(out["1"][1], out["0"][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [65536, 300001])
def test_relu_dot64_kernels_match_pytorch(rows):
    """mapdn_relu_dot64_* (round 5): v = relu(pre) @ w^T + b on rows of 64 as one pass — values and the three gradients against
    torch.relu + F.linear with autograd; deterministic (fixed-order block reduction)."""
    from mapdn_amd.learner import _ReluDot64
    g = torch.Generator(device="cuda:0"); g.manual_seed(rows)
    pre = (torch.randn(rows, 64, device="cuda:0", generator=g) * 1.5).requires_grad_(True)
    w = (torch.randn(1, 64, device="cuda:0", generator=g) * 0.3).requires_grad_(True)
    b = torch.randn(1, device="cuda:0", generator=g).requires_grad_(True)
    dv = torch.randn(rows, 1, device="cuda:0", generator=g)
    v = _ReluDot64.apply(pre, w, b)
    v.backward(dv)
    got = [v.detach().clone(), pre.grad.clone(), w.grad.clone(), b.grad.clone()]
    pre.grad = w.grad = b.grad = None
    ref = torch.nn.functional.linear(torch.relu(pre), w, b)
    ref.backward(dv)
    assert torch.allclose(got[0], ref.detach(), rtol=1e-5, atol=1e-5)
    assert torch.equal(got[1], pre.grad)                               # dpre = [pre > 0] dv w: one product per element, same bits
    scale = rows ** 0.5
    assert (got[2] - w.grad).abs().max().item() < 1e-5 * scale * 10 and (got[3] - b.grad).abs().max().item() < 1e-5 * scale * 10
    pre.grad = None
    v2 = _ReluDot64.apply(pre, w, b); v2.backward(dv)
    assert torch.equal(v2.detach(), got[0]) and torch.equal(pre.grad, got[1])


def test_splitk_weight_gradient_helper_on_column_slices():
    """round 6: learner._splitk_dw (the K = rows products behind _PolicyTrunk / tall_linear_w): dy^T x in blocks, dy possibly a COLUMN
    SLICE of a wider tensor (the GRU's gate-gradient tensor [rows, 256] is read as [:, :192], [:, :128], [:, 192:])"""
    from mapdn_amd.learner import _splitk_dw
    g = torch.Generator().manual_seed(0)
    rows = 1000
    dg = torch.randn(rows, 256, generator=g, dtype=torch.float64)
    x = torch.randn(rows, 64, generator=g, dtype=torch.float64)
    for sl in (slice(0, 192), slice(0, 128), slice(192, 256), slice(0, 256)):
        for blk in (64, 128, 1000, 4096):
            got = _splitk_dw(dg[:, sl], x, blk=blk)
            assert torch.allclose(got, dg[:, sl].t() @ x, rtol=0, atol=1e-10), (sl, blk)


def test_policy_means_grad_only_falls_back_to_the_modules_on_cpu():
    """DDPGNet.policy(..., means_grad_only=True) without a GPU is the stock route: same means, hidden state returned"""
    torch.manual_seed(0)
    args = make_alg_args(3, 5, 1)
    net = DDPGNet(args, "maddpg")
    obs, hid = torch.randn(7, 3, 5), torch.randn(7, 3, 64)
    m1, s1, h1 = net.policy(obs, hid)
    m2, s2, h2 = net.policy(obs, hid, means_grad_only=True)
    assert torch.equal(m1, m2) and torch.equal(h1, h2) and torch.equal(s1, s2)
