"""The policy update's agent trunk as HIP launches (round 6; csrc/policy.hip: mapdn_policy_forward_train, csrc/policy_bwd.hip:
mapdn_policy_backward; learner.py::_PolicyTrunk): fc1 + id column -> LayerNorm -> ReLU -> GRUCell -> fc2 of agents/rnn_agent.py:5-32 as
trained by models/maddpg.py:103-125 — means and the gradient of every policy parameter against the PyTorch modules in float64."""
import copy

import pytest
import torch

from mapdn_amd.learner import DDPGNet, make_alg_args


@pytest.mark.gpu
@pytest.mark.parametrize("n,o,b,ids", [(6, 26, 700, True), (22, 58, 300, True), (38, 82, 1000, True), (22, 58, 257, False), (38, 82, 9000, True)])
def test_policy_trunk_forward_and_every_gradient(n, o, b, ids, monkeypatch):
    dev = torch.device("cuda:0")
    torch.manual_seed(n + o)
    args = make_alg_args(n, o, 1, agent_id=ids)
    net = DDPGNet(args, "maddpg").to(dev)
    with torch.no_grad():
        ag = net.policy_dicts[0]
        ag.layernorm.weight.copy_(1.0 + 0.3 * torch.randn(64)); ag.layernorm.bias.copy_(0.2 * torch.randn(64))
        ag.fc2.weight.mul_(3.0)
    net64 = copy.deepcopy(net).double()
    g = torch.Generator(device="cpu").manual_seed(b)
    obs = torch.randn(b, n, o, generator=g).to(dev)
    hid = (0.5 * torch.randn(b, n, 64, generator=g)).to(dev)
    dm = torch.randn(b, n, 1, generator=g).to(dev)
    params = [p for _, p in net.policy_dicts.named_parameters()]
    names = [k for k, _ in net.policy_dicts.named_parameters()]
    means, log_std, hidden = net.policy(obs, hid, means_grad_only=True)
    assert hidden is None and means.shape == (b, n, 1) and means.grad_fn is not None
    assert "PolicyTrunk" in type(means.grad_fn.next_functions[0][0]).__name__            # (behind the view to [b, n, 1])
    got = torch.autograd.grad(means, params, dm)
    m2, _, _ = net.policy(obs, hid, means_grad_only=True)
    got2 = torch.autograd.grad(m2, params, dm)
    assert torch.equal(means, m2) and all(torch.equal(x, y) for x, y in zip(got, got2))            # deterministic
    ref, _, _ = net64.policy(obs.double(), hid.double())
    want = torch.autograd.grad(ref, [p for _, p in net64.policy_dicts.named_parameters()], dm.double())
    assert float((means.double() - ref).abs().max()) < 5e-6 * max(1.0, float(ref.abs().max()))
    rows = b * n
    for name, a, w in zip(names, got, want):
        assert a.shape == w.shape, name
        err, scale = float((a.double() - w).abs().max()), max(1e-3, float(w.abs().max()))
        assert err <= 4e-7 * max(1.0, rows ** 0.5) * max(scale, 1.0) + 2e-5 * scale, (name, err, scale)
    # and the stock f32 route (flag off) agrees too
    monkeypatch.setenv("MAPDN_FUSED_POLICY_TRAIN", "0")
    m3, _, h3 = net.policy(obs, hid, means_grad_only=True)
    assert h3 is not None and torch.allclose(m3, means, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["maddpg", "iddpg"])
def test_policy_loss_gradients_with_and_without_the_fused_trunk(alg, monkeypatch):
    dev = torch.device("cuda:0")
    n, o, h, bs = 38, 30, 64, 512
    torch.manual_seed(3)
    args = make_alg_args(n, o, 1, hid_size=h, reward_normalisation=False)
    net = DDPGNet(args, alg, DDPGNet(args, alg).to(dev)).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)      # noqa: E731
    batch = dict(state=r(bs, n, o), action=torch.tanh(r(bs, n, 1)), reward=r(bs, 1).expand(bs, n).contiguous(), next_state=r(bs, n, o),
                 done=(torch.rand(bs, 1, generator=g) < 0.2).float().to(dev), last_step=torch.zeros(bs, 1, device=dev),
                 action_avail=torch.ones(bs, n, 1, device=dev), last_hid=0.3 * r(bs, n, h), hid=0.3 * r(bs, n, h))
    pol = [p for name, p in net.named_parameters() if name.startswith("policy_dicts")]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MAPDN_FUSED_POLICY_TRAIN", flag)
        pl, _, (means, log_stds) = net.get_loss(batch, want=("policy",))
        res[flag] = (pl.item(), torch.autograd.grad(pl, pol), means.detach().clone())
    assert abs(res["1"][0] - res["0"][0]) < 2e-6 * max(1.0, abs(res["0"][0]))
    assert torch.allclose(res["1"][2], res["0"][2], rtol=1e-4, atol=1e-5)
    for a, b in zip(res["1"][1], res["0"][1]):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-6), ((a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.gpu
def test_policy_trunk_at_two_million_rows_is_the_sum_of_its_halves():
    """a size-independent property at a batch far beyond what the float64 modules can replay quickly (65 536 x 38 = 2.5 M rows of the 322-bus
    shape): means of the full batch are bit-identical to those of its halves, parameter gradients their sum to f32 summation order"""
    dev = torch.device("cuda:0")
    n, o, b = 38, 82, 65536
    torch.manual_seed(1)
    net = DDPGNet(make_alg_args(n, o, 1), "maddpg").to(dev)
    g = torch.Generator(device=dev); g.manual_seed(2)
    obs = torch.randn(b, n, o, device=dev, generator=g)
    hid = 0.5 * torch.randn(b, n, 64, device=dev, generator=g)
    dm = torch.randn(b, n, 1, device=dev, generator=g) / (b * n)
    params = [p for _, p in net.policy_dicts.named_parameters()]
    means, _, _ = net.policy(obs, hid, means_grad_only=True)
    full = torch.autograd.grad(means, params, dm)
    parts = []
    for lo, hi in ((0, b // 2), (b // 2, b)):
        m, _, _ = net.policy(obs[lo:hi], hid[lo:hi], means_grad_only=True)
        assert torch.equal(m, means[lo:hi])
        parts.append(torch.autograd.grad(m, params, dm[lo:hi]))
    for a, x, y in zip(full, parts[0], parts[1]):
        ref = x.double() + y.double()
        assert float((a.double() - ref).abs().max()) <= 3e-5 * max(float(ref.abs().max()), 1e-12)
