"""The one-launch critic head (round 6; csrc/critic.hip, mapdn_critic_head_*): v = relu(relu(LayerNorm(x)) W2^T + b2) . w3 + b3 on rows of
64 — critics/mlp_critic.py:22-36 behind models/maddpg.py:35-79 / models/iddpg.py:32-58, trained by learning_algorithms/ddpg.py:15-39 —
against the PyTorch modules it replaces: values and EVERY gradient, for read rows and for rows formed as base[b] + per_n[i]."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mapdn_amd.learner import DDPGNet, MLPCritic, make_alg_args


def _critic(dev, seed):
    torch.manual_seed(seed)
    args = make_alg_args(3, 5, 1)
    cr = MLPCritic(7, 1, args)
    with torch.no_grad():
        cr.layernorm.weight.copy_(1.0 + 0.3 * torch.randn(64)); cr.layernorm.bias.copy_(0.2 * torch.randn(64))
        cr.fc2.weight.copy_(0.2 * torch.randn(64, 64)); cr.fc2.bias.copy_(0.1 * torch.randn(64))
        cr.fc3.weight.copy_(0.3 * torch.randn(1, 64)); cr.fc3.bias.copy_(0.1 * torch.randn(1))
    return cr.to(dev)


def _stock(cr, x):
    """the reference's modules in float64 (critics/mlp_critic.py:22-36)"""
    c = cr.double() if False else cr
    ln, fc2, fc3 = c.layernorm, c.fc2, c.fc3
    xn = F.relu(F.layer_norm(x.double(), (64,), ln.weight.double(), ln.bias.double(), ln.eps))
    return F.linear(F.relu(F.linear(xn, fc2.weight.double(), fc2.bias.double())), fc3.weight.double(), fc3.bias.double())


def _close(a, b, tol, what):
    scale = max(1.0, float(b.detach().abs().max()))
    err = float((a.double() - b.double()).abs().max())
    assert err <= tol * scale, (what, err, scale)


HEAD_PARAMS = ("layernorm.weight", "layernorm.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1024, 4101, 300007])
def test_head_on_read_rows_matches_the_modules(rows):
    from mapdn_amd.learner import critic_head
    dev = torch.device("cuda:0")
    cr = _critic(dev, rows)
    prm = [dict(cr.named_parameters())[k] for k in HEAD_PARAMS]
    g = torch.Generator(device="cpu").manual_seed(rows)
    x = (1.5 * torch.randn(rows, 64, generator=g) + 0.3).to(dev).requires_grad_(True)
    dv = torch.randn(rows, 1, generator=g).to(dev)
    v = critic_head(cr, x)
    got = torch.autograd.grad(v, [x] + prm, dv)
    v2 = critic_head(cr, x)                                          # deterministic: same bits on a second run
    got2 = torch.autograd.grad(v2, [x] + prm, dv)
    assert torch.equal(v, v2) and all(torch.equal(a, b) for a, b in zip(got, got2))
    ref = _stock(cr, x)
    want = torch.autograd.grad(ref, [x] + prm, dv.double())
    _close(v, ref, 2e-6, "v")
    _close(got[0], want[0], 2e-6, "dx")
    for name, a, b in zip(HEAD_PARAMS, got[1:], want[1:]):
        assert a.shape == b.shape
        _close(a, b, 3e-7 * max(1.0, rows ** 0.5), name)           # f32 sums over `rows` terms in another order
    # parameters that do not require grad: dx alone, same bits
    frozen = [p.detach() for p in prm]
    from mapdn_amd.learner import _CriticHead
    v4 = _CriticHead.apply(x, None, frozen[0], frozen[1], cr.layernorm.eps, frozen[2], frozen[3], frozen[4], frozen[5])
    (dx4,) = torch.autograd.grad(v4, [x], dv)
    assert torch.equal(v4, v) and torch.equal(dx4, got[0])


@pytest.mark.gpu
@pytest.mark.parametrize("nb,n", [(27, 38), (700, 38), (4096, 6), (9001, 22), (70000, 3), (3000, 50), (600, 88), (200, 1)])
def test_head_on_formed_rows_matches_the_modules(nb, n):
    """x[b, i] = base[b] + per_n[i] never exists: values, dbase (summed over the agents IN the kernel), dper_n (summed over the batch) and
    the parameter gradients against autograd through the materialised sum"""
    from mapdn_amd.learner import critic_head
    dev = torch.device("cuda:0")
    cr = _critic(dev, nb + n)
    prm = [dict(cr.named_parameters())[k] for k in HEAD_PARAMS]
    g = torch.Generator(device="cpu").manual_seed(nb * 100 + n)
    base = (1.2 * torch.randn(nb, 64, generator=g)).to(dev).requires_grad_(True)
    pern = (0.8 * torch.randn(n, 64, generator=g)).to(dev).requires_grad_(True)
    dv = torch.randn(nb * n, 1, generator=g).to(dev)
    v = critic_head(cr, base, pern)
    got = torch.autograd.grad(v, [base, pern] + prm, dv)
    x = (base.unsqueeze(1) + pern.unsqueeze(0)).reshape(nb * n, 64)        # the one f32 add the kernel performs
    ref = _stock(cr, x)
    want = torch.autograd.grad(ref, [base, pern] + prm, dv.double())
    _close(v, ref, 2e-6, "v")
    _close(got[0], want[0], 3e-6, "dbase")
    _close(got[1], want[1], 3e-7 * max(1.0, nb ** 0.5) * 4, "dper_n")
    for name, a, b in zip(HEAD_PARAMS, got[2:], want[2:]):
        _close(a, b, 3e-7 * max(1.0, (nb * n) ** 0.5), name)
    # and the read-rows kernel on the materialised sum gives the same VALUES bit for bit (same arithmetic per row)
    assert torch.equal(critic_head(cr, x.detach()), v)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["maddpg", "iddpg"])
def test_learner_losses_and_gradients_with_and_without_the_head(alg, monkeypatch):
    """DDPGNet.get_loss on one batch with MAPDN_FUSED_HEAD on / off: same losses; value-loss gradients of every critic parameter and
    policy-loss gradients of every POLICY parameter equal to f32 summation-order accuracy (the policy update through the central
    critic differentiates with respect to the agents' own actions only — models/maddpg.py:52-58 — and no longer produces the critic
    parameters' gradients, which utilities/trainer.py:73-98 never reads)."""
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
    val = [p for name, p in net.named_parameters() if name.startswith("value_dicts")]
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MAPDN_FUSED_HEAD", flag)
        pl, vl, _ = net.get_loss(batch)
        gp = torch.autograd.grad(pl, pol, retain_graph=True)
        gv = torch.autograd.grad(vl, val)
        res[flag] = (pl.item(), vl.item(), gp, gv)
    assert abs(res["1"][0] - res["0"][0]) < 2e-6 * max(1.0, abs(res["0"][0])) and abs(res["1"][1] - res["0"][1]) < 2e-6 * max(1.0, abs(res["0"][1]))
    for which in (2, 3):
        for a, b in zip(res["1"][which], res["0"][which]):
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-6), (which, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.gpu
def test_own_action_gradient_matches_the_reference_construction():
    """models/maddpg.py:41-66 builds, per agent i, the critic input [all obs | one-hot i | all actions with every action but i's detached].
    Here: that construction literally (a [b, n, n (o + 1) + n] tensor, autograd through the stock modules) against the head's
    own-action route — values and d loss / d action."""
    dev = torch.device("cuda:0")
    n, o, bs = 6, 11, 300
    torch.manual_seed(8)
    args = make_alg_args(n, o, 1, reward_normalisation=False)
    net = DDPGNet(args, "maddpg").to(dev)
    cr = net.value_dicts[0]
    g = torch.Generator(device="cpu").manual_seed(9)
    obs = torch.randn(bs, n, o, generator=g).to(dev)
    act = torch.tanh(torch.randn(bs, n, 1, generator=g)).to(dev).requires_grad_(True)
    dv = torch.randn(bs, n, 1, generator=g).to(dev)
    v = net.value(obs, act, own_action_only=True)
    (dact,) = torch.autograd.grad(v, [act], dv)
    # the reference's input, agent by agent
    obs_rep = obs.reshape(bs, 1, n * o).expand(bs, n, n * o)
    eye = torch.eye(n, device=dev).unsqueeze(0).expand(bs, n, n)
    act_rep = act.reshape(bs, 1, n).expand(bs, n, n)
    mask = torch.eye(n, device=dev).unsqueeze(0)
    act_in = act_rep * mask + act_rep.detach() * (1 - mask)
    inp = torch.cat((obs_rep, eye, act_in), dim=-1).double()
    x = F.linear(inp, cr.fc1.weight.double(), cr.fc1.bias.double()).reshape(bs * n, 64)
    ref = _stock(cr, x).view(bs, n, 1)
    (dref,) = torch.autograd.grad(ref, [act], dv.double())
    _close(v, ref, 3e-6, "v")
    _close(dact, dref, 3e-6, "dact")


@pytest.mark.gpu
def test_head_refuses_bad_arguments():
    from mapdn_amd import _lib
    lib = _lib.load()
    t = torch.zeros(64 * 64, device="cuda:0")
    p = t.data_ptr()
    assert lib.mapdn_critic_head_forward(p, p, 7, p, p, 1e-5, p, p, p, p, p, 20, None) == -1     # rows not a multiple of n
    assert lib.mapdn_critic_head_forward(None, None, 1, p, p, 1e-5, p, p, p, p, p, 16, None) == -1
    assert lib.mapdn_critic_head_scratch_floats(0, 1, 0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("nb,n,formed,weighted", [(700, 38, True, False), (5000, 22, True, True), (300, 6, False, True), (40000, 1, False, False)])
def test_fused_value_loss_matches_the_modules(nb, n, formed, weighted):
    """mapdn_critic_head_mse: sum_rows w (returns - v)^2 and every gradient of it out of one launch (no forward launch), against the
    same expression through the stock modules with autograd; the incoming gradient scales the result."""
    from mapdn_amd.learner import _CriticHeadMSE
    dev = torch.device("cuda:0")
    cr = _critic(dev, nb + n + 1)
    prm = [dict(cr.named_parameters())[k] for k in HEAD_PARAMS]
    g = torch.Generator(device="cpu").manual_seed(nb * 10 + n)
    rows = nb * n
    if formed:
        base = (1.2 * torch.randn(nb, 64, generator=g)).to(dev).requires_grad_(True)
        pern = (0.8 * torch.randn(n, 64, generator=g)).to(dev).requires_grad_(True)
        x = (base.unsqueeze(1) + pern.unsqueeze(0)).reshape(rows, 64)
        leaves = [base, pern]
    else:
        xin = (1.2 * torch.randn(rows, 64, generator=g)).to(dev).requires_grad_(True)
        x, pern, leaves = xin, None, [xin]
    ret = torch.randn(rows, generator=g).to(dev)
    valid = (torch.rand(nb, generator=g) < 0.7).float().to(dev) if weighted else None
    if valid is None:
        scale, wrow, wfull = torch.full((1,), 1.0 / rows, device=dev), None, torch.full((rows,), 1.0 / rows, device=dev, dtype=torch.float64)
    else:
        scale = (1.0 / (valid.sum().clamp(min=1.0) * n)).reshape(1)
        wrow = valid if formed else valid.repeat_interleave(n)
        wfull = valid.repeat_interleave(n).double() * scale.double()
    ln = cr.layernorm
    loss = _CriticHeadMSE.apply(leaves[0], pern if formed else None, ln.weight, ln.bias, ln.eps, cr.fc2.weight, cr.fc2.bias, cr.fc3.weight,
                                cr.fc3.bias, ret, wrow, scale)
    got = torch.autograd.grad(3.0 * loss, leaves + prm)
    ref = (wfull * (ret.double() - _stock(cr, x).view(-1)) ** 2).sum()
    want = torch.autograd.grad(3.0 * ref, leaves + prm)
    assert abs(loss.item() - ref.item()) <= 3e-6 * max(1.0, abs(ref.item()))
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape
        err, ref_max = float((a.double() - b).abs().max()), float(b.abs().max())
        assert err <= 2e-4 * ref_max, (i, err, ref_max)                  # (the weights are 1 / rows: gradients of order 1e-6 — relative bar)


@pytest.mark.gpu
def test_many_agent_nets_fall_back_to_the_unfused_route():
    """formed rows keep an [n][64] accumulator per wavefront in LDS: beyond 88 agents the central critic takes the round-5 route (and the
    library itself refuses such a launch by code instead of overrunning LDS)"""
    from mapdn_amd import _lib
    from mapdn_amd.learner import HEAD_MAX_FORMED_N, critic_head_ok
    dev = torch.device("cuda:0")
    cr = _critic(dev, 1)
    base = torch.randn(64, 64, device=dev)
    assert critic_head_ok(cr, base, 64 * HEAD_MAX_FORMED_N, HEAD_MAX_FORMED_N) and not critic_head_ok(cr, base, 64 * 120, 120)
    n, rows = 200, 200 * 64
    t = torch.zeros(max(rows, 64 * 64) * 64, device=dev)
    p = t.data_ptr()
    lib = _lib.load()
    scratch = torch.zeros(max(1, lib.mapdn_critic_head_scratch_floats(rows, n, 1)), device=dev)
    assert lib.mapdn_critic_head_backward(p, p, p, n, p, p, 1e-5, p, p, p, p, p, p, scratch.data_ptr(), rows, 1, None) == -1
    args = make_alg_args(120, 4, 1, reward_normalisation=False)
    net = DDPGNet(args, "maddpg").to(dev)
    v = net.value(torch.randn(32, 120, 4, device=dev), torch.randn(32, 120, 1, device=dev))
    v.mean().backward()
    assert v.shape == (32, 120, 1) and net.value_dicts[0].fc2.weight.grad is not None


@pytest.mark.gpu
def test_head_at_the_end_to_end_size_is_the_sum_of_its_halves():
    """BASELINE configs[4]'s per-GPU batch (32 steps x 8192 envs x 38 agents = 9 961 472 rows, the size bench.py times): a size-independent
    property instead of an oracle — per-row arithmetic does not depend on the launch shape, so v and dbase of the full batch are
    BIT-identical to those of its two halves, and the parameter gradients are their sum to f32 summation order."""
    from mapdn_amd.learner import critic_head
    dev = torch.device("cuda:0")
    nb, n = 32 * 8192, 38
    cr = _critic(dev, 7)
    prm = [dict(cr.named_parameters())[k] for k in HEAD_PARAMS]
    g = torch.Generator(device=dev); g.manual_seed(3)
    base = torch.randn(nb, 64, device=dev, generator=g).requires_grad_(True)
    pern = (0.7 * torch.randn(n, 64, device=dev, generator=g)).requires_grad_(True)
    dv = torch.randn(nb * n, 1, device=dev, generator=g) / (nb * n)
    v = critic_head(cr, base, pern)
    full = torch.autograd.grad(v, [base, pern] + prm, dv)
    h = nb // 2
    parts = []
    for lo, hi in ((0, h), (h, nb)):
        b2 = base.detach()[lo:hi].clone().requires_grad_(True)
        vv = critic_head(cr, b2, pern)
        assert torch.equal(vv, v[lo * n:hi * n])
        parts.append(torch.autograd.grad(vv, [b2, pern] + prm, dv[lo * n:hi * n]))
    assert torch.equal(torch.cat((parts[0][0], parts[1][0])), full[0])                        # dbase: row-local, bit-identical
    for a, b, c in zip(full[1:], parts[0][1:], parts[1][1:]):
        ref = b.double() + c.double()
        assert float((a.double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-12)
    assert torch.isfinite(v).all()
