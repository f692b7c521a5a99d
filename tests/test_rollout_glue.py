"""csrc/rollout.hip (round 6): the glue of one batched rollout step as three launches — exploration sample + availability mask +
translate_action (utilities/util.py:52-76, 123-132; maddpg.py:92-93), the running sums of reward / info over the live envs
(models/model.py:243-248) and the replay insertion (utilities/replay_buffer.py:25-29) — against the PyTorch chains they replace."""
import numpy as np
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("scale,bias,bound,std", [(0.8, 0.0, True, 1.0), (0.6, 0.0, True, 1.0), (0.5, 0.25, True, 0.7), (0.8, 0.1, False, 1.0)])
def test_explore_actions_is_the_pytorch_chain_bit_for_bit(scale, bias, bound, std):
    import math
    from mapdn_amd import _lib
    from mapdn_amd.learner import DDPGNet, make_alg_args
    from mapdn_amd.rollout import translate_action
    dev = torch.device("cuda:0")
    B, n = 300, 22
    args = make_alg_args(n, 10, 1, scale, bias, action_enforcebound=bound, fixed_policy_std=std)
    net = DDPGNet(args, "maddpg").to(dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    means = 1.5 * torch.randn(B, n, 1, device=dev, generator=g)
    avail = (torch.rand(B, n, 1, device=dev, generator=g) > 0.1).float()
    log_std = torch.full_like(means, math.log(std))
    torch.manual_seed(11)
    act, _ = net._select_action(means, log_std, "train", True)           # draws randn_like(means) from the global CUDA generator
    pol = (1.0 - (avail == 0).to(act.dtype)) * act
    actual = translate_action(act.squeeze(-1), scale, bias)
    torch.manual_seed(11)
    eps = torch.randn_like(means)
    a2, p2, t2 = torch.empty_like(means), torch.empty_like(means), torch.empty_like(means)
    stdv = float(torch.tensor(math.log(std), dtype=torch.float32).exp())
    _lib.check(_lib.load().mapdn_explore_actions(means.data_ptr(), eps.data_ptr(), avail.data_ptr(), stdv, int(bound), scale, bias, a2.data_ptr(),
                                                 p2.data_ptr(), t2.data_ptr(), means.numel(), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(a2, act) and torch.equal(p2, pol) and torch.equal(t2.squeeze(-1), actual)


@pytest.mark.gpu
def test_rollout_stats_kernel():
    from mapdn_amd import _lib
    dev = torch.device("cuda:0")
    B = 5000
    g = torch.Generator(device="cpu").manual_seed(0)
    info = torch.randn(B, 11, generator=g, dtype=torch.float64).to(dev)
    reward = torch.randn(B, generator=g, dtype=torch.float64).to(dev)
    alive = (torch.rand(B, generator=g) > 0.3).to(dev)
    done = (torch.rand(B, generator=g) > 0.8).to(dev)
    sums = torch.arange(13, dtype=torch.float64, device=dev)
    out = torch.empty_like(alive)
    _lib.check(_lib.load().mapdn_rollout_stats(info.data_ptr(), reward.data_ptr(), alive.data_ptr(), done.data_ptr(), out.data_ptr(), sums.data_ptr(), B,
                                               torch.cuda.current_stream().cuda_stream))
    w = alive.double()
    want = torch.arange(13, dtype=torch.float64, device=dev) + torch.cat(((info * w.unsqueeze(-1)).sum(0), (reward * w).sum().view(1), w.sum().view(1)))
    assert torch.allclose(sums, want, rtol=1e-13, atol=1e-12) and torch.equal(out, alive & ~done)


@pytest.mark.gpu
def test_replay_insertion_in_one_launch_fills_the_same_ring(monkeypatch):
    from mapdn_amd.replay import TransReplayBuffer
    dev = torch.device("cuda:0")
    B, n, o = 64, 6, 26

    def run(flag):
        monkeypatch.setenv("MAPDN_FUSED_ROLLOUT", flag[0])
        monkeypatch.setenv("MAPDN_REPLAY_ASYNC", flag[1])
        g = torch.Generator(device="cpu").manual_seed(1)
        rb = TransReplayBuffer(5 * B + 32, device=dev, window=2 * B)             # not a multiple of B: insertions wrap mid-batch
        for t in range(13):
            tr = dict(state=torch.randn(B, n, o, generator=g).to(dev), action=torch.randn(B, n, 1, generator=g).to(dev),
                      reward=torch.randn(B, n, generator=g).to(dev), done=(torch.rand(B, 1, generator=g) > 0.5).float().to(dev),
                      hid=torch.randn(B, n, 64, generator=g).to(dev), valid=(torch.rand(B, generator=g) > 0.5).to(dev))
            rb.add_experience(tr)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in rb.store.items()}, len(rb)
    b, lb = run("00")
    for flags in ("11", "10"):                                    # one launch on a side stream (round 6) / on the current stream
        a, la = run(flags)
        assert la == lb
        for k in a:
            assert torch.equal(a[k][:5 * B + 32], b[k][:5 * B + 32]) and torch.equal(a[k][5 * B + 32:], b[k][5 * B + 32:]), (flags, k)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["maddpg", "iddpg"])
def test_training_episode_with_and_without_the_fused_glue(alg, monkeypatch):
    """one seeded training run of two episodes on the GPU env, MAPDN_FUSED_ROLLOUT on / off: the exploration noise comes from the same
    torch generator draw, the actions are bit-identical, so the replay contents, every update and the final state_dict are too; the
    logged means agree to f64 summation order"""
    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.learner import PGTrainer, make_alg_args
    from mapdn_amd.netspec import make_case
    dev = torch.device("cuda:0")
    net, prof = make_case("case33")

    def run(flag, async_insert="1"):
        monkeypatch.setenv("MAPDN_FUSED_ROLLOUT", flag)
        monkeypatch.setenv("MAPDN_REPLAY_ASYNC", async_insert)
        torch.manual_seed(5); np.random.seed(5)
        env = VoltageControlBatch(net, prof, dict(episode_limit=24, action_scale=0.8, action_bias=0.0, voltage_barrier_type="bowl", seed=0),
                                  n_envs=64, device=dev, copy=True)
        args = make_alg_args(env.n_agents, env.obs_size, 1, 0.8, 0.0, max_steps=24, batch_size=256, replay_buffer_size=64 * 16,
                             behaviour_update_freq=8, target_update_freq=16, num_eval_episodes=64)
        tr = PGTrainer(args, alg, env, device=dev, data_parallel=False)
        stat = {}
        for _ in range(2):
            tr.train_process(stat)
        env.close()
        return {k: v.clone() for k, v in tr.behaviour_net.state_dict().items()}, stat
    a, sa = run("1")
    b, sb = run("0")
    c, _ = run("1", async_insert="0")
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    for k in sb:
        assert abs(sa[k] - sb[k]) <= 1e-11 * max(1.0, abs(sb[k])), k


@pytest.mark.gpu
def test_reading_the_ring_right_after_an_asynchronous_insertion_sees_it(monkeypatch):
    """the insertion runs on a side stream: every read path (`store`, get_batch, get_single) must wait for it, and the memory of a source
    that the caller drops right after the call must not be handed out again before the copy has read it (record_stream) — sources are
    freed and same-sized tensors of garbage allocated immediately.  (Contract: a source is not modified IN PLACE after add_experience.)"""
    from mapdn_amd.replay import TransReplayBuffer
    monkeypatch.setenv("MAPDN_FUSED_ROLLOUT", "1"); monkeypatch.setenv("MAPDN_REPLAY_ASYNC", "1")
    dev = torch.device("cuda:0")
    B = 256
    rb = TransReplayBuffer(8 * B, device=dev, window=3 * B)
    big = torch.empty(64 << 20, device=dev)                       # some work for the main stream between the calls
    for t in range(40):
        x = torch.full((B, 6, 64), float(t), device=dev)
        y = torch.full((B, 1), float(-t), device=dev)
        rb.add_experience(dict(state=x, done=y))
        del x, y                                                  # dropped at once: the allocator would reuse their blocks ...
        junk = [torch.full((B, 6, 64), 1e9, device=dev), torch.full((B, 1), 1e9, device=dev)]      # ... for these, on the main stream
        big.add_(1.0)
        k = t % 3
        if k == 0:
            last = rb.get_batch(B, start=len(rb) - B)
        elif k == 1:
            last = {f: v[((rb._tail + len(rb) - B) % rb.size):][:B] for f, v in rb.store.items()}
        else:
            last = {f: v.unsqueeze(0) for f, v in rb.get_single(-1).items()}
        assert float(last["state"].min()) == float(last["state"].max()) == float(t), t
        assert float(last["done"].max()) == float(-t)
