"""MADDPG / IDDPG learners for the batched env (BASELINE.json configs[4]).

What is learned, and every quirk of how, follows the reference (file:line cited at each piece):
`models/maddpg.py`, `models/iddpg.py`, `learning_algorithms/ddpg.py`, `models/model.py`,
`agents/rnn_agent.py`, `critics/mlp_critic.py`, `utilities/trainer.py`, `utilities/util.py`,
defaults from `args/default.yaml` + `args/alg_args/{maddpg,iddpg}.yaml`.  Module and parameter names
are the reference's (`policy_dicts.0.fc1.weight`, `value_dicts.0.fc3.bias`, `target_net.…`,
`batchnorm.…`), so a reference `model.pt` (`{"model_state_dict": …}`, train.py:119) loads with
`strict=True` and vice versa.

What is different is how it runs: tensors in, tensors out, no Python lists or host round trips —
a batch is a dict of device tensors straight from `mapdn_amd.replay`; the centralised MADDPG critic
never materialises the reference's [batch, n, n·obs] input (maddpg.py:41-66) — its first layer is
evaluated as (shared observation term) + (agent-id column) + (joint-action term), with the
"other agents' actions are detached" rule (maddpg.py:52-58) kept by a zero-valued, gradient-carrying
own-action term; data-parallel ranks average gradients through one flat RCCL all-reduce per update.

Only the configuration the DDPG family trains with exists: continuous actions, deterministic
(non-Gaussian) policy head.  Anything else raises.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import INFO_KEYS
from .replay import TransReplayBuffer
from .rollout import translate_action

# args/default.yaml:6-51 merged with args/alg_args/maddpg.yaml (== iddpg.yaml)
ALG_DEFAULTS = dict(
    gumbel_softmax=False, epsilon_softmax=False, softmax_eps=None, episodic=False, cuda=True, grad_clip_eps=1.0,
    save_model_freq=40, replay_warmup=0, policy_lrate=1.0e-4, value_lrate=1.0e-4, mixer_lrate=None, target=True,
    target_lr=0.1, entr=1.0e-3, max_steps=240, batch_size=32, replay=True, replay_buffer_size=5.0e3,
    agent_type="rnn", agent_id=True, shared_params=True, layernorm=True, mixer=False, gaussian_policy=False,
    LOG_STD_MIN=0.0, LOG_STD_MAX=0.5, fixed_policy_std=1.0, hid_activation="relu", init_type="normal", init_std=0.1,
    action_enforcebound=True, double_q=True, clip_c=1.0, gamma=0.99, hid_size=64, continuous=True,
    normalize_advantages=False, train_episodes_num=400, behaviour_update_freq=60, target_update_freq=120,
    policy_update_epochs=1, value_update_epochs=10, mixer_update_epochs=None, reward_normalisation=True,
    eval_freq=20, num_eval_episodes=10,
)

Batch = Dict[str, torch.Tensor]


def make_alg_args(agent_num: int, obs_size: int, action_dim: int = 1, action_scale: float = 0.8,
                  action_bias: float = 0.0, **overrides) -> SimpleNamespace:
    """The `args` namedtuple train.py:64-67 assembles, as a namespace."""
    d = dict(ALG_DEFAULTS)
    unknown = set(overrides) - set(d)
    if unknown:
        raise KeyError(f"unknown algorithm argument(s): {sorted(unknown)}")
    d.update(overrides)
    d.update(agent_num=int(agent_num), obs_size=int(obs_size), action_dim=int(action_dim),
             action_scale=float(action_scale), action_bias=float(action_bias))
    a = SimpleNamespace(**d)
    if not a.continuous or a.gaussian_policy or a.mixer or a.episodic or a.agent_type != "rnn":
        raise NotImplementedError("the DDPG-family learners here cover continuous, non-Gaussian, recurrent, "
                                  "transition-update training (args/alg_args/maddpg.yaml, iddpg.yaml)")
    return a


def _activation(name: str):
    if name == "relu":
        return F.relu
    if name == "tanh":
        return torch.tanh
    raise ValueError(f"hid_activation {name!r}")


class _LayerNorm64(torch.autograd.Function):
    """LayerNorm over 64 features with the ReLU that follows it in both reference modules fused in (agents/rnn_agent.py:16-21,
    critics/mlp_critic.py:22-27), forward and backward as hand-written HIP kernels (libmapdn_hip.so: mapdn_layernorm64_*): at the
    reference's update intensity a batch is millions of rows of 64, where the stock kernels run at an eighth of the HBM rate."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu):
        from . import _lib
        lib = _lib.load()
        x2 = x.reshape(-1, 64).contiguous()
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(x.device):
            _lib.check(lib.mapdn_layernorm64_forward(x2.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                     rows, float(eps), int(relu), torch.cuda.current_stream(x.device).cuda_stream))
        ctx.save_for_backward(x2, w, b, mean, rstd)
        ctx.relu, ctx.shape = bool(relu), x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        lib = _lib.load()
        x2, w, b, mean, rstd = ctx.saved_tensors
        rows = x2.shape[0]
        dy2 = dy.reshape(-1, 64).contiguous()
        dx = torch.empty_like(x2)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        with torch.cuda.device(x2.device):                  # the block count follows the CU count of the CURRENT device: ask inside the context
            partial = torch.empty(lib.mapdn_layernorm64_backward_blocks(rows) * 128, dtype=torch.float32, device=x2.device)
            _lib.check(lib.mapdn_layernorm64_backward(dy2.data_ptr(), x2.data_ptr(), w.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                      dx.data_ptr(), dw.data_ptr(), db.data_ptr(), partial.data_ptr(), rows, int(ctx.relu),
                                                      torch.cuda.current_stream(x2.device).cuda_stream))
        return dx.view(ctx.shape), dw, db, None, None


class _TallLinear(torch.autograd.Function):
    """y = x W^T + b for x of MILLIONS of rows and a small weight (the 64 -> 64 and 64 -> 1 layers of the critic / agent trunks on a
    batch of 32 x 8192 transitions x 38 agents = 10 M rows).  Forward and dX are ordinary GEMMs; the WEIGHT gradient dW = dy^T x is a
    [out, in] product with the 10 M rows as its reduction dimension — the BLAS back end runs it on two workgroups (8.4 ms and 5.6 ms
    per call, the two largest GEMM entries of profiles/e2e/r04_e2e_reference_kernel_stats.txt, against 1.3 ms for the forward).  Here
    the rows are cut into blocks whose partial products are one batched GEMM, then summed (split-K by hand): same sum, other order."""

    BLOCK_ROWS = 16384

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ weight
        if ctx.needs_input_grad[1]:
            rows, blk = x.shape[0], _TallLinear.BLOCK_ROWS
            full = rows // blk * blk
            dy2, x2 = dy.reshape(rows, -1).contiguous(), x.reshape(rows, -1).contiguous()      # (a permuted upstream gradient is not viewable)
            dw = torch.bmm(dy2[:full].view(-1, blk, dy2.shape[1]).transpose(1, 2), x2[:full].view(-1, blk, x2.shape[1])).sum(0)
            if full < rows:
                dw = dw + dy2[full:].t() @ x2[full:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.reshape(-1, dy.shape[-1]).sum(0)
        return dx, dw, db


def tall_linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """lin(x); with the hand-split weight gradient when x is a tall 2-D batch on the GPU (>= 2^18 rows, <= 128 features each way)"""
    if x.dim() == 2 and _tall_ok(x, x.shape[0], lin.weight):
        return _TallLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


def tall_linear_w(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """x W^T (+ b) for a weight given as a tensor (a column slice of a larger layer): _TallLinear on a tall GPU batch under autograd.
    The central critic's observation block (K = n * obs = 2204-3116 inputs, 2^18 rows of reduction) included: its weight gradient as 16
    batched products + a sum is 0.3 ms faster per update than the back end's single GEMM (round 6, same-box A/B: value update 0.216 ->
    0.205 s per episode); MAPDN_TALL_LINEAR_WIDE=0 keeps the stock backward for weights wider than 128 inputs."""
    if (x.dim() == 2 and x.is_cuda and x.shape[0] >= (1 << 18) and torch.is_grad_enabled() and weight.requires_grad and weight.shape[0] <= 256
            and os.environ.get("MAPDN_TALL_LINEAR", "1") != "0"
            and (weight.shape[1] <= 128 or os.environ.get("MAPDN_TALL_LINEAR_WIDE", "1") != "0")):
        return _TallLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def _tall_ok(x: torch.Tensor, rows: int, weight: torch.Tensor) -> bool:
    return (x.is_cuda and rows >= (1 << 18) and weight.shape[0] <= 256 and weight.shape[1] <= 128 and torch.is_grad_enabled()
            and weight.requires_grad and os.environ.get("MAPDN_TALL_LINEAR", "1") != "0")


_GRU_FUSED_OK: Dict[str, bool] = {}


def _gru_fused_ok(device) -> bool:
    """one-time self-check of the route gru_cell_tall takes (input / hidden projections as _TallLinear, then ATen's fused GRU cell on
    the pre-computed gates) against nn.GRUCell, values and gradients; any exception or mismatch disables the route for the process"""
    key = str(device)
    if key not in _GRU_FUSED_OK:
        ok = False
        try:
            g = torch.Generator(device="cpu").manual_seed(1)
            with torch.random.fork_rng(devices=[]):          # nn.GRUCell's initialisation draws from the GLOBAL generator: leave the caller's stream alone
                cell = nn.GRUCell(8, 8).to(device)
            x = torch.randn(32, 8, generator=g).to(device).requires_grad_(True)
            h = torch.randn(32, 8, generator=g).to(device).requires_grad_(True)
            ref = cell(x, h)
            gr = torch.autograd.grad(ref.square().sum(), [x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh])
            out = torch.ops.aten._thnn_fused_gru_cell(_TallLinear.apply(x, cell.weight_ih, None), _TallLinear.apply(h, cell.weight_hh, None),
                                                      h, cell.bias_ih, cell.bias_hh)[0]
            go = torch.autograd.grad(out.square().sum(), [x, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh])
            ok = bool(torch.allclose(out, ref, rtol=1e-5, atol=1e-6)) and all(bool(torch.allclose(a, b, rtol=1e-4, atol=1e-5)) for a, b in zip(go, gr))
        except Exception:                      # noqa: BLE001 — an ATen signature change must not break training: the stock cell is used
            ok = False
        _GRU_FUSED_OK[key] = ok
    return _GRU_FUSED_OK[key]


def gru_cell_tall(rnn: nn.GRUCell, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """rnn(x, h); on a tall GPU batch under autograd the two projections go through _TallLinear (their weight gradients are products
    with millions of rows of reduction) and ATen's fused cell does the gate arithmetic, as inside nn.GRUCell"""
    if x.dim() == 2 and _tall_ok(x, x.shape[0], rnn.weight_ih) and rnn.bias and _gru_fused_ok(x.device):
        ig = _TallLinear.apply(x, rnn.weight_ih, None)
        hg = _TallLinear.apply(h, rnn.weight_hh, None)
        return torch.ops.aten._thnn_fused_gru_cell(ig, hg, h, rnn.bias_ih, rnn.bias_hh)[0]
    return rnn(x, h)


class _LayerNorm64BC(torch.autograd.Function):
    """relu(LayerNorm(base[b] + per_n[i])) for every (b, i), rows ordered (b, i), WITHOUT materialising the [b, n, 64] sum — the central
    critic's first layer (DDPGNet._value_central) is exactly such a sum: W_obs·obs_all + bias per batch element plus the agent's id
    column.  HIP kernels mapdn_layernorm64_bc_* (csrc/policy.hip): the row is formed in registers (one f32 add, as PyTorch's broadcast
    add would), forward and backward; the backward hands back dx per formed row, reduced here over the agents / over the batch."""

    @staticmethod
    def forward(ctx, base, per_n, weight, bias, eps, relu):
        from . import _lib
        lib = _lib.load()
        b2, p2 = base.contiguous(), per_n.contiguous()
        w, bb = weight.detach().contiguous(), bias.detach().contiguous()
        nb, n = b2.shape[0], p2.shape[0]
        rows = nb * n
        y = torch.empty(rows, 64, dtype=torch.float32, device=base.device)
        mean = torch.empty(rows, dtype=torch.float32, device=base.device)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(base.device):
            _lib.check(lib.mapdn_layernorm64_bc_forward(b2.data_ptr(), p2.data_ptr(), n, w.data_ptr(), bb.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                        rstd.data_ptr(), rows, float(eps), int(relu), torch.cuda.current_stream(base.device).cuda_stream))
        ctx.save_for_backward(b2, p2, w, bb, mean, rstd)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        lib = _lib.load()
        b2, p2, w, bb, mean, rstd = ctx.saved_tensors
        nb, n = b2.shape[0], p2.shape[0]
        rows = nb * n
        dy2 = dy.reshape(rows, 64).contiguous()
        dx = torch.empty_like(dy2)
        dw, db = torch.empty_like(w), torch.empty_like(bb)
        with torch.cuda.device(b2.device):
            partial = torch.empty(lib.mapdn_layernorm64_backward_blocks(rows) * 128, dtype=torch.float32, device=b2.device)
            _lib.check(lib.mapdn_layernorm64_bc_backward(dy2.data_ptr(), b2.data_ptr(), p2.data_ptr(), n, w.data_ptr(), bb.data_ptr(), mean.data_ptr(),
                                                         rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), partial.data_ptr(), rows,
                                                         int(ctx.relu), torch.cuda.current_stream(b2.device).cuda_stream))
        dx3 = dx.view(nb, n, 64)
        dbase = dx3.sum(1) if ctx.needs_input_grad[0] else None
        dpern = dx3.sum(0) if ctx.needs_input_grad[1] else None
        return dbase, dpern, dw, db, None, None


class _ReluDot64(torch.autograd.Function):
    """v = relu(pre) @ w^T + b for pre [rows, 64], w [1, 64]: the critic's activation after fc2 and its one-output fc3 as one pass
    over the pre-activation (mapdn_relu_dot64_*, csrc/policy.hip); relu(pre) is never materialised, forward or backward."""

    @staticmethod
    def forward(ctx, pre, weight, bias):
        from . import _lib
        lib = _lib.load()
        p2 = pre.contiguous()
        w = weight.detach().reshape(64).contiguous()
        rows = p2.shape[0]
        v = torch.empty(rows, 1, dtype=torch.float32, device=pre.device)
        with torch.cuda.device(pre.device):
            _lib.check(lib.mapdn_relu_dot64_forward(p2.data_ptr(), w.data_ptr(), 0.0, v.data_ptr(), rows,
                                                    torch.cuda.current_stream(pre.device).cuda_stream))
        ctx.save_for_backward(p2, w)
        return v + bias.detach()          # (the bias stays a device tensor: no host read of a parameter inside the training loop)

    @staticmethod
    def backward(ctx, dv):
        from . import _lib
        lib = _lib.load()
        p2, w = ctx.saved_tensors
        rows = p2.shape[0]
        dv2 = dv.reshape(rows).contiguous()
        dpre = torch.empty_like(p2)
        dw, db = torch.empty(64, dtype=torch.float32, device=p2.device), torch.empty(64, dtype=torch.float32, device=p2.device)
        with torch.cuda.device(p2.device):
            partial = torch.empty(lib.mapdn_layernorm64_backward_blocks(rows) * 128, dtype=torch.float32, device=p2.device)
            _lib.check(lib.mapdn_relu_dot64_backward(dv2.data_ptr(), p2.data_ptr(), w.data_ptr(), dpre.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                     partial.data_ptr(), rows, torch.cuda.current_stream(p2.device).cuda_stream))
        return dpre, dw.view(1, 64), db[:1].clone()


def relu_dot64_ok(act, fc3: nn.Linear, pre: torch.Tensor) -> bool:
    return (pre.is_cuda and pre.dtype == torch.float32 and pre.dim() == 2 and pre.shape[1] == 64 and pre.shape[0] >= 64 * 1024 and act is F.relu
            and fc3.out_features == 1 and fc3.bias is not None and os.environ.get("MAPDN_FUSED_RELU_DOT", "1") != "0")


class _CriticHead(torch.autograd.Function):
    """v = relu(relu(LayerNorm(x)) W2^T + b2) . w3 + b3 on rows of 64 — everything of the critic behind its first layer
    (critics/mlp_critic.py:22-36) — as ONE HIP launch forward and one backward (libmapdn_hip.so: mapdn_critic_head_*, csrc/critic.hip:
    fp32 MFMA, the 16-row tile stays in registers from the LayerNorm input to v / from dv to dx and the parameter gradients; the backward
    recomputes the forward, so nothing but the inputs is saved).  x is read ([rows, 64], per_n None) or formed as base[b] + per_n[i]
    (rows ordered (b, i): the central critic); the backward then returns dbase / dper_n already summed over the agents / the batch."""

    @staticmethod
    def forward(ctx, x, per_n, ln_w, ln_b, eps, w2, b2, w3, b3):
        from . import _lib
        lib = _lib.load()
        x2 = x.contiguous()
        pn = per_n.contiguous() if per_n is not None else None
        n = pn.shape[0] if pn is not None else 1
        rows = x2.shape[0] * n
        prm = tuple(t.detach().contiguous() for t in (ln_w, ln_b, w2, b2, w3.reshape(64), b3.reshape(1)))
        v = torch.empty(rows, 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.mapdn_critic_head_forward(x2.data_ptr(), pn.data_ptr() if pn is not None else None, n, prm[0].data_ptr(), prm[1].data_ptr(),
                                                     float(eps), prm[2].data_ptr(), prm[3].data_ptr(), prm[4].data_ptr(), prm[5].data_ptr(), v.data_ptr(),
                                                     rows, torch.cuda.current_stream(x.device).cuda_stream))
        ctx.save_for_backward(x2, pn, *prm)
        ctx.eps, ctx.rows, ctx.n = float(eps), rows, n
        return v

    @staticmethod
    def backward(ctx, dv):
        from . import _lib
        lib = _lib.load()
        x2, pn, g, b, w2, b2, w3, b3 = ctx.saved_tensors
        rows, n, formed = ctx.rows, ctx.n, pn is not None
        need = ctx.needs_input_grad
        param_grads = bool(need[2] or need[3] or need[5] or need[6] or need[7] or need[8])
        dv2 = dv.reshape(rows).contiguous()
        dx = torch.empty_like(x2)
        dev = x2.device
        with torch.cuda.device(dev):
            grads = torch.empty(4416 + (n * 64 if formed else 0), dtype=torch.float32, device=dev)
            scratch = torch.empty(max(1, lib.mapdn_critic_head_scratch_floats(rows, n, int(formed))), dtype=torch.float32, device=dev)
            _lib.check(lib.mapdn_critic_head_backward(dv2.data_ptr(), x2.data_ptr(), pn.data_ptr() if formed else None, n, g.data_ptr(), b.data_ptr(),
                                                      ctx.eps, w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), dx.data_ptr(), grads.data_ptr(),
                                                      scratch.data_ptr(), rows, int(param_grads), torch.cuda.current_stream(dev).cuda_stream))
        dpn = grads[4416:].view(n, 64) if formed and need[1] else None
        if not param_grads:
            return dx, dpn, None, None, None, None, None, None, None
        return (dx, dpn, grads[4096:4160], grads[4160:4224], None, grads[:4096].view(64, 64), grads[4224:4288], grads[4288:4352].view(1, 64),
                grads[4352:4353])


class _CriticHeadOwnAction(torch.autograd.Function):
    """The central critic's value as a function of every agent's OWN action only (models/maddpg.py:52-58: the other agents' actions
    enter detached): forward = _CriticHead on base[b] + id_column[i] (the own-action term (a - a.detach()) W_act[:, i] is zero in
    value), backward = d loss / d act[b, i] = dx[b, i, :] . W_act[:, i] straight from the kernel (mapdn_critic_head_backward_dot);
    no [b, n, 64] tensor exists in either direction and the critic's parameters receive no gradient — the policy optimiser does not
    own them (utilities/trainer.py:26-27, 73-98)."""

    @staticmethod
    def forward(ctx, act, base, per_n, dot_w, ln_w, ln_b, eps, w2, b2, w3, b3):
        from . import _lib
        lib = _lib.load()
        x2, pn, dw = base.detach().contiguous(), per_n.detach().contiguous(), dot_w.detach().contiguous()
        n = pn.shape[0]
        rows = x2.shape[0] * n
        prm = tuple(t.detach().contiguous() for t in (ln_w, ln_b, w2, b2, w3.reshape(64), b3.reshape(1)))
        v = torch.empty(rows, 1, dtype=torch.float32, device=base.device)
        with torch.cuda.device(base.device):
            _lib.check(lib.mapdn_critic_head_forward(x2.data_ptr(), pn.data_ptr(), n, prm[0].data_ptr(), prm[1].data_ptr(), float(eps), prm[2].data_ptr(),
                                                     prm[3].data_ptr(), prm[4].data_ptr(), prm[5].data_ptr(), v.data_ptr(), rows,
                                                     torch.cuda.current_stream(base.device).cuda_stream))
        ctx.save_for_backward(x2, pn, dw, *prm)
        ctx.eps, ctx.rows, ctx.n, ctx.act_shape = float(eps), rows, n, act.shape
        return v

    @staticmethod
    def backward(ctx, dv):
        from . import _lib
        lib = _lib.load()
        x2, pn, dw, g, b, w2, b2, w3, b3 = ctx.saved_tensors
        rows, n, dev = ctx.rows, ctx.n, x2.device
        dv2 = dv.reshape(rows).contiguous()
        dact = torch.empty(rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mapdn_critic_head_backward_dot(dv2.data_ptr(), x2.data_ptr(), pn.data_ptr(), n, g.data_ptr(), b.data_ptr(), ctx.eps,
                                                          w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), dw.data_ptr(), dact.data_ptr(), rows,
                                                          torch.cuda.current_stream(dev).cuda_stream))
        return (dact.view(ctx.act_shape),) + (None,) * 10


class _CriticHeadMSE(torch.autograd.Function):
    """The value loss of the DDPG family (learning_algorithms/ddpg.py:36-38 == models/maddpg.py:122-124),
        loss = sum_rows w[row] (returns[row] - v[row])^2,    w[row] = scale * wrow[row / n]   (wrow None: 1),
    with v the critic head of _CriticHead — as ONE launch that produces the loss AND every gradient of it (mapdn_critic_head_mse):
    d loss / d v = -2 w (returns - v) is known per row the moment v is, so the separate forward launch (a fifth 64 x 64 product over
    all rows) disappears.  backward only scales the stored gradients by the incoming one (1 for `loss.backward()`)."""

    @staticmethod
    def forward(ctx, x, per_n, ln_w, ln_b, eps, w2, b2, w3, b3, returns, wrow, scale):
        from . import _lib
        lib = _lib.load()
        x2 = x.detach().contiguous()
        pn = per_n.detach().contiguous() if per_n is not None else None
        n = pn.shape[0] if pn is not None else 1
        rows, formed, dev = x2.shape[0] * n, pn is not None, x2.device
        prm = tuple(t.detach().contiguous() for t in (ln_w, ln_b, w2, b2, w3.reshape(64), b3.reshape(1)))
        ret = returns.detach().reshape(rows).contiguous().float()
        wr = wrow.detach().contiguous().float() if wrow is not None else None
        sc = scale.detach().reshape(1).contiguous().float()
        dx = torch.empty_like(x2)
        with torch.cuda.device(dev):
            grads = torch.empty(4416 + (n * 64 if formed else 0), dtype=torch.float32, device=dev)
            scratch = torch.empty(max(1, lib.mapdn_critic_head_scratch_floats(rows, n, int(formed))), dtype=torch.float32, device=dev)
            _lib.check(lib.mapdn_critic_head_mse(ret.data_ptr(), wr.data_ptr() if wr is not None else None, sc.data_ptr(), x2.data_ptr(),
                                                 pn.data_ptr() if formed else None, n, prm[0].data_ptr(), prm[1].data_ptr(), float(eps), prm[2].data_ptr(),
                                                 prm[3].data_ptr(), prm[4].data_ptr(), prm[5].data_ptr(), dx.data_ptr(), grads.data_ptr(), scratch.data_ptr(),
                                                 rows, torch.cuda.current_stream(dev).cuda_stream))
        ctx.save_for_backward(dx, grads)
        ctx.n, ctx.formed = n, formed
        return grads[4353].clone()

    @staticmethod
    def backward(ctx, g):
        dx, grads = ctx.saved_tensors
        n, need = ctx.n, ctx.needs_input_grad
        gs = grads * g
        return (dx * g if need[0] else None, gs[4416:].view(n, 64) if ctx.formed and need[1] else None, gs[4096:4160], gs[4160:4224], None,
                gs[:4096].view(64, 64), gs[4224:4288], gs[4288:4352].view(1, 64), gs[4352:4353], None, None, None)


HEAD_MAX_FORMED_N = 88      # formed rows keep one [n][64] LDS accumulator per wavefront: n <= 88 fits the 160 KB of a CU with four of them


def critic_head_ok(cr: "MLPCritic", x: torch.Tensor, rows: int, formed_n: int = 0) -> bool:
    """the one-launch critic head covers the reference's default critic (LayerNorm, ReLU, hidden size 64, one output) in fp32 on the GPU
    (formed_n: the number of agents when the rows are formed as base[b] + per_n[i])"""
    return (formed_n <= HEAD_MAX_FORMED_N and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[-1] == 64 and rows >= 1024 and rows < 2 ** 31 and cr.use_ln
            and cr.act is F.relu and cr.layernorm.elementwise_affine and cr.layernorm.bias is not None and cr.fc2.in_features == 64
            and cr.fc2.out_features == 64 and cr.fc2.bias is not None and cr.fc3.out_features == 1 and cr.fc3.bias is not None
            and cr.fc2.weight.dtype == torch.float32 and os.environ.get("MAPDN_FUSED_HEAD", "1") != "0")


def critic_head(cr: "MLPCritic", x: torch.Tensor, per_n: Optional[torch.Tensor] = None) -> torch.Tensor:
    ln = cr.layernorm
    return _CriticHead.apply(x, per_n, ln.weight, ln.bias, ln.eps, cr.fc2.weight, cr.fc2.bias, cr.fc3.weight, cr.fc3.bias)


def layernorm_act_bc(ln: nn.LayerNorm, act, base: torch.Tensor, per_n: torch.Tensor):
    """act(LayerNorm(base.unsqueeze(1) + per_n.unsqueeze(0))).reshape(b * n, 64) through the broadcast-input kernels, or None when the
    case is not theirs (the caller then forms the sum and takes layernorm_act)"""
    if (base.is_cuda and base.dtype == torch.float32 and per_n.dtype == torch.float32 and base.dim() == 2 and per_n.dim() == 2
            and base.shape[-1] == 64 and per_n.shape[-1] == 64 and act is F.relu and ln.elementwise_affine and ln.bias is not None
            and ln.weight.dtype == torch.float32 and base.shape[0] * per_n.shape[0] >= 1024
            and os.environ.get("MAPDN_FUSED_LN", "1") != "0" and os.environ.get("MAPDN_FUSED_LN_BC", "1") != "0"):
        return _LayerNorm64BC.apply(base, per_n, ln.weight, ln.bias, ln.eps, True)
    return None


def layernorm_act(ln: nn.LayerNorm, act, x: torch.Tensor) -> torch.Tensor:
    """act(LayerNorm(x)): one HIP launch each way for the reference's default shape (64 features, ReLU, fp32, on the GPU), the
    PyTorch modules otherwise (MAPDN_FUSED_LN=0 forces them)."""
    if (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 64 and act is F.relu and ln.elementwise_affine and ln.bias is not None
            and ln.weight.dtype == torch.float32 and x.numel() >= 64 * 1024 and os.environ.get("MAPDN_FUSED_LN", "1") != "0"):
        return _LayerNorm64.apply(x, ln.weight, ln.bias, ln.eps, True)
    return act(ln(x))


def _splitk_dw(dy2: torch.Tensor, x2: torch.Tensor, blk: int = 16384) -> torch.Tensor:
    """dy2^T x2 ([out, in]) for [rows, out] x [rows, in] with millions of rows: the rows cut into blocks whose partial products are one
    batched GEMM, then summed (what _TallLinear.backward does for its weight).  dy2 may be a column slice of a wider tensor."""
    rows = x2.shape[0]
    full = rows // blk * blk
    dw = None
    if full:
        a = dy2[:full].unflatten(0, (-1, blk)).transpose(1, 2)
        dw = torch.bmm(a, x2[:full].unflatten(0, (-1, blk))).sum(0)
    if full < rows:
        rest = dy2[full:].t() @ x2[full:]
        dw = rest if dw is None else dw + rest
    return dw


class _PolicyTrunk(torch.autograd.Function):
    """means = fc2(GRUCell(relu(LayerNorm(fc1(obs) + id column)), last_hid)) of the shared recurrent agent (agents/rnn_agent.py:5-32) for the
    policy UPDATE (models/maddpg.py:103-125: only the means carry gradient — the loss does not read the new hidden state), forward and
    backward as HIP launches: mapdn_policy_forward_train (the rollout's one-launch forward; keeps x1 = the LayerNorm input) and
    mapdn_policy_backward (csrc/policy_bwd.hip: recomputes the trunk, writes d gate pre-activations / xn / dx1 and the small gradients).
    What is left for the BLAS back end are the three K = rows weight-gradient products, run block-wise (_splitk_dw).  The PyTorch route
    moved ~100 GB per update through the projections, ATen's fused cell and its workspace."""

    @staticmethod
    def forward(ctx, obs2, hid2, n_agents, ids, eps, w1, b1, lnw, lnb, w_ih, w_hh, b_ih, b_hh, w2, b2):
        from . import _lib
        lib = _lib.load()
        obs2, hid2 = obs2.contiguous(), hid2.contiguous()
        rows, o = obs2.shape
        prm = tuple(t.detach().contiguous() for t in (w1, b1, lnw, lnb, w_ih, w_hh, b_ih, b_hh, w2, b2))
        means = torch.empty(rows, dtype=torch.float32, device=obs2.device)
        x1 = torch.empty(rows, 64, dtype=torch.float32, device=obs2.device)
        with torch.cuda.device(obs2.device):
            _lib.check(lib.mapdn_policy_forward_train(obs2.data_ptr(), hid2.data_ptr(), *(t.data_ptr() for t in prm), means.data_ptr(), None,
                                                      x1.data_ptr(), rows, int(n_agents), o, int(ids), float(eps),
                                                      torch.cuda.current_stream(obs2.device).cuda_stream))
        ctx.save_for_backward(obs2, hid2, x1, *prm)
        ctx.n_agents, ctx.ids, ctx.eps = int(n_agents), int(ids), float(eps)
        return means

    @staticmethod
    def backward(ctx, dmeans):
        from . import _lib
        lib = _lib.load()
        obs2, hid2, x1, w1, b1, lnw, lnb, w_ih, w_hh, b_ih, b_hh, w2, b2 = ctx.saved_tensors
        rows, o = obs2.shape
        n, dev = ctx.n_agents, obs2.device
        dm = dmeans.reshape(rows).contiguous()
        dx1 = torch.empty_like(x1)
        xn = torch.empty_like(x1)
        dg = torch.empty(rows, 256, dtype=torch.float32, device=dev)
        small = torch.empty(512, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            scratch = torch.empty(max(1, lib.mapdn_policy_backward_scratch_floats(rows)), dtype=torch.float32, device=dev)
            _lib.check(lib.mapdn_policy_backward(dm.data_ptr(), x1.data_ptr(), hid2.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), ctx.eps, w_ih.data_ptr(),
                                                 w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), w2.data_ptr(), dx1.data_ptr(), dg.data_ptr(), xn.data_ptr(),
                                                 small.data_ptr(), scratch.data_ptr(), rows, torch.cuda.current_stream(dev).cuda_stream))
        dw_ih = _splitk_dw(dg[:, :192], xn)
        dw_hh = torch.cat((_splitk_dw(dg[:, :128], hid2), _splitk_dw(dg[:, 192:], hid2)), 0)
        db_ih = small[:192].clone()
        db_hh = torch.cat((small[:128], small[192:256]))
        dw1 = torch.empty_like(w1)
        dw1[:, :o] = _splitk_dw(dx1, obs2)
        per_agent = dx1.view(-1, n, 64).sum(0)                   # [n, 64]: sum over the batch for every agent
        if ctx.ids:
            dw1[:, o:] = per_agent.t()
        db1 = per_agent.sum(0)
        return (None, None, None, None, None, dw1, db1, small[256:320], small[320:384], dw_ih, dw_hh, db_ih, db_hh, small[384:448].view(1, 64),
                small[448:449])


class RNNAgent(nn.Module):
    """fc1 -> LayerNorm -> act -> GRUCell -> fc2 (agents/rnn_agent.py:5-32)."""

    def __init__(self, input_shape: int, args):
        super().__init__()
        self.hid_size = args.hid_size
        self.fc1 = nn.Linear(input_shape, args.hid_size)
        if args.layernorm:
            self.layernorm = nn.LayerNorm(args.hid_size)
        self.rnn = nn.GRUCell(args.hid_size, args.hid_size)
        self.fc2 = nn.Linear(args.hid_size, args.action_dim)
        self.use_ln = bool(args.layernorm)
        self.act = _activation(args.hid_activation)

    def trunk(self, x: torch.Tensor, hidden: torch.Tensor):
        """x: pre-activation of fc1, [rows, hid]"""
        x = layernorm_act(self.layernorm, self.act, x) if self.use_ln else self.act(x)
        h = gru_cell_tall(self.rnn, x, hidden.reshape(-1, self.hid_size))
        return tall_linear(self.fc2, h), h

    def forward(self, inputs, hidden):
        a, h = self.trunk(self.fc1(inputs), hidden)
        return a, None, h


class MLPCritic(nn.Module):
    """fc1 -> LayerNorm -> act -> fc2 -> act -> fc3 (critics/mlp_critic.py:7-36)."""

    def __init__(self, input_shape: int, output_shape: int, args):
        super().__init__()
        self.fc1 = nn.Linear(input_shape, args.hid_size)
        if args.layernorm:
            self.layernorm = nn.LayerNorm(args.hid_size)
        self.fc2 = nn.Linear(args.hid_size, args.hid_size)
        self.fc3 = nn.Linear(args.hid_size, output_shape)
        self.use_ln = bool(args.layernorm)
        self.act = _activation(args.hid_activation)

    def trunk(self, x: torch.Tensor):
        if critic_head_ok(self, x, x.shape[0]):
            return critic_head(self, x), None                      # (no caller uses the hidden activation)
        x = layernorm_act(self.layernorm, self.act, x) if self.use_ln else self.act(x)
        return self.head(x)

    def head(self, x: torch.Tensor):
        """what follows the first layer's LayerNorm + activation: fc2 -> act -> fc3"""
        pre = tall_linear(self.fc2, x)
        if relu_dot64_ok(self.act, self.fc3, pre):
            return _ReluDot64.apply(pre, self.fc3.weight, self.fc3.bias), None      # (no caller uses the hidden activation)
        h = self.act(pre)
        return tall_linear(self.fc3, h), h

    def forward(self, inputs, hidden=None):
        return self.trunk(self.fc1(inputs))


class DDPGNet(nn.Module):
    """`MADDPG(Model)` (models/maddpg.py:10) or `IDDPG(Model)` (models/iddpg.py:9): behaviour net holding
    its target net, the per-agent reward BatchNorm (models/model.py:26) and the DDPG loss."""

    def __init__(self, args, alg: str = "maddpg", target_net: Optional["DDPGNet"] = None):
        super().__init__()
        if alg not in ("maddpg", "iddpg"):
            raise KeyError(alg)                                    # models/model_registry.py:14-25
        self.args, self.alg = args, alg
        self.n_, self.obs_dim, self.act_dim, self.hid_dim = args.agent_num, args.obs_size, args.action_dim, args.hid_size
        self._fused_fits = None                   # does the one-launch HIP policy forward have a launch shape for this obs width?
        n, o, a = self.n_, self.obs_dim, self.act_dim
        ids = n if args.agent_id else 0
        self.batchnorm = nn.BatchNorm1d(n)
        # advantage normalisation: MADDPG re-uses `batchnorm` (maddpg.py:17,120); IDDPG's lives in its DDPG
        # helper object, which is not an nn.Module — so it is NOT part of the state_dict (ddpg.py:10,34)
        self.__dict__["_adv_batchnorm"] = self.batchnorm if alg == "maddpg" else nn.BatchNorm1d(n)
        critic_in = (o + a) * n + ids if alg == "maddpg" else o + a + ids     # maddpg.py:20-24, iddpg.py:19-23
        copies = 1 if args.shared_params else n
        self.value_dicts = nn.ModuleList([MLPCritic(critic_in, 1, args) for _ in range(copies)])
        self.policy_dicts = nn.ModuleList([RNNAgent(o + ids, args) for _ in range(copies)])   # model.py:141-164
        self.apply(self._init_weights)
        if target_net is not None:
            self.target_net = target_net
            self.reload_params_to_target()

    # ---- initialisation / target handling (models/model.py:28-48,169-177) ----------------------
    def _init_weights(self, m):
        if type(m) == nn.Linear:
            if self.args.init_type == "normal":
                nn.init.normal_(m.weight, 0.0, self.args.init_std)
            elif self.args.init_type == "orthogonal":
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain(self.args.hid_activation))

    def reload_params_to_target(self):
        self.target_net.policy_dicts.load_state_dict(self.policy_dicts.state_dict())
        self.target_net.value_dicts.load_state_dict(self.value_dicts.state_dict())

    @torch.no_grad()
    def update_target(self):
        """soft update over every state_dict entry of the policy and value nets (model.py:34-48)"""
        lr = self.args.target_lr
        for mine, theirs in ((self.policy_dicts, self.target_net.policy_dicts), (self.value_dicts, self.target_net.value_dicts)):
            src = mine.state_dict()
            for name, param in theirs.state_dict().items():
                param.copy_((1 - lr) * param + lr * src[name])

    def init_hidden(self, batch: int = 1):
        return self.policy_dicts[0].fc1.weight.new_zeros(batch, self.n_, self.hid_dim)    # rnn_agent.py:22-24

    # ---- policy (models/model.py:101-139) --------------------------------------------------------
    def _fused_policy_ok(self, obs: torch.Tensor, last_hid: torch.Tensor) -> bool:
        """the one-launch HIP forward (libmapdn_hip.so: mapdn_policy_forward) covers the reference's default agent — shared
        parameters, LayerNorm, ReLU, hidden size 64, one action — for inference on the GPU in fp32"""
        a = self.args
        if not (not torch.is_grad_enabled() and obs.is_cuda and obs.dtype == torch.float32 and last_hid.dtype == torch.float32
                and a.shared_params and a.layernorm and a.hid_activation == "relu" and self.hid_dim == 64 and self.act_dim == 1
                and os.environ.get("MAPDN_FUSED_POLICY", "1") != "0"):
            return False
        if self._fused_fits is None:              # the parameter set + activations of one tile must fit a CU's LDS (wide observations,
            from . import _lib                    # e.g. history > 1 on the 322-bus net, do not): those keep the PyTorch modules
            self._fused_fits = bool(_lib.load().mapdn_policy_forward_fits(self.obs_dim, self.n_ if a.agent_id else 0))
        return self._fused_fits

    def _fused_policy(self, obs: torch.Tensor, last_hid: torch.Tensor, need_hidden: bool = True):
        from . import _lib
        ag = self.policy_dicts[0]
        b, n, o = obs.shape[0], self.n_, self.obs_dim
        obs_c, hid_c = obs.contiguous(), last_hid.contiguous()
        means = torch.empty(b, n, 1, dtype=torch.float32, device=obs.device)
        hid = torch.empty(b, n, self.hid_dim, dtype=torch.float32, device=obs.device) if need_hidden else None
        keep = []                                  # contiguous views stay referenced until the launch is enqueued

        def P(t):
            keep.append(t.detach().contiguous())
            return keep[-1].data_ptr()
        ids = n if self.args.agent_id else 0
        with torch.cuda.device(obs.device):
            _lib.check(_lib.load().mapdn_policy_forward(
                obs_c.data_ptr(), hid_c.data_ptr(), P(ag.fc1.weight), P(ag.fc1.bias), P(ag.layernorm.weight), P(ag.layernorm.bias),
                P(ag.rnn.weight_ih), P(ag.rnn.weight_hh), P(ag.rnn.bias_ih), P(ag.rnn.bias_hh), P(ag.fc2.weight), P(ag.fc2.bias),
                means.data_ptr(), hid.data_ptr() if hid is not None else None, b * n, n, o, ids, float(ag.layernorm.eps),
                torch.cuda.current_stream(obs.device).cuda_stream))
        return means, torch.full_like(means, math.log(self.args.fixed_policy_std)), hid

    def _fused_policy_train_ok(self, obs: torch.Tensor, last_hid: torch.Tensor) -> bool:
        a = self.args
        if not (torch.is_grad_enabled() and obs.is_cuda and obs.dtype == torch.float32 and last_hid.dtype == torch.float32 and a.shared_params
                and a.layernorm and a.hid_activation == "relu" and self.hid_dim == 64 and self.act_dim == 1 and not last_hid.requires_grad
                and not obs.requires_grad and obs.shape[0] * self.n_ >= 4096 and os.environ.get("MAPDN_FUSED_POLICY_TRAIN", "1") != "0"):
            return False
        if self._fused_fits is None:
            from . import _lib
            self._fused_fits = bool(_lib.load().mapdn_policy_forward_fits(self.obs_dim, self.n_ if a.agent_id else 0))
        return self._fused_fits

    def policy(self, obs: torch.Tensor, last_hid: torch.Tensor, means_grad_only: bool = False):
        """obs [b, n, o], last_hid [b, n, h] -> means [b, n, a], log_stds, hiddens [b, n, h].  means_grad_only: the caller differentiates the
        means alone (the policy loss; the new hidden state is then not returned) — the HIP forward / backward pair of _PolicyTrunk."""
        b, n, o = obs.shape[0], self.n_, self.obs_dim
        if self._fused_policy_ok(obs, last_hid):
            return self._fused_policy(obs, last_hid, need_hidden=not means_grad_only)
        if means_grad_only and self._fused_policy_train_ok(obs, last_hid):
            ag = self.policy_dicts[0]
            means = _PolicyTrunk.apply(obs.reshape(b * n, o), last_hid.reshape(b * n, -1), n, n if self.args.agent_id else 0, ag.layernorm.eps,
                                       ag.fc1.weight, ag.fc1.bias, ag.layernorm.weight, ag.layernorm.bias, ag.rnn.weight_ih, ag.rnn.weight_hh,
                                       ag.rnn.bias_ih, ag.rnn.bias_hh, ag.fc2.weight, ag.fc2.bias).view(b, n, 1)
            return means, torch.full_like(means, math.log(self.args.fixed_policy_std)), None
        if self.args.shared_params:
            ag = self.policy_dicts[0]
            w = ag.fc1.weight
            if _tall_ok(obs, b * n, w):                                  # (the policy update on a replay batch: weight gradient over b * n rows)
                x = _TallLinear.apply(obs.reshape(b * n, o), w[:, :o], ag.fc1.bias).view(b, n, -1)
            else:
                x = F.linear(obs, w[:, :o], ag.fc1.bias)                 # observation columns
            if self.args.agent_id:
                x = x + w[:, o:].t().unsqueeze(0)                        # one-hot id i selects column o + i
            means, hid = ag.trunk(x.reshape(b * n, -1), last_hid)
            means, hid = means.view(b, n, -1), hid.view(b, n, -1)
            log_stds = torch.full_like(means, math.log(self.args.fixed_policy_std))       # model.py:119-120
        else:
            ms, hs = [], []
            for i, ag in enumerate(self.policy_dicts):
                w = ag.fc1.weight
                x = F.linear(obs[:, i], w[:, :o], ag.fc1.bias)
                if self.args.agent_id:
                    x = x + w[:, o + i]
                m, h = ag.trunk(x, last_hid[:, i])
                ms.append(m); hs.append(h)
            means, hid = torch.stack(ms, 1), torch.stack(hs, 1)
            log_stds = torch.zeros_like(means)                            # model.py:134-135
        return means, log_stds, hid

    # ---- critics ---------------------------------------------------------------------------------
    def value(self, obs: torch.Tensor, act: torch.Tensor, own_action_only: bool = False) -> torch.Tensor:
        """obs [b, n, o], act [b, n, a] -> [b, n, 1].  own_action_only: the caller differentiates with respect to `act` alone (the policy
        loss): the central critic may then skip the gradients of its own parameters, which the policy optimiser never reads."""
        return self._value_central(obs, act, own_action_only) if self.alg == "maddpg" else self._value_independent(obs, act)

    def _independent_first_layer(self, cr, obs, act):
        """IDDPG's shared critic, first layer on [obs_i | id_i | act_i] (iddpg.py:32-58) as  W_obs obs + id column + W_act act, [b, n, h];
        on a tall GPU batch the observation product takes _TallLinear (its weight gradient is a [h, o] product with b * n rows of
        reduction, which the BLAS back end runs on two workgroups)"""
        b, n, o = obs.shape[0], self.n_, self.obs_dim
        ids = n if self.args.agent_id else 0
        w = cr.fc1.weight
        if _tall_ok(obs, b * n, w):
            x = _TallLinear.apply(obs.reshape(b * n, o), w[:, :o], cr.fc1.bias)
            if self.act_dim == 1:        # a one-column product is an outer product: one fused multiply-add pass instead of a GEMM + an add
                x = torch.addcmul(x, act.reshape(b * n, 1), w[:, o + ids:].t()).view(b, n, -1)
            else:
                x = (x + _TallLinear.apply(act.reshape(b * n, -1), w[:, o + ids:], None)).view(b, n, -1)
        else:
            x = F.linear(obs, w[:, :o], cr.fc1.bias) + F.linear(act, w[:, o + ids:])
        if ids:
            x = x + w[:, o:o + n].t().unsqueeze(0)
        return x

    def _value_independent(self, obs, act):
        """IDDPG (iddpg.py:32-58): critic input [obs_i | id_i | act_i]"""
        b, n, o = obs.shape[0], self.n_, self.obs_dim
        ids = n if self.args.agent_id else 0

        def first_layer(cr, ob, ac, who):
            w = cr.fc1.weight
            x = F.linear(ob, w[:, :o], cr.fc1.bias) + F.linear(ac, w[:, o + ids:])
            if ids:
                x = x + (w[:, o:o + n].t().unsqueeze(0) if who is None else w[:, o + who])
            return x

        if self.args.shared_params:
            cr = self.value_dicts[0]
            v, _ = cr.trunk(self._independent_first_layer(cr, obs, act).reshape(b * n, -1))
            return v.view(b, n, -1)
        return torch.stack([cr.trunk(first_layer(cr, obs[:, i], act[:, i], i))[0] for i, cr in enumerate(self.value_dicts)], 1)

    def _value_central(self, obs, act, own_action_only=False):
        """MADDPG (maddpg.py:35-79): agent i's critic sees [all obs | id_i | all actions] and only its OWN
        action carries gradient.  First layer = W_obs·obs_all + W_id[:, i] + W_act·act_all, where the joint
        action enters detached and agent i's own (act_i - act_i.detach()) — zero in value — restores its
        gradient path; no [b, n, n·o] tensor is ever built."""
        b, n, o, a = obs.shape[0], self.n_, self.obs_dim, self.act_dim
        ids = n if self.args.agent_id else 0
        obs_all, act_all = obs.reshape(b, n * o), act.reshape(b, n * a)
        own = (act - act.detach()) if act.requires_grad else None

        def first_layer(cr, who):
            w = cr.fc1.weight
            w_act = w[:, n * o + ids:]
            base = F.linear(obs_all, w[:, :n * o], cr.fc1.bias) + F.linear(act_all.detach(), w_act)      # [b, h]
            if who is None:                                              # all agents at once: [b, n, h]
                x = base.unsqueeze(1)
                if ids:
                    x = x + w[:, n * o:n * o + n].t().unsqueeze(0)
                else:
                    x = x.expand(b, n, -1)
                if own is not None:
                    x = x + torch.einsum("bna,hna->bnh", own, w_act.reshape(-1, n, a))
                return x
            x = base + w[:, n * o + who] if ids else base
            if own is not None:
                x = x + F.linear(own[:, who], w_act[:, who * a:(who + 1) * a])
            return x

        if self.args.shared_params:
            cr = self.value_dicts[0]
            if ids and cr.use_ln and own is not None and a == 1 and own_action_only and critic_head_ok(cr, obs_all.new_empty(0, 64), b * n, n):
                # the policy update (maddpg.py:52-58, 103-125): value = the head on base[b] + id_column[i] (the own-action term is zero in
                # value), gradient = d/d act[b, i] only, from the kernel; the critic's own parameters are not differentiated
                w = cr.fc1.weight.detach()
                with torch.no_grad():
                    base = F.linear(obs_all, w[:, :n * o], cr.fc1.bias.detach()) + F.linear(act_all.detach(), w[:, n * o + ids:])
                ln = cr.layernorm
                v = _CriticHeadOwnAction.apply(act, base, w[:, n * o:n * o + n].t(), w[:, n * o + ids:].t(), ln.weight, ln.bias, ln.eps,
                                               cr.fc2.weight, cr.fc2.bias, cr.fc3.weight, cr.fc3.bias)
                return v.view(b, n, 1)
            if ids and own is None and cr.use_ln:
                # no gradient path through the actions (value loss, target values): the first layer's output is base[b] + id_column[i] —
                # LayerNorm + ReLU straight from the two small operands, the [b, n, h] sum is never written
                w = cr.fc1.weight
                base = tall_linear_w(obs_all, w[:, :n * o], cr.fc1.bias) + tall_linear_w(act_all.detach(), w[:, n * o + ids:])
                if critic_head_ok(cr, base, b * n, n):
                    return critic_head(cr, base, w[:, n * o:n * o + n].t()).view(b, n, 1)
                xn = layernorm_act_bc(cr.layernorm, cr.act, base, w[:, n * o:n * o + n].t())
                if xn is not None:
                    return cr.head(xn)[0].view(b, n, 1)
            v, _ = cr.trunk(first_layer(cr, None).reshape(b * n, -1))
            return v.view(b, n, 1)
        return torch.stack([cr.trunk(first_layer(cr, i))[0] for i, cr in enumerate(self.value_dicts)], 1)

    def value_mse(self, obs, act, returns, valid=None):
        """mean over (batch, agent) of (returns - Q(obs, act))^2 — the value loss of maddpg.py:122-124 / ddpg.py:36-38 — weighted by
        `valid` [b] when given (1 everywhere = the reference).  With the default critic on the GPU the loss and its gradients come out of
        ONE head launch (_CriticHeadMSE); otherwise value() + the plain expression."""
        b, n, o = obs.shape[0], self.n_, self.obs_dim
        a = self.args
        ids = n if a.agent_id else 0
        if (a.shared_params and torch.is_grad_enabled() and not act.requires_grad and obs.is_cuda
                and os.environ.get("MAPDN_FUSED_MSE", "1") != "0"):
            cr = self.value_dicts[0]
            w = cr.fc1.weight
            x = per_n = None
            if self.alg == "maddpg" and ids and cr.use_ln:
                x = tall_linear_w(obs.reshape(b, n * o), w[:, :n * o], cr.fc1.bias) + tall_linear_w(act.reshape(b, n * self.act_dim), w[:, n * o + ids:])
                per_n = w[:, n * o:n * o + n].t()
            elif self.alg == "iddpg":
                x = self._independent_first_layer(cr, obs, act).reshape(b * n, -1)
            if x is not None and critic_head_ok(cr, x, b * n, n if per_n is not None else 0):
                if valid is None:
                    scale, wrow = x.new_full((1,), 1.0 / (b * n)), None
                else:
                    vf = valid.float().view(-1)
                    scale = (1.0 / (vf.sum().clamp(min=1.0) * n)).reshape(1)
                    wrow = vf if per_n is not None else vf.repeat_interleave(n)
                ln = cr.layernorm
                return _CriticHeadMSE.apply(x, per_n, ln.weight, ln.bias, ln.eps, cr.fc2.weight, cr.fc2.bias, cr.fc3.weight, cr.fc3.bias,
                                            returns, wrow, scale)
        values = self.value(obs, act).view(-1, n)
        d2 = (returns - values).pow(2)
        return d2.mean() if valid is None else (d2 * valid.float().view(-1, 1)).sum() / (valid.float().sum().clamp(min=1.0) * d2.shape[1])

    # ---- action selection (maddpg.py:81-101 == iddpg.py:60-80; utilities/util.py:52-98) ----------
    def get_actions(self, state, status, exploration, actions_avail, target=False, last_hid=None, means_grad_only=False):
        net = self.target_net if (target and self.args.target) else self
        means, log_stds, hiddens = net.policy(state, last_hid, means_grad_only)
        if means.size(-1) > 1:                                   # maddpg.py:85-87 (action_dim > 1 sums over agents)
            means_, log_stds_ = means.sum(dim=1, keepdim=True), log_stds.sum(dim=1, keepdim=True)
        else:
            means_, log_stds_ = means, log_stds
        actions, log_prob = self._select_action(means_, log_stds_, status, exploration)
        restore_mask = 1.0 - (actions_avail == 0).to(actions.dtype)
        return actions, restore_mask * actions, log_prob, (means, log_stds), hiddens

    def _select_action(self, mean, log_std, status, exploration):
        a = self.args
        if status == "train":
            if not exploration:
                return mean, None
            std = log_std.exp()
            if a.action_enforcebound:                            # util.py:57-66
                x_t = mean + std * torch.randn_like(mean)        # Normal(mean, std).rsample()
                y_t = torch.tanh(x_t)
                log_prob = -((x_t - mean) ** 2) / (2 * std ** 2) - log_std - math.log(math.sqrt(2 * math.pi))
                return y_t, log_prob - torch.log(1 - y_t.pow(2) + 1e-6)
            x_t = std * torch.randn_like(mean)                   # util.py:67-76
            log_prob = -(x_t ** 2) / (2 * std ** 2) - log_std - math.log(math.sqrt(2 * math.pi))
            return mean + x_t, log_prob
        if status == "test":                                     # util.py:80-87
            return (torch.tanh(mean) if a.action_enforcebound else mean), None
        raise ValueError(status)

    # ---- loss (models/maddpg.py:103-125 == learning_algorithms/ddpg.py:15-39) ----------------------
    def normalise_reward(self, reward: torch.Tensor) -> torch.Tensor:
        """model.py:316-317: BatchNorm1d over the batch, per agent.  Data-parallel ranks (PGTrainer sets `_dp_all_reduce`) normalise with
        the statistics of the UNION of their batches — one small all-reduce of (count, sum, sum of squares) — so that an N-rank update
        equals the one-rank update on the concatenated batch and the running statistics stay identical on every replica."""
        if not self.args.reward_normalisation:
            return reward
        bn, red = self.batchnorm, self.__dict__.get("_dp_all_reduce")
        if red is None or not bn.training:
            return bn(reward)
        x = reward.double()
        st = torch.cat((x.new_full((1,), float(x.shape[0])), x.sum(0), (x * x).sum(0)))
        red(st)
        cnt, n = st[0], x.shape[1]
        mean = st[1:1 + n] / cnt
        var = (st[1 + n:] / cnt - mean * mean).clamp(min=0.0)                      # biased, as BatchNorm normalises with
        with torch.no_grad():
            m = bn.momentum if bn.momentum is not None else 0.1
            bn.running_mean.mul_(1 - m).add_(m * mean.to(bn.running_mean.dtype))
            bn.running_var.mul_(1 - m).add_(m * (var * cnt / (cnt - 1).clamp(min=1.0)).to(bn.running_var.dtype))
            bn.num_batches_tracked += 1
        y = (reward - mean.to(reward.dtype)) * torch.rsqrt(var.to(reward.dtype) + bn.eps)
        return y * bn.weight + bn.bias

    def get_loss(self, batch: Batch, want=("policy", "value")):
        """batch: state/next_state [bs, n, o], action/action_avail [bs, n, a], reward [bs, n], done [bs, 1],
        last_hid/hid [bs, n, h] float32 (the `unpack_data` tensors, model.py:304-319); optional `valid`
        [bs] weights the means (1 everywhere = the reference).  Returns (policy_loss, value_loss,
        (means, log_stds)); a loss not in `want` is None and its forward passes are skipped."""
        n = self.n_
        state, actions, next_state = batch["state"], batch["action"], batch["next_state"]
        avail, last_hid, hid = batch["action_avail"], batch["last_hid"], batch["hid"]
        rewards = self.normalise_reward(batch["reward"].float())
        done = batch["done"].float().view(-1, 1)
        valid = batch.get("valid")
        wmean = (lambda t: t.mean()) if valid is None else \
            (lambda t: (t * valid.float().view(-1, 1)).sum() / (valid.float().sum().clamp(min=1.0) * t.shape[1]))
        policy_loss = value_loss = action_out = None
        if "policy" in want:
            _, actions_pol, _, action_out, _ = self.get_actions(state, "train", False, avail, False, last_hid, means_grad_only=True)
            advantages = self.value(state, actions_pol, own_action_only=True).view(-1, n)
            if self.args.normalize_advantages:
                advantages = self._adv_batchnorm.to(advantages.device)(advantages)
            policy_loss = wmean(-advantages)
        if "value" in want:
            with torch.no_grad():
                if "next_value_cached" in batch:                 # PGTrainer precomputed both for the whole replay ring (same values)
                    next_values = batch["next_value_cached"].view(-1, n)
                else:
                    if "next_action_cached" in batch:
                        next_actions = batch["next_action_cached"]
                    else:
                        _, next_actions, _, _, _ = self.get_actions(next_state, "train", False, avail,
                                                                    not self.args.double_q, hid)
                    next_values = self.target_net.value(next_state, next_actions).view(-1, n)
                returns = rewards + self.args.gamma * (1 - done) * next_values
            value_loss = self.value_mse(state, actions, returns, valid)
        return policy_loss, value_loss, action_out


def normal_entropy(log_stds: torch.Tensor) -> torch.Tensor:
    """Normal(mean, std).entropy().mean() (utilities/util.py:37-38)"""
    return (0.5 + 0.5 * math.log(2 * math.pi) + log_stds).mean()


class PGTrainer:
    """`PGTrainer` (utilities/trainer.py:9-108) + the rollout/update schedule of `Model.train_process`,
    `transition_update` and `evaluation` (models/model.py:39-70,197-302) for a batch of B envs.

    One `run()` is one episode of every env.  `steps` counts batched steps, so with B == 1 the update
    schedule is the reference's; with B envs each update still draws `batch_size` transitions (a
    contiguous window of the (step, env)-ordered replay, see mapdn_amd/replay.py).  With
    torch.distributed initialised, gradients are averaged over ranks (one flat all-reduce per
    update) before clipping, so every rank applies the same step to identical replicas."""

    def __init__(self, args, alg: str, env, device=None, data_parallel: Optional[bool] = None):
        self.args, self.env = args, env
        self.device = torch.device(device if device is not None else getattr(env, "device", "cpu"))
        target = DDPGNet(args, alg).to(self.device) if args.target else None
        self.behaviour_net = DDPGNet(args, alg, target).to(self.device)
        self.replay_buffer = TransReplayBuffer(int(args.replay_buffer_size), device=self.device, window=int(args.batch_size))   # windows are views
        rms = dict(alpha=0.99, eps=1e-5)                                                   # trainer.py:26-27
        self.policy_optimizer = torch.optim.RMSprop(self.behaviour_net.policy_dicts.parameters(), lr=args.policy_lrate, **rms)
        self.value_optimizer = torch.optim.RMSprop(self.behaviour_net.value_dicts.parameters(), lr=args.value_lrate, **rms)
        self.steps = 0
        self.episodes = 0
        self.entr = args.entr
        import torch.distributed as dist
        self._dist = dist if (data_parallel if data_parallel is not None else (dist.is_available() and dist.is_initialized())) else None
        self.collectives = {"broadcast": 0, "all_reduce_grads": 0, "all_reduce_flag": 0, "all_reduce_reward_stats": 0}   # issued so far
        if self._dist is not None:
            def _bn_reduce(t):
                self._collective(self._dist.all_reduce, t)
                self.collectives["all_reduce_reward_stats"] += 1
            self.behaviour_net.__dict__["_dp_all_reduce"] = _bn_reduce
        if self._dist is not None:                               # identical replicas to start from
            for t in self.behaviour_net.state_dict().values():
                self._collective(self._dist.broadcast, t, src=0)
                self.collectives["broadcast"] += 1

    def _collective(self, fn, t: torch.Tensor, **kw):
        """fn(t, **kw) in place.  RCCL ("nccl") takes the device tensor as it is; any other backend (gloo: the CPU / one-GPU pre-flight of
        the N-rank path) gets a host copy and the result is written back."""
        if t.is_cuda and self._dist.get_backend() != "nccl":
            h = t.detach().cpu()
            fn(h, **kw)
            t.copy_(h)
        else:
            fn(t, **kw)

    # ---- one optimiser step (trainer.py:73-98) ---------------------------------------------------
    def _all_reduce_grads(self, params):
        if self._dist is None:
            return
        grads = [p.grad for p in params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._collective(self._dist.all_reduce, flat)
        self.collectives["all_reduce_grads"] += 1
        flat /= self._dist.get_world_size()
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()

    def replica_fingerprint(self) -> torch.Tensor:
        """[2] float64: (sum of every state_dict value, sum of their bit patterns read as integers) — equal on every rank exactly when the
        replicas are (bit-)identical, which data-parallel training must keep them (same broadcast start, same averaged gradients, same
        optimiser arithmetic).  replicas_identical() gathers and compares it."""
        tot, bits = torch.zeros((), dtype=torch.float64, device=self.device), torch.zeros((), dtype=torch.float64, device=self.device)
        for v in self.behaviour_net.state_dict().values():
            if v.dtype == torch.float32:
                tot += v.double().sum(); bits += v.contiguous().view(torch.int32).double().sum()
            else:
                tot += v.double().sum(); bits += v.double().sum()
        return torch.stack((tot, bits))

    def replicas_identical(self) -> bool:
        if self._dist is None:
            return True
        fp = self.replica_fingerprint()
        world = self._dist.get_world_size()
        mine = fp if self._dist.get_backend() == "nccl" else fp.cpu()
        out = [torch.empty_like(mine) for _ in range(world)]
        self._dist.all_gather(out, mine)
        return all(torch.equal(o, out[0]) for o in out)

    def _apply(self, optimizer, loss, stat, key):
        optimizer.zero_grad()
        loss.backward()
        params = optimizer.param_groups[0]["params"]
        self._all_reduce_grads(params)
        norm = torch.nn.utils.clip_grad_norm_(params, self.args.grad_clip_eps)          # util.py:161-163
        optimizer.step()
        stat[f"mean_train_{key}_grad_norm"] = norm
        stat[f"mean_train_{key}_loss"] = loss.detach()

    def value_transition_process(self, stat, batch: Batch):
        _, value_loss, _ = self.behaviour_net.get_loss(batch, want=("value",))
        self._apply(self.value_optimizer, value_loss, stat, "value")

    def policy_transition_process(self, stat, batch: Batch):
        policy_loss, _, (means, log_stds) = self.behaviour_net.get_loss(batch, want=("policy",))
        stat["mean_train_policy_loss_raw"] = policy_loss.detach()
        if self.entr > 0:                                        # trainer.py:39-49
            entropy = normal_entropy(log_stds)
            policy_loss = policy_loss - self.entr * entropy
            stat["mean_train_entropy"] = entropy.detach()
        self._apply(self.policy_optimizer, policy_loss, stat, "policy")

    # ---- optional per-phase device timing of an episode (examples/train_ddpg.py --phases): CUDA events on the current stream around
    # replay insertion, batch sampling, the value / policy updates and the target update; the rollout is the rest of the episode
    profile_phases = False

    class _Phase:
        def __init__(self, trainer, name):
            self.t, self.name = trainer, name

        def __enter__(self):
            if self.t.profile_phases and self.t.device.type == "cuda":
                self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()
            return self

        def __exit__(self, *exc):
            if self.t.profile_phases and self.t.device.type == "cuda":
                self.b.record()
                self.t._phase_events.append((self.name, self.a, self.b))

    def _phase(self, name):
        if not hasattr(self, "_phase_events"):
            self._phase_events = []
        return PGTrainer._Phase(self, name)

    def phase_seconds(self):
        """{phase: seconds} accumulated since the last call (synchronises); empty unless profile_phases"""
        out = {}
        ev = getattr(self, "_phase_events", [])
        if ev:
            torch.cuda.synchronize(self.device)
            for name, a, b in ev:
                out[name] = out.get(name, 0.0) + a.elapsed_time(b) * 1e-3
            ev.clear()
        return out

    def value_replay_process(self, stat):
        with self._phase("sample"):
            batch = self.replay_buffer.get_batch(self.args.batch_size)
        with self._phase("value_update"):
            self.value_transition_process(stat, batch)

    def policy_replay_process(self, stat):
        with self._phase("sample"):
            batch = self.replay_buffer.get_batch(self.args.batch_size)
        with self._phase("policy_update"):
            self.policy_transition_process(stat, batch)

    def _cache_targets(self) -> bool:
        """The value loss needs pi(next_state) and Q_target(next_state, pi(next_state)) of every sampled transition (maddpg.py:103-125).
        The value epochs of one update round (models/model.py:46-49: ten of them) train the behaviour CRITIC only: neither the policy
        that produces those actions nor the target critic that values them changes between the epochs, and their windows overlap
        heavily (32 of the ring's 64 steps each).  Both are therefore computed ONCE for the whole ring, in chunks of one batch — the
        same kernels on the same inputs as the per-epoch passes, same values — and handed out with the sampled windows as two more
        (temporary) fields of the replay store.  Only when that is the cheaper order: a ring longer than the transitions the round's
        value epochs sample (the reference's own defaults: a 5000-transition ring against 10 x 32) keeps the per-epoch passes.
        MAPDN_CACHE_NEXT_ACTIONS=0 disables it."""
        rb, net, a = self.replay_buffer, self.behaviour_net, self.args
        if not isinstance(rb, TransReplayBuffer) or os.environ.get("MAPDN_CACHE_NEXT_ACTIONS", "1") == "0" or len(rb) == 0:
            return False
        st = rb.store
        if not all(k in st for k in ("next_state", "action_avail", "hid", "action")):
            return False
        n_ring = rb.size if len(rb) == rb.size else len(rb)      # (the ring fills from position 0: a partly filled ring is [0, len))
        if n_ring > int(a.value_update_epochs) * int(a.batch_size):
            return False
        cache = torch.empty_like(st["action"])
        values = torch.empty(st["action"].shape[0], net.n_, dtype=torch.float32, device=cache.device) if a.target else None
        chunk = max(int(a.batch_size), 1)
        with torch.no_grad():
            for lo in range(0, n_ring, chunk):
                hi = min(lo + chunk, n_ring)
                _, na, _, _, _ = net.get_actions(st["next_state"][lo:hi], "train", False, st["action_avail"][lo:hi], not a.double_q, st["hid"][lo:hi],
                                                 means_grad_only=True)          # (only the actions are read: the new hidden state is not stored)
                cache[lo:hi] = na
                if values is not None:
                    values[lo:hi] = net.target_net.value(st["next_state"][lo:hi], na).view(-1, net.n_)
            if rb.window:
                m = min(rb.window, n_ring)
                cache[rb.size:rb.size + m] = cache[:m]
                if values is not None:
                    values[rb.size:rb.size + m] = values[:m]
        st["next_action_cached"] = cache
        if values is not None:
            st["next_value_cached"] = values
        return True

    def transition_update(self, trans: Batch, stat):
        """models/model.py:39-70 with replay=True, mixer=False"""
        a = self.args
        with self._phase("replay_insert"):
            self.replay_buffer.add_experience(trans)
        if self.steps > a.replay_warmup and len(self.replay_buffer) >= a.batch_size and self.steps % a.behaviour_update_freq == 0:
            cached = a.value_update_epochs > 1 and self._cache_targets()
            try:
                for _ in range(a.value_update_epochs):
                    self.value_replay_process(stat)
            finally:
                if cached:
                    self.replay_buffer.store.pop("next_action_cached", None)
                    self.replay_buffer.store.pop("next_value_cached", None)
            for _ in range(a.policy_update_epochs):
                self.policy_replay_process(stat)
        if a.target and self.steps % a.target_update_freq == 0:
            with self._phase("target_update"):
                self.behaviour_net.update_target()

    # ---- rollouts (models/model.py:197-302): episodes run through the package's one rollout loop (rollout.BatchedRollout) ----
    def _episode(self, stat, train: bool):
        env, net, a = self.env, self.behaviour_net, self.args
        B, dv = env.n_envs, self.device
        prefix = "mean_train_" if train else "mean_test_"
        avail = env.get_avail_actions().to(dv)

        fused_explore = (train and self.device.type == "cuda" and a.action_dim == 1 and a.continuous and not a.gaussian_policy
                         and os.environ.get("MAPDN_FUSED_ROLLOUT", "1") != "0")
        if fused_explore:
            from . import _lib
            lib = _lib.load()
            # std = exp(log_std) with log_std = log(fixed_policy_std) (model.py:119-120, util.py:56), rounded as the f32 tensors are
            std = float(torch.tensor(math.log(a.fixed_policy_std), dtype=torch.float32).exp()) if a.shared_params else 1.0

        def policy(obs, last_hid):
            if fused_explore:
                # get_actions(status="train", exploration=True) without the log-probability nobody reads in this loop (maddpg.py:81-101,
                # util.py:52-76) and with translate_action (util.py:123-132) in the same launch: mapdn_explore_actions draws nothing itself —
                # eps is torch.randn_like(means), the same draw from the same generator as Normal(mean, std).rsample()
                means, _, hid = net.policy(obs, last_hid)
                means = means.contiguous()
                eps = torch.randn_like(means)
                action, action_pol, actual = torch.empty_like(means), torch.empty_like(means), torch.empty_like(means)
                av = avail.expand_as(means).contiguous() if avail.shape != means.shape else avail.contiguous()
                with torch.cuda.device(dv):
                    _lib.check(lib.mapdn_explore_actions(means.data_ptr(), eps.data_ptr(), av.data_ptr(), std, int(bool(a.action_enforcebound)),
                                                         float(a.action_scale), float(a.action_bias), action.data_ptr(), action_pol.data_ptr(),
                                                         actual.data_ptr(), means.numel(), torch.cuda.current_stream(dv).cuda_stream))
                return action, hid, dict(action_pol=action_pol, last_hid=last_hid, hid=hid, actual=actual)
            if train:
                action, action_pol, _, _, hid = net.get_actions(obs, "train", True, avail, False, last_hid)
            else:
                action, action_pol, _, _, hid = net.get_actions(obs, "test", False, avail, False, last_hid)
            return action, hid, dict(action_pol=action_pol, last_hid=last_hid, hid=hid)

        def on_step(t, obs, action, reward, done, info, next_obs, alive, aux):
            action_pol, last_hid, hid = aux["action_pol"], aux["last_hid"], aux["hid"]
            trans = dict(state=obs, action=action_pol, reward=reward.float().unsqueeze(-1).expand(B, net.n_).contiguous(),
                         next_state=next_obs, done=done.view(B, 1).float(),
                         last_step=(done | (t == a.max_steps - 1)).view(B, 1).float(), action_avail=avail,
                         last_hid=last_hid, hid=hid, valid=alive.clone())
            self.transition_update(trans, stat)
            self.steps += 1

        def agree(flag):                                        # data-parallel ranks must agree on the early exit: `steps` drives
            if self._dist is None:                              # the update schedule and every update is a collective
                return flag
            if self._dist.get_backend() != "nccl":
                flag = flag.cpu()
            self._dist.all_reduce(flag, op=self._dist.ReduceOp.MAX)
            self.collectives["all_reduce_flag"] += 1
            return flag

        from .rollout import BatchedRollout
        ro = BatchedRollout(env, policy, a.max_steps, on_step=on_step if train else None, store_window=False, all_alive_reduce=agree,
                            action_scale=a.action_scale, action_bias=a.action_bias)
        _, ep = ro.run(prefix=prefix, hidden=net.init_hidden(B))
        for k, v in ep.items():
            stat[k] = v if train else stat.get(k, 0.0) + v

    def train_process(self, stat):
        self._episode(stat, train=True)
        self.episodes += 1
        for k, v in list(stat.items()):
            if torch.is_tensor(v):
                stat[k] = float(v)

    def evaluation(self, stat):
        """num_eval_episodes episodes in total, B at a time (model.py:265-302)"""
        rounds = max(1, -(-self.args.num_eval_episodes // self.env.n_envs))
        test = {}
        for _ in range(rounds):
            self._episode(test, train=False)
        stat.update({k: v / rounds for k, v in test.items()})

    def run(self, stat, episode: int):
        """trainer.py:110-113"""
        self.train_process(stat)
        if episode % self.args.eval_freq == self.args.eval_freq - 1 or episode == 0:
            self.evaluation(stat)

    def save(self, path):
        torch.save({"model_state_dict": self.behaviour_net.state_dict()}, path)          # train.py:119

    def load(self, path):
        self.behaviour_net.load_state_dict(torch.load(path, map_location=self.device)["model_state_dict"])
