"""Batched, GPU-resident rollout driver — the direct caller of the hot path.

Mirrors `Model.train_process` / `Model.evaluation` of the reference (models/model.py:197-302) for B
envs at once: policy forward, `translate_action` (utilities/util.py:123-132), env step, next obs and
statistics all stay on the device — the reference crosses host<->GPU twice per env step
(models/model.py:211,214).  The transition window is the GPU-resident counterpart of the fields of
`Transition` (models/model.py:18) the DDPG-style learners consume.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from ._lib import INFO_KEYS


def translate_action(action: torch.Tensor, action_scale: float, action_bias: float) -> torch.Tensor:
    """utilities/util.py:123-132 (continuous branch), batched and on device:
    clamp to [-1, 1], then map linearly onto [bias - scale, bias + scale]."""
    cp = torch.clamp(action, min=-1.0, max=1.0)
    low = action_bias - action_scale
    high = action_bias + action_scale
    return 0.5 * (cp + 1.0) * (high - low) + low


@dataclass
class TransitionWindow:
    """[T, B, ...] tensors on the env's device (models/model.py:18 fields: state, action, reward,
    next_state, done, last_step); `reward` is repeated per agent as at models/model.py:217."""
    state: torch.Tensor        # [T, B, n_agents, obs]   f32
    action: torch.Tensor       # [T, B, n_agents, 1]     f32 (policy space, before translate_action)
    reward: torch.Tensor       # [T, B, n_agents]        f32
    next_state: torch.Tensor   # [T, B, n_agents, obs]   f32
    done: torch.Tensor         # [T, B]                  bool
    last_step: torch.Tensor    # [T, B]                  bool  (done or t == max_steps-1, model.py:225)
    steps: int = 0


class BatchedRollout:
    """The one rollout loop of the package (the trainer's episodes run through it too).

    policy: callable (obs [B, n, obs_size] f32, hidden) -> (action, hidden) or (action, hidden, aux); action [B, n] or [B, n, 1]
    in [-1, 1] space; `aux` (any object) is handed to `on_step` — the learner passes the pre-noise action and the
    hidden states a `Transition` stores (models/model.py:18).
    on_step: optional callable(t, obs, action, reward, done, info, next_obs, alive, aux) called after every batched step
    (the learner's `transition_update`, models/model.py:39-70).
    store_window: keep the [T, B, ...] `TransitionWindow` (off for the learner: its replay buffer is the store).
    all_alive_reduce: optional callable(flag int32 tensor) -> flag, for data-parallel ranks that must agree on the early exit.
    own: how observations handed out by the env are taken over (clone for envs that reuse their output buffers)."""

    def __init__(self, env, policy, max_steps: int = 240, on_step=None, store_window: bool = True, all_alive_reduce=None,
                 action_scale=None, action_bias=None):
        self.env, self.policy, self.max_steps = env, policy, int(max_steps)
        self.on_step, self.all_alive_reduce = on_step, all_alive_reduce
        self.action_scale = float(env.args["action_scale"] if action_scale is None else action_scale)
        self.action_bias = float(env.args["action_bias"] if action_bias is None else action_bias)
        B, n, o, dv, T = env.n_envs, env.n_agents, env.obs_size, env.device, self.max_steps
        self.win = None
        if store_window:
            self.win = TransitionWindow(
                state=torch.empty(T, B, n, o, device=dv), action=torch.empty(T, B, n, 1, device=dv),
                reward=torch.empty(T, B, n, device=dv), next_state=torch.empty(T, B, n, o, device=dv),
                done=torch.empty(T, B, dtype=torch.bool, device=dv), last_step=torch.empty(T, B, dtype=torch.bool, device=dv))
        self._own = (lambda t: t) if getattr(env, "copy", False) else (lambda t: t.clone())   # copy=True envs hand out fresh tensors

    @torch.no_grad()
    def run(self, prefix: str = "mean_train_", hidden=None, add_noise: bool = True):
        """One episode for every env.  Returns (window or None, stat): stat[prefix + key] = mean over steps and
        envs of every info key and of the reward (models/model.py:243-248,257-261), computed on device."""
        env, win, T = self.env, self.win, self.max_steps
        obs, _ = env.reset()
        obs = self._own(obs.float())
        info_sum = torch.zeros(len(INFO_KEYS), dtype=torch.float64, device=env.device)
        rew_sum = torch.zeros((), dtype=torch.float64, device=env.device)
        alive_steps = torch.zeros((), dtype=torch.float64, device=env.device)
        alive = torch.ones(env.n_envs, dtype=torch.bool, device=env.device)
        # the per-step bookkeeping as one launch (csrc/rollout.hip: k_rollout_stats) on the GPU: sums = info[11] | reward | live count
        fused = torch.device(env.device).type == "cuda" and len(INFO_KEYS) == 11 and os.environ.get("MAPDN_FUSED_ROLLOUT", "1") != "0"
        if fused:
            from . import _lib
            lib = _lib.load()
            sums = torch.zeros(13, dtype=torch.float64, device=env.device)
            alive_next = torch.ones_like(alive)
        t = 0
        for t in range(T):
            out = self.policy(obs, hidden)
            action, hidden, aux = out if len(out) == 3 else (out[0], out[1], None)
            action = action.reshape(env.n_envs, env.n_agents, 1).float()
            if isinstance(aux, dict) and aux.get("actual") is not None:     # the policy already ran translate_action (fused with its sampling)
                actual = aux["actual"].reshape(env.n_envs, env.n_agents)
            else:
                actual = translate_action(action.squeeze(-1), self.action_scale, self.action_bias)
            reward, done, info = env.step(actual) if add_noise else env.step(actual, add_noise=False)
            nxt = self._own(env.get_obs().float())
            if win is not None:
                win.state[t].copy_(obs); win.action[t].copy_(action)
                win.reward[t].copy_(reward.float().unsqueeze(-1).expand(-1, env.n_agents))
                win.next_state[t].copy_(nxt); win.done[t].copy_(done)
                win.last_step[t].copy_(done | (t == T - 1))
            use = fused and info.dtype == torch.float64 and reward.dtype == torch.float64 and done.dtype == torch.bool \
                and info.is_contiguous() and reward.is_contiguous() and done.is_contiguous()
            if use:                                             # frozen (already terminated) envs do not count
                with torch.cuda.device(env.device):
                    _lib.check(lib.mapdn_rollout_stats(info.data_ptr(), reward.data_ptr(), alive.data_ptr(), done.data_ptr(), alive_next.data_ptr(),
                                                       sums.data_ptr(), env.n_envs, torch.cuda.current_stream(env.device).cuda_stream))
            else:
                w = alive.double()
                info_sum += (info.double() * w.unsqueeze(-1)).sum(0); rew_sum += (reward.double() * w).sum(); alive_steps += w.sum()
            if self.on_step is not None:
                with torch.enable_grad():
                    self.on_step(t, obs, action, reward, done, info, nxt, alive, aux)
            if use:
                alive, alive_next = alive_next, alive           # (the step's own `alive` stayed intact for on_step)
            else:
                alive = alive & ~done.bool()
            obs = nxt
            if t % 16 == 15:                                    # the only host sync, once per 16 steps
                flag = alive.any().to(torch.int32)
                if self.all_alive_reduce is not None:
                    flag = self.all_alive_reduce(flag)
                if not bool(flag):
                    break
        if win is not None:
            win.steps = t + 1
        if fused:
            info_sum = info_sum + sums[:11]; rew_sum = rew_sum + sums[11]; alive_steps = alive_steps + sums[12]
        denom = torch.clamp(alive_steps, min=1.0)
        stat = {prefix + k: float(v) for k, v in zip(INFO_KEYS, (info_sum / denom).tolist())}
        stat[prefix + "reward"] = float(rew_sum / denom)
        return win, stat
