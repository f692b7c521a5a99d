"""Python host side of the MI355X VoltageControl environment.

``VoltageControlBatch``  B independent env instances on one GPU, tensor in / tensor out.
``VoltageControl``       B = 1 adapter with the exact PyMARL surface and Python types of the
                         reference class (environments/var_voltage_control/voltage_control_env.py:24),
                         so `train.py:62`, `test.py:68` and `code_examples.py:32` keep working.

All numeric work happens in libmapdn_hip.so (hand-written gfx950 kernels) through the C ABI of
include/mapdn.h; PyTorch only provides device memory and the stream.  There is no CPU fallback.
"""
from __future__ import annotations

import os
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from ._lib import INFO_KEYS, N_INFO
from .marl_env_api import MultiAgentEnv
from .netspec import NetSpec, Profiles, make_case


def convert(dictionary):
    """voltage_control_env.py:14-15"""
    return namedtuple('GenericDict', dictionary.keys())(**dictionary)


class ActionSpace(object):
    """voltage_control_env.py:18-21"""

    def __init__(self, low, high):
        self.low = low
        self.high = high


DEFAULT_ARGS = dict(  # args/env_args/var_voltage_control.yaml:3-20
    voltage_barrier_type="l1", voltage_weight=1.0, q_weight=0.1, line_weight=None, dq_dv_weight=None,
    history=1, pv_scale=1.0, demand_scale=1.0,
    state_space=["pv", "demand", "reactive", "vm_pu", "va_degree"],
    v_upper=1.05, v_lower=0.95, episode_limit=240, action_scale=0.8, action_bias=0.0,
    mode="distributed", reset_action=True, seed=0,
    auto_reset=False,     # not a reference key: per-env automatic restart of terminated envs (include/mapdn.h, mapdn_env_config)
)

_SCENARIOS = {"case33_3min_final": "case33", "case141_3min_final": "case141", "case322_3min_final": "case322",
              "case33": "case33", "case141": "case141", "case322": "case322"}


def _as_dict(kwargs):
    if isinstance(kwargs, dict):
        return dict(kwargs)
    if hasattr(kwargs, "_asdict"):
        return dict(kwargs._asdict())
    raise TypeError("env args must be a dict or a namedtuple")


class VoltageControlBatch:
    """B env instances sharing one network; state lives on the GPU in env-minor SoA arrays.

    reset()/step()/get_obs()/get_state() mirror the reference methods with a leading batch axis:
      obs    float32 [B, n_agents, obs_size]      (prep_obs's `.float()` hand-off, utilities/util.py:147)
      state  float32 [B, state_size]
      reward float64 [B], terminated bool [B], info float64 [B, 11] (columns = INFO_KEYS)
    Returned tensors are views of preallocated buffers that the next call overwrites
    (pass ``copy=True`` to get fresh tensors).
    """

    def __init__(self, net: NetSpec, profiles: Profiles, args=None, n_envs: int = 1, device=None,
                 env_id_offset: int = 0, max_reset_tries: int = 3, copy: bool = False,
                 obs_dtype=torch.float32, tuning: dict | None = None):
        a = dict(DEFAULT_ARGS)
        a.update(_as_dict(args or {}))
        if a["mode"] != "distributed":
            # the reference's decentralised mode raises KeyError('sgen0') at voltage_control_env.py:239
            raise NotImplementedError("only mode='distributed' exists (the reference's decentralised get_obs raises KeyError)")
        if a["line_weight"] is None and a["q_weight"] is None:
            raise NotImplementedError("Please at least give one weight, either q_weight or line_weight.")  # :617
        self.args = a
        self.net, self.profiles = net, profiles
        self.n_envs = int(n_envs)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("VoltageControlBatch needs a ROCm GPU device (no CPU fallback)")
        self.copy = copy
        self.obs_dtype = obs_dtype
        self.max_reset_tries = int(max_reset_tries)
        self.history = int(a["history"])
        self.episode_limit = int(a["episode_limit"])
        self.action_space = ActionSpace(low=-a["action_scale"] + a["action_bias"], high=a["action_scale"] + a["action_bias"])
        self._lib = _lib.load()
        cnet, self._keep = _lib.make_cnetspec(net)
        ccfg = _lib.make_cconfig(a, env_id_offset, tuning)     # tuning: per-handle launch / solver fields of mapdn_env_config
        h = _lib.C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self._lib.mapdn_create(_lib.C.byref(cnet), _lib.C.byref(ccfg), self.n_envs, dev_index, _lib.C.byref(h)))
        self._h = h
        dims = _lib.CDims()
        _lib.check(self._lib.mapdn_dims(self._h, _lib.C.byref(dims)), self._h)
        self.n_bus, self.n_line, self.n_load, self.n_sgen = dims.n_bus, dims.n_line, dims.n_load, dims.n_sgen
        self.n_agents, self.n_actions = dims.n_agents, dims.n_actions
        self._obs_size1, self.state_size = dims.obs_size, dims.state_size
        self.obs_size = self._obs_size1 * self.history          # :303-315 stacks `history` frames
        self.max_zone_size = dims.max_zone_size
        self.is_radial = bool(dims.is_radial)                   # False: meshed net, general-topology solver (k_nr_dense)
        p = profiles
        _lib.check(self._lib.mapdn_set_profiles(
            self._h, _lib._p(p.pv, _lib._pd), _lib._p(p.load_p, _lib._pd), _lib._p(p.load_q, _lib._pd),
            p.n_rows, p.time_delta_min, p.days), self._h)
        B, dv = self.n_envs, self.device
        self._reward = torch.zeros(B, dtype=torch.float64, device=dv)
        self._term = torch.zeros(B, dtype=torch.bool, device=dv)     # 1 byte per env, written as 0/1 by the kernel
        self._info = torch.zeros(B, N_INFO, dtype=torch.float64, device=dv)
        self._obs = {}
        self._state = {}
        self._obs_hist = None
        self._was_reset = False
        self._stepped = False
        self._obs_fresh = None                                      # dtype of the obs buffer that holds the current state's obs
        # step() = mapdn_step_obs (False, or MAPDN_FUSED_STEP=0 for A/B runs: mapdn_step, then mapdn_get_obs)
        self.fused_step = os.environ.get("MAPDN_FUSED_STEP", "1") != "0"

    # ---- plumbing -------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mapdn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, t):
        return t.clone() if self.copy else t

    @staticmethod
    def _code(dtype):
        if dtype == torch.float32:
            return _lib.F32
        if dtype == torch.float64:
            return _lib.F64
        raise TypeError("dtype must be torch.float32 or torch.float64")

    # ---- PyMARL-style surface, batched ----------------------------------------------------------
    def start_rows(self):
        """episode start row of every env (`start` of voltage_control_env.py:445), int64 [B]"""
        out = torch.empty(self.n_envs, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_start_rows(self._h, out.data_ptr(), self._stream()), self._h)
        return out

    def auto_reset_mask(self):
        """bool [B]: envs that the last step() call (re)started (args['auto_reset']); their reward / terminated / info of
        that call are 0 and the obs that follows is the first obs of the new episode"""
        out = torch.empty(self.n_envs, dtype=torch.bool, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_auto_reset_mask(self._h, out.data_ptr(), self._stream()), self._h)
        return out

    def episode_returns(self):
        """sum_rewards of the running episode (voltage_control_env.py:203), float64 [B]"""
        out = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_returns(self._h, out.data_ptr(), self._stream()), self._h)
        return out

    def reset(self, start_rows=None, add_noise=True, reset_time=True):
        """reset() (voltage_control_env.py:96-135).  start_rows: optional int64 [B] table rows;
        reset_time=False re-uses the previous episode's start (:110-113)."""
        sr = None
        if start_rows is None and not reset_time and self._was_reset:
            start_rows = self.start_rows()
        if start_rows is not None:
            sr = torch.as_tensor(start_rows, dtype=torch.int64, device=self.device).contiguous()
            assert sr.shape == (self.n_envs,)
            # the episode window start+1 .. start+episode_limit must lie inside the table: the reference slices past the
            # end and dies with IndexError on the empty row (:446-447, :473-475); the kernels also refuse such rows
            lo, hi = int(sr.min()), int(sr.max())
            if lo < 0 or hi + self.episode_limit + 1 >= self.profiles.n_rows:
                raise IndexError(f"start row {lo if lo < 0 else hi}: the {self.episode_limit}-step episode window leaves the "
                                 f"{self.profiles.n_rows}-row profile table")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_reset(self._h, sr.data_ptr() if sr is not None else None, int(add_noise),
                                             self.max_reset_tries, self._stream()), self._h)
        self._was_reset = True
        self._stepped = False
        self._obs_fresh = None
        if self.history > 1:
            self._obs_hist = None
        return self.get_obs(), self.get_state()

    def manual_reset(self, day, hour, interval):
        """manual_reset (voltage_control_env.py:137-176): same start for every env, no noise."""
        day, hour, interval = int(day), int(hour), int(interval)
        if not (0 <= hour < 24 and 0 <= interval < self.profiles.intervals_per_hour and day >= 0):
            raise ValueError(f"manual_reset(day={day}, hour={hour}, interval={interval}): need day >= 0, 0 <= hour < 24, "
                             f"0 <= interval < {self.profiles.intervals_per_hour}")
        row = self.profiles.start_row(day, hour, interval)
        return self.reset(start_rows=torch.full((self.n_envs,), row, dtype=torch.int64), add_noise=False)

    def step(self, actions, add_noise=True):
        """step (voltage_control_env.py:178-211).  actions: [B, n_sgen] float32/float64 CUDA tensor,
        already scaled to [bias-scale, bias+scale] (utilities/util.py:123-132); not clipped (:553)."""
        a = actions
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a), device=self.device)
        if a.device != self.device:
            a = a.to(self.device)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float64)
        a = a.reshape(self.n_envs, self.n_sgen).contiguous()
        # step() and the get_obs() that always follows it (models/model.py:216,219) as one C call: the obs of the new
        # state lands in the preallocated buffer of self.obs_dtype and get_obs() hands it out without another launch
        if not self.fused_step:                                     # plain mapdn_step; get_obs() launches its own gather
            with torch.cuda.device(self.device):
                _lib.check(self._lib.mapdn_step(self._h, a.data_ptr(), self._code(a.dtype), int(add_noise),
                                                self._reward.data_ptr(), self._term.data_ptr(), self._info.data_ptr(),
                                                self._stream()), self._h)
            self._stepped = True
            self._obs_fresh = None
            return self._out(self._reward), self._out(self._term), self._out(self._info)
        buf = self._obs_buf(self.obs_dtype)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_step_obs(self._h, a.data_ptr(), self._code(a.dtype), int(add_noise),
                                                self._reward.data_ptr(), self._term.data_ptr(), self._info.data_ptr(),
                                                buf.data_ptr(), self._code(self.obs_dtype), self._stream()), self._h)
        self._stepped = True
        self._obs_fresh = self.obs_dtype
        return self._out(self._reward), self._out(self._term), self._out(self._info)

    def _obs_buf(self, dtype):
        buf = self._obs.get(dtype)
        if buf is None:
            buf = self._obs[dtype] = torch.empty(self.n_envs, self.n_agents, self._obs_size1, dtype=dtype, device=self.device)
        return buf

    def get_obs(self, dtype=None):
        dtype = dtype or self.obs_dtype
        buf = self._obs_buf(dtype)
        if self._obs_fresh != dtype:                                # (after step() the obs of this dtype is already there)
            with torch.cuda.device(self.device):
                _lib.check(self._lib.mapdn_get_obs(self._h, buf.data_ptr(), self._code(dtype), self._stream()), self._h)
        if self.history > 1:                                        # :303-315: [zeros | older frames | newest]
            o1, key = self._obs_size1, (dtype, "hist")
            pair = self._obs.get(key)
            if pair is None:                                        # two preallocated stacked frames, used alternately
                pair = self._obs[key] = [torch.zeros(self.n_envs, self.n_agents, o1 * self.history, dtype=dtype, device=self.device)
                                         for _ in range(2)]
            if self._obs_hist is None:                              # first get_obs of the episode: no history yet
                pair[0].zero_()
                self._obs_hist = 0
            cur, new = pair[self._obs_hist], pair[1 - self._obs_hist]
            new[..., :-o1].copy_(cur[..., o1:])
            new[..., -o1:].copy_(buf)
            if self.args.get("auto_reset") and self._stepped:          # a restarted env begins with an empty history (:103-104)
                new[..., :-o1].masked_fill_(self.auto_reset_mask().view(-1, 1, 1), 0)
            self._obs_hist = 1 - self._obs_hist
            return self._out(new)
        return self._out(buf)

    def get_state(self, dtype=None):
        dtype = dtype or self.obs_dtype
        buf = self._state.get(dtype)
        if buf is None:
            buf = self._state[dtype] = torch.empty(self.n_envs, self.state_size, dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_state(self._h, buf.data_ptr(), self._code(dtype), self._stream()), self._h)
        return self._out(buf)

    def get_avail_actions(self):
        """[B, n_agents, 1] ones (distributed mode, voltage_control_env.py:345-357)"""
        return torch.ones(self.n_envs, self.n_agents, self.n_actions, device=self.device)

    def get_obs_size(self):
        return self.obs_size

    def get_state_size(self):
        return self.state_size

    def get_total_actions(self):
        return self.n_actions

    def get_num_of_agents(self):
        return self.n_agents

    def get_env_info(self):
        """environments/multiagentenv.py:61-67"""
        return {"state_shape": self.get_state_size(), "obs_shape": self.get_obs_size(),
                "n_actions": self.get_total_actions(), "n_agents": self.n_agents,
                "episode_limit": self.episode_limit}

    # ---- extras ---------------------------------------------------------------------------------
    _RESULT_KEYS = ("vm_pu", "va_degree", "p_mw", "q_mvar", "pl_mw", "sgen_p", "sgen_q")

    def results(self, keys=None):
        """tester getters (voltage_control_env.py:625-647) for the whole batch, float64.  `keys`: subset of
        vm_pu va_degree p_mw q_mvar pl_mw sgen_p sgen_q (default all); only those are transposed out."""
        keys = self._RESULT_KEYS if keys is None else tuple(keys)
        B, dv, f64 = self.n_envs, self.device, torch.float64
        width = dict(vm_pu=self.n_bus, va_degree=self.n_bus, p_mw=self.n_bus, q_mvar=self.n_bus, pl_mw=self.n_line,
                     sgen_p=self.n_sgen, sgen_q=self.n_sgen)
        out = {k: (torch.empty(B, width[k], dtype=f64, device=dv) if self.copy or B > 1 else self._res_buf(k, width[k])) for k in keys}
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_results(self._h, *[(out[k].data_ptr() if k in out else None) for k in self._RESULT_KEYS],
                                                   self._stream()), self._h)
        return out

    def _res_buf(self, k, w):
        b = self._obs.get(("res", k))
        if b is None:
            b = self._obs[("res", k)] = torch.empty(self.n_envs, w, dtype=torch.float64, device=self.device)
        return b

    def loads(self):
        lp = torch.empty(self.n_envs, self.n_load, dtype=torch.float64, device=self.device)
        lq = torch.empty_like(lp)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_get_loads(self._h, lp.data_ptr(), lq.data_ptr(), self._stream()), self._h)
        return lp, lq

    def solve(self, p_load, q_load, p_sgen, q_sgen):
        """Pure batched power flow == pp.runpp on explicit MW/MVAr inputs ([B, nl] / [B, ns], float64)."""
        f64, dv, B = torch.float64, self.device, self.n_envs
        ins = [torch.as_tensor(x, dtype=f64, device=dv).contiguous() for x in (p_load, q_load, p_sgen, q_sgen)]
        assert ins[0].shape == (B, self.n_load) and ins[2].shape == (B, self.n_sgen)
        vm = torch.empty(B, self.n_bus, dtype=f64, device=dv)
        va = torch.empty_like(vm)
        it = torch.empty(B, dtype=torch.int32, device=dv)
        cv = torch.empty(B, dtype=torch.uint8, device=dv)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_solve_only(self._h, *[t.data_ptr() for t in ins], vm.data_ptr(), va.data_ptr(),
                                                  it.data_ptr(), cv.data_ptr(), self._stream()), self._h)
        return vm, va, it, cv.bool()

    def ybus_dense(self):
        """Ybus over the ELECTRICAL nodes (n_nodes x n_nodes; == buses unless closed bus-bus switches fuse some: include/mapdn.h)"""
        nn = self.geometry()["n_nodes"]
        out = np.zeros((nn, nn, 2))
        _lib.check(self._lib.mapdn_get_ybus_dense(self._h, _lib._p(out, _lib._pd)), self._h)
        return out[..., 0] + 1j * out[..., 1]

    def obs_index(self):
        n = self.n_agents * self._obs_size1
        kind = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        _lib.check(self._lib.mapdn_get_obs_index(self._h, _lib._p(kind, _lib._pi), _lib._p(idx, _lib._pi)), self._h)
        return kind.reshape(self.n_agents, -1), idx.reshape(self.n_agents, -1)

    def geometry(self):
        """the NR launch geometry mapdn_create settled on (solver, waves, envs per workgroup, LDS residency, rows, model time)"""
        return _lib.nr_geometry(self._h)

    def stats(self):
        rf, mi, mx = _lib.C.c_int64(), _lib.C.c_double(), _lib.C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mapdn_stats(self._h, _lib.C.byref(rf), _lib.C.byref(mi), _lib.C.byref(mx), self._stream()), self._h)
        return dict(reset_failures=rf.value, mean_nr_iters=mi.value, max_nr_iters=mx.value)

    def profile_stats(self):
        """(stdv [n_sgen + 2 n_load], s_max [n_sgen]) as the library derived them from the profile tables (:70-72, :518-520)"""
        sd, sm = np.empty(self.n_sgen + 2 * self.n_load), np.empty(self.n_sgen)
        _lib.check(self._lib.mapdn_get_profile_stats(self._h, _lib._p(sd, _lib._pd), _lib._p(sm, _lib._pd)), self._h)
        return sd, sm

    def nr_timing(self, enable=True):
        _lib.check(self._lib.mapdn_nr_timing(self._h, int(enable)), self._h)

    def nr_time_ms(self):
        ms, n = _lib.C.c_double(), _lib.C.c_int64()
        _lib.check(self._lib.mapdn_nr_time_ms(self._h, _lib.C.byref(ms), _lib.C.byref(n)), self._h)
        return ms.value, n.value


# ------------------------------------------------------------------------------------------------
def _resolve_data(args):
    """(NetSpec, Profiles) for the constructor kwargs of the reference class.

    The reference loads `data_path/model.p` (pandapower pickle) and three CSVs
    (voltage_control_env.py:400-438).  Those files are an external download that is not available
    offline, so: explicit `net`/`profiles` kwargs win; otherwise the last component of `data_path`
    selects the synthetic network of the same shape.
    """
    if args.get("net") is not None and args.get("profiles") is not None:
        return args["net"], args["profiles"]
    dp = str(args.get("data_path", ""))
    ps, ds = float(args.get("pv_scale", 1.0)), float(args.get("demand_scale", 1.0))       # :415,426,437
    if os.path.isdir(dp) and (os.path.exists(os.path.join(dp, "netspec.npz")) or os.path.exists(os.path.join(dp, "model.p"))):
        from .data import load_scenario
        return load_scenario(dp, ps, ds, hv_init=args.get("hv_init"))                     # real scenario directory (hv_init: see load_scenario)
    name = os.path.basename(os.path.normpath(dp))
    if name not in _SCENARIOS:
        raise FileNotFoundError(f"unknown scenario {name!r}: pass net=/profiles=, a directory with netspec.npz + the three "
                                f"CSV tables, or a data_path ending in one of {sorted(_SCENARIOS)}")
    import warnings
    warnings.warn(f"data_path {dp!r} holds no netspec.npz / model.p: using the SYNTHETIC {_SCENARIOS[name]} feeder of the same shape "
                  f"(mapdn_amd.netspec.make_case) — results are NOT comparable with runs on the real MAPDN data; pass "
                  f"net=/profiles= or a scenario directory to silence this", RuntimeWarning, stacklevel=3)
    net, prof = make_case(_SCENARIOS[name], seed=0)
    if ps != 1.0 or ds != 1.0:
        prof = Profiles(pv=prof.pv * ps, load_p=prof.load_p * ds, load_q=prof.load_q * ds,
                        time_delta_min=prof.time_delta_min, days=prof.days)
    return net, prof


class VoltageControl(MultiAgentEnv):
    """Drop-in for the reference class: same constructor, methods, return types (B = 1).

        state, global_state = env.reset()
        for t in range(240):
            actions = agents.get_actions(state)
            reward, done, info = env.step(actions)
            next_state = env.get_obs()
    """

    def __init__(self, kwargs, device=None):
        args = _as_dict(kwargs)
        self.args = convert({k: v for k, v in args.items() if k not in ("net", "profiles")})
        net, prof = _resolve_data(args)
        self.data_path = args.get("data_path")
        self._b = VoltageControlBatch(net, prof, {k: v for k, v in args.items() if k in DEFAULT_ARGS}, n_envs=1,
                                      device=device, copy=False, obs_dtype=torch.float64)
        b = self._b
        self.episode_limit = b.episode_limit
        self.voltage_barrier_type = b.args["voltage_barrier_type"]
        self.voltage_weight, self.q_weight, self.line_weight = b.args["voltage_weight"], b.args["q_weight"], b.args["line_weight"]
        self.v_upper, self.v_lower = b.args["v_upper"], b.args["v_lower"]
        self.action_space = b.action_space
        self.history = b.history
        self.state_space = b.args["state_space"]
        self.n_actions = 1                                            # :80
        self.n_agents = b.n_agents                                    # :81
        self.s_max = prof.s_max(1.2)
        print(f"This is the s_max: \n{self.s_max}")                  # :521
        # ---- latency path of the drop-in call pattern (utilities/tester.py:48-55: numpy action in, Python scalars + list of numpy
        # out, once per step).  reward | info[11] | terminated | obs live in ONE device buffer (the batch object's output tensors
        # are views of it) mirrored by ONE pinned host buffer: a step() is one async H2D of the action, the four kernels, one
        # async D2H of the packed outputs and a single stream synchronisation; the get_obs() that follows costs nothing.
        n, o1 = b.n_agents, b._obs_size1
        # Offsets: info 32-byte aligned, obs 64-byte aligned — k_gather's write phase stores 4 consecutive f64 columns as one
        # 32-byte vector (kernels.hip, `V4`), which assumes the alignment torch.empty gave the old stand-alone obs tensor
        # (torch device allocations and pinned host allocations are >= 256-byte aligned; asserted below)
        off_info = 32
        off_term = off_info + 8 * N_INFO
        off_obs = (off_term + 8 + 63) // 64 * 64
        self._pk_off = dict(reward=0, info=off_info, term=off_term, obs=off_obs)
        nbytes = self._pk_off["obs"] + 8 * n * o1
        self._pk_dev = torch.zeros(nbytes, dtype=torch.uint8, device=b.device)
        self._pk_host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        self._pk_np = self._pk_host.numpy()
        assert self._pk_dev.data_ptr() % 64 == 0 and self._pk_host.data_ptr() % 64 == 0, "packed step buffers must be 64-byte aligned"
        po = self._pk_off
        b._reward = self._pk_dev[po["reward"]:po["reward"] + 8].view(torch.float64)
        b._info = self._pk_dev[po["info"]:po["info"] + 8 * N_INFO].view(torch.float64).view(1, N_INFO)
        b._term = self._pk_dev[po["term"]:po["term"] + 1].view(torch.bool)
        b._obs[torch.float64] = self._pk_dev[po["obs"]:].view(torch.float64).view(1, n, o1)
        self._act_host = torch.zeros(1, b.n_sgen, dtype=torch.float64).pin_memory()
        self._act_dev = torch.zeros(1, b.n_sgen, dtype=torch.float64, device=b.device)
        self._packed = b.fused_step and os.environ.get("MAPDN_DROPIN_PACKED", "1") != "0"
        # zero-copy variant (history == 1): pinned host memory is mapped into the device's address space on ROCm, so the kernels
        # read the action straight from the pinned host buffer and the solver epilogue / k_gather store reward, info, terminated
        # and the obs straight into it — no H2D / D2H copy launches at all, one stream synchronisation per step
        self._zero_copy = self._packed and self.history == 1 and os.environ.get("MAPDN_DROPIN_ZEROCOPY", "1") != "0"
        if self._zero_copy:
            po, hp = self._pk_off, self._pk_host.data_ptr()
            self._zc = dict(reward=hp + po["reward"], info=hp + po["info"], term=hp + po["term"], obs=hp + po["obs"],
                            act=self._act_host.data_ptr())
        self._host_obs_valid = False
        agents_obs, state = self.reset()
        self.obs_size = agents_obs[0].shape[0]                        # :87
        self.state_size = state.shape[0]
        self.steps = 1
        self.sum_rewards = 0

    def reset(self, reset_time=True):
        """reset (:96-135): the reference re-draws until the start is solvable (`while not solvable`, :108); so does this
        (the library tries `max_reset_tries` starts per call), giving up loudly after 100 calls instead of spinning forever."""
        self._host_obs_valid = False
        self.steps = 1
        self.sum_rewards = 0
        for _ in range(100):
            obs, state = self._b.reset(reset_time=reset_time)
            if self._b.stats()["reset_failures"] == 0:
                return self._obs_list(obs), state[0].cpu().numpy()
            # (reset_time=False keeps the time stamp, but every pass still draws new noise and a new initial action, :118-122,
            # so it can succeed at the same stamp: keep looping like the reference, up to the same cap)
        raise RuntimeError("reset(): no solvable start found (the reference would keep looping at voltage_control_env.py:108)")

    def manual_reset(self, day, hour, interval):
        self._host_obs_valid = False
        self.steps = 1
        self.sum_rewards = 0
        obs, state = self._b.manual_reset(day, hour, interval)
        if self._b.stats()["reset_failures"]:
            # the reference loops forever here (:151-174: same start, no noise => same failure); fail loudly instead
            raise RuntimeError("manual_reset(): the power flow of this start is not solvable")
        return self._obs_list(obs), state[0].cpu().numpy()

    def step(self, actions, add_noise=True):
        if self._packed:
            self._act_host.numpy()[0, :] = np.asarray(actions, dtype=np.float64).reshape(-1)
            if self._zero_copy:
                b, z = self._b, self._zc
                with torch.cuda.device(b.device):
                    _lib.check(b._lib.mapdn_step_obs(b._h, z["act"], _lib.F64, int(add_noise), z["reward"], z["term"], z["info"],
                                                     z["obs"], _lib.F64, b._stream()), b._h)
                b._stepped = True
                b._obs_fresh = None                                   # (the batch object's own device-side obs buffer was not filled)
            else:
                self._act_dev.copy_(self._act_host, non_blocking=True)
                self._b.step(self._act_dev, add_noise=add_noise)      # enqueues only; outputs land in the packed device buffer
                self._pk_host.copy_(self._pk_dev, non_blocking=True)
            torch.cuda.current_stream(self._b.device).synchronize()   # the one synchronisation of the step
            po, h = self._pk_off, self._pk_np
            reward = float(h[po["reward"]:po["reward"] + 8].view(np.float64)[0])
            terminated = bool(h[po["term"]])
            vals = h[po["info"]:po["info"] + 8 * N_INFO].view(np.float64).copy()
            self._host_obs_valid = self.history == 1                  # (history > 1 stacks frames on the device: get_obs() goes there)
        else:
            a = torch.as_tensor(np.asarray(actions, dtype=np.float64).reshape(1, -1))
            r, t, info = self._b.step(a, add_noise=add_noise)
            reward = float(r[0].item())
            terminated = bool(t[0].item())
            vals = info[0].cpu().numpy()
        self.steps += 1
        self.sum_rewards += reward
        if terminated:
            print(f"Episode terminated at time: {self.steps} with return: {self.sum_rewards:2.4f}.")   # :209
        return reward, terminated, {k: float(v) for k, v in zip(INFO_KEYS, vals)}

    def _obs_list(self, obs):
        o = obs[0].double().cpu().numpy()
        return [o[i].copy() for i in range(o.shape[0])]

    def get_obs(self):
        if self._packed and self._host_obs_valid:                                           # the obs of the last step() is on the host
            n, o1 = self.n_agents, self._b._obs_size1
            o = self._pk_np[self._pk_off["obs"]:].view(np.float64).reshape(n, o1)
            return [o[i].copy() for i in range(n)]
        return self._obs_list(self._b.get_obs())

    def get_obs_agent(self, agent_id):
        return self.get_obs()[agent_id]

    def get_state(self):
        return self._b.get_state()[0].cpu().numpy()

    def get_obs_size(self):
        return self.obs_size

    def get_state_size(self):
        return self.state_size

    def get_action(self):
        """uniform over [low, high) — :334-338 (host numpy RNG, as in the reference)"""
        return np.random.uniform(low=self.action_space.low, high=self.action_space.high, size=(self._b.n_sgen,))

    def get_total_actions(self):
        return self.n_actions

    def get_avail_actions(self):
        return np.expand_dims(np.array([self.get_avail_agent_actions(i) for i in range(self.n_agents)]), axis=0)

    def get_avail_agent_actions(self, agent_id):
        return [1]                                                    # :356-357

    def get_num_of_agents(self):
        return self.n_agents

    def _res(self, key):
        return self._b.results((key,))[key][0].cpu().numpy()

    def _get_voltage(self):
        return self._res("vm_pu")

    def _get_res_bus_v(self):
        return self._res("vm_pu")

    def _get_res_bus_active(self):
        return self._res("p_mw")

    def _get_res_bus_reactive(self):
        return self._res("q_mvar")

    def _get_res_line_loss(self):
        return self._res("pl_mw")

    def _get_sgen_active(self):
        return self._res("sgen_p")

    def _get_sgen_reactive(self):
        return self._res("sgen_q")

    def render(self, mode="human"):
        raise NotImplementedError("rendering (pyglet/plotly GUI of the reference) is out of scope")

    def close(self):
        self._b.close()
