"""ctypes binding of libmapdn_hip.so (C ABI: include/mapdn.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.  Nothing in
the product path imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .netspec import NetSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmapdn_hip.so")

N_INFO = 11
INFO_KEYS = (
    "percentage_of_v_out_of_control", "percentage_of_lower_than_lower_v",
    "percentage_of_higher_than_upper_v", "totally_controllable_ratio",
    "average_voltage_deviation", "average_voltage", "max_voltage_drop_deviation",
    "max_voltage_rise_deviation", "total_line_loss", "q_loss", "destroy",
)
BARRIER_IDS = dict(l1=0, l2=1, courant_beltrami=2, bowl=3, bump=4)
SS_BITS = dict(pv=1, demand=2, reactive=4, vm_pu=8, va_degree=16)
F32, F64 = 0, 1

EXPORTS = (
    "mapdn_last_error", "mapdn_create", "mapdn_destroy", "mapdn_dims", "mapdn_set_profiles", "mapdn_reset",
    "mapdn_step", "mapdn_get_start_rows", "mapdn_get_returns", "mapdn_get_obs", "mapdn_get_state", "mapdn_get_results", "mapdn_get_loads",
    "mapdn_solve_only", "mapdn_get_ybus_dense", "mapdn_get_obs_index", "mapdn_get_schedule", "mapdn_get_flat_factors", "mapdn_stats", "mapdn_nr_timing",
    "mapdn_nr_time_ms", "mapdn_get_auto_reset_mask", "mapdn_dense_solve", "mapdn_step_obs", "mapdn_get_sparse_program", "mapdn_policy_forward",
    "mapdn_policy_forward_fits", "mapdn_layernorm64_forward", "mapdn_layernorm64_backward", "mapdn_layernorm64_backward_blocks",
    "mapdn_get_nr_geometry", "mapdn_debug_stream", "mapdn_build_info", "mapdn_layernorm64_bc_forward", "mapdn_layernorm64_bc_backward", "mapdn_relu_dot64_forward", "mapdn_relu_dot64_backward",
    "mapdn_critic_head_forward", "mapdn_critic_head_scratch_floats", "mapdn_critic_head_backward", "mapdn_critic_head_backward_dot", "mapdn_critic_head_mse", "mapdn_get_profile_stats",
    "mapdn_explore_actions", "mapdn_rollout_stats", "mapdn_copy_segments",
    "mapdn_policy_forward_train", "mapdn_policy_backward", "mapdn_policy_backward_scratch_floats",
)

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)


class CNetSpec(C.Structure):
    _fields_ = [
        ("n_bus", C.c_int32), ("bus_vn_kv", _pd), ("bus_zone", _pi),
        ("n_line", C.c_int32), ("line_from_bus", _pi), ("line_to_bus", _pi),
        ("line_r_ohm_per_km", _pd), ("line_x_ohm_per_km", _pd), ("line_c_nf_per_km", _pd),
        ("line_g_us_per_km", _pd), ("line_length_km", _pd), ("line_parallel", _pi), ("line_in_service", _pu8),
        ("n_branch_pu", C.c_int32), ("br_from_bus", _pi), ("br_to_bus", _pi), ("br_r_pu", _pd), ("br_x_pu", _pd),
        ("br_b_pu", _pd), ("br_ratio", _pd), ("br_shift_deg", _pd),
        ("n_shunt", C.c_int32), ("shunt_bus", _pi), ("shunt_p_mw", _pd), ("shunt_q_mvar", _pd),
        ("n_load", C.c_int32), ("load_bus", _pi),
        ("n_sgen", C.c_int32), ("sgen_bus", _pi), ("sgen_zone", _pi),
        ("ext_grid_bus", C.c_int32), ("ext_grid_vm_pu", C.c_double), ("sn_mva", C.c_double), ("f_hz", C.c_double),
        ("br_g_pu", _pd), ("load_scaling", _pd), ("sgen_scaling", _pd), ("bus_alias", _pi),
    ]


class CEnvConfig(C.Structure):
    _fields_ = [
        ("barrier_type", C.c_int32), ("voltage_weight", C.c_double), ("q_weight", C.c_double),
        ("line_weight", C.c_double), ("use_line_weight", C.c_int32), ("use_q_weight", C.c_int32),
        ("v_lower", C.c_double), ("v_upper", C.c_double), ("episode_limit", C.c_int32),
        ("action_low", C.c_double), ("action_high", C.c_double), ("reset_action", C.c_int32),
        ("state_space", C.c_int32), ("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("auto_reset", C.c_int32),
        # launch / solver tuning (0 = automatic) and the runpp options the reference leaves at their defaults — include/mapdn.h
        ("nr_solver", C.c_int32), ("nr_waves", C.c_int32), ("nr_lanes", C.c_int32), ("nr_lean", C.c_int32),
        ("nr_h_lds", C.c_int32), ("nr_g_lds", C.c_int32), ("nr_rec_lds", C.c_int32), ("nr_flat_lds", C.c_int32), ("nr_line_lds", C.c_int32),
        ("nr_mm_pass", C.c_int32), ("sp_lanes", C.c_int32), ("inject_full", C.c_int32),
        ("nr_check_dx", C.c_double), ("nr_check_quad", C.c_double), ("debug_geometry", C.c_int32),
        ("tolerance_mva", C.c_double), ("tolerance_is_pu", C.c_int32), ("nr_init", C.c_int32),
        ("fuse_inject", C.c_int32), ("overlap_advance", C.c_int32), ("xcd_map", C.c_int32),
    ]


# keys of the `tuning` dict of VoltageControlBatch (== the appended fields of mapdn_env_config)
TUNING_INT = ("nr_solver", "nr_waves", "nr_lanes", "nr_lean", "nr_h_lds", "nr_g_lds", "nr_rec_lds", "nr_flat_lds", "nr_line_lds",
              "nr_mm_pass", "sp_lanes", "inject_full", "debug_geometry", "tolerance_is_pu", "nr_init", "fuse_inject", "overlap_advance", "xcd_map")
TUNING_F64 = ("nr_check_dx", "nr_check_quad", "tolerance_mva")
NR_SOLVERS = dict(auto=0, tree=0, sparse=1, dense=2)
GEOMETRY_KEYS = ("solver", "waves", "lanes", "lean", "rows", "h_lds", "g_lds", "rec_lds", "flat_lds", "line_lds", "mm_pass",
                 "lds_bytes", "workgroups", "resident_per_cu", "rounds", "model_ns", "fuse_inject", "n_fused_buses", "n_nodes", "reserved")


class CDims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "n_envs", "n_bus", "n_line", "n_load", "n_sgen", "n_agents", "n_actions", "obs_size", "state_size",
        "n_info", "is_radial", "max_zone_size")]


class MapdnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmapdn_hip error {code}: {msg}")
        self.code = code


_lib = None


def _safe_source_hash(b):
    try:
        return b.source_hash()
    except OSError:
        return "sources unreadable"


def load():
    """Load the shared library (once).  Raises if it was not built — never falls back to CPU."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it FIRST so that our
    # NEEDED libamdhip64.so.7 resolves to the runtime torch already loaded: one HIP runtime per
    # process, shared streams and device pointers.  (Loading ours first drags in /opt/rocm's copy
    # and the second runtime then fails with "no ROCm-capable device".)
    import torch  # noqa: F401
    if "MAPDN_LIB_PATH" not in os.environ:
        # not a fallback: a missing library — or one whose embedded source hash differs from the sources beside it (a prebuilt
        # .so shipped with edited sources) — is compiled from the same HIP sources in-tree (hipcc cross-compiles gfx950), once,
        # race-free when several ranks start together; raise if that is impossible
        from . import build as _build
        try:
            is_stale = _build.stale()
        except OSError:                       # a binary-only install: no csrc/ beside the library to hash — the library is what there is
            is_stale = not os.path.exists(LIB_PATH)
            if not is_stale:
                import warnings
                warnings.warn(f"mapdn_amd/csrc is not readable: {LIB_PATH} is loaded without the source-hash check", RuntimeWarning, stacklevel=2)
        if is_stale:
            try:
                _build.build_locked()
            except Exception as exc:
                raise ImportError(
                    f"{LIB_PATH} is missing or was built from other sources than mapdn_amd/csrc (hash {_build.library_hash()} vs "
                    f"{_safe_source_hash(_build)}) and could not be rebuilt ({exc}); build it with `python -m mapdn_amd.build` "
                    "(hipcc --offload-arch=gfx950); there is no CPU fallback") from exc
    lib = C.CDLL(os.environ.get("MAPDN_LIB_PATH", LIB_PATH))      # override: A/B experiments with debug builds only
    if "MAPDN_LIB_PATH" in os.environ:                              # an OLDER build for a same-box A/B may lack the newest exports
        for name in EXPORTS:
            if not hasattr(lib, name):
                def _missing(*a, _n=name, **k):
                    raise AttributeError(f"{os.environ['MAPDN_LIB_PATH']} does not export {_n}")
                lib.__dict__[name] = _missing
    vp = C.c_void_p
    lib.mapdn_last_error.restype = C.c_char_p
    lib.mapdn_last_error.argtypes = [vp]
    lib.mapdn_build_info.restype = C.c_char_p
    lib.mapdn_build_info.argtypes = []
    lib.mapdn_create.argtypes = [C.POINTER(CNetSpec), C.POINTER(CEnvConfig), C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.mapdn_destroy.argtypes = [vp]
    lib.mapdn_destroy.restype = None
    lib.mapdn_dims.argtypes = [vp, C.POINTER(CDims)]
    lib.mapdn_set_profiles.argtypes = [vp, _pd, _pd, _pd, C.c_int64, C.c_int32, C.c_int32]
    lib.mapdn_get_profile_stats.argtypes = [vp, _pd, _pd]
    lib.mapdn_reset.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    lib.mapdn_step.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.mapdn_step_obs.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int32, vp]
    lib.mapdn_get_start_rows.argtypes = [vp, vp, vp]
    lib.mapdn_get_returns.argtypes = [vp, vp, vp]
    lib.mapdn_get_auto_reset_mask.argtypes = [vp, vp, vp]
    lib.mapdn_get_obs.argtypes = [vp, vp, C.c_int32, vp]
    lib.mapdn_get_state.argtypes = [vp, vp, C.c_int32, vp]
    lib.mapdn_get_results.argtypes = [vp] + [vp] * 7 + [vp]
    lib.mapdn_get_loads.argtypes = [vp, vp, vp, vp]
    lib.mapdn_solve_only.argtypes = [vp] + [vp] * 8 + [vp]
    lib.mapdn_get_ybus_dense.argtypes = [vp, _pd]
    lib.mapdn_get_obs_index.argtypes = [vp, _pi, _pi]
    lib.mapdn_get_schedule.argtypes = [vp, C.c_int32, _pi, _pi, _pi]
    lib.mapdn_get_flat_factors.argtypes = [vp, _pd, _pi]
    lib.mapdn_get_nr_geometry.argtypes = [vp, _pi]
    lib.mapdn_debug_stream.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.mapdn_get_sparse_program.argtypes = [vp, C.c_int32, _pi, _pi, _pi, _pi]
    lib.mapdn_policy_forward.argtypes = [vp] * 14 + [C.c_int32] * 4 + [C.c_float, vp]
    lib.mapdn_policy_forward_fits.argtypes = [C.c_int32, C.c_int32]
    lib.mapdn_policy_forward_train.argtypes = [vp] * 15 + [C.c_int32] * 4 + [C.c_float, vp]
    lib.mapdn_policy_backward_scratch_floats.argtypes = [C.c_int64]
    lib.mapdn_policy_backward_scratch_floats.restype = C.c_int64
    lib.mapdn_policy_backward.argtypes = [vp] * 5 + [C.c_float] + [vp] * 10 + [C.c_int64, vp]
    lib.mapdn_layernorm64_forward.argtypes = [vp] * 6 + [C.c_int64, C.c_float, C.c_int32, vp]
    lib.mapdn_layernorm64_backward_blocks.argtypes = [C.c_int64]
    lib.mapdn_layernorm64_backward.argtypes = [vp] * 10 + [C.c_int64, C.c_int32, vp]
    lib.mapdn_layernorm64_bc_forward.argtypes = [vp, vp, C.c_int32] + [vp] * 5 + [C.c_int64, C.c_float, C.c_int32, vp]
    lib.mapdn_layernorm64_bc_backward.argtypes = [vp, vp, vp, C.c_int32] + [vp] * 8 + [C.c_int64, C.c_int32, vp]
    lib.mapdn_relu_dot64_forward.argtypes = [vp, vp, C.c_float, vp, C.c_int64, vp]
    lib.mapdn_relu_dot64_backward.argtypes = [vp] * 7 + [C.c_int64, vp]
    lib.mapdn_critic_head_forward.argtypes = [vp, vp, C.c_int32, vp, vp, C.c_float] + [vp] * 5 + [C.c_int64, vp]
    lib.mapdn_critic_head_scratch_floats.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.mapdn_critic_head_scratch_floats.restype = C.c_int64
    lib.mapdn_critic_head_backward.argtypes = [vp, vp, vp, C.c_int32, vp, vp, C.c_float] + [vp] * 7 + [C.c_int64, C.c_int32, vp]
    lib.mapdn_explore_actions.argtypes = [vp, vp, vp, C.c_float, C.c_int32, C.c_double, C.c_double, vp, vp, vp, C.c_int64, vp]
    lib.mapdn_rollout_stats.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, vp]
    lib.mapdn_copy_segments.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64), C.c_int32, vp]
    lib.mapdn_critic_head_mse.argtypes = [vp] * 5 + [C.c_int32, vp, vp, C.c_float] + [vp] * 7 + [C.c_int64, vp]
    lib.mapdn_critic_head_backward_dot.argtypes = [vp, vp, vp, C.c_int32, vp, vp, C.c_float] + [vp] * 6 + [C.c_int64, vp]
    lib.mapdn_dense_solve.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp]
    lib.mapdn_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32), vp]
    lib.mapdn_nr_timing.argtypes = [vp, C.c_int32]
    lib.mapdn_nr_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    for name in EXPORTS:
        if name not in ("mapdn_last_error", "mapdn_destroy", "mapdn_build_info"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != 0:
        msg = load().mapdn_last_error(handle)
        raise MapdnError(rc, msg.decode() if msg else "?")


def _p(a, typ):
    return a.ctypes.data_as(typ) if a.size else C.cast(None, typ)


def make_cnetspec(net: NetSpec):
    """Returns (CNetSpec, keepalive) — keepalive holds the numpy arrays the struct points into."""
    keep = net  # NetSpec arrays are contiguous with the right dtypes (NetSpec.__post_init__)
    s = CNetSpec()
    s.n_bus = net.n_bus
    s.bus_vn_kv = _p(net.bus_vn_kv, _pd)
    s.bus_zone = _p(net.bus_zone, _pi)
    s.n_line = net.n_line
    s.line_from_bus = _p(net.line_from_bus, _pi)
    s.line_to_bus = _p(net.line_to_bus, _pi)
    s.line_r_ohm_per_km = _p(net.line_r_ohm_per_km, _pd)
    s.line_x_ohm_per_km = _p(net.line_x_ohm_per_km, _pd)
    s.line_c_nf_per_km = _p(net.line_c_nf_per_km, _pd)
    s.line_g_us_per_km = _p(net.line_g_us_per_km, _pd)
    s.line_length_km = _p(net.line_length_km, _pd)
    s.line_parallel = _p(net.line_parallel, _pi)
    s.line_in_service = _p(net.line_in_service, _pu8)
    s.n_branch_pu = net.n_branch_pu
    s.br_from_bus = _p(net.br_from_bus, _pi)
    s.br_to_bus = _p(net.br_to_bus, _pi)
    s.br_r_pu = _p(net.br_r_pu, _pd)
    s.br_x_pu = _p(net.br_x_pu, _pd)
    s.br_b_pu = _p(net.br_b_pu, _pd)
    s.br_ratio = _p(net.br_ratio, _pd)
    s.br_shift_deg = _p(net.br_shift_deg, _pd)
    s.n_shunt = int(net.shunt_bus.shape[0])
    s.shunt_bus = _p(net.shunt_bus, _pi)
    s.shunt_p_mw = _p(net.shunt_p_mw, _pd)
    s.shunt_q_mvar = _p(net.shunt_q_mvar, _pd)
    s.n_load = net.n_load
    s.load_bus = _p(net.load_bus, _pi)
    s.n_sgen = net.n_sgen
    s.sgen_bus = _p(net.sgen_bus, _pi)
    s.sgen_zone = _p(net.sgen_zone, _pi)
    s.ext_grid_bus = int(net.ext_grid_bus)
    s.ext_grid_vm_pu = float(net.ext_grid_vm_pu)
    s.sn_mva = float(net.sn_mva)
    s.f_hz = float(net.f_hz)
    s.br_g_pu = _p(net.br_g_pu, _pd)
    s.load_scaling = _p(net.load_scaling, _pd)
    s.sgen_scaling = _p(net.sgen_scaling, _pd)
    s.bus_alias = _p(net.bus_alias, _pi) if net.has_fused_buses else C.cast(None, _pi)
    return s, keep


def make_cconfig(args: dict, env_id_offset: int = 0, tuning: dict | None = None) -> CEnvConfig:
    """mapdn_env_config from the reference's constructor kwargs; `tuning` fills the appended launch / solver fields
    (include/mapdn.h: nr_waves, nr_lanes, nr_lean, nr_*_lds, nr_mm_pass, nr_solver = 'sparse' | 'dense', nr_init, tolerance_mva ...)."""
    c = CEnvConfig()
    for k, v in (tuning or {}).items():
        if k == "nr_solver" and isinstance(v, str):
            v = NR_SOLVERS[v]
        if k in TUNING_INT:
            setattr(c, k, int(v))
        elif k in TUNING_F64:
            setattr(c, k, float(v))
        else:
            raise KeyError(f"unknown tuning key {k!r} (known: {TUNING_INT + TUNING_F64})")
    bt = args.get("voltage_barrier_type", "l1")
    if bt not in BARRIER_IDS:
        raise KeyError(bt)   # reference: Voltage_Barrier[name] KeyError (voltage_barrier_backend.py:8)
    c.barrier_type = BARRIER_IDS[bt]
    c.voltage_weight = float(args.get("voltage_weight", 1.0))
    qw, lw = args.get("q_weight", 0.1), args.get("line_weight", None)
    c.q_weight = 0.0 if qw is None else float(qw)
    c.line_weight = 0.0 if lw is None else float(lw)
    c.use_q_weight = int(qw is not None)
    c.use_line_weight = int(lw is not None)
    c.v_lower = float(args.get("v_lower", 0.95))
    c.v_upper = float(args.get("v_upper", 1.05))
    c.episode_limit = int(args["episode_limit"])
    c.action_low = -float(args["action_scale"]) + float(args["action_bias"])
    c.action_high = float(args["action_scale"]) + float(args["action_bias"])
    c.reset_action = int(bool(args.get("reset_action", True)))
    ss = args.get("state_space", ["pv", "demand", "reactive", "vm_pu", "va_degree"])
    c.state_space = sum(SS_BITS[k] for k in set(ss) if k in SS_BITS)
    c.seed = int(args.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF
    c.env_id_offset = int(env_id_offset)
    c.auto_reset = int(bool(args.get("auto_reset", False)))
    return c


def nr_geometry(handle) -> dict:
    """what mapdn_create settled on for this handle (mapdn_get_nr_geometry)"""
    out = (C.c_int32 * 20)()
    check(load().mapdn_get_nr_geometry(handle, out), handle)
    return dict(zip(GEOMETRY_KEYS, [int(x) for x in out]))
