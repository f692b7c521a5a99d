"""The environment protocol the MARL training loop programs against (the drop-in boundary).

The reference's base class (environments/multiagentenv.py:1-67, PyMARL heritage) is a list of methods that
raise NotImplementedError.  Here the same protocol is written down as data — which calls a caller may make,
what they return, and whether an env must provide them — and the base class is generated from that table, so
`isinstance(env, MultiAgentEnv)` and the reference's call sites keep working while the table doubles as the
checklist `tests/test_capi_cpu.py` uses to verify that `VoltageControl` covers the whole surface.
"""
from __future__ import annotations

from typing import Dict, NamedTuple, Tuple


class Call(NamedTuple):
    returns: str          # what the reference's callers expect back
    required: bool        # False: the base class supplies a harmless default
    args: Tuple[str, ...] = ()


#: method name -> contract (reference line numbers refer to environments/multiagentenv.py)
PROTOCOL: Dict[str, Call] = {
    "reset": Call("(list of per-agent observations, global state)", True),                      # :41-43
    "step": Call("(reward, terminated, info dict)", True, ("actions",)),                        # :3-5
    "get_obs": Call("list with one observation vector per agent", True),                        # :7-9
    "get_obs_agent": Call("observation vector of one agent", True, ("agent_id",)),              # :11-13
    "get_obs_size": Call("length of an observation vector", True),                              # :15-17
    "get_state": Call("global state vector", True),                                             # :19-20
    "get_state_size": Call("length of the global state vector", True),                          # :22-24
    "get_avail_actions": Call("availability mask for every agent", True),                       # :26-27
    "get_avail_agent_actions": Call("availability mask of one agent", True, ("agent_id",)),     # :29-31
    "get_total_actions": Call("number of actions per agent", True),                             # :33-36
    "render": Call("None", True),                                                               # :45-46
    "seed": Call("None", True),                                                                 # :51-52
    "save_replay": Call("None", True),                                                          # :54-55
    "get_stats": Call("dict", True),                                                            # :38-39 (unused by MAPDN)
    "get_agg_stats": Call("dict", False, ("stats",)),                                           # default {}
    "close": Call("None", False),                                                               # default no-op
}


def _missing(name: str, call: Call):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.{name}() -> {call.returns}")
    method.__name__ = name
    method.__doc__ = f"returns {call.returns}"
    return method


class MultiAgentEnv:
    """Base class with the reference's method set; subclasses override what PROTOCOL marks required."""

    n_agents: int
    episode_limit: int

    def get_agg_stats(self, stats):
        return {}

    def close(self):
        return None

    def get_env_info(self):
        """sizes a learner needs to build its networks (multiagentenv.py:57-67)"""
        info = dict(n_agents=self.n_agents, episode_limit=self.episode_limit)
        info.update(state_shape=self.get_state_size(), obs_shape=self.get_obs_size(), n_actions=self.get_total_actions())
        return info


for _name, _call in PROTOCOL.items():
    if _call.required:
        setattr(MultiAgentEnv, _name, _missing(_name, _call))
