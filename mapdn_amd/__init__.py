"""mapdn_amd — MI355X-native batched `VoltageControl` hot path of Future-Power-Networks/MAPDN.

    env.py        VoltageControlBatch (B envs, device tensors) and VoltageControl (B = 1 drop-in)
    csrc/         HIP kernels + C ABI (include/mapdn.h -> libmapdn_hip.so), built by build.py
    netspec.py    NetSpec / Profiles (the pandapower columns and CSV tables the path reads), synthetic cases
    data.py       on-disk formats (reference CSV layout, netspec.npz, scenario directories)
    sharding.py   env-batch sharding over GPUs, end-of-rollout gather
    rollout.py, replay.py, learner.py, tester.py   the callers either side of the path

Nothing here imports torch or loads the shared library at package-import time; there is no CPU fallback.
"""
__all__ = ["env", "netspec", "data", "sharding", "rollout", "replay", "learner", "tester", "marl_env_api", "build"]
