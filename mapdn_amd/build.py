"""Build libmapdn_hip.so in-tree with hipcc for gfx950 (no torch, no cmake)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmapdn_hip.so")
SOURCES = ["plan.cpp", "kernels.hip", "dense.hip", "sparse.hip", "policy.hip", "capi.hip"]
HEADERS = ["plan.hpp", "kernels.hpp", "nr_common.hpp", os.path.join("..", "..", "include", "mapdn.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-o", tmp]
    cmd += os.environ.get("MAPDN_EXTRA_FLAGS", "").split()        # debug builds only (e.g. -DMAPDN_NR_STAMPS)
    # .cpp host files are compiled as plain C++ by hipcc; .hip as HIP
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)          # atomic: a concurrent loader never sees a half-written library
    return LIB


def build_locked(timeout: float = 600.0) -> str:
    """build() guarded by a lock file, for several processes (ranks) that find the library missing."""
    import time
    lock = LIB + ".lock"
    t0 = time.time()
    while True:
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
            break
        except FileExistsError:
            if os.path.exists(LIB) and not os.path.exists(lock):
                return LIB
            if time.time() - t0 > timeout:
                raise TimeoutError(f"waiting for {lock}")
            time.sleep(0.5)
            if os.path.exists(LIB) and not os.path.exists(lock):
                return LIB
    try:
        os.close(fd)
        return build() if _stale() else LIB
    finally:
        try:
            os.remove(lock)
        except OSError:
            pass


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
