"""Build libmapdn_hip.so in-tree with hipcc for gfx950 (no torch, no cmake)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("MAPDN_BUILD_OUT") or os.path.join(HERE, "libmapdn_hip.so")   # MAPDN_BUILD_OUT: debug variants beside the product library
SOURCES = ["plan.cpp", "kernels.hip", "dense.hip", "sparse.hip", "policy.hip", "capi.hip"]
# k_nr_tree's instantiations (nr_inst_list.hpp) are compiled as NR_PARTS objects from ONE source, in parallel
NR_INST_SOURCE, NR_PARTS = "nr_inst.hip", 4
HEADERS = ["plan.hpp", "kernels.hpp", "philox.hpp", "nrmath.hpp", "nr_common.hpp", "nr_tree.hpp", "nr_inst_list.hpp", NR_INST_SOURCE,
           os.path.join("..", "..", "include", "mapdn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc -c every translation unit (in parallel: the NR instantiations dominate), then link the shared library."""
    if not force and not _stale():
        return LIB
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MAPDN_EXTRA_FLAGS", "").split()        # debug builds only (e.g. -DMAPDN_NR_STAMPS)
    objdir = tempfile.mkdtemp(prefix="mapdn_build_")
    jobs = []                                                      # (command, object)
    for p in range(NR_PARTS):                                      # longest first
        o = os.path.join(objdir, f"nr_inst_{p}.o")
        jobs.append(([hipcc] + FLAGS + extra + [f"-DNR_INST_PART={p}", "-c", os.path.join(CSRC, NR_INST_SOURCE), "-o", o], o))
    for s in SOURCES:                                              # .cpp host files are compiled as plain C++ by hipcc; .hip as HIP
        o = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        jobs.append(([hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, s), "-o", o], o))

    def run(job):
        if verbose:
            print(" ".join(job[0]), file=sys.stderr)
        subprocess.run(job[0], check=True)
        return job[1]
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            objs = list(ex.map(run, jobs))
        tmp = LIB + f".tmp{os.getpid()}"
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.run(link, check=True)
        os.replace(tmp, LIB)      # atomic: a concurrent loader never sees a half-written library
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
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
