"""Build libmapdn_hip.so in-tree with hipcc for gfx950 (no torch, no cmake)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("MAPDN_BUILD_OUT") or os.path.join(HERE, "libmapdn_hip.so")   # MAPDN_BUILD_OUT: debug variants beside the product library
SOURCES = ["plan.cpp", "kernels.hip", "dense.hip", "sparse.hip", "policy.hip", "policy_bwd.hip", "critic.hip", "rollout.hip", "capi.hip"]
# k_nr_tree's instantiations (nr_inst_list.hpp) are compiled as NR_PARTS objects from ONE source, in parallel
NR_INST_SOURCE, NR_PARTS = "nr_inst.hip", 4
HEADERS = ["plan.hpp", "colstats.hpp", "rowtile.hpp", "kernels.hpp", "philox.hpp", "nrmath.hpp", "nr_common.hpp", "nr_tree.hpp", "nr_inst_list.hpp", NR_INST_SOURCE,
           os.path.join("..", "..", "include", "mapdn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
HASH_TAG = b"MAPDN_SRC_HASH="


def _extra_flags():
    return os.environ.get("MAPDN_EXTRA_FLAGS", "").split()        # debug builds only (e.g. -DMAPDN_NR_STAMPS)


def source_hash() -> str:
    """sha256 over the CONTENT of every source / header (by name, sorted) and the compiler flags: what the library on disk must
    have been built from.  Content, not mtimes: a checkout, a copy to another box or `touch` do not change it; an edit does."""
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(name).encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS + _extra_flags()).encode())
    return h.hexdigest()


def library_hash(path: str | None = None) -> str | None:
    """the source hash embedded in a built library (mapdn_build_info(), capi.hip), read from the file without loading it"""
    path = path or LIB
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    i = data.find(HASH_TAG)
    while i >= 0:
        digest = data[i + len(HASH_TAG): i + len(HASH_TAG) + 64]
        if len(digest) == 64 and all(c in b"0123456789abcdef" for c in digest):
            return digest.decode()
        i = data.find(HASH_TAG, i + 1)
    return None


def stale() -> bool:
    """True when the library is missing or was built from other sources / flags than the ones on disk"""
    return library_hash() != source_hash()


_stale = stale          # (older name)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc -c every translation unit (in parallel: the NR instantiations dominate), then link the shared library."""
    if not force and not stale():
        return LIB
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = _extra_flags()
    digest = source_hash()
    objdir = tempfile.mkdtemp(prefix="mapdn_build_")
    jobs = []                                                      # (command, object)
    for p in range(NR_PARTS):                                      # longest first
        o = os.path.join(objdir, f"nr_inst_{p}.o")
        jobs.append(([hipcc] + FLAGS + extra + [f"-DNR_INST_PART={p}", "-c", os.path.join(CSRC, NR_INST_SOURCE), "-o", o], o))
    for s in SOURCES:                                              # .cpp host files are compiled as plain C++ by hipcc; .hip as HIP
        o = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        more = [f'-DMAPDN_SRC_HASH="{digest}"'] if s == "capi.hip" else []
        jobs.append(([hipcc] + FLAGS + extra + more + ["-c", os.path.join(CSRC, s), "-o", o], o))

    def run(job):
        if verbose:
            print(" ".join(job[0]), file=sys.stderr)
        subprocess.run(job[0], check=True)
        return job[1]
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            objs = list(ex.map(run, jobs))
        tmp = LIB + f".tmp{os.getpid()}"
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [f for f in extra if f.startswith("-fsanitize") or f == "-fno-gpu-sanitize"] \
            + ["-o", tmp] + objs
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.run(link, check=True)
        if library_hash(tmp) != digest:
            raise RuntimeError("the linked library does not carry the source hash it was built with")
        os.replace(tmp, LIB)      # atomic: a concurrent loader never sees a half-written library
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return LIB


def build_locked(timeout: float = 900.0) -> str:
    """build() guarded by a lock file, for several processes (ranks) that find the library missing or stale at the same time:
    exactly one of them compiles, the others wait for it and then use its result."""
    import time
    lock = LIB + ".lock"
    t0 = time.time()
    while True:
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            if time.time() - t0 > timeout:
                raise TimeoutError(f"waiting for {lock} (left behind by a killed build? remove it)")
            time.sleep(0.25)
            if not os.path.exists(lock) and not stale():
                return LIB
            continue
        try:
            os.write(fd, str(os.getpid()).encode())
            os.close(fd)
            return build() if stale() else LIB
        finally:
            try:
                os.remove(lock)
            except OSError:
                pass


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
