"""CPU: the Philox4x32-10 the wide kernels draw from (mapdn_amd/csrc/philox.hpp) compiled for the HOST — the header is written so
that g++ and hipcc compile the same source — pinned on the Random123 known-answer vectors (kat_vectors: zero / all-ones / pi-digits
counter and key), and the oracle's restatement (oracle/philox.py) pinned on the same vectors and on the compiled header for random
(env, draw, stream, block) counters, including the 53-bit uniform it feeds Box-Muller with."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import philox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = {"philox0": ((0, 0, 0, 0, 0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),       # Random123 kat_vectors, philox4x32 10 rounds
       "philox1": ((0xffffffff,) * 6, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
       "philox2": ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0), "d16cfe09 94fdcceb 5001e420 24126ea1")}


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("philox") / "philox_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mapdn_amd", "csrc"),
                        os.path.join(ROOT, "tests", "philox_check.cpp"), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, check=True)
    return dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())


def test_header_known_answers(exe):
    rep = _run(exe)
    for k, (_, want) in KAT.items():
        assert rep[k] == want


def test_oracle_known_answers():
    for args, want in KAT.values():
        x = philox.philox4x32_10(*args)
        assert " ".join(f"{int(np.asarray(v).reshape(-1)[0]):08x}" for v in x) == want


def test_header_equals_oracle_on_env_counters(exe):
    rng = np.random.default_rng(5)
    for _ in range(25):
        seed = int(rng.integers(0, 2**63))
        env, draw, block = (int(rng.integers(0, 2**32)) for _ in range(3))
        stream = int(rng.integers(0, 5))
        k0, k1 = philox._key(seed)
        rep = _run(exe, *(f"{v:x}" for v in (env, draw, stream, block, k0, k1)))
        x = [np.asarray(v).reshape(-1) for v in philox.philox4x32_10(env, draw, stream, block, k0, k1)]
        assert rep["block"] == " ".join(f"{int(v[0]):08x}" for v in x)
        u = [float(s) for s in rep["u53"].split()]
        assert u[0] == float(philox._u53(x[0], x[1])[0]) and u[1] == float(philox._u53(x[2], x[3])[0])
