"""CPU: the small elementary functions of the solver kernels (mapdn_amd/csrc/nrmath.hpp) compiled for the HOST — the header is
written so that g++ and hipcc compile the same source — and measured against long-double libm on the ranges the kernels use:
the Taylor sincos of a Newton angle step (|x| <= 0.5), the Cody-Waite form for large steps, the bowl barrier inside its band."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("nrmath") / "nrmath_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-I", os.path.join(ROOT, "mapdn_amd", "csrc"),
                        os.path.join(ROOT, "tests", "nrmath_check.cpp"), "-o", exe, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("host compiler without -mfma: " + r.stderr[:200])
    run = subprocess.run([exe], capture_output=True, text=True)
    if run.returncode != 0:
        pytest.skip("host CPU without FMA")
    return dict(line.split(" ", 1) for line in run.stdout.strip().splitlines())


def test_polynomial_sincos_of_a_newton_step(report):
    assert float(report["sincos_small_rel"]) < 2.5e-16          # ~1 ulp on |x| <= 0.5


def test_large_step_sincos(report):
    """Cody-Waite reduction by pi/2 + the same Taylor pair: identical to the polynomial form inside |x| <= 0.5 (geometry independence
    does not depend on which path a wave took), <= 2e-16 absolute up to 1e5 rad, still accurate at 1e8"""
    assert report["mid_equals_small_inside"].strip() == "1"
    assert float(report["sincos_mid_abs_1e5"]) < 2e-16
    assert float(report["sincos_mid_abs_1e8"]) < 1e-15


def test_bowl_barrier_inside_its_band(report):
    assert float(report["bowl_abs"]) < 5e-17                   # values are 1e-4 .. 4e-3: ~1 ulp of 0.04
