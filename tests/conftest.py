import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libmapdn_hip.so, built in-tree if stale (hipcc cross-compiles gfx950 without a GPU)."""
    from mapdn_amd import build, _lib
    build.build()
    return _lib.load()


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need the device: skip them (instead of failing in mapdn_create) on a box without one"""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no ROCm GPU on this box (run with -m gpu on the MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
