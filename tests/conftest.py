import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Both native pieces are built in-tree: the product (hipcc, gfx950 cross-compile works
    without a GPU) and the test oracle (gcc)."""
    from mujoco_sim_amd import build

    build.build()
    build.build_oracle()


@pytest.fixture(scope="session")
def lib():
    from mujoco_sim_amd import capi

    return capi.load()


def has_gpu():
    try:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False
