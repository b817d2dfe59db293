import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# kd-trees of kd-trees go through the per-tree kernels (rpt_nest_trace) only when a pass is large enough to pay for their
# launches (api.cpp nest_min_paths, 6 Mi paths); the fixtures are a few thousand paths, so the tests ask for that route at
# any size — test_small_passes_of_nest_scenes_walk_in_kernel covers the default
os.environ.setdefault("RPTGPU_NEST_MIN_PATHS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def gpu_available():
    from rpt_amd import device_count
    try:
        return device_count() > 0
    except Exception:
        return False
