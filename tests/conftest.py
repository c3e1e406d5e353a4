import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a wedged GPU must fail the test quickly instead of hanging the whole run (pytest-timeout is installed)
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(180))


@pytest.fixture(scope="session")
def bsfm():
    import bundler_sfm_amd
    return bundler_sfm_amd


@pytest.fixture(scope="session")
def gpu_bsfm(bsfm):
    if bsfm.lib.bsfm_device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    return bsfm
