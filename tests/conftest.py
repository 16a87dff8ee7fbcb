import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """libsslam_hip.so, built on demand (hipcc cross-compiles without a GPU)."""
    from semantic_slam_amd import _lib
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    return _lib.load_library()


@pytest.fixture(scope="session")
def gpu_lib(hip_lib):
    if hip_lib.sslam_device_count() < 1:
        pytest.fail("test marked gpu but no HIP device is visible (the product has no CPU fallback)")
    return hip_lib
