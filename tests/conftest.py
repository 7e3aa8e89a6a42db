import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svt-av1-1_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own kernels (oracle/_ref); present in the build container and, as a prebuilt
    .so, on the GPU box.  Tests that need it skip when it is absent."""
    from oracle.binding import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    return Reference()


@pytest.fixture(scope="session")
def hip_ctx():
    import svtav1_hip
    ctx = svtav1_hip.Context(0)
    yield ctx
    ctx.close()
