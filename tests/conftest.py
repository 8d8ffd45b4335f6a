import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A device context; GPU tests FAIL (not skip) when the HIP library or device is missing."""
    from locityper_amd import api
    assert api.device_count() >= 1, "no HIP device visible: -m gpu tests must run on a GPU box"
    ctx = api.Context(0)
    yield ctx
    ctx.close()
