import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libr2l_hip.so must match the sources: (re)build it when a source or header changed (sha1 stamps make this a
    no-op otherwise).  There is no non-HIP fallback to fall back to, so a stale or missing library would fail every test."""
    from r2l_amd import build as B
    if os.path.exists(B.HIPCC):
        B.build(verbose=False)
    yield
