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


R2L_SWITCHES = ("R2L_FORCE_VARIANT", "R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2", "R2L_COOPF_TILES", "R2L_DW_EXACT")


def use_family(monkeypatch, **cfg):
    """Select the kernel family of everything built from here on THROUGH r2l_config (include/r2l_hip.h): the keywords become
    the defaults of the AUTO fields of every engine's config (r2l_amd.engine.DEFAULT_CONFIG: precision, tiling, coop_tiles,
    dw_mode), handed to the *_cfg entry points call by call.  The R2L_* environment switches are cleared: they are the
    library's test / A-B overrides of AUTO fields and are exercised on purpose by
    tests/test_forward_gpu.py::test_explicit_config_selects_the_family only."""
    from r2l_amd import engine
    for k in R2L_SWITCHES:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(engine, "DEFAULT_CONFIG", dict(cfg))
