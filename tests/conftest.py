import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {k: np.load(os.path.join(d, k + ".npz")) for k in ("ops", "stack", "segment", "segment_full", "ops_unsigned", "segment_whu")
            if os.path.exists(os.path.join(d, k + ".npz"))}


@pytest.fixture
def tuning_env(monkeypatch):
    """Set a tuning switch of libsemstereo_hip.so for one test.  The library reads its switches from the environment once
    per process (ss::tuning()), so a change needs ss_reload_tuning(); undone (and reloaded) at teardown."""
    import semstereo_amd
    lib = semstereo_amd._lib.load()

    def setenv(name, value):
        monkeypatch.setenv(name, value)
        assert lib.ss_reload_tuning() == 0
    yield setenv
    monkeypatch.undo()
    lib.ss_reload_tuning()


@pytest.fixture
def deferral_on(monkeypatch):
    """Tests ABOUT the deferred handles (which rules fire under an untouched forward()) switch them on for their duration,
    whatever SS_DEFER says: the rest of the suite honours the switch."""
    from semstereo_amd import deferred as dfr
    monkeypatch.setattr(dfr, "ENABLED", True)
    return dfr
