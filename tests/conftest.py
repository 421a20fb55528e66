import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture
def golden():
    return load_golden


class _Switches:
    """The library reads its NRHIP_* A/B switches once, at load (csrc/common.h: struct Tuning); a test that flips one sets the
    variable AND has the library read it again -- and once more when the test is over and the variable is restored."""

    def __init__(self, monkeypatch):
        self.mp = monkeypatch

    def _reload(self):
        from neurad_studio_amd import ops

        ops.reload_tuning()

    def set(self, name, value):
        self.mp.setenv(name, value)
        self._reload()

    def unset(self, name):
        self.mp.delenv(name, raising=False)
        self._reload()


@pytest.fixture
def switches(monkeypatch):
    s = _Switches(monkeypatch)
    yield s
    monkeypatch.undo()
    s._reload()
