import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy access to the fixtures captured from the reference (tests/golden/gen_golden.py)."""

    def __init__(self):
        self._files = {}
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            self.cases = json.load(f)
        extra = os.path.join(GOLDEN, "cases_cross.json")           # non-square alphabets (tests/golden/gen_golden_cross.py)
        if os.path.exists(extra):
            with open(extra) as f:
                self.cases.update(json.load(f))

    def __getitem__(self, name):
        if name not in self._files:
            self._files[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return self._files[name]

    def input(self, name, dtype):
        return np.ascontiguousarray(self["inputs"][name + "_E"].astype(dtype))


_GOLDEN = Golden()


@pytest.fixture(scope="session")
def golden():
    return _GOLDEN


def golden_cases(group):
    return _GOLDEN.cases[group]


class _Forms:
    """The library's form switches (``qh_set_form``: test hooks of include/qampy_hip.h) for the duration of a test: ``forms.set("trainer", "direct")``;
    everything a test set goes back to automatic when it ends.  (Rounds 1-5 steered the kernels through environment variables read in the launch paths.)"""

    def __init__(self):
        self._touched = set()

    def set(self, key, value):
        from qampy_amd import _lib
        _lib.set_form(key, value)
        self._touched.add(key)

    def reset(self, key=None):
        from qampy_amd import _lib
        for k in ([key] if key else list(self._touched)):
            _lib.set_form(k, None)
            self._touched.discard(k)


@pytest.fixture
def forms():
    f = _Forms()
    yield f
    f.reset()


CT = {"c64": np.complex64, "c128": np.complex128}
RT = {"c64": np.float32, "c128": np.float64}

# Stated tolerances.  complex128: the only differences are the order of a handful of roundings;
# complex64: the reference's own c64-vs-c128 spread is ~3e-6, a different summation order of the
# 82-term dot product moves single steps by a few float32 ulp which the LMS recursion keeps bounded.
TOL = {
    "c128": dict(rtol=1e-9, atol=1e-11),
    "c64": dict(rtol=1e-4, atol=1e-4),
}
