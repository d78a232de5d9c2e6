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
