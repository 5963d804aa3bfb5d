import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def occ_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "occupancy_golden.npz"))


@pytest.fixture(scope="session")
def ship_cfg():
    from benchpush_amd.config import default_cfg, ship_ice_physics_params
    cfg = default_cfg("ship_ice")
    return cfg, ship_ice_physics_params(cfg)
