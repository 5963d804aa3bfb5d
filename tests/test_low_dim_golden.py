"""Low-dimensional observations of the gym adapters vs. the reference classes' generate_observation_low_dim
(tests/golden/make_golden_low_dim.py; maze_NAMO_env.py:488-504, box_delivery_env.py:1025-1037, area_clearing.py:908-919,
ship_ice_env.py:358-370)."""
import json
import os

import numpy as np
import pytest

from benchpush_amd.envs.box_delivery import low_dim_observation
from benchpush_amd.envs.maze_namo import MazeNAMO

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "low_dim_golden.json")) as f:
        return json.load(f)


def test_box_vectors_match_reference(golden):
    P = [np.asarray(p) for p in golden["polys"]]
    for key in ("box_delivery", "area_clearing", "ship_ice"):
        assert np.array_equal(low_dim_observation(P), np.asarray(golden[key])), key


def test_maze_vector_keeps_the_reference_layout(golden):
    P = [np.asarray(p) for p in golden["polys"]]
    got = MazeNAMO._low_dim(None, P, golden["robot"])
    ref = np.asarray(golden["maze"])
    assert np.array_equal(got, ref)
    assert got[0] == golden["robot"][0] and got[1] == golden["robot"][1]
    assert got[-2] == 0.0 and got[-1] == 0.0          # the reference's loop never writes the last pair (nor obstacle 0)
