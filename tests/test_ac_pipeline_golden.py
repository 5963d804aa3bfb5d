"""area-clearing's non-physics pipeline against the reference's own AreaClearingEnv code (tests/golden/make_golden_ac_pipeline.py):
configuration space, goal-point map, overhead map, observation, boxes_completed and obs_to_goal_difference ran in the reference's
code on scene states taken from the oracle, with the absent third-party primitives supplied by this repository's restatements."""
import hashlib
import json
import os

import numpy as np

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd.config import default_cfg
from oracle import oracle_bd as ob

HERE = os.path.dirname(__file__)
G = json.load(open(os.path.join(HERE, "golden", "ac_pipeline_golden.json")))
Z = np.load(os.path.join(HERE, "golden", "ac_pipeline_golden.npz"))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_reference_pipeline_on_oracle_states():
    for ci, c in enumerate(G["cases"]):
        cfg = default_cfg("area_clearing")
        cfg.env = c["layout"]
        trial = A.generate_trials(cfg, 2)[1]
        o = ob.OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
        o.reset(trial, observe=False)
        info = None
        for a in c["actions"]:
            _, _, _, _, info = o.step(a, observe=False)
        obs = o.observe()
        m = o.maps()
        assert _sha(m["cspace"]) == c["cspace_sha"] and _sha(np.stack([m["edt_i"], m["edt_j"]]).astype(np.int32)) == c["edt_sha"]
        assert _sha(m["small_free"]) == c["small_sha"] and _sha(m["recept"]) == c["goal_map_sha"] and _sha(m["overhead"]) == c["overhead_sha"]
        ref = Z["obs%d" % ci]
        for ch in range(4):
            assert int((ref[..., ch] != obs[..., ch]).sum()) <= 6, (c["layout"], ch)
        assert int(info["box_count"]) == c["num_completed"]
        assert abs(info["diff_reward"] - c["last_diff_reward"]) < 1e-12
        # the cleared boxes are drawn in their own class: count them through the overhead raster of the oracle
        assert ((m["overhead"] == np.float32(7 / 8)).any()) == any(c["statuses"])
