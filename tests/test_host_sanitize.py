"""CPU AddressSanitizer + UndefinedBehaviorSanitizer pass over the host-side product code (bp_host_geom.hpp, bp_host_bd.hpp: loaders' hulls and mass
properties, fillPoly, disk dilation, EDT indices, spfa, maze maps), which is otherwise compiled only inside the .hip translation unit.
tools/host_sanitize/host_sanitize.cpp replays the load-time sequences of bp_load_scenarios / bp_load_maze / bp_bd_load on the arrays the Python envs
hand to those entry points, for every shipped layout family: 100 ship-ice trials at 10-50 %, both maze versions, the four box-delivery obstacle
configurations and the four area-clearing layouts.  GPU sanitizers are not available on the pool; this is the CPU build only."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "host_sanitize", "host_sanitize.cpp")


def _b(a, dt):
    return np.ascontiguousarray(a, dt).tobytes()


def _ship_records():
    from benchpush_amd import _lib
    from benchpush_amd.config import default_cfg, merge_user_cfg, ship_ice_physics_params
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.scenario import pack_trials
    out = []
    for conc in (0.1, 0.2, 0.3, 0.4, 0.5):
        cfg = merge_user_cfg(default_cfg("ship_ice"), {"concentration": conc})
        bc = _lib.make_config(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
        pk = pack_trials(default_trials(conc, 20, base_seed=int(conc * 100)), max_verts=_lib.MAXV)
        T, F, V = pk["verts"].shape[:3]
        out.append(struct.pack("<i", 1) + bytes(bc) + struct.pack("<iii", T, F, V) + _b(pk["verts"], np.float64) + _b(pk["counts"], np.int32) +
                   _b(pk["centres"], np.float64) + _b(pk["starts"], np.float64) + _b(pk["nfloes"], np.int32))
    return out


def _maze_records():
    from benchpush_amd import _lib
    from benchpush_amd.config import default_cfg, maze_physics_params, maze_walls, merge_user_cfg
    from benchpush_amd.envs.maze_namo import _maze_cfg
    from benchpush_amd.maze_scenario import generate_layout
    out = []
    for version, nbox, rs in ((1, 20, False), (2, 8, True)):
        cfg = _maze_cfg({"maze_version": version, "num_obstacles": nbox, "random_start": rs})
        P = maze_physics_params(cfg)
        rv = cfg.robot.vertices
        head = ((rv[0][0] + rv[3][0]) / 2, (rv[0][1] + rv[3][1]) / 2)
        tail = ((rv[1][0] + rv[2][0]) / 2, (rv[1][1] + rv[2][1]) / 2)
        bc = _lib.make_config(P, rv, head, tail, env_kind=_lib.ENV_MAZE, wheel_vertices=cfg.robot.wheel_vertices, obstacle_size=cfg.obstacle_size)
        walls = np.ascontiguousarray(maze_walls(cfg), np.float64)
        layouts = [generate_layout(cfg, walls, 3 + t) for t in range(6)]
        centres = np.stack([np.asarray(l["centres"], np.float64).reshape(nbox, 2) for l in layouts])
        start = np.stack([np.asarray(l["start"], np.float64).reshape(3) for l in layouts])
        gh, gw = int(round(bc.map_h * bc.m_to_pix)), int(round(bc.map_w * bc.m_to_pix))
        out.append(struct.pack("<i", 2) + bytes(bc) + struct.pack("<iiiii", len(layouts), nbox, len(walls), gh, gw) + _b(centres, np.float64) +
                   _b(walls, np.float64) + _b(start, np.float64))
    return out


def _bd_record(bcfg, trials, nbox, ns):
    def pad(a, fill=0):
        o = np.full((ns,) + a.shape[1:], fill, a.dtype)
        o[: len(a)] = a
        return o
    starts = np.stack([t["start"] for t in trials])
    boxes = np.stack([t["boxes"] for t in trials])
    sv = np.stack([pad(np.asarray(t["statics"][0])) for t in trials])
    sc = np.stack([pad(np.asarray(t["statics"][1])) for t in trials])
    sp = np.stack([pad(np.asarray(t["statics"][2])) for t in trials])
    sr = np.stack([pad(np.asarray(t["statics"][3])) for t in trials])
    st = np.stack([pad(np.asarray(t["statics"][4]), 3) for t in trials])
    return (struct.pack("<i", 3) + bytes(bcfg) + struct.pack("<iii", len(trials), nbox, ns) + _b(starts, np.float64) + _b(boxes, np.float64) +
            _b(sv, np.float64) + _b(sc, np.int32) + _b(sp, np.float64) + _b(sr, np.float64) + _b(st, np.int32))


def _box_records():
    from benchpush_amd import _lib
    from benchpush_amd.box_delivery_scenario import box_delivery_params, box_delivery_physics_params, generate_trials
    from benchpush_amd.envs.box_delivery import _bd_cfg
    out = []
    for oc in ("small_empty", "small_columns", "large_columns", "large_divider"):
        cfg = _bd_cfg({"env": {"obstacle_config": oc}})
        bd = box_delivery_params(cfg)
        trials = generate_trials(cfg, 4, 5)
        nbox = len(trials[0]["boxes"])
        bd["num_boxes"] = nbox
        bcfg = _lib.make_bd_config(box_delivery_physics_params(cfg), bd, cfg)
        out.append(_bd_record(bcfg, trials, nbox, max(len(t["statics"][1]) for t in trials)))
    return out


def _area_records():
    from benchpush_amd import _lib
    from benchpush_amd.area_clearing_scenario import (area_clearing_params, area_clearing_physics_params, env_layout, generate_trials, goal_points)
    from benchpush_amd.envs.area_clearing import _ac_cfg
    out = []
    for name in ("clear_env", "clear_env_small", "walled_env", "walled_env_with_columns"):
        cfg = _ac_cfg({"env": name})
        bd = area_clearing_params(cfg)
        trials = generate_trials(cfg, 3, 2)
        nbox = len(trials[0]["boxes"])
        bd["num_boxes"] = nbox
        lay = env_layout(cfg)
        bcfg = _lib.make_bd_config(area_clearing_physics_params(cfg), bd, cfg)
        _lib.fill_area_geometry(bcfg, lay.boundary, lay.outer_boundary, cfg.agent.footprint_vertices, goal_points(cfg))
        bcfg.distance_scale_max = bd["distance_scale_max"]
        out.append(_bd_record(bcfg, trials, nbox, max(len(t["statics"][1]) for t in trials)))
    return out


def test_host_side_loaders_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-Wall", "-Wextra", "-Wno-unused-function", "-o", exe, SRC])
    case = tmp_path / "cases.bin"
    recs = _ship_records() + _maze_records() + _box_records() + _area_records()
    case.write_bytes(b"".join(recs))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([exe, str(case)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    assert "host-sanitize-ok %d" % len(recs) in out and "runtime error" not in err and "AddressSanitizer" not in err, (out[-2000:], err[-4000:])
    assert out.count("ship-ice:") == 5 and out.count("maze:") == 2 and out.count("box-delivery:") == 4 and out.count("area-clearing:") == 4
