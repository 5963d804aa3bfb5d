"""Layouts for maze-NAMO-v0: box placement by rejection sampling, restating MazeNAMO.generate_obstacles
(benchpush/environments/maze_NAMO/maze_NAMO_env.py:271-322) on a seeded ``random.Random`` (the reference draws from
the unseeded module-level generator).  The wall test is pymunk's ``space.point_query(p, max_distance, ShapeFilter())``
against the Segment(radius 0.5) walls, i.e. Chipmunk's ``distance(p, segment) - radius < max_distance``.
"""
import math
import random as _random

import numpy as np


def _seg_dist(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    t = ((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy)
    t = min(1.0, max(0.0, t))
    cx, cy = ax + t * dx, ay + t * dy
    return math.hypot(px - cx, py - cy)


def point_query_hits_wall(walls, x, y, max_distance, wall_radius=0.5):
    return any(_seg_dist(x, y, *w) - wall_radius < max_distance for w in walls)


def generate_boxes(cfg, walls, rng):
    """Box centres [n, 2] (maze_NAMO_env.py:271-311).  Note the reference's quirk: walls are only tested from inside
    the loop over previously placed boxes, so the first box is never tested against walls."""
    if not cfg.randomize_obstacles:
        return np.array([[8.5, 11], [10, 9], [11.25, 11.5], [6, 10], [3.5, 8.5]], np.float64)
    need, dmin = cfg.num_obstacles, cfg.min_obs_dist
    lo, hi = 0, cfg.env.length
    out = []
    while len(out) < need:
        cx = rng.random() * (hi - lo) + lo
        cy = rng.random() * (hi - lo) + lo
        overlapped = False
        for px, py in out:
            if ((cx - px) ** 2 + (cy - py) ** 2) ** 0.5 <= dmin:
                overlapped = True
                break
            if point_query_hits_wall(walls, cx, cy, dmin):
                overlapped = True
                break
        if not overlapped:
            out.append([cx, cy])
    return np.array(out, np.float64)


def maze_start(cfg):
    """Fixed start poses of maze_NAMO_env.py:241-245."""
    if cfg.maze_version == 1:
        return (11.25, 3.75, math.pi / 2)
    return (16.66, 16.66, 3 * math.pi / 2)


def random_start(cfg, walls, rng):
    """cfg.random_start (maze_NAMO_env.py:229-238): rejection-sample (x, y) in [1, start_range] until the point keeps
    robot.min_obstacle_dist from every wall (point_query against the walls only: the boxes are drawn afterwards and are not tested
    against the robot); heading 3 pi / 2.  Drawn before the boxes, like the reference, from the layout's generator."""
    while True:
        x = 1 + rng.random() * (cfg.start_x_range - 1)
        y = 1 + rng.random() * (cfg.start_y_range - 1)
        if not point_query_hits_wall(walls, x, y, cfg.robot.min_obstacle_dist):
            return (x, y, np.pi * 3 / 2)


def generate_layout(cfg, walls, seed):
    """One episode's layout on ``random.Random(seed)`` in the reference's draw order: start pose (if cfg.random_start), then boxes."""
    rng = _random.Random(seed)
    start = random_start(cfg, walls, rng) if cfg.get("random_start", False) else maze_start(cfg)
    return {"centres": generate_boxes(cfg, walls, rng), "walls": np.array(walls, np.float64), "start": start}
