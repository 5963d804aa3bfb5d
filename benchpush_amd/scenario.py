"""Ice-field scenarios for ship-ice-v0.

The reference env unpickles ``ice_environments/experiments_<conc>_100_r06_d40x12.pk`` at construction
(ship_ice_env.py:76-80) and picks ``trial = experiment[episode_idx % 100]`` at every reset (:188-198).  Those
pickles are missing large blobs in the reference checkout, so fields are synthesised here with the same
schema:

    {'meta_data': {...}, 'exp': {conc: {trial_idx: {'goal': ..., 'ship_state': (x, y, theta),
                                                  'obstacles': [{'vertices': (n,2) f64, 'centre': (x, y),
                                                                 'radius': r, 'area': a}, ...]}}}}

``generate_polygon`` restates the reference's random convex polygon sampler (geometry/polygon.py:53-146,
Valtr's algorithm) call-for-call on a ``random.Random`` stream, so that seeding it like the reference's global
``random`` reproduces the reference's polygons (pinned by tests/golden/polygon_golden.json).  Field layout
(circle placement) follows the structure of ``generate_rand_exp`` (ship_ice_nav_mujoco/ship_ice_utils.py:779-887):
non-overlapping circles -> one polygon per circle in the circle's bounding square -> clamp vertices to the
channel -> adjust to the requested concentration; the circle packer (third-party ``packcircles``) is replaced by
seeded rejection sampling.
"""
import math
import pickle
import random as _random

import numpy as np

__all__ = ["poly_area", "poly_centroid", "generate_polygon", "generate_ice_field", "generate_experiment",
           "load_experiment", "pack_trials"]


def poly_area(vertices):
    """Shoelace area (reference: geometry/polygon.py:25-29)."""
    x, y = np.asarray(vertices).T
    return 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))


def poly_centroid(vertices):
    """Centroid with the reference's abs() (geometry/polygon.py:32-41)."""
    x, y = np.asarray(vertices).T
    A = poly_area(vertices)
    u = x * np.roll(y, 1) - np.roll(x, 1) * y
    return np.abs((1 / (6 * A) * np.dot(x + np.roll(x, 1), u), 1 / (6 * A) * np.dot(y + np.roll(y, 1), u)))


def generate_polygon(diameter, origin=(0, 0), num_vertices_range=(10, 20), rng=None):
    """Random convex polygon inside a ``diameter`` square, centroid moved to ``origin``.

    Restates geometry/polygon.py:53-146 (non-circular branch) with the same sequence of draws on ``rng``
    (a ``random.Random``; defaults to the module-level ``random`` like the reference).
    """
    rng = _random if rng is None else rng
    n = rng.randint(*num_vertices_range)
    xs = sorted(rng.uniform(0, diameter) for _ in range(n))
    ys = sorted(rng.uniform(0, diameter) for _ in range(n))
    x_min, x_max, y_min, y_max = xs[0], xs[-1], ys[0], ys[-1]

    def chain(vals, lo, hi):
        last_a = last_b = lo
        out = []
        for i in range(1, n - 1):
            val = vals[i]
            if bool(rng.getrandbits(1)):
                out.append(val - last_a)
                last_a = val
            else:
                out.append(last_b - val)
                last_b = val
        out.append(hi - last_a)
        out.append(last_b - hi)
        return out

    x_vec = chain(xs, x_min, x_max)
    y_vec = chain(ys, y_min, y_max)
    rng.shuffle(y_vec)
    pairs = sorted(zip(x_vec, y_vec), key=lambda p: np.arctan2(p[0], p[1]))
    min_px = min_py = 0
    x = y = 0
    pts = []
    for px, py in pairs:
        pts.append((x, y))
        x += px
        y += py
        min_px = min(min_px, x)
        min_py = min(min_py, y)
    pts = np.asarray(pts) + np.array([x_min - min_px, y_min - min_py]).T
    pts -= poly_centroid(pts) - np.asarray(origin)
    return pts


def generate_ice_field(concentration, seed, map_w=12.0, map_h=40.0, min_y=3.0, min_r=0.45, max_r=0.70,
                       start=None, goal_y=9.0, tol=0.01, max_floes=None):
    """One synthetic trial. Deterministic in ``seed``.  Returns a trial dict in the reference pickle schema."""
    nrng = np.random.default_rng(seed)
    prng = _random.Random(seed)
    region = map_w * (map_h - min_y)
    target = concentration * region
    circles = []
    cell = 2 * max_r
    grid = {}

    def free(x, y, r):
        ci, cj = int(x // cell), int(y // cell)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for (ox, oy, orr) in grid.get((ci + di, cj + dj), ()):
                    if (ox - x) ** 2 + (oy - y) ** 2 < (orr + r) ** 2:
                        return False
        return True

    obstacles = []
    area = 0.0
    fails = 0
    while area < target - tol * region and fails < 20000:
        if max_floes is not None and len(obstacles) >= max_floes:
            break
        r = float(nrng.uniform(min_r, max_r))
        x = float(nrng.uniform(0.0, map_w))
        y = float(nrng.uniform(min_y, map_h))
        if not free(x, y, r):
            fails += 1
            continue
        verts = generate_polygon(2 * r, (x, y), rng=prng)
        # intersection with the channel, as generate_rand_exp does (ship_ice_utils.py:835-842)
        verts[:, 0] = np.clip(verts[:, 0], 0.0, map_w)
        verts[:, 1] = np.clip(verts[:, 1], min_y, map_h)
        a = float(poly_area(verts))
        if a == 0.0:
            fails += 1
            continue
        if area + a > target + tol * region:
            fails += 1
            continue
        grid.setdefault((int(x // cell), int(y // cell)), []).append((x, y, r))
        circles.append((x, y, r))
        obstacles.append({"vertices": verts, "centre": (x, y), "radius": r, "area": a})
        area += a
    if start is None:
        start = (float(nrng.uniform(1.0, map_w - 1.0)), 1.0, math.pi / 2)
    return {"goal": (0, goal_y), "ship_state": tuple(start), "obstacles": obstacles}


def generate_experiment(concentration, num_trials, base_seed=0, **kw):
    """Experiment dict with the reference's on-disk layout (consumer: ship_ice_env.py:76-80,188-198)."""
    exp = {i: generate_ice_field(concentration, base_seed + i, **kw) for i in range(num_trials)}
    return {"meta_data": {"concentration": concentration, "map_shape": (kw.get("map_h", 40.0), kw.get("map_w", 12.0)),
                          "synthetic": True, "base_seed": base_seed},
            "exp": {concentration: exp}}


def load_experiment(path, concentration):
    """Load a reference ``experiments_*.pk`` file and return its ``{trial_idx: trial}`` dict."""
    with open(path, "rb") as f:
        ddict = pickle.load(f)
    return ddict["exp"][concentration]


def pack_trials(trials, max_verts=24):
    """Flatten trials into the arrays ``bp_load_scenarios`` takes.

    Returns dict(verts [T, F, max_verts, 2] f64 raw world vertices, counts [T, F] i32 (0 = unused slot),
    centres [T, F, 2] f64, starts [T, 3] f64, nfloes [T] i32).  Floes are kept in trial order; zero-area floes
    are kept here and dropped by the loader exactly like ship_ice_env.py:206.
    """
    T = len(trials)
    F = max(1, max(len(t["obstacles"]) for t in trials))
    verts = np.zeros((T, F, max_verts, 2), np.float64)
    counts = np.zeros((T, F), np.int32)
    centres = np.zeros((T, F, 2), np.float64)
    starts = np.zeros((T, 3), np.float64)
    nfl = np.zeros((T,), np.int32)
    for t, trial in enumerate(trials):
        starts[t] = trial["ship_state"]
        nfl[t] = len(trial["obstacles"])
        for f, ob in enumerate(trial["obstacles"]):
            v = np.asarray(ob["vertices"], np.float64)
            if len(v) > max_verts:
                raise ValueError("floe with %d vertices exceeds max_verts=%d" % (len(v), max_verts))
            verts[t, f, : len(v)] = v
            counts[t, f] = len(v)
            centres[t, f] = ob["centre"]
    return dict(verts=verts, counts=counts, centres=centres, starts=starts, nfloes=nfl)
