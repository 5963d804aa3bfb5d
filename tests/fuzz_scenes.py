"""Random small ship-ice scenes for the differential fuzz of the sub-step (tests/test_gpu_fuzz.py).

There are three hand-maintained restatements of one Chipmunk sub-step -- oracle/bp_oracle.c, csrc/bp_physics.hpp (one env per wavefront) and
csrc/bp_physics_pair.hpp (two envs per wavefront) -- and a fix to the narrow phase has to be made in all three.  The suite's parity tests hold them together
on the trajectories of generated ice fields; this generator aims at the configurations those rarely reach: 0-12 bodies, floes that overlap the ship's bow
and each other from the first sub-step (no settle), facing edges that are parallel exactly or to 1e-12 ... 1e-4 rad with partial tangential overlap and
gaps within +-1e-9 of the radius sum, corners that face corners, degenerate inputs (collinear points, duplicate vertices: the loaders take the convex hull).
A scene is a trial dict in the reference's pickle schema ({'ship_state', 'obstacles': [{'vertices', 'centre', 'radius'}]}).
"""
import math

import numpy as np

SHIP_HEAD = (1.0, 0.0)   # cfg.ship.head: the bow in ship coordinates (configs/ship_ice.yaml)


def _rot(v, th):
    c, s = math.cos(th), math.sin(th)
    v = np.asarray(v, np.float64)
    return np.stack([c * v[..., 0] - s * v[..., 1], s * v[..., 0] + c * v[..., 1]], -1)


def _ob(verts):
    v = np.asarray(verts, np.float64)
    return {"vertices": v, "centre": (float(v[:, 0].mean()), float(v[:, 1].mean())), "radius": float(np.abs(v - v.mean(0)).max())}


def _blob(rng, centre, r):
    """3..14 points on a wobbly circle; not necessarily convex (the loaders take the hull), sometimes with a duplicate or a collinear point."""
    n = int(rng.integers(3, 15))
    ang = np.sort(rng.uniform(0, 2 * math.pi, n))
    rad = r * rng.uniform(0.55, 1.0, n)
    v = np.stack([rad * np.cos(ang), rad * np.sin(ang)], -1) + np.asarray(centre)
    k = rng.integers(0, 6)
    if k == 0 and n < 14:
        v = np.concatenate([v, v[:1]])                      # duplicate vertex
    elif k == 1 and n < 14:
        v = np.concatenate([v, [(v[0] + v[1]) / 2]])         # point on an edge of the polygon
    return _ob(v)


def _rect(centre, w, h, phi):
    c = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    return _rot(c, phi) + np.asarray(centre)


GAPS = (-0.05, -1e-3, -1e-9, 0.0, 1e-9, 1e-3)
TILTS = (0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-4, -1e-4)


def make_scene(seed, radius):
    """Scene number `seed` for a handle whose shapes are rounded by `radius` (0 or 0.02)."""
    rng = np.random.default_rng(seed)
    th = math.pi / 2 + rng.uniform(-0.6, 0.6)
    sx, sy = rng.uniform(3.0, 9.0), rng.uniform(2.0, 5.0)
    bow = np.array([sx, sy]) + _rot(SHIP_HEAD, th)
    fwd = np.array([math.cos(th), math.sin(th)])
    left = np.array([-fwd[1], fwd[0]])
    obs = []
    fam = seed % 4
    rsum = 2 * radius
    if fam == 1:
        # facing edges, parallel exactly or nearly, partial tangential overlap, gap around the radius sum; the first box sits on the bow and is pushed into the second
        phi = th + rng.uniform(-0.4, 0.4)
        wa, ha, wb, hb = rng.uniform(0.3, 0.9, 4)
        u = np.array([math.cos(phi), math.sin(phi)])
        ca = bow + u * (wa / 2 + rng.uniform(-0.03, 0.03)) + left * rng.uniform(-0.2, 0.2)
        gap = rsum + GAPS[int(rng.integers(len(GAPS)))]
        tilt = TILTS[int(rng.integers(len(TILTS)))]
        off = rng.uniform(0.2, 0.8) * (ha + hb) / 2 * (1 if rng.random() < 0.5 else -1)
        cb = ca + u * (wa / 2 + gap + wb / 2) + np.array([-u[1], u[0]]) * off
        obs += [_ob(_rect(ca, wa, ha, phi)), _ob(_rect(cb, wb, hb, phi + tilt))]
    elif fam == 2:
        # corner against corner along the ship's heading, distance around the radius sum
        a, b = rng.uniform(0.3, 0.7, 2)
        phi = th + math.pi / 4 + rng.uniform(-0.05, 0.05)
        ca = bow + fwd * (a / math.sqrt(2) + rng.uniform(-0.02, 0.03))
        d = rsum + GAPS[int(rng.integers(len(GAPS)))]
        tip = ca + fwd * (a / math.sqrt(2))
        cb = tip + fwd * (d + b / math.sqrt(2)) + left * rng.choice([0.0, 1e-9, -1e-9, 1e-3])
        obs += [_ob(_rect(ca, a, a, phi)), _ob(_rect(cb, b, b, phi + rng.choice([0.0, 1e-9, math.pi / 2])))]
    elif fam == 3:
        # a chain of near-identical boxes in front of the bow: several colours, two-contact manifolds on every link
        phi = th + rng.choice([0.0, 1e-9, 0.02])
        u = np.array([math.cos(phi), math.sin(phi)])
        w = rng.uniform(0.25, 0.5)
        c = bow + u * (w / 2 - rng.uniform(0.0, 0.04))
        for _ in range(int(rng.integers(2, 6))):
            obs.append(_ob(_rect(c, w, rng.uniform(0.3, 0.8), phi)))
            c = c + u * (w + rsum + GAPS[int(rng.integers(len(GAPS)))])
    # a random cluster around the bow (family 0: nothing else; the others: a few extras), 0 .. 12 bodies in all
    nmax = 12 - len(obs)
    nextra = int(rng.integers(0, nmax + 1)) if fam == 0 else int(rng.integers(0, min(4, nmax) + 1))
    for _ in range(nextra):
        c = bow + fwd * rng.uniform(-0.6, 1.6) + left * rng.uniform(-1.0, 1.0)
        obs.append(_blob(rng, c, rng.uniform(0.12, 0.5)))
    return {"ship_state": (float(sx), float(sy), float(th)), "goal": (0.0, 9.0), "obstacles": obs}


def make_scenes(n, radius, base_seed=0):
    return [make_scene(base_seed + i, radius) for i in range(n)]


def fuzz_params(params, radius, substeps):
    """Physics parameters of a fuzz handle: `substeps` sub-steps of the reference's 2 ms per env step, ONE settle sub-step instead of the reference's 1000
    (the constructed contacts are still there when the first step starts; the one sub-step is the space's first, in which every shape is new to the
    broadphase and every pair is tested -- a reset without it has no counterpart in the reference, and the HIP path, which rebuilds the moving list of a
    resumed env from its velocities, would rightly never look at pairs of floes that lie still), shapes rounded by `radius`."""
    p = dict(params)
    p.update(settle_steps=1, poly_radius=float(radius), steps=int(substeps), dt=float(p["dt"]) / int(p["steps"]) * int(substeps))
    return p
