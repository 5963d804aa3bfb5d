"""Self-consistency and known-answer tests of the oracle's restated Chipmunk step (no pymunk here: parity unpinned)."""
import math

import numpy as np
import pytest

from oracle import oracle as orc


def _env(ship_cfg, **over):
    cfg, P = ship_cfg
    return orc.OracleShipIce(dict(P, **over), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail), cfg


def _square(cx, cy, h=0.5):
    v = np.array([[cx - h, cy - h], [cx + h, cy - h], [cx + h, cy + h], [cx - h, cy + h]])
    return {"vertices": v, "centre": (cx, cy), "radius": h, "area": 4 * h * h}


def test_sincos_within_2ulp_of_libm():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-10, 10, 4000), [0.0, math.pi / 2, math.pi, 1e-9, -math.pi / 2]])
    for x in xs:
        s, c = orc.sincos(x)
        assert abs(s - math.sin(x)) <= 2 * np.spacing(abs(math.sin(x))) + 1e-300
        assert abs(c - math.cos(x)) <= 2 * np.spacing(abs(math.cos(x))) + 1e-300


def test_convex_hull_is_ccw_and_equals_scipy_hull_set():
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(1)
    for _ in range(30):
        pts = rng.uniform(-1, 1, (17, 2))
        h = orc.convex_hull(pts)
        ref = pts[ConvexHull(pts).vertices]
        assert {tuple(p) for p in h} == {tuple(p) for p in ref}
        x, y = h.T
        assert 0.5 * (np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) > 0  # counter-clockwise
    # ship outline of the reference config: 17 listed vertices -> 7-vertex hull
    from benchpush_amd.config import default_cfg
    assert len(orc.convex_hull(np.array(default_cfg("ship_ice").ship.vertices))) == 7


def _poly_dist(a, b):
    """Brute-force distance between two convex polygons (0 if they overlap), independent of the oracle."""
    def seg_pt(p, q, x):
        d = q - p
        t = np.clip(np.dot(x - p, d) / np.dot(d, d), 0, 1)
        return np.linalg.norm(x - (p + t * d))
    def inside(poly, x):
        s = [(poly[(i + 1) % len(poly)][0] - poly[i][0]) * (x[1] - poly[i][1]) - (poly[(i + 1) % len(poly)][1] - poly[i][1]) * (x[0] - poly[i][0])
             for i in range(len(poly))]
        return all(v >= 0 for v in s)
    if any(inside(a, x) for x in b) or any(inside(b, x) for x in a):
        return 0.0
    d = min(seg_pt(a[i], a[(i + 1) % len(a)], x) for i in range(len(a)) for x in b)
    return min(d, min(seg_pt(b[i], b[(i + 1) % len(b)], x) for i in range(len(b)) for x in a))


def test_narrowphase_touching_decision_and_normal():
    import random
    from benchpush_amd.scenario import generate_polygon
    rng = random.Random(5)
    nrng = np.random.default_rng(5)
    r = 0.02
    hits = 0
    for _ in range(300):
        a = orc.convex_hull(generate_polygon(1.0, (0.0, 0.0), rng=rng))
        b = orc.convex_hull(generate_polygon(1.0, tuple(nrng.uniform(-1.3, 1.3, 2)), rng=rng))
        cnt, n, p1, p2, h = orc.collide(a, r, b, r)
        d = _poly_dist(a, b)
        if abs(d - 2 * r) < 1e-9:
            continue
        if cnt > 0:
            hits += 1
            assert d <= 2 * r + 1e-12
            assert abs(np.linalg.norm(n) - 1) < 1e-12
            if d > 0:  # separated cores: the normal is the direction of the closest points, from A to B
                assert np.dot(n, b.mean(0) - a.mean(0)) > 0
            assert cnt <= 2 and all(np.dot(p2[k] - p1[k], n) <= 1e-15 for k in range(cnt))
        elif d == 0.0:
            pytest.fail("overlapping polygons reported as not touching")
    assert hits > 20


def test_box_pushed_by_kinematic_ship_moves_with_it(ship_cfg):
    """Quasi-static pushing (damping 0): a floe dead ahead of the ship ends up moving at the ship's speed."""
    env, cfg = _env(ship_cfg)
    trial = {"obstacles": [_square(6.0, 3.0)], "ship_state": (6.0, 1.0, math.pi / 2)}
    env.reset(trial, observe=False)
    y0 = env.bodies()[1, 1]
    for _ in range(6):
        env.step(0.0, observe=False)
    b = env.bodies()
    assert b[0, 1] == pytest.approx(1.0 + 6 * 0.8 * 0.3, abs=1e-9)      # ship is kinematic: unaffected by contacts
    assert b[0, 0] == pytest.approx(6.0, abs=1e-12)
    assert b[1, 1] > y0 + 0.5                                            # the floe was pushed north
    gap = (b[1, 1] - 0.5) - (b[0, 1] + 1.0)                              # floe bottom face vs ship bow tip
    assert -0.1 - 1e-9 <= gap <= 0.04 + 1e-9                             # within [ -slop, r1 + r2 ]
    assert b[1, 4] == pytest.approx(0.3, rel=0.05)                       # pushed at the ship's speed


def test_resting_overlap_within_slop_produces_no_motion(ship_cfg):
    env, cfg = _env(ship_cfg)
    # two squares overlapping by 0.05 < slop 0.1, far from the ship
    trial = {"obstacles": [_square(3.0, 20.0), _square(3.95, 20.0)], "ship_state": (9.0, 1.0, math.pi / 2)}
    env.reset(trial, observe=False)
    b0 = env.bodies().copy()
    env.step(0.3, observe=False)
    b1 = env.bodies()
    assert np.array_equal(b0[1:, :3], b1[1:, :3])
    assert env.stats()["arb_max"] == 1  # the pair is a (cold) arbiter


def test_deep_overlap_is_pushed_apart_by_bias(ship_cfg):
    env, cfg = _env(ship_cfg)
    trial = {"obstacles": [_square(3.0, 20.0), _square(3.7, 20.0)], "ship_state": (9.0, 1.0, math.pi / 2)}
    env.reset(trial, observe=False)
    b = env.bodies()
    assert b[2, 0] - b[1, 0] > 0.7 + 1e-6  # separated during the 1000 settle sub-steps
    assert abs(b[1, 1] - 20.0) < 1e-9


def test_damping_known_answer_a_released_floe_coasts_geometrically(ship_cfg):
    """cpBodyUpdateVelocity with `space.damping` d (ship_ice_env.py:120): a body without contacts keeps v * d^dt per sub-step.  A floe dead ahead is pushed for
    three steps, then the ship turns away at full rate and the floe runs free: its speed shrinks by d^(400 dt) = d^0.8 per env step -- a known answer that does not
    depend on the contact model -- until the ship's stern swings back into it."""
    cfg, P = ship_cfg
    d = 0.5
    env, _ = _env(ship_cfg, damping_pow=math.pow(d, cfg.dt / cfg.sim.steps))
    trial = {"obstacles": [_square(6.0, 2.6, 0.3)], "ship_state": (6.0, 1.0, math.pi / 2)}
    env.reset(trial, observe=False)
    speeds = []
    for t in range(10):
        env.step(0.0 if t < 3 else 1.0, observe=False)
        b = env.bodies()
        speeds.append(float(np.hypot(b[1, 3], b[1, 4])))
    assert speeds[2] == pytest.approx(0.3, rel=0.05)                        # pushed at the ship's speed
    for t in range(4, 10):                                                  # free flight: steps 4..9
        assert speeds[t] / speeds[t - 1] == pytest.approx(d ** 0.8, rel=1e-9), t


def test_zero_area_floe_is_dropped(ship_cfg):
    env, cfg = _env(ship_cfg)
    deg = {"vertices": np.array([[1.0, 5.0], [2.0, 5.0], [3.0, 5.0]]), "centre": (2.0, 5.0), "radius": 1.0, "area": 0.0}
    env.reset({"obstacles": [deg, _square(6.0, 10.0)], "ship_state": (6.0, 1.0, math.pi / 2)}, observe=False)
    assert env.nf == 1 and len(env.bodies()) == 2


def test_empty_field_and_termination_flags(ship_cfg):
    env, cfg = _env(ship_cfg)
    env.reset({"obstacles": [], "ship_state": (6.0, 1.0, math.pi / 2)}, observe=False)
    total = 0.0
    for t in range(40):
        obs, r, term, info = env.step(0.0)
        total += r
        if term:
            break
    assert term and info["trial_success"] == 1.0 and info["y"] >= 9.0
    assert t == 33  # 1 + 0.24 * 34 = 9.16 >= 9
    assert r == pytest.approx(200.0)  # terminal reward, no heading term once past the goal line
    # boundary: steer hard left from near the wall -> -50 and termination without success
    env.reset({"obstacles": [], "ship_state": (0.3, 1.0, math.pi / 2)}, observe=False)
    for t in range(40):
        obs, r, term, info = env.step(1.0)
        if term:
            break
    assert term and info["trial_success"] == 0.0 and info["boundary_violated"] == 1.0 and info["x"] < 0


def test_yaw_limit_freezes_rotation(ship_cfg):
    env, cfg = _env(ship_cfg, goal_y=1000.0, map_w=1000.0)
    env.reset({"obstacles": [], "ship_state": (500.0, 1.0, math.pi / 2)}, observe=False)
    for t in range(12):
        obs, r, term, info = env.step(-1.0, observe=False)
    assert info["yaw_violated"] == 1.0 and info["theta"] <= 0.0 and info["theta"] > -0.01


def test_sweep_broadphase_equals_all_pairs(ship_cfg):
    from benchpush_amd.scenario import generate_ice_field
    cfg, P = ship_cfg
    tr = generate_ice_field(0.3, 3, min_r=0.40, max_r=0.58)
    envs = [orc.OracleShipIce(P, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail, brute_force=b) for b in (False, True)]
    for e in envs:
        e.reset(tr, observe=False)
    rng = np.random.default_rng(3)
    for t in range(4):
        a = rng.uniform(-1, 1)
        o = [e.step(a) for e in envs]
        assert np.array_equal(o[0][0], o[1][0]) and o[0][1] == o[1][1]
        assert np.array_equal(envs[0].bodies(), envs[1].bodies())


def test_work_nonnegative_and_determinism(ship_cfg):
    from benchpush_amd.scenario import generate_ice_field
    cfg, P = ship_cfg
    tr = generate_ice_field(0.5, 11, min_r=0.40, max_r=0.58)
    outs = []
    for rep in range(2):
        e = orc.OracleShipIce(P, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
        e.reset(tr, observe=False)
        rng = np.random.default_rng(9)
        seq = []
        for t in range(5):
            obs, r, term, info = e.step(rng.uniform(-1, 1), observe=False)
            assert info["work"] >= 0.0
            seq.append((r, info["total_work"], info["n_contact_pts"]))
        outs.append((seq, e.bodies().copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def test_contact_points_known_answers_worked_by_hand():
    """cpCollide / ContactPoints (Chipmunk2D 7.0.3 cpCollision.c) on two configurations worked out by hand from the published algorithm
    (support edges, d_e* cross products, clamped lerps, radii along the normal), radius 0.02 per shape as in the reference
    (sim_utils.py:144, ship.py:90).
      parallel edges: unit square A, unit square B shifted by (1.03, 0.25): n = (1, 0); the overlapping stretch y in [0.25, 1] of the two
        facing edges gives two contacts, each 0.01 deep (gap 0.03 - radii 0.04).
      vertex-vertex: B is a diamond whose left vertex sits 0.03 from A's corner (1, 1) in the direction 30 degrees: n = (cos 30, sin 30);
        A's support edge is its right edge (n.x > n.y), B's the lower-left one; only the pair of end points (1,1) / B's vertex is within reach."""
    r = 0.02
    A = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    B = A + np.array([1.03, 0.25])
    cnt, n, p1, p2, h = orc.collide(A, r, B, r)
    assert cnt == 2 and np.allclose(n, [1.0, 0.0], atol=0, rtol=0)
    assert np.allclose(p1, [[1.02, 0.25], [1.02, 1.0]], atol=1e-12) and np.allclose(p2, [[1.01, 0.25], [1.01, 1.0]], atol=1e-12)
    assert np.allclose((p2 - p1) @ n, [-0.01, -0.01], atol=1e-12) and h[0] != h[1]
    c30, s30 = np.cos(np.pi / 6), np.sin(np.pi / 6)
    L = np.array([1.0, 1.0]) + 0.03 * np.array([c30, s30])
    D = np.array([L + [0.5, -0.5], L + [1.0, 0.0], L + [0.5, 0.5], L])       # CCW: bottom, right, top, left
    cnt, n, p1, p2, h = orc.collide(A, r, D, r)
    assert cnt == 1 and np.allclose(n, [c30, s30], atol=1e-12)
    assert np.allclose(p1[0], [1.0 + r * c30, 1.0 + r * s30], atol=1e-9) and np.allclose(p2[0], L - r * np.array([c30, s30]), atol=1e-9)
    assert abs(float((p2[0] - p1[0]) @ n) - (0.03 - 2 * r)) < 1e-9
    # just out of reach: 0.0401 between the two vertices -> no contact; the same shapes swapped give the mirrored normal
    L2 = np.array([1.0, 1.0]) + 0.0401 * np.array([c30, s30])
    D2 = np.array([L2 + [0.5, -0.5], L2 + [1.0, 0.0], L2 + [0.5, 0.5], L2])
    assert orc.collide(A, r, D2, r)[0] == 0
    cnt, n2, q1, q2, _ = orc.collide(D, r, A, r)
    assert cnt == 1 and np.allclose(n2, [-c30, -s30], atol=1e-12) and np.allclose(q1[0], p2[0], atol=1e-9) and np.allclose(q2[0], p1[0], atol=1e-9)


# ---- hardening of the unpinned narrow phase: the exact closest-feature query that stands in for Chipmunk's GJK / EPA
# (cpCollision.c ClosestPoints: the touching decision d <= r1 + r2 and the direction of the closest points are all that reach ContactPoints) is
# held to a brute-force distance computation that shares nothing with it -----------------------------------------------------------------
def _closest_brute(a, b):
    """(distance, unit direction from A's closest point to B's, kind) of two convex polygons / segments given as CCW vertex arrays; distance 0 and
    direction None when they intersect.  Every (vertex, edge) combination of both shapes, vectorised; kind = 'vv' if the closest points are two vertices."""
    def seg_pairs(P, Q):   # closest point on every edge of P to every vertex of Q
        A0, A1 = P, np.roll(P, -1, 0)
        d = A1 - A0
        dd = np.maximum((d * d).sum(1), 1e-300)
        t = np.clip(((Q[None, :, :] - A0[:, None, :]) * d[:, None, :]).sum(2) / dd[:, None], 0.0, 1.0)
        C = A0[:, None, :] + t[:, :, None] * d[:, None, :]
        D = np.linalg.norm(Q[None, :, :] - C, axis=2)
        k = np.unravel_index(np.argmin(D), D.shape)
        return D[k], C[k], Q[k[1]], t[k]

    def inside(P, x):
        if len(P) < 3:
            return False
        e = np.roll(P, -1, 0) - P
        return bool(np.all(e[:, 0] * (x[1] - P[:, 1]) - e[:, 1] * (x[0] - P[:, 0]) >= 0))

    def cross_any(P, Q):   # proper or touching intersection of any edge pair
        for i in range(len(P)):
            p, r = P[i], P[(i + 1) % len(P)] - P[i]
            for j in range(len(Q)):
                q, s = Q[j], Q[(j + 1) % len(Q)] - Q[j]
                den = r[0] * s[1] - r[1] * s[0]
                if den == 0.0:
                    continue
                t = ((q[0] - p[0]) * s[1] - (q[1] - p[1]) * s[0]) / den
                u = ((q[0] - p[0]) * r[1] - (q[1] - p[1]) * r[0]) / den
                if 0 <= t <= 1 and 0 <= u <= 1:
                    return True
        return False
    if any(inside(a, x) for x in b) or any(inside(b, x) for x in a) or cross_any(a, b):
        return 0.0, None, "overlap"
    dA, cA, qB, tA = seg_pairs(a, b)      # point on A's boundary, vertex of B
    dB, cB, qA, tB = seg_pairs(b, a)      # point on B's boundary, vertex of A
    if dA <= dB:
        d, pa, pb, t = dA, cA, qB, tA
    else:
        d, pa, pb, t = dB, qA, cB, tB
    return float(d), (pb - pa) / d, ("vv" if t in (0.0, 1.0) else "ve")


def _rot(p, ang, about=(0.0, 0.0)):
    c, s = math.cos(ang), math.sin(ang)
    q = np.asarray(p, float) - about
    return np.stack([c * q[:, 0] - s * q[:, 1], s * q[:, 0] + c * q[:, 1]], 1) + about


def _check_pair(a, ra, b, rb, stats, eps=1e-9):
    d, ndir, kind = _closest_brute(a, b)
    cnt, n, p1, p2, _ = orc.collide(a, ra, b, rb)
    rs = ra + rb
    if d < rs - eps:
        assert cnt > 0, ("missed contact", d, rs, kind, a.tolist(), b.tolist())
        stats["touch"] += 1
    elif d > rs + eps:
        assert cnt == 0, ("phantom contact", d, rs, kind, a.tolist(), b.tolist())
        stats["apart"] += 1
    if cnt > 0:
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and cnt <= 2
        assert all(np.dot(p2[k] - p1[k], n) <= 1e-15 for k in range(cnt))
        if d > 1e-7:     # separated cores: the normal is the direction of the closest points (unique for convex sets), from A to B
            assert np.linalg.norm(n - ndir) < 1e-9 * max(1.0, 1e-3 / d), ("normal", n, ndir, d, kind, a.tolist(), b.tolist())
            stats[kind] += 1
            # the contact points sit on the two rounded surfaces; the deepest one realises the distance: dist = d - (ra + rb)
            deepest = min(np.dot(p2[k] - p1[k], n) for k in range(cnt))
            assert abs(deepest - (d - rs)) < 1e-9, ("depth", deepest, d - rs, kind)


def test_narrowphase_against_brute_force_on_10000_convex_pairs():
    """Random convex pairs moved to gaps around r1 + r2 (incl. just inside / just outside), constructed vertex-vertex and (near-)parallel-edge
    configurations, radii 0 and 0.02: touching decision both ways, normal = brute-force closest-points direction, depth = d - (r1 + r2)."""
    import collections
    import random
    from benchpush_amd.scenario import generate_polygon
    rng = random.Random(11)
    nrng = np.random.default_rng(11)
    stats = collections.Counter()
    polys = [orc.convex_hull(generate_polygon(rng.uniform(0.6, 2.0), (0.0, 0.0), rng=rng)) for _ in range(400)]
    n_pairs = 0
    # (1) random pairs, B pushed along the closest direction to a prescribed gap: exact for convex sets while the gap stays positive
    for it in range(7000):
        a = polys[rng.randrange(len(polys))]
        b = _rot(polys[rng.randrange(len(polys))], rng.uniform(0, 2 * math.pi)) + nrng.uniform(-2.5, 2.5, 2)
        r = 0.02 if it % 3 else 0.0
        d, ndir, _ = _closest_brute(a, b)
        if ndir is not None:
            gap = rng.choice([rng.uniform(1e-6, 2 * r + 0.05), 2 * r - 1e-7, 2 * r + 1e-7, 2 * r * rng.random(), d])
            if gap > 0:
                b = b + (gap - d) * ndir
        _check_pair(a, r, b, r, stats)
        n_pairs += 1
    # (2) vertex against vertex: B's vertex j put on the outward bisector of A's vertex i at gap g, B turned so that -u lies inside j's normal cone
    for it in range(1500):
        a, b0 = polys[rng.randrange(len(polys))], polys[rng.randrange(len(polys))]
        i, j = rng.randrange(len(a)), rng.randrange(len(b0))
        def cone(P, k):
            e0, e1 = P[k] - P[k - 1], P[(k + 1) % len(P)] - P[k]
            n0, n1 = np.array([e0[1], -e0[0]]) / np.linalg.norm(e0), np.array([e1[1], -e1[0]]) / np.linalg.norm(e1)
            return n0, n1
        n0, n1 = cone(a, i)
        w = rng.uniform(0.05, 0.95)
        u = n0 * w + n1 * (1 - w)
        u /= np.linalg.norm(u)
        m0, m1 = cone(b0, j)
        v = m0 * 0.5 + m1 * 0.5
        v /= np.linalg.norm(v)
        ang = math.atan2(-u[1], -u[0]) - math.atan2(v[1], v[0])       # turn B so that its vertex bisector points along -u
        b = _rot(b0, ang + rng.uniform(-0.02, 0.02))
        r = 0.02 if it % 4 else 0.0
        g = rng.choice([rng.uniform(1e-5, 0.039), 0.04 - 1e-7, 0.04 + 1e-7, rng.uniform(0.041, 0.2)])
        b = b + (a[i] + g * u - b[j])
        _check_pair(a, r, b, r, stats)
        n_pairs += 1
    # (3) an edge of B parallel (exactly, and off by 1e-12 .. 1e-4 rad) to an edge of A, facing it at gap g with partial tangential overlap
    for it in range(1500):
        a, b0 = polys[rng.randrange(len(polys))], polys[rng.randrange(len(polys))]
        i, j = rng.randrange(len(a)), rng.randrange(len(b0))
        ea, eb = a[i] - a[i - 1], b0[j] - b0[j - 1]
        ang = math.atan2(-ea[1], -ea[0]) - math.atan2(eb[1], eb[0]) + rng.choice([0.0, 1e-12, -1e-12, 1e-9, 1e-6, -1e-4])
        b = _rot(b0, ang)
        na = np.array([ea[1], -ea[0]]) / np.linalg.norm(ea)
        r = 0.02 if it % 4 else 0.0
        g = rng.choice([rng.uniform(1e-5, 0.039), 0.04 - 1e-7, 0.04 + 1e-7, 0.0401, rng.uniform(0.05, 0.3)])
        mid_b = 0.5 * (b[j] + b[j - 1])
        b = b + (a[i - 1] + rng.uniform(-0.2, 1.2) * ea + g * na - mid_b)
        _check_pair(a, r, b, r, stats)
        n_pairs += 1
    assert n_pairs == 10000
    assert stats["vv"] > 800 and stats["ve"] > 800 and stats["touch"] > 3000 and stats["apart"] > 1500, dict(stats)


def test_narrowphase_segments_against_brute_force():
    """The 2-vertex hulls of the maze walls (pymunk.Segment radius 0.5, sim_utils.py:174-181) against boxes / robot outlines (radius 0.02) and against
    each other: the same three properties."""
    import collections
    import random
    rng = random.Random(3)
    stats = collections.Counter()
    box = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    octa = orc.convex_hull(np.array([[0.7, -0.5], [0.55, -0.6], [-0.55, -0.6], [-0.7, -0.5], [-0.7, 0.5], [-0.55, 0.6], [0.55, 0.6], [0.7, 0.5]]))   # the maze robot, counter-clockwise like every hull the envs build
    for it in range(3000):
        L = rng.uniform(0.5, 8.0)
        seg = _rot(np.array([[0.0, 0.0], [L, 0.0]]), rng.uniform(0, 2 * math.pi))
        other = _rot(box if it % 2 else octa, rng.uniform(0, 2 * math.pi)) + np.array([rng.uniform(-1.5, L + 1.5), rng.uniform(-2.0, 2.0)])
        ra, rb = 0.5, 0.02
        if it % 5 == 0:
            other = _rot(np.array([[0.0, 0.0], [rng.uniform(0.5, 4.0), 0.0]]), rng.uniform(0, 2 * math.pi)) + other.mean(0)
            rb = 0.5
        d, ndir, _ = _closest_brute(seg, other)
        if ndir is not None:
            gap = rng.choice([rng.uniform(1e-6, ra + rb + 0.1), ra + rb - 1e-7, ra + rb + 1e-7, d])
            other = other + (gap - d) * ndir
        if it % 2:
            _check_pair(seg, ra, other, rb, stats)
        else:
            _check_pair(other, rb, seg, ra, stats)
    assert stats["touch"] > 800 and stats["apart"] > 500 and stats["vv"] > 100 and stats["ve"] > 300, dict(stats)
