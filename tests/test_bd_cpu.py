"""box-delivery-v0 oracle (oracle/bp_oracle_bd.c): pieces pinned against the real scipy/numpy in this image, restated
third-party pieces checked against small independent Python restatements and invariants, and env-level known answers."""
import math

import numpy as np
import pytest
from scipy import ndimage
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import dijkstra

from benchpush_amd import box_delivery_scenario as S
from benchpush_amd.config import default_cfg
from oracle import oracle_bd as ob
from oracle.oracle import lib as _orclib

import ctypes as C


def _sincos(x):
    s, c = C.c_double(), C.c_double()
    _orclib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def _env(cfg=None, **bp_over):
    cfg = cfg or default_cfg("box_delivery")
    bp = S.box_delivery_params(cfg)
    bp.update(bp_over)
    return cfg, ob.OracleBoxDelivery(S.box_delivery_physics_params(cfg), bp, cfg)


# ---------------------------------------------------------------- libm replacements
def test_atan2_and_mod_against_libm():
    rng = np.random.RandomState(0)
    worst = 0.0
    for _ in range(20000):
        y, x = rng.uniform(-10, 10), rng.uniform(-10, 10)
        if rng.rand() < 0.1:
            y *= 1e-9
        a, b = ob.atan2(y, x), math.atan2(y, x)
        worst = max(worst, abs(a - b) / max(np.spacing(abs(b)), 5e-324))
    assert worst <= 1.0                      # fdlibm's atan2 is within 1 ulp of the correctly rounded value
    assert ob.atan2(0.0, 1.0) == 0.0 and ob.atan2(0.0, -1.0) == math.pi and ob.atan2(1.0, 0.0) == math.pi / 2
    for _ in range(5000):
        a, b = rng.uniform(-50, 50), 2 * math.pi
        assert ob.pymod(a, b) == float(np.mod(a, b))


# ---------------------------------------------------------------- scipy-pinned rasters
def test_rotate_order0_matches_scipy():
    """scipy.ndimage.rotate(order=0, reshape=True): same shape, and at most 2 differing border pixels per image (scipy takes
    cos/sin from cosdg/sindg, the restatement from bp_sincos)."""
    rng = np.random.RandomState(1)
    for t in range(24):
        h, w = (318, 318) if t % 3 else (rng.randint(200, 318), rng.randint(200, 318))
        img = rng.rand(h, w).astype(np.float32)
        deg = rng.uniform(-400, 400)
        ref = ndimage.rotate(img, deg, order=0)
        s, c = _sincos(math.radians(deg))
        out = ob.rotate0(img, c, s)
        assert out.shape == ref.shape
        assert int((out != ref).sum()) <= 2


def _disk(r):
    a = np.arange(-r, r + 1)
    X, Y = np.meshgrid(a, a)
    return (X ** 2 + Y ** 2 <= r ** 2).astype(np.uint8)


@pytest.mark.parametrize("oc", ["small_empty", "small_columns", "large_divider"])
def test_configuration_space_and_edt_match_scipy(oc):
    cfg = default_cfg("box_delivery")
    cfg.env.obstacle_config = oc
    cfg, o = _env(cfg)
    o.reset(S.generate_trials(cfg, 1)[0], observe=False)
    m = o.maps()
    obst = np.ones((o.H, o.W), np.float32)
    si, sj = int(o.H / 2 - o.SH / 2), int(o.W / 2 - o.SW / 2)
    obst[si:si + o.SH, sj:sj + o.SW] = 1 - m["small_free"]
    for key, rad in (("cspace", S.robot_radius(cfg)), ("cspace_thin", max(cfg.agent.length, cfg.agent.width) / 2)):
        ref = 1 - ndimage.binary_dilation(obst, _disk(int(np.floor(rad * o.bd["ppm"])))).astype(np.float32)
        assert np.array_equal(ref, m[key])
    idx = ndimage.distance_transform_edt(1 - m["cspace"], return_distances=False, return_indices=True)
    assert np.array_equal(idx[0], m["edt_i"]) and np.array_equal(idx[1], m["edt_j"])


def test_edt_tie_rule_on_random_maps():
    rng = np.random.RandomState(3)
    for _ in range(10):
        free = (rng.rand(40, 50) < 0.05).astype(np.float32)
        idx = ndimage.distance_transform_edt(1 - free, return_distances=False, return_indices=True)
        ii, jj = ob.edt_indices(free)
        assert np.array_equal(idx[0], ii) and np.array_equal(idx[1], jj)


# ---------------------------------------------------------------- restated third-party pieces (unpinned)
def _py_sk_line(r0, c0, r1, c1):
    """Bresenham exactly as skimage.draw.line documents/implements it (independent Python restatement)."""
    steep = 0
    r, c = r0, c0
    dr, dc = abs(r1 - r0), abs(c1 - c0)
    sc = 1 if (c1 - c) > 0 else -1
    sr = 1 if (r1 - r) > 0 else -1
    if dr > dc:
        steep = 1
        c, r = r, c
        dc, dr = dr, dc
        sc, sr = sr, sc
    d = 2 * dr - dc
    rr, cc = [], []
    for _ in range(dc):
        if steep:
            rr.append(c); cc.append(r)
        else:
            rr.append(r); cc.append(c)
        while d >= 0:
            r += sr
            d -= 2 * dc
        c += sc
        d += 2 * dr
    rr.append(r1); cc.append(c1)
    return np.array(rr), np.array(cc)


def test_sk_line():
    rng = np.random.RandomState(2)
    for _ in range(300):
        r0, c0, r1, c1 = (int(v) for v in rng.randint(0, 60, 4))
        rr, cc = ob.sk_line(r0, c0, r1, c1)
        pr, pc = _py_sk_line(r0, c0, r1, c1)
        assert np.array_equal(rr, pr) and np.array_equal(cc, pc)
        assert (rr[0], cc[0]) == (r0, c0) and (rr[-1], cc[-1]) == (r1, c1)
        assert np.all(np.maximum(np.abs(np.diff(rr)), np.abs(np.diff(cc))) == 1) or len(rr) == 1


def _py_approx_polygon(coords, tol):
    """Douglas-Peucker as in skimage.measure.approximate_polygon, written with numpy on the same deterministic
    atan2/sincos the oracle uses."""
    coords = np.asarray(coords, np.int64)
    n = len(coords)
    chain = np.zeros(n, bool)
    chain[0] = chain[-1] = True
    stack = [(0, n - 1)]
    while stack:
        start, end = stack.pop()
        r0, c0 = coords[start]
        r1, c1 = coords[end]
        dr, dc = r1 - r0, c1 - c0
        ang = -ob.atan2(float(dr), float(dc))
        sa, ca = _sincos(ang)
        sd = float(c0) * sa + float(r0) * ca
        seg = coords[start + 1:end]
        if len(seg) == 0:
            continue
        dr0, dc0 = seg[:, 0] - r0, seg[:, 1] - c0
        dr1, dc1 = seg[:, 0] - r1, seg[:, 1] - c1
        perp = (dr0 * dr + dc0 * dc > 0) & (-dr1 * dr - dc1 * dc > 0)
        d = np.minimum(np.sqrt((dc0 ** 2 + dr0 ** 2).astype(np.float64)), np.sqrt((dc1 ** 2 + dr1 ** 2).astype(np.float64)))
        d[perp] = np.abs((seg[perp, 0] * ca + seg[perp, 1] * sa) - sd)
        if np.any(d > tol):
            k = start + int(np.argmax(d)) + 1
            stack.append((k, end)); stack.append((start, k))
            chain[k] = True
    return coords[chain]


def test_approximate_polygon():
    rng = np.random.RandomState(4)
    for _ in range(60):
        n = rng.randint(2, 200)
        steps = rng.randint(-1, 2, size=(n, 2))
        pts = np.cumsum(steps, axis=0) + 100
        got = ob.approx_polygon(pts, 1.0)
        assert np.array_equal(got, _py_approx_polygon(pts, 1.0))
        assert tuple(got[0]) == tuple(pts[0]) and tuple(got[-1]) == tuple(pts[-1])
    line = np.stack([np.arange(50), 2 * np.arange(50)], 1)
    assert len(ob.approx_polygon(line, 1.0)) == 2


def test_fill_poly_invariants():
    rng = np.random.RandomState(5)
    for _ in range(40):
        c = rng.uniform(15, 45, 2)
        ang = rng.uniform(0, 2 * np.pi)
        hw = rng.uniform(2, 12, 2)
        loc = np.array([[hw[0], hw[1]], [-hw[0], hw[1]], [-hw[0], -hw[1]], [hw[0], -hw[1]]])
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        pts = (loc @ R.T + c).astype(np.int32)
        img = np.zeros((60, 60), np.float32)
        ob.fill_poly(img, pts, 0.5)
        for x, y in pts:
            assert img[y, x] == 0.5                      # vertices are drawn
        # every lattice point strictly inside the integer polygon is filled; nothing farther than 1 px outside is
        ys, xs = np.mgrid[0:60, 0:60]
        inside = np.ones((60, 60), bool)
        far = np.zeros((60, 60), bool)
        P = pts.astype(np.float64)
        area2 = sum(P[i - 1][0] * P[i][1] - P[i][0] * P[i - 1][1] for i in range(4))
        sgn = 1.0 if area2 > 0 else -1.0
        for i in range(4):
            a, b = P[i - 1], P[i]
            cr = sgn * ((b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0]))
            L = max(np.hypot(*(b - a)), 1e-9)
            inside &= cr > 0
            far |= cr / L < -1.0
        if abs(area2) > 1:
            assert np.all(img[inside] == 0.5)
            assert not np.any(img[far] == 0.5)
    sq = np.zeros((20, 20), np.float32)
    ob.fill_poly(sq, [(3, 4), (10, 4), (10, 9), (3, 9)], 1.0)
    assert sq.sum() == 8 * 6 and sq[4:10, 3:11].all()     # axis-aligned rectangle: closed on both ends


def test_spfa_fixed_point_and_parents():
    rng = np.random.RandomState(6)
    H, W = 48, 64
    free = (rng.rand(H, W) > 0.25).astype(np.float32)
    free[10:14, 5:40] = 0
    src = tuple(np.argwhere(free > 0)[7])
    dist, par, parq = ob.spfa(free, src)
    # float64 Dijkstra on the same graph agrees to float32 accuracy and on reachability
    idx = -np.ones((H, W), np.int64)
    cells = np.argwhere(free > 0)
    idx[cells[:, 0], cells[:, 1]] = np.arange(len(cells))
    rows, cols, w = [], [], []
    for di, dj, ln in [(-1, -1, 2 ** 0.5), (-1, 0, 1), (-1, 1, 2 ** 0.5), (0, 1, 1), (1, 1, 2 ** 0.5), (1, 0, 1), (1, -1, 2 ** 0.5), (0, -1, 1)]:
        for (i, j) in cells:
            a, b = i + di, j + dj
            if 0 <= a < H and 0 <= b < W and free[a, b] > 0:
                rows.append(idx[i, j]); cols.append(idx[a, b]); w.append(ln)
    g = coo_matrix((w, (rows, cols)), shape=(len(cells), len(cells))).tocsr()
    ref = dijkstra(g, indices=idx[src])
    mine = dist[cells[:, 0], cells[:, 1]]
    reach = np.isfinite(ref)
    assert np.allclose(mine[reach], ref[reach], rtol=1e-5, atol=1e-4)
    assert np.all(mine[~reach] == 0)
    # least fixed point in float32: no edge can improve any cell, every reached cell has a tight parent
    SQ2 = np.float32(np.sqrt(np.float32(2.0)))
    for (i, j) in cells:
        if (i, j) == tuple(src) or dist[i, j] == 0:
            continue
        p = par[i, j]
        assert p >= 0
        pi, pj = divmod(int(p), W)
        ln = SQ2 if (pi != i and pj != j) else np.float32(1)
        assert np.float32(dist[pi, pj] + ln) == dist[i, j]
        for di, dj in [(-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1)]:
            a, b = i + di, j + dj
            if 0 <= a < H and 0 <= b < W and free[a, b] > 0 and ((a, b) == tuple(src) or dist[a, b] > 0):
                l2 = SQ2 if (di and dj) else np.float32(1)
                assert np.float32(dist[a, b] + l2) >= dist[i, j]
    # the queue-order parents of the C++ original are tight as well (possibly a different equal-cost neighbour)
    same = (par == parq)[free > 0].mean()
    assert same > 0.5


# ---------------------------------------------------------------- numpy float32 observation arithmetic
def test_observation_channels_against_numpy_scipy():
    """Recompute the four channels the way box_delivery_env.py:1045-1138 does, with the real scipy rotate and numpy float32
    arithmetic on the oracle's global maps; allow the few border pixels that the cosdg/bp_sincos difference can move."""
    cfg, o = _env()
    trials = S.generate_trials(cfg, 2)
    o.reset(trials[1], observe=False)
    rng = np.random.RandomState(0)
    for _ in range(3):
        o.step(rng.uniform(-1, 1), observe=False)
    obs = o.observe()
    m = o.maps()
    st = o.shape_states()[0]
    x, y, h = st[0], st[1], st[2]
    ppm, lp = o.bd["ppm"], o.lp

    def local_map(g):
        cw = int(np.ceil(lp * np.sqrt(2) / 2) * 2)
        rot = 90 - np.degrees(h)
        pi = int(np.floor(-y * ppm + g.shape[0] / 2)); pj = int(np.floor(x * ppm + g.shape[1] / 2))
        crop = g[pi - cw // 2:pi + cw // 2, pj - cw // 2:pj + cw // 2]
        r = ndimage.rotate(crop, rot, order=0)
        return r[r.shape[0] // 2 - lp // 2:r.shape[0] // 2 + lp // 2, r.shape[1] // 2 - lp // 2:r.shape[1] // 2 + lp // 2]

    def sp_channel(raw):
        g = raw.copy()
        g /= ppm
        g /= (np.sqrt(2) * lp) / ppm
        g *= cfg.env.shortest_path_channel_scale
        return g

    pi = int(np.clip(np.floor(o.H / 2 - y * ppm), 0, o.H - 1)); pj = int(np.clip(np.floor(o.W / 2 + x * ppm), 0, o.W - 1))
    raw, _, _ = ob.spfa(m["cspace"], (m["edt_i"][pi, pj], m["edt_j"][pi, pj]))
    ch = [local_map(m["overhead"]), None, local_map(sp_channel(raw)), local_map(m["recept"].copy())]
    for k in (2, 3):
        ch[k] = ch[k] - ch[k].min()
    for k in (0, 2, 3):
        ref = (ch[k] * 255).astype(np.uint8)
        assert ref.shape == (lp, lp)
        assert int((ref != obs[..., k]).sum()) <= 4, k
    assert obs[..., 1].max() == 255 and obs[..., 1][lp // 2, lp // 2] == 255 and obs[..., 1][0, 0] == 0
    assert set(np.unique(obs[..., 0])).issubset({0, 31, 95, 127, 191})    # k/8 * 255 for k in 0,1,3,4,6


# ---------------------------------------------------------------- env-level known answers
def test_trials_follow_the_reference_random_stream():
    cfg = default_cfg("box_delivery")
    t = S.generate_trials(cfg, 2)
    rs = np.random.RandomState(42)
    size = max(cfg.agent.length, cfg.agent.width)
    assert t[0]["start"][0] == rs.uniform(-5 + size, 5 - size)
    assert len(t[0]["boxes"]) == 10 and len(t[0]["statics"][1]) == 1 + 4 + 9
    d = np.linalg.norm(t[0]["boxes"][:, None, :2] - t[0]["boxes"][None, :, :2], axis=2) + np.eye(10) * 9
    assert d.min() > cfg.boxes.min_box_dist


def test_push_box_into_receptacle():
    cfg, o = _env(num_boxes=2)
    tr = dict(S.generate_trials(cfg, 1)[0])
    tr["start"] = np.array([1.5, 1.75, 0.0]); tr["boxes"] = np.array([[2.6, 1.75, 0.3], [-3.0, -1.0, 0.0]])
    o.reset(tr, observe=False)
    _, r0, term, _, info = o.step(1.0, observe=False)
    assert 0 < r0 < 1 and info["substeps"] > 500 and not term
    # straight drive along +x; the two until-still sub-steps keep the last commanded 0.6 m/s (kinematic body)
    assert abs(info["robot_distance"] + 2 * 0.6 * 0.002 - (info["x"] - 1.5)) < 1e-9 and abs(info["y"] - 1.75) < 1e-9
    _, r1, term, _, info = o.step(1.0, observe=False)
    assert r1 > 10 and list(o.alive()) == [0, 1] and info["cumulative_boxes"] == 1 and info["inactivity"] == 0 and not term
    obs, r2, _, _, info = o.step(1.0)
    assert r2 == 0.0 and info["num_boxes_left"] == 1
    assert (obs[..., 0] == 95).sum() > 0 and (obs[..., 0] == 191).sum() > 0   # receptacle (3/8) and robot (6/8) are in view


def test_boxes_stay_in_room_and_state_is_finite():
    cfg, o = _env()
    o.reset(S.generate_trials(cfg, 3)[2], observe=False)
    rng = np.random.RandomState(9)
    for _ in range(25):
        _, r, term, trunc, info = o.step(rng.uniform(-1, 1), observe=False)
        st = o.shape_states()
        assert np.isfinite(st).all() and np.isfinite(r)
        boxes = st[6:16, :2]
        assert np.all(np.abs(boxes[:, 0]) < 5.05) and np.all(np.abs(boxes[:, 1]) < 2.55)
        assert 2 <= info["substeps"] <= 10002 + 10002


def test_inverted_receptacle_map_equals_the_numpy_statement_of_the_reference():
    """box_delivery_env.py:1126-1128 on float32 arrays: global_map += 1 - cspace; global_map[global_map == (1 - cspace)] = 1."""
    cfg = default_cfg("box_delivery")
    cfg.env.obstacle_config = "small_columns"
    trial = S.generate_trials(cfg, 1)[0]
    _, plain = _env(cfg)
    plain.reset(trial, observe=False)
    cfg.env.invert_receptacle_map = True
    _, inv = _env(cfg)
    inv.reset(trial, observe=False)
    pm, im = plain.maps(), inv.maps()
    g = pm["recept"].copy()
    g += 1 - pm["cspace"]
    g[g == (1 - pm["cspace"])] = 1
    assert g.dtype == np.float32 and np.array_equal(g, im["recept"])
    assert (im["recept"][pm["cspace"] == 0] == 1).all() and not np.array_equal(pm["recept"], im["recept"])


def test_chokepoint_layout_has_a_diagonal_only_corridor_and_skips_buckets():
    """The regression layout of tests/chokepoint_layout.py really is what its docstring says (oracle maps + oracle spfa): the pocket and the room are
    8-connected but not 4-connected, and the search from the robot leaves distance buckets empty between non-empty ones."""
    from collections import deque
    from benchpush_amd import box_delivery_scenario as S
    from benchpush_amd.config import default_cfg
    from oracle.oracle_bd import OracleBoxDelivery, spfa
    from chokepoint_layout import make_chokepoint_trial
    cfg = default_cfg("box_delivery")
    cfg.boxes.num_boxes_small = 3
    tr = make_chokepoint_trial(cfg)
    bp = S.box_delivery_params(cfg)
    bp["num_boxes"] = len(tr["boxes"])
    o = OracleBoxDelivery(S.box_delivery_physics_params(cfg), bp, cfg)
    o.reset(tr, observe=False)
    m = o.maps()
    cs = m["cspace"] > 0.5
    H, W = cs.shape

    def reach(conn8, src):
        seen = np.zeros_like(cs)
        seen[src] = True
        dq = deque([src])
        while dq:
            i, j = dq.popleft()
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    if (di or dj) and (conn8 or di == 0 or dj == 0):
                        a, b = i + di, j + dj
                        if 0 <= a < H and 0 <= b < W and cs[a, b] and not seen[a, b]:
                            seen[a, b] = True
                            dq.append((a, b))
        return seen

    ppm = 224 / 10.0
    pi, pj = int(np.floor(-tr["start"][1] * ppm + H / 2)), int(np.floor(tr["start"][0] * ppm + W / 2))
    src = (int(m["edt_i"][pi, pj]), int(m["edt_j"][pi, pj]))
    assert cs[src]
    r8, r4 = reach(True, src), reach(False, src)
    assert r8.sum() == cs.sum() and r4.sum() < 0.2 * cs.sum()          # everything is reachable, but only over diagonal steps
    dist, _, _ = spfa(m["cspace"], src)
    d = dist[r8]
    cnt = np.bincount(np.floor(d).astype(int))
    empties = [k for k in range(len(cnt) - 1) if cnt[k] == 0]
    assert len(empties) >= 8 and all(cnt[k + 1] > 0 for k in empties)    # single empty buckets between non-empty ones
