"""Layouts for area-clearing-v0: start pose, box placement, walls and goal points, restating
AreaClearingEnv.init_area_clearing_env / generate_obstacles / generate_walls / generate_static_obstacles / _compute_boundary_goals
(benchpush/environments/area_clearing/area_clearing.py:225-264,361-561).  The reference draws from the unseeded module-level
``random``; here trial t uses ``random.Random(base_seed + t)`` with the same draw order (start x, then box centres).
shapely pieces are restated for the shipped layouts: ``create_polygon_from_line`` = the flat-capped 0.2 m wide rectangle around
the segment, boundary goal lines = the boundary edges minus the walls' 0.1 m buffers (``cut_edge_by_wall``: exact where a wall crosses an
edge, GEOS' polygon of the round cap where a wall's end reaches one -- ``segment_buffer_ring``, restated from GEOS' published offset-curve algorithm, unpinned; none of the
shipped walls does either), ``interpolate`` = linear.
"""
import math
import random as _random

import numpy as np

OBSTACLE = 3


def env_layout(cfg):
    return cfg.envs[cfg.env]


def robot_radius(cfg):
    return ((cfg.agent.length ** 2 + cfg.agent.width ** 2) ** 0.5 / 2) * 1.2   # area_clearing.py:188


def _extent(v):
    xs, ys = [p[0] for p in v], [p[1] for p in v]
    return min(xs), max(xs), min(ys), max(ys)


def line_rectangle(line, width=0.2):
    """create_polygon_from_line (common/geometry/polygon.py:202-208): LineString.buffer(width/2, cap flat, join mitre)."""
    (x0, y0), (x1, y1) = line
    dx, dy = x1 - x0, y1 - y0
    L = math.hypot(dx, dy)
    nx, ny = -dy / L * width / 2, dx / L * width / 2
    return [[x0 + nx, y0 + ny], [x1 + nx, y1 + ny], [x1 - nx, y1 - ny], [x0 - nx, y0 - ny]]


def _dist_to_segment(px, py, a, b):
    ax, ay = a
    bx, by = b
    dx, dy = bx - ax, by - ay
    L2 = dx * dx + dy * dy
    t = 0.0 if L2 == 0.0 else max(0.0, min(1.0, ((px - ax) * dx + (py - ay) * dy) / L2))
    return math.hypot(px - (ax + t * dx), py - (ay + t * dy))


def segment_buffer_ring(wall, r=0.1, quad_segs=16):
    """The outline of ``LineString(wall).buffer(r)`` as GEOS builds it for a two-point line with round caps (shapely's defaults: 16 segments per quadrant,
    area_clearing.py:237-238), restated from the published algorithm of GEOS' OffsetCurveBuilder / OffsetSegmentGenerator (computeLineBufferCurve,
    computeOffsetSegment, addLineEndCap, addDirectedFillet) -- the library is not in this image, so this is unpinned:
    left offset of a -> b at b; the cap at b as a clockwise fan of ``quad_segs * 2`` chords from angle + pi/2 to angle - pi/2 (fan point i at
    b + r (cos, sin)(angle + pi/2 - i * pi / (2 quad_segs)); the fan's first point coincides with the offset point and is dropped as GEOS drops vertices closer
    than r * 1e-6 to their predecessor); right offset at b; then the same from b -> a.  Vertices in ring order, not closed."""
    (ax, ay), (bx, by) = wall
    ring = []

    def add(pt):
        if ring and math.hypot(pt[0] - ring[-1][0], pt[1] - ring[-1][1]) < r * 1.0e-6:
            return
        ring.append(pt)

    def offset_p1(p0_, p1_, side_sign):           # computeOffsetSegment(...).p1
        dx, dy = p1_[0] - p0_[0], p1_[1] - p0_[1]
        ln = math.sqrt(dx * dx + dy * dy)
        ux, uy = side_sign * r * dx / ln, side_sign * r * dy / ln
        return (p1_[0] - uy, p1_[1] + ux)

    for p0_, p1_ in (((ax, ay), (bx, by)), ((bx, by), (ax, ay))):
        add(offset_p1(p0_, p1_, 1.0))                                            # addLastSegment
        angle = math.atan2(p1_[1] - p0_[1], p1_[0] - p0_[0])                      # addLineEndCap, CAP_ROUND
        add(offset_p1(p0_, p1_, 1.0))
        start, total = angle + math.pi / 2.0, math.pi
        nseg = int(total / (math.pi / 2.0 / quad_segs) + 0.5)
        inc = total / nseg
        for i in range(nseg):                                                    # addDirectedFillet, clockwise
            a_ = start - i * inc
            add((p1_[0] + r * math.cos(a_), p1_[1] + r * math.sin(a_)))
        add(offset_p1(p0_, p1_, -1.0))
    if math.hypot(ring[0][0] - ring[-1][0], ring[0][1] - ring[-1][1]) < r * 1.0e-6:
        ring.pop()
    return ring


def _cut_edge_by_ring(p0, p1, ring):
    """``LineString([p0, p1]).difference(Polygon(ring))`` for a convex ring: the pieces of the edge outside it, in the direction of the edge.  Crossing points are
    the exact rational intersections of the edge with the ring's segments rounded to the nearest double (GEOS intersects in double-double arithmetic, which
    rounds the same way except in contrived cases); a coordinate along which the edge does not move stays the edge's own."""
    from fractions import Fraction as F
    x0, y0, x1, y1 = F(p0[0]), F(p0[1]), F(p1[0]), F(p1[1])
    ex, ey = x1 - x0, y1 - y0
    n = len(ring)
    # orientation of the ring, then "inside" = on the inner side of every edge (convex)
    area2 = sum(F(ring[i][0]) * F(ring[(i + 1) % n][1]) - F(ring[(i + 1) % n][0]) * F(ring[i][1]) for i in range(n))
    sgn = 1 if area2 > 0 else -1

    def inside(px, py):
        for i in range(n):
            qx, qy, sx, sy = F(ring[i][0]), F(ring[i][1]), F(ring[(i + 1) % n][0]), F(ring[(i + 1) % n][1])
            if sgn * ((sx - qx) * (py - qy) - (sy - qy) * (px - qx)) < 0:
                return False
        return True

    ts = []
    for i in range(n):
        qx, qy, sx, sy = F(ring[i][0]), F(ring[i][1]), F(ring[(i + 1) % n][0]), F(ring[(i + 1) % n][1])
        dx, dy = sx - qx, sy - qy
        den = ex * dy - ey * dx
        if den == 0:
            continue
        t = ((qx - x0) * dy - (qy - y0) * dx) / den
        u = ((qx - x0) * ey - (qy - y0) * ex) / den
        if 0 <= t <= 1 and 0 <= u <= 1:
            ts.append(t)
    in0, in1 = inside(x0, y0), inside(x1, y1)
    if not ts:
        return [] if (in0 and in1) else [(list(p0), list(p1))]

    def point(t):
        return [p0[0] if ex == 0 else float(x0 + t * ex), p0[1] if ey == 0 else float(y0 + t * ey)]

    out = []
    if not in0 and min(ts) > 0:
        out.append((list(p0), point(min(ts))))
    if not in1 and max(ts) < 1:
        out.append((point(max(ts)), list(p1)))
    return out


def cut_edge_by_wall(p0, p1, wall, r=0.1):
    """``LineString([p0, p1]).difference(LineString(wall).buffer(r))`` (area_clearing.py:236-240) for one boundary edge and one wall: the parts of the
    edge outside the wall's 0.1 m buffer, in the direction of the edge.  The buffer of a segment is convex (a stadium), so it covers one interval of
    the edge.  Where the interval ends on a *straight side* of the buffer (a wall that crosses the edge) the end point is the exact intersection with
    the offset line; where it ends on a round cap (a wall whose END reaches the edge, area_clearing.py:225-262) the buffer is the polygon GEOS makes of the cap
    -- 16 chords per quadrant, ``segment_buffer_ring`` -- and the end point is the crossing with that polygon (up to 0.12 mm inside the circle)."""
    (ax, ay), (bx, by) = wall
    wx, wy = bx - ax, by - ay
    wl = math.hypot(wx, wy)
    ex, ey = p1[0] - p0[0], p1[1] - p0[1]
    inside0 = _dist_to_segment(p0[0], p0[1], wall[0], wall[1]) <= r
    inside1 = _dist_to_segment(p1[0], p1[1], wall[0], wall[1]) <= r
    ts = []       # parameters of the edge where it crosses the buffer's outline, with the kind of outline
    nx, ny = -wy / wl, wx / wl
    for sgn in (1.0, -1.0):   # the two straight sides: points a + s * w + sgn * r * n, 0 <= s <= 1
        ox, oy = ax + sgn * r * nx, ay + sgn * r * ny
        den = ex * wy - ey * wx
        if den == 0.0:
            continue
        t = ((ox - p0[0]) * wy - (oy - p0[1]) * wx) / den
        sp = ((p0[0] + t * ex - ox) * wx + (p0[1] + t * ey - oy) * wy) / (wl * wl)
        if 0.0 <= t <= 1.0 and 0.0 <= sp <= 1.0:
            # the crossing point from both parametrisations; a coordinate along which one of the two lines does not move is exact there (axis-aligned
            # edges and walls, the usual case, come out as GEOS computes them: (edge x, wall y +- 0.1))
            qx = p0[0] if ex == 0.0 else (ox if wx == 0.0 else p0[0] + t * ex)
            qy = p0[1] if ey == 0.0 else (oy if wy == 0.0 else p0[1] + t * ey)
            ts.append((t, "side", (qx, qy)))
    for cx, cy, outward in ((ax, ay, -1.0), (bx, by, 1.0)):   # the two caps: the half circles beyond the wall's ends
        fx, fy = p0[0] - cx, p0[1] - cy
        qa, qb, qc = ex * ex + ey * ey, 2 * (fx * ex + fy * ey), fx * fx + fy * fy - r * r
        disc = qb * qb - 4 * qa * qc
        if disc <= 0.0:
            continue
        for t in ((-qb - math.sqrt(disc)) / (2 * qa), (-qb + math.sqrt(disc)) / (2 * qa)):
            hx, hy = p0[0] + t * ex - cx, p0[1] + t * ey - cy
            if 0.0 <= t <= 1.0 and (hx * wx + hy * wy) * outward > 0.0:
                ts.append((t, "cap", None))
    if not ts and not inside0:
        return [(list(p0), list(p1))]                      # the wall's buffer does not reach this edge
    if inside0 and inside1 and not ts:
        return []   # the whole edge lies inside the buffer (convex: both ends inside, no crossing): LineString.difference is empty, the length filter drops it
    if any(k == "cap" for _, k, _ in ts):
        return _cut_edge_by_ring(p0, p1, segment_buffer_ring(wall, r))
    first = None if inside0 else min(ts, key=lambda c: c[0])
    last = None if inside1 else max(ts, key=lambda c: c[0])
    out = []
    if first is not None and first[0] > 0.0:
        out.append((list(p0), list(first[2])))
    if last is not None and last[0] < 1.0:
        out.append((list(last[2]), list(p1)))
    return out


def boundary_goal_lines(cfg):
    """_compute_boundary_goals (area_clearing.py:225-255): the boundary edges minus every wall's 0.1 m buffer, pieces longer than 0.1 m, in edge order."""
    lay = env_layout(cfg)
    b = lay.boundary
    lines = [([float(b[i][0]), float(b[i][1])], [float(b[(i + 1) % len(b)][0]), float(b[(i + 1) % len(b)][1])]) for i in range(len(b))]
    for w in (lay.walls if "walls" in lay else []):
        wall = ((float(w[0][0]), float(w[0][1])), (float(w[1][0]), float(w[1][1])))
        nxt = []
        for q0, q1 in lines:      # the reference cuts edge i by every wall in turn; a piece of an edge stays in the edge's place
            nxt.extend(cut_edge_by_wall(q0, q1, wall))
        lines = nxt
    return [(q0, q1) for q0, q1 in lines if math.hypot(q1[0] - q0[0], q1[1] - q0[1]) > 0.1]


def goal_points(cfg, interpolated_points=10):
    """10 points per boundary goal line at ((k + 1/2) / 10) of its length (area_clearing.py:257-262)."""
    pts = []
    for p0, p1 in boundary_goal_lines(cfg):
        length = math.hypot(p1[0] - p0[0], p1[1] - p0[1])
        for k in range(int(interpolated_points)):
            d = ((k + 1 / 2) / interpolated_points) * length
            f = d / length
            pts.append([p0[0] + (p1[0] - p0[0]) * f, p0[1] + (p1[1] - p0[1]) * f])
    return np.array(pts, np.float64)


def static_shapes(cfg):
    """generate_walls + generate_static_obstacles: Poly(space.static_body, verts, radius=0.1), friction 0.99, collision type 3.
    Order as the reference adds them to update_configuration_space (wall_shapes + static_obs_shapes)."""
    lay = env_layout(cfg)
    ob = lay.outer_boundary
    polys = []
    for w in (lay.walls if "walls" in lay else []):
        polys.append(line_rectangle(w, 0.2))
    room_length, room_width = abs(ob[0][0]) * 2, abs(ob[0][1]) * 2
    T = 24
    for x, y, length, width in [(-room_length / 2 - T / 2, 0, T, room_width), (room_length / 2 + T / 2, 0, T, room_width),
                                (0, -room_width / 2 - T / 2, room_length + 2 * T, T), (0, room_width / 2 + T / 2, room_length + 2 * T, T)]:
        polys.append([[x - length / 2, y - width / 2], [x + length / 2, y - width / 2], [x + length / 2, y + width / 2], [x - length / 2, y + width / 2]])
    for so in (lay.static_obstacles if "static_obstacles" in lay else []):
        if len(so) != 4:
            raise NotImplementedError("static obstacles are quadrilaterals in the shipped layouts")
        polys.append([list(p) for p in so])
    n = len(polys)
    return (np.array(polys, np.float64), np.full(n, 4, np.int32), np.zeros((n, 3), np.float64), np.full(n, 0.1, np.float64), np.full(n, OBSTACLE, np.int32))


def generate_trial(cfg, seed):
    rng = _random.Random(seed)
    lay = env_layout(cfg)
    bx0, bx1, by0, by1 = _extent(lay.boundary)
    if cfg.random_start:
        x_start = (bx0 + 1) + rng.random() * ((bx1 - bx0) - 2)
        start = (x_start, by0 + 1.0, np.pi / 2)
    else:
        start = ((bx0 + bx1) / 2, by0 + 1.0, np.pi / 2)
    need, dmin = cfg.num_obstacles, cfg.min_obs_dist
    lo_x, hi_x, lo_y, hi_y = bx0 + 1, bx1 - 1, by0 + 1, by1 - 1
    boxes = []
    while len(boxes) < need:
        cx = rng.random() * (hi_x - lo_x) + lo_x
        cy = rng.random() * (hi_y - lo_y) + lo_y
        if all(((cx - px) ** 2 + (cy - py) ** 2) ** 0.5 > dmin for px, py in boxes):
            boxes.append([cx, cy])
    b3 = np.zeros((need, 3), np.float64)
    b3[:, :2] = np.array(boxes, np.float64)
    return dict(start=np.array(start, np.float64), boxes=b3, statics=static_shapes(cfg))


def generate_trials(cfg, num_trials, base_seed=0):
    return [generate_trial(cfg, base_seed + t) for t in range(num_trials)]


def area_clearing_params(cfg):
    """Scalar parameters shared by the oracle and the C ABI (same struct as box-delivery, task = 1)."""
    lay = env_layout(cfg)
    ox0, ox1, oy0, oy1 = _extent(lay.outer_boundary)
    map_width, map_height = ox1 - ox0, oy1 - oy0
    lp = int(lay.local_map_pixel_width)
    return dict(
        # PositionController(cfg, robot_radius, map_width, map_height, ...) bounds x by map_height / 2 and y by map_width / 2, and
        # the padded room is (map_width, map_height) rows x columns (area_clearing.py:956-960): room_length/room_width carry those roles
        room_length=float(map_height), room_width=float(map_width), recept_x=0.0, recept_y=0.0, recept_size=0.0,
        ppm=lp / float(lay.local_map_width), local_px=lp, local_w=float(lay.local_map_width),
        robot_radius=robot_radius(cfg), robot_half_width=robot_radius(cfg),
        step_size=float(cfg.agent.movement_step_size), target_speed=float(cfg.controller.target_speed), ctrl_dt=float(cfg.controller.dt),
        steps=int(cfg.sim.steps), partial_rewards_scale=0.0, goal_reward=0.0, collision_penalty=0.0, non_movement_penalty=0.0,
        correct_direction_reward_scale=1.0, use_correct_direction_reward=0, inactivity_cutoff=0, ministep_size=2.5, sp_channel_scale=1.0,
        invert_receptacle_map=0, num_boxes=int(cfg.num_obstacles), step_limit=5000,
        action_type={'heading': 0, 'position': 1, 'velocity': 2}[cfg.agent.action_type],
        task=1, omega_scale=0.5, v_scale=5.0, lfc=float(cfg.controller.Lfc),        # area_clearing.py:903-906, config.yaml:138
        yaw_rate_step=(np.pi / 2) / 15, t_max=int(cfg.sim.t_max),                    # :198, config.yaml:105
        boundary_penalty=-0.25, box_cleared_reward=10.0, box_putback_penalty=-10.0, truncation_penalty=0.0, terminal_reward=50.0,
        pushing_mult=0.2, distance_scale_max=0.5)                                    # constants of area_clearing.py:37-49


def area_clearing_physics_params(cfg):
    dt_sub = cfg.controller.dt / cfg.sim.steps
    return dict(dt=float(cfg.controller.dt), steps=int(cfg.sim.steps), iterations=int(cfg.sim.iterations), persistence=3, settle_steps=1000,
                damping_pow=math.pow(float(cfg.sim.damping), dt_sub), bias_coef=1.0 - math.pow(math.pow(1.0 - 0.1, 60.0), dt_sub), slop=0.1)
