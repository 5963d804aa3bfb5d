"""Layouts for area-clearing-v0: start pose, box placement, walls and goal points, restating
AreaClearingEnv.init_area_clearing_env / generate_obstacles / generate_walls / generate_static_obstacles / _compute_boundary_goals
(benchpush/environments/area_clearing/area_clearing.py:225-264,361-561).  The reference draws from the unseeded module-level
``random``; here trial t uses ``random.Random(base_seed + t)`` with the same draw order (start x, then box centres).
shapely pieces are restated for the shipped layouts: ``create_polygon_from_line`` = the flat-capped 0.2 m wide rectangle around
the segment, boundary goal lines = the boundary edges (none of the shipped walls cuts an edge), ``interpolate`` = linear.
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


def goal_points(cfg, interpolated_points=10):
    lay = env_layout(cfg)
    b = lay.boundary
    walls = lay.walls if "walls" in lay else []
    for w in walls:   # the restatement covers layouts whose walls (buffered by 0.1) stay clear of the boundary edges
        rect = line_rectangle(w, 0.2)
        x0, x1, y0, y1 = _extent(rect)
        bx0, bx1, by0, by1 = _extent(b)
        touches = (x0 <= bx1 and x1 >= bx0 and y0 <= by1 and y1 >= by0) and not (x0 > bx0 and x1 < bx1 and y0 > by0 and y1 < by1)
        if touches:
            raise NotImplementedError("walls that cut the clearance boundary need shapely's difference()")
    pts = []
    for i in range(len(b)):
        p0, p1 = b[i], b[(i + 1) % len(b)]
        length = math.hypot(p1[0] - p0[0], p1[1] - p0[1])
        if not length > 0.1:
            continue
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
