"""Layouts for box-delivery-v0: room boundary, random robot start and box placement, restating
BoxDeliveryEnv.get_random_robot_start / generate_boundary / generate_boxes
(benchpush/environments/box_delivery/box_delivery_env.py:313-562) on ``np.random.RandomState`` exactly as the reference
draws them (one generator per env, seeded with ``cfg.misc.random_seed``, consumed episode after episode), and the
static shapes of ``generate_sim_bounds`` / ``create_corners`` (benchpush/common/utils/sim_utils.py:75-135).
"""
import math

import numpy as np

OBSTACLE, RECEPTACLE = 3, 4   # collision types (box_delivery_env.py:259-265)


def room_dims(cfg):
    small = cfg.env.obstacle_config.split("_")[0] == "small"
    return (float(cfg.env.room_length), float(cfg.env.room_width_small if small else cfg.env.room_width_large),
            int(cfg.boxes.num_boxes_small if small else cfg.boxes.num_boxes_large))


def robot_radius(cfg):
    return ((cfg.agent.length ** 2 + cfg.agent.width ** 2) ** 0.5 / 2) * 1.2   # box_delivery_env.py:122


def _rect(x, y, length, width):
    return [[x - length / 2, y - width / 2], [x + length / 2, y - width / 2], [x + length / 2, y + width / 2], [x - length / 2, y + width / 2]]


def random_start(cfg, rs):
    L, Wd, _ = room_dims(cfg)
    size = max(cfg.agent.length, cfg.agent.width)
    x = rs.uniform(-L / 2 + size, L / 2 - size)
    y = rs.uniform(-Wd / 2 + size, Wd / 2 - size)
    h = rs.uniform(0, 2 * np.pi)
    return (x, y, h)


def generate_boundary(cfg, rs, start):
    """List of dicts like the reference's boundary_dicts (+ possibly a re-drawn start for 'large_divider')."""
    L, Wd, _ = room_dims(cfg)
    T = float(cfg.env.wall_thickness)
    size = float(cfg.env.receptacle_width)
    rx, ry = L / 2 - size / 2, Wd / 2 - size / 2
    out = [dict(type="receptacle", position=(rx, ry), vertices=_rect(rx, ry, size, size), length=size, width=size)]
    for x, y, length, width in [(-L / 2 - T / 2, 0, T, Wd), (L / 2 + T / 2, 0, T, Wd),
                                (0, -Wd / 2 - T / 2, L + 2 * T, T), (0, Wd / 2 + T / 2, L + 2 * T, T)]:
        out.append(dict(type="wall", position=(x, y), vertices=_rect(x, y, length, width)))
    rr = robot_radius(cfg)
    oc = cfg.env.obstacle_config
    if oc in ("small_columns", "large_columns"):
        num = rs.randint(1, 3 if oc == "small_columns" else 8)
        cl = cw = 1
        buf, mind = 0.8, 2
        cols = []
        for _ in range(num):
            for _ in range(100):
                x = rs.uniform(-L / 2 + 2 * buf + cl / 2, L / 2 - 2 * buf - cl / 2)
                y = rs.uniform(-Wd / 2 + 2 * buf + cw / 2, Wd / 2 - 2 * buf - cw / 2)
                if ((x - rx) ** 2 + (y - ry) ** 2) ** 0.5 <= mind / 2 + size / 2:
                    break
                if ((x - start[0]) ** 2 + (y - start[1]) ** 2) ** 0.5 <= mind / 2 + rr:
                    break
                overlapped = False
                for px, py in cols:
                    if ((x - px) ** 2 + (y - py) ** 2) ** 0.5 <= mind:
                        overlapped = True
                        break
                if not overlapped:
                    cols.append([x, y])
                    break
        for x, y in cols:
            out.append(dict(type="column", position=(x, y), vertices=_rect(x, y, cl, cw), length=cl, width=cw))
    elif oc == "large_divider":
        dl, dw, buf = 8, 0.5, 3.5
        new = []
        for _ in range(100):
            for _ in range(100):
                x = L / 2 - dl / 2
                y = rs.uniform(-Wd / 2 + buf + dw / 2, Wd / 2 - buf - dw / 2)
                if not ((x - start[0]) ** 2 + (y - start[1]) ** 2) ** 0.5 <= 3 * rr:
                    new.append([x, y])
                    break
            if len(new) == 1:
                break
            start = random_start(cfg, rs)
        for x, y in new:
            out.append(dict(type="divider", position=(x, y), vertices=_rect(x, y, dl, dw), length=dl, width=dw))
    elif oc != "small_empty":
        raise ValueError(f"Invalid obstacle config: {oc}")
    for i, (x, y) in enumerate([(-L / 2, Wd / 2), (L / 2, Wd / 2), (L / 2, -Wd / 2), (-L / 2, -Wd / 2)]):
        if i == 1:
            continue
        out.append(dict(type="corner", position=(x, y), heading=-np.radians(i * 90)))
    for ob in list(out):
        if ob["type"] == "divider":
            (x, y), w = ob["position"], ob["width"]
            for pos, hd in zip([(L / 2, y - w / 2), (L / 2, y + w / 2)], [-90, 180]):
                out.append(dict(type="corner", position=pos, heading=np.radians(hd)))
    return out, start


def generate_boxes(cfg, rs, boundary):
    """[n, 3] = x, y, heading (box_delivery_env.py:509-562)."""
    L, Wd, n = room_dims(cfg)
    half = cfg.boxes.box_size / 2
    dmin = cfg.boxes.min_box_dist
    lo_x, hi_x, lo_y, hi_y = -L / 2 + half, L / 2 - half, -Wd / 2 + half, Wd / 2 - half
    boxes = []
    while len(boxes) < n:
        cx, cy = rs.uniform(lo_x, hi_x), rs.uniform(lo_y, hi_y)
        h = rs.uniform(0, 2 * np.pi)
        overlapped = False
        for ob in boundary:
            if ob["type"] in ("corner", "wall"):
                continue
            if ob["type"] == "divider":
                if abs(cy - ob["position"][1]) <= (dmin / 2 + ob["width"] / 2) * 1.2:
                    overlapped = True
                    break
            elif ((cx - ob["position"][0]) ** 2 + (cy - ob["position"][1]) ** 2) ** 0.5 <= (dmin / 2 + ob["width"] / 2) * 1.2:
                overlapped = True
                break
        for px, py, _ in boxes:
            if ((cx - px) ** 2 + (cy - py) ** 2) ** 0.5 <= dmin:
                overlapped = True
                break
        if not overlapped:
            boxes.append([cx, cy, h])
    return np.array(boxes, np.float64)


_S, _C = 0.5 * np.sin(22.5 * np.pi / 180), 0.5 * np.cos(22.5 * np.pi / 180)
CORNER_TRIANGLES = [[(0, 0), (0, -1), (_S, -1 + _C)], [(0, 0), (_S, -1 + _C), (1 - _C, -_S)], [(0, 0), (1, 0), (1 - _C, -_S)]]  # sim_utils.py:96-110


def static_shapes(boundary):
    """generate_sim_bounds order: non-corner shapes first (radius 0, body at the origin), then 3 triangles per corner
    (radius 0.02, body at the corner with its heading).  Returns verts [n,4,2], counts, poses [n,3], radii, types."""
    verts, counts, poses, radii, types = [], [], [], [], []
    for ob in boundary:
        if ob["type"] == "corner":
            continue
        verts.append(np.array(ob["vertices"], np.float64)); counts.append(4); poses.append([0.0, 0.0, 0.0]); radii.append(0.0)
        types.append(RECEPTACLE if ob["type"] == "receptacle" else OBSTACLE)
    for ob in boundary:
        if ob["type"] != "corner":
            continue
        for tri in CORNER_TRIANGLES:
            v = np.zeros((4, 2)); v[:3] = np.array(tri, np.float64)
            verts.append(v); counts.append(3); poses.append([ob["position"][0], ob["position"][1], float(ob["heading"])]); radii.append(0.02)
            types.append(OBSTACLE)
    return (np.array(verts, np.float64), np.array(counts, np.int32), np.array(poses, np.float64), np.array(radii, np.float64),
            np.array(types, np.int32))


def generate_trials(cfg, num_trials, seed=None):
    """Consecutive episodes of one env: the reference keeps a single RandomState across resets (box_delivery_env.py:149)."""
    rs = np.random.RandomState(cfg.misc.random_seed if seed is None else seed)
    trials = []
    for _ in range(num_trials):
        start = random_start(cfg, rs) if cfg.agent.random_start else (5, 1.5, np.pi * 3 / 2)
        boundary, start = generate_boundary(cfg, rs, start)
        boxes = generate_boxes(cfg, rs, boundary)
        trials.append(dict(start=np.array(start, np.float64), boxes=boxes, boundary=boundary, statics=static_shapes(boundary)))
    return trials


def box_delivery_params(cfg):
    """Scalar parameters shared by the oracle and the C ABI."""
    L, Wd, n = room_dims(cfg)
    size = float(cfg.env.receptacle_width)
    sam = cfg.train.job_type == "sam"
    lp = int(cfg.env.local_map_pixel_width_sam if sam else cfg.env.local_map_pixel_width)
    rw = cfg.rewards_sam if sam else cfg.rewards
    return dict(
        room_length=L, room_width=Wd, recept_x=L / 2 - size / 2, recept_y=Wd / 2 - size / 2, recept_size=size,
        ppm=lp / float(cfg.env.local_map_width), local_px=lp, local_w=float(cfg.env.local_map_width),
        robot_radius=robot_radius(cfg), robot_half_width=max(cfg.agent.length, cfg.agent.width) / 2,
        step_size=float(cfg.agent.step_size), target_speed=float(cfg.controller.target_speed), ctrl_dt=float(cfg.controller.dt),
        steps=int(cfg.sim.steps),
        partial_rewards_scale=float(rw.partial_rewards_scale), goal_reward=float(rw.goal_reward), collision_penalty=float(rw.collision_penalty),
        non_movement_penalty=float(rw.non_movement_penalty), correct_direction_reward_scale=float(rw.correct_direction_reward_scale),
        use_correct_direction_reward=int(bool(cfg.train.use_correct_direction_reward)),
        inactivity_cutoff=int(cfg.misc.inactivity_cutoff_sam if sam else cfg.misc.inactivity_cutoff),
        ministep_size=float(cfg.misc.ministep_size), sp_channel_scale=float(cfg.env.shortest_path_channel_scale),
        invert_receptacle_map=int(bool(cfg.env.invert_receptacle_map)), num_boxes=n, step_limit=10000,
        action_type={'heading': 0, 'position': 1, 'velocity': 2}[cfg.agent.action_type],
        task=0, omega_scale=3.0, v_scale=2.0, lfc=0.0)   # apply_controller: omega*3, v*2 (box_delivery_env.py:887-889); DP Lfc default 0


def box_delivery_physics_params(cfg):
    """Chipmunk step parameters (defaults as in config.ship_ice_physics_params; sub-step dt = controller.dt / sim.steps)."""
    dt_sub = cfg.controller.dt / cfg.sim.steps
    return dict(dt=float(cfg.controller.dt), steps=int(cfg.sim.steps), iterations=int(cfg.sim.iterations), persistence=3, settle_steps=1000,
                damping_pow=math.pow(float(cfg.sim.damping), dt_sub), bias_coef=1.0 - math.pow(math.pow(1.0 - 0.1, 60.0), dt_sub), slop=0.1)
