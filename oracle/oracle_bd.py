"""ctypes wrapper around oracle/bp_oracle_bd.c (CPU restatement of box-delivery-v0's env.step() path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C

import numpy as np

from .oracle import OrcParams, _p, lib

BD_INFO_KEYS = ["x", "y", "theta", "cumulative_distance", "cumulative_boxes", "cumulative_reward", "total_work", "ministeps", "inactivity",
                "robot_hit_obstacle", "substeps", "robot_distance", "boxes_distance", "num_waypoints", "num_boxes_left", "work"]


class BdParams(C.Structure):
    _fields_ = [("room_length", C.c_double), ("room_width", C.c_double), ("recept_x", C.c_double), ("recept_y", C.c_double),
                ("recept_size", C.c_double), ("ppm", C.c_double), ("local_px", C.c_int), ("local_w", C.c_double),
                ("robot_radius", C.c_double), ("robot_half_width", C.c_double), ("step_size", C.c_double),
                ("target_speed", C.c_double), ("ctrl_dt", C.c_double), ("steps", C.c_int),
                ("partial_rewards_scale", C.c_double), ("goal_reward", C.c_double), ("collision_penalty", C.c_double),
                ("non_movement_penalty", C.c_double), ("correct_direction_reward_scale", C.c_double),
                ("use_correct_direction_reward", C.c_int), ("inactivity_cutoff", C.c_int), ("ministep_size", C.c_double),
                ("sp_channel_scale", C.c_double), ("invert_receptacle_map", C.c_int), ("num_boxes", C.c_int), ("step_limit", C.c_int),
                ("action_type", C.c_int), ("task", C.c_int), ("omega_scale", C.c_double), ("v_scale", C.c_double), ("lfc", C.c_double),
                ("yaw_rate_step", C.c_double), ("t_max", C.c_int), ("boundary_penalty", C.c_double), ("box_cleared_reward", C.c_double),
                ("box_putback_penalty", C.c_double), ("truncation_penalty", C.c_double), ("terminal_reward", C.c_double),
                ("pushing_mult", C.c_double), ("distance_scale_max", C.c_double)]


_ready = False


def bdlib():
    global _ready
    L = lib()
    if not _ready:
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.orc_bd_create.restype = vp
        L.orc_bd_create.argtypes = [C.POINTER(OrcParams), C.POINTER(BdParams)]
        L.orc_bd_destroy.argtypes = [vp]
        L.orc_bd_dims.argtypes = [vp, vp]
        L.orc_bd_reset.restype = ci
        L.orc_bd_reset.argtypes = [vp, vp, vp, vp, vp, ci, vp, cd, cd, ci, vp, vp, vp, vp, vp]
        L.orc_bd_step.argtypes = [vp, cd, vp, C.POINTER(cd), C.POINTER(ci), C.POINTER(ci), vp]
        L.orc_bd_step2.argtypes = [vp, cd, cd, vp, C.POINTER(cd), C.POINTER(ci), C.POINTER(ci), vp]
        L.orc_bd_observe.argtypes = [vp, vp]
        L.orc_bd_get_maps.argtypes = [vp] + [vp] * 7
        L.orc_bd_physics.restype = vp
        L.orc_bd_physics.argtypes = [vp]
        L.orc_bd_num_alive.restype = ci
        L.orc_bd_num_alive.argtypes = [vp]
        L.orc_bd_get_alive.argtypes = [vp, vp]
        L.orc_bd_last_waypoints.restype = ci
        L.orc_bd_last_waypoints.argtypes = [vp, vp]
        L.orc_bd_shortest_path.restype = ci
        L.orc_bd_shortest_path.argtypes = [vp, vp, vp, ci, vp]
        L.orc_bd_world_verts.argtypes = [vp, vp, vp]
        L.orc_bd_execute_path.argtypes = [vp, ci, vp, vp]
        L.orc_bd_local_map.argtypes = [vp, vp, cd, cd, cd, vp]
        L.orc_bd_controller_trace.argtypes = [vp, cd, cd, cd, ci, vp, vp]
        L.orc_bd_plan.restype = ci
        L.orc_bd_plan.argtypes = [vp, ci, ci, vp, vp, C.POINTER(cd)]
        L.orc_bd_set_all_free.argtypes = [vp]
        L.orc_ac_set_geometry.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp]
        L.orc_atan2.restype = cd
        L.orc_atan2.argtypes = [cd, cd]
        L.orc_pymod.restype = cd
        L.orc_pymod.argtypes = [cd, cd]
        L.orc_bd_fill_poly.argtypes = [vp, ci, ci, ci, vp, vp, C.c_float]
        L.orc_bd_sk_line.restype = ci
        L.orc_bd_sk_line.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, vp, vp]
        L.orc_bd_approx_polygon.argtypes = [ci, vp, vp, cd, vp]
        L.orc_bd_spfa.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp]
        L.orc_bd_dilate.argtypes = [vp, ci, ci, ci, vp]
        L.orc_bd_edt.argtypes = [vp, ci, ci, vp, vp]
        L.orc_bd_rotate0.argtypes = [vp, ci, ci, cd, cd, C.POINTER(ci), C.POINTER(ci), vp]
        L.orc_bd_point_in_shape.restype = ci
        L.orc_bd_point_in_shape.argtypes = [vp, ci, cd, cd]
        _ready = True
    return L


class OracleBoxDelivery:
    """Single-env oracle with the reference's reset()/step() shape (box_delivery_env.py:578-830), heading actions."""

    def __init__(self, phys_params, bd_params, cfg):
        self.L = bdlib()
        p = OrcParams()
        for k, v in phys_params.items():
            setattr(p, k, v)
        b = BdParams()
        for k, v in bd_params.items():
            setattr(b, k, v)
        self.bd = dict(bd_params)
        self.h = self.L.orc_bd_create(C.byref(p), C.byref(b))
        d = np.zeros(4, np.int32)
        self.L.orc_bd_dims(self.h, _p(d))
        self.H, self.W, self.SH, self.SW = (int(x) for x in d)
        self.lp = int(bd_params["local_px"])
        self.robot_verts = np.ascontiguousarray(cfg.agent.vertices, np.float64)
        self.wheel_verts = np.ascontiguousarray(cfg.agent.wheel_vertices, np.float64)
        self.bumper_verts = np.ascontiguousarray(cfg.agent.front_bumper_vertices, np.float64)
        self.half = float(cfg.boxes.box_size) / 2
        self.density = float(cfg.boxes.box_density)
        self.nbox = 0

    def __del__(self):
        try:
            if self.h:
                self.L.orc_bd_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self, trial, observe=True):
        verts, counts, poses, radii, types = trial["statics"]
        boxes = np.ascontiguousarray(trial["boxes"], np.float64)
        start = np.ascontiguousarray(trial["start"], np.float64)
        self.nbox = len(boxes)
        self.nstatic = len(counts)
        self.ns = self.L.orc_bd_reset(self.h, _p(start), _p(self.robot_verts), _p(self.wheel_verts), _p(self.bumper_verts), self.nbox, _p(boxes),
                                      self.half, self.density, self.nstatic, _p(np.ascontiguousarray(verts)), _p(np.ascontiguousarray(counts)),
                                      _p(np.ascontiguousarray(poses)), _p(np.ascontiguousarray(radii)), _p(np.ascontiguousarray(types)))
        return self.observe() if observe else None

    def observe(self):
        obs = np.zeros((self.lp, self.lp, 4), np.uint8)
        self.L.orc_bd_observe(self.h, _p(obs))
        return obs

    def step(self, action, observe=True):
        """action: heading in [-1, 1], a position index, or (linear, angular) for 'velocity' (bd_params['action_type'] 0 / 1 / 2)."""
        obs = np.zeros((self.lp, self.lp, 4), np.uint8) if observe else None
        r, t, tr = C.c_double(), C.c_int(), C.c_int()
        info = np.zeros(len(BD_INFO_KEYS), np.float64)
        a = np.asarray(action, np.float64).reshape(-1)
        self.L.orc_bd_step2(self.h, float(a[0]), float(a[1]) if len(a) > 1 else 0.0, _p(obs) if observe else None, C.byref(r), C.byref(t),
                            C.byref(tr), _p(info))
        return obs, r.value, bool(t.value), bool(tr.value), dict(zip(BD_INFO_KEYS, info.tolist()))

    def maps(self):
        N = (self.H, self.W)
        out = dict(cspace=np.zeros(N, np.float32), cspace_thin=np.zeros(N, np.float32), edt_i=np.zeros(N, np.int32), edt_j=np.zeros(N, np.int32),
                   recept=np.zeros(N, np.float32), small_free=np.zeros((self.SH, self.SW), np.float32), overhead=np.zeros(N, np.float32))
        self.L.orc_bd_get_maps(self.h, *(_p(out[k]) for k in ("cspace", "cspace_thin", "edt_i", "edt_j", "recept", "small_free", "overhead")))
        return out

    def shape_states(self):
        out = np.zeros((self.ns, 9), np.float64)
        self.L.orc_get_shape_states(self.L.orc_bd_physics(self.h), _p(out))
        return out

    def alive(self):
        out = np.zeros(self.nbox, np.int32)
        self.L.orc_bd_get_alive(self.h, _p(out))
        return out

    def last_waypoints(self):
        out = np.zeros((64, 3), np.float64)
        n = self.L.orc_bd_last_waypoints(self.h, _p(out))
        return out[:n]

    def plan(self, spatial_action):
        """get_waypoints_to_spatial_action for a local-map pixel index from the robot's current pose: (waypoints [n, 3], move_sign)."""
        st = self.shape_states()[0]
        pose = np.array([st[0], st[1], float(np.mod(st[2] + np.pi, 2 * np.pi) - np.pi)], np.float64)
        pose[2] = bdlib().orc_pymod(st[2] + np.pi, 2 * np.pi) - np.pi
        row, col = divmod(int(spatial_action), self.lp)
        out = np.zeros((64, 3), np.float64)
        ms = C.c_double()
        n = self.L.orc_bd_plan(self.h, col, row, _p(pose), _p(out), C.byref(ms))
        return out[:n], ms.value

    def world_verts(self):
        """Current world vertices of every shape: list of [n, 2] arrays (robot main, 4 wheels, bumper, boxes, statics)."""
        out = np.zeros((self.ns, 4, 2), np.float64)
        cnt = np.zeros(self.ns, np.int32)
        self.L.orc_bd_world_verts(self.h, _p(out), _p(cnt))
        return [out[i, : cnt[i]].copy() for i in range(self.ns)]

    def execute_path(self, waypoints):
        """Run execute_robot_path for waypoints [n, 3] from the current pose: dict(robot_distance, turn_angle, final, sim_steps)."""
        w = np.ascontiguousarray(waypoints, np.float64)
        out = np.zeros(6, np.float64)
        self.L.orc_bd_execute_path(self.h, len(w), _p(w), _p(out))
        return dict(robot_distance=out[0], turn_angle=out[1], final=out[2:5].copy(), sim_steps=int(out[5]))

    def local_map(self, gmap, x, y, h):
        g = np.ascontiguousarray(gmap, np.float32)
        assert g.shape == (self.H, self.W)
        out = np.zeros((self.lp, self.lp), np.float32)
        self.L.orc_bd_local_map(self.h, _p(g), float(x), float(y), float(h), _p(out))
        return out

    def shortest_path(self, s, t, check_straight=False):
        out = np.zeros((64, 2), np.float64)
        n = self.L.orc_bd_shortest_path(self.h, _p(np.ascontiguousarray(s, np.float64)), _p(np.ascontiguousarray(t, np.float64)), int(check_straight), _p(out))
        return out[:n]


AC_INFO_KEYS = ["x", "y", "theta", "total_work", "collision_reward", "diff_reward", "box_completed_reward", "box_count", "ministeps",
                "robot_hit_obstacle", "substeps", "robot_distance", "t", "num_waypoints", "work", "pushing_reward"]


class OracleAreaClearing(OracleBoxDelivery):
    """area-clearing-v0 on the same restated engine (area_clearing.py:563-778)."""

    def __init__(self, phys_params, ac_params, cfg):
        class _Shim:   # OracleBoxDelivery reads cfg.agent.* and cfg.boxes.*
            pass
        shim = _Shim()
        shim.agent = cfg.agent
        shim.boxes = _Shim()
        shim.boxes.box_size = 2 * float(cfg.obstacle_size)
        shim.boxes.box_density = float(cfg.sim.obstacle_density)
        super().__init__(phys_params, ac_params, shim)
        from benchpush_amd import area_clearing_scenario as A
        lay = A.env_layout(cfg)
        bd = np.ascontiguousarray(lay.boundary, np.float64)
        ob = np.ascontiguousarray(lay.outer_boundary, np.float64)
        gp = np.ascontiguousarray(A.goal_points(cfg), np.float64)
        fp = np.ascontiguousarray(cfg.agent.footprint_vertices, np.float64)
        self.goal_points = gp
        self.L.orc_ac_set_geometry(self.h, len(bd), _p(bd), len(ob), _p(ob), len(gp), _p(gp), _p(fp), None)

    def step(self, action, observe=True):
        obs, r, t, tr, info = super().step(action, observe)
        vals = [info[k] for k in BD_INFO_KEYS]
        return obs, r, t, tr, dict(zip(AC_INFO_KEYS, vals))


# ---- primitive hooks (unit tests) ----
def fill_poly(img, pts, color):
    L = bdlib()
    px = np.ascontiguousarray([p[0] for p in pts], np.int64); py = np.ascontiguousarray([p[1] for p in pts], np.int64)
    L.orc_bd_fill_poly(_p(img), img.shape[0], img.shape[1], len(pts), _p(px), _p(py), float(color))
    return img


def sk_line(r0, c0, r1, c1):
    L = bdlib()
    n = max(abs(r1 - r0), abs(c1 - c0)) + 1
    rr = np.zeros(n + 2, np.int64); cc = np.zeros(n + 2, np.int64)
    k = L.orc_bd_sk_line(r0, c0, r1, c1, _p(rr), _p(cc))
    return rr[:k], cc[:k]


def approx_polygon(coords, tol):
    L = bdlib()
    c = np.ascontiguousarray(coords, np.int64)
    cr = np.ascontiguousarray(c[:, 0]); cc = np.ascontiguousarray(c[:, 1])
    keep = np.zeros(len(c), np.uint8)
    L.orc_bd_approx_polygon(len(c), _p(cr), _p(cc), float(tol), _p(keep))
    return c[keep.astype(bool)]


def spfa(cmap, source):
    L = bdlib()
    m = np.ascontiguousarray(cmap, np.float32)
    dist = np.zeros(m.shape, np.float32); par = np.zeros(m.shape, np.int32); parq = np.zeros(m.shape, np.int32)
    L.orc_bd_spfa(_p(m), m.shape[0], m.shape[1], int(source[0]), int(source[1]), _p(dist), _p(par), _p(parq))
    return dist, par, parq


def dilate_disk(img, r):
    L = bdlib()
    m = np.ascontiguousarray(img, np.float32)
    out = np.zeros_like(m)
    L.orc_bd_dilate(_p(m), m.shape[0], m.shape[1], int(r), _p(out))
    return out


def edt_indices(cspace):
    L = bdlib()
    m = np.ascontiguousarray(cspace, np.float32)
    ii = np.zeros(m.shape, np.int32); jj = np.zeros(m.shape, np.int32)
    L.orc_bd_edt(_p(m), m.shape[0], m.shape[1], _p(ii), _p(jj))
    return ii, jj


def rotate0(img, c, s):
    L = bdlib()
    m = np.ascontiguousarray(img, np.float32)
    oh, ow = C.c_int(), C.c_int()
    L.orc_bd_rotate0(_p(m), m.shape[0], m.shape[1], float(c), float(s), C.byref(oh), C.byref(ow), None)
    out = np.zeros((oh.value, ow.value), np.float32)
    L.orc_bd_rotate0(_p(m), m.shape[0], m.shape[1], float(c), float(s), C.byref(oh), C.byref(ow), _p(out))
    return out


def atan2(y, x):
    return bdlib().orc_atan2(float(y), float(x))


def pymod(a, b):
    return bdlib().orc_pymod(float(a), float(b))


def controller_trace(wp2, lfc, target_speed, dt, poses):
    """DP controller over a pose sequence: rows = omega, vx, vy, setpoint x, setpoint y."""
    L = bdlib()
    w = np.ascontiguousarray(wp2, np.float64); ps = np.ascontiguousarray(poses, np.float64)
    out = np.zeros((len(ps), 5), np.float64)
    L.orc_bd_controller_trace(_p(w), float(lfc), float(target_speed), float(dt), len(ps), _p(ps), _p(out))
    return out


def plan_on_free_map(phys_params, bd_params, x_pixel, y_pixel, pose):
    """get_waypoints_to_spatial_action on an all-free configuration space: (waypoints [n, 3], move_sign)."""
    L = bdlib()
    p = OrcParams()
    for k, v in phys_params.items():
        setattr(p, k, v)
    b = BdParams()
    for k, v in bd_params.items():
        setattr(b, k, v)
    h = L.orc_bd_create(C.byref(p), C.byref(b))
    L.orc_bd_set_all_free(h)
    out = np.zeros((64, 3), np.float64)
    ms = C.c_double()
    n = L.orc_bd_plan(h, int(x_pixel), int(y_pixel), _p(np.ascontiguousarray(pose, np.float64)), _p(out), C.byref(ms))
    L.orc_bd_destroy(h)
    return out[:n], ms.value
