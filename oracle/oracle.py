"""ctypes wrapper around oracle/bp_oracle.c (the CPU restatement of the ship-ice env.step() path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (benchpush_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbp_oracle.so")
MAXV = 24
OBS_SHAPE = (4, 150, 150)
INFO_KEYS = ["x", "y", "theta", "total_work", "work", "collision_reward", "scaled_collision_reward", "dist_reward",
             "trial_success", "boundary_violated", "yaw_violated", "total_ke", "total_impulse", "n_post_solve",
             "n_contact_pts", "n_first_contact"]


class OrcParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("steps", C.c_int), ("iterations", C.c_int), ("persistence", C.c_int),
                ("settle_steps", C.c_int), ("brute_force", C.c_int), ("damping_pow", C.c_double),
                ("bias_coef", C.c_double), ("slop", C.c_double), ("target_speed", C.c_double),
                ("max_yaw_rate", C.c_double), ("map_w", C.c_double), ("map_h", C.c_double), ("goal_y", C.c_double),
                ("m_to_pix", C.c_double), ("density", C.c_double), ("poly_radius", C.c_double),
                ("elasticity", C.c_double), ("friction", C.c_double), ("beta", C.c_double),
                ("boundary_penalty", C.c_double), ("terminal_reward", C.c_double), ("local_w", C.c_double),
                ("local_h", C.c_double), ("vshift", C.c_double), ("obs_range", C.c_double),
                ("goal_x", C.c_double), ("goal_reach", C.c_double), ("k_increment", C.c_double), ("wall_radius", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("bp_oracle.c", "bp_oracle_bd.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.restype = C.c_int
        L.orc_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.orc_observe.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_num_shapes.restype = C.c_int
        L.orc_num_shapes.argtypes = [C.c_void_p]
        for name in ("orc_get_bodies", "orc_get_mass", "orc_get_stats", "orc_get_info"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_world_polys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_local_polys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_poly_area.restype = C.c_double
        L.orc_poly_area.argtypes = [C.c_int, C.c_void_p]
        L.orc_poly_centroid.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_convex_hull.restype = C.c_int
        L.orc_convex_hull.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_collide.restype = C.c_int
        L.orc_collide.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_double, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_total_work.restype = C.c_double
        L.orc_total_work.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_crop.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_maze_reset.restype = C.c_int
        L.orc_maze_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_int, C.c_void_p]
        L.orc_maze_step.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.orc_maze_observe.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_shape_states.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_maze_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_maze_set_dist_map.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_draw_polygon.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double]
        L.orc_cv_line.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, C.c_double]
        L.orc_nd_rotate.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_observe_global.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.orc_costmap.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_bench.restype = C.c_double
        L.orc_bench.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_long)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleShipIce:
    """Single-env oracle with the reference's reset()/step() shape (ship_ice_env.py:223-355)."""

    def __init__(self, params, ship_vertices, head, tail, brute_force=False):
        self.L = lib()
        p = OrcParams()
        for k, v in params.items():
            setattr(p, k, v)
        p.brute_force = int(brute_force)
        self.params = dict(params)
        self.h = self.L.orc_create(C.byref(p))
        self.ship_vertices = np.ascontiguousarray(ship_vertices, np.float64)
        self.head = np.ascontiguousarray(head, np.float64)
        self.tail = np.ascontiguousarray(tail, np.float64)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_solve_order(self, mode, seed=0):
        """Arbiter sweep order of the solver: 0 = (colour, key) [what the GPU path and every parity test use], 1 = ascending key,
        2 = order in which the broadphase sweep met the pairs, 3 = seeded random permutation per sub-step, 4 = descending key."""
        self.L.orc_set_solve_order.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        self.L.orc_set_solve_order(self.h, int(mode), int(seed))

    def reset(self, trial, start=None, observe=True):
        obs_list = trial["obstacles"]
        counts = np.array([len(o["vertices"]) for o in obs_list], np.int32)
        verts = np.ascontiguousarray(np.concatenate([np.asarray(o["vertices"], np.float64) for o in obs_list])
                                     if len(obs_list) else np.zeros((0, 2)), np.float64)
        centres = np.ascontiguousarray([o["centre"] for o in obs_list], np.float64).reshape(-1, 2)
        st = np.ascontiguousarray(trial["ship_state"] if start is None else start, np.float64)
        self.nf = self.L.orc_reset(self.h, len(obs_list), _p(verts), _p(counts), _p(centres),
                                   len(self.ship_vertices), _p(self.ship_vertices), _p(self.head), _p(self.tail), _p(st))
        obs = None
        if observe:
            obs = np.zeros(OBS_SHAPE, np.uint8)
            self.L.orc_observe(self.h, _p(obs))
        return obs, self.info()

    def step(self, action, observe=True):
        obs = np.zeros(OBS_SHAPE, np.uint8) if observe else None
        r = C.c_double()
        t = C.c_int()
        info = np.zeros(len(INFO_KEYS), np.float64)
        self.L.orc_step(self.h, float(action), _p(obs) if observe else None, C.byref(r), C.byref(t), _p(info))
        return obs, r.value, bool(t.value), dict(zip(INFO_KEYS, info.tolist()))

    def observe_global(self, grid_m=0.2):
        """Planner observation (egocentric_obs: false): uint8 [2, map_h/grid, map_w/grid]."""
        shape = (2, int(self.params["map_h"] / grid_m), int(self.params["map_w"] / grid_m))
        obs = np.zeros(shape, np.uint8)
        self.L.orc_observe_global(self.h, float(grid_m), _p(obs))
        return obs

    def costmap(self, scale, m, n, alpha=10.0, ship_mass=1.0, horizon=None, margin=1, ship_pos_y=0.0, vs=1.0):
        """CostMap(scale, m, n, alpha, ship_mass, horizon, margin).update(info['obs'], ship_pos_y, vs).cost_map (common/cost_map.py)."""
        out = np.zeros((int(m * scale), int(n * scale)), np.float64)
        self.L.orc_costmap(self.h, float(scale), int(m), int(n), float(alpha), float(ship_mass), float(horizon or 0.0), int(margin),
                           float(ship_pos_y), float(vs), _p(out))
        return out

    def info(self):
        info = np.zeros(len(INFO_KEYS), np.float64)
        self.L.orc_get_info(self.h, _p(info))
        return dict(zip(INFO_KEYS, info.tolist()))

    def bodies(self):
        n = self.L.orc_num_shapes(self.h)
        out = np.zeros((n, 9), np.float64)
        self.L.orc_get_bodies(self.h, _p(out))
        return out

    def mass(self):
        n = self.L.orc_num_shapes(self.h)
        out = np.zeros((n, 5), np.float64)
        self.L.orc_get_mass(self.h, _p(out))
        return out

    def world_polys(self):
        n = self.L.orc_num_shapes(self.h)
        out = np.zeros((n, MAXV, 2), np.float64)
        cnt = np.zeros(n, np.int32)
        self.L.orc_get_world_polys(self.h, _p(out), _p(cnt))
        return out, cnt

    def local_polys(self):
        n = self.L.orc_num_shapes(self.h)
        v = np.zeros((n, MAXV, 2), np.float64)
        nn = np.zeros((n, MAXV, 2), np.float64)
        self.L.orc_get_local_polys(self.h, _p(v), _p(nn))
        return v, nn

    def stats(self):
        out = np.zeros(8, np.int64)
        self.L.orc_get_stats(self.h, _p(out))
        return dict(zip(["substeps", "pairs_bb", "narrow", "arb_sum", "arb_max", "moving_sum", "hot_sum", "ncol_max"], out.tolist()))


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def poly_area(v):
    v = np.ascontiguousarray(v, np.float64)
    return lib().orc_poly_area(len(v), _p(v))


def poly_centroid(v):
    v = np.ascontiguousarray(v, np.float64)
    out = np.zeros(2)
    lib().orc_poly_centroid(len(v), _p(v), _p(out))
    return out


def convex_hull(v):
    v = np.ascontiguousarray(v, np.float64)
    out = np.zeros((len(v), 2))
    n = lib().orc_convex_hull(len(v), _p(v), _p(out))
    return out[:n]


def collide(a, ra, b, rb):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    n = np.zeros(2)
    p1 = np.zeros((2, 2))
    p2 = np.zeros((2, 2))
    h = np.zeros(2, np.uint32)
    c = lib().orc_collide(len(a), _p(a), ra, len(b), _p(b), rb, _p(n), _p(p1), _p(p2), _p(h))
    return c, n, p1[:c], p2[:c], h[:c]


def total_work(polys_a, polys_b):
    counts = np.array([len(p) for p in polys_a], np.int32)
    a = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64) for p in polys_a]))
    b = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64) for p in polys_b]))
    return lib().orc_total_work(len(counts), _p(counts), _p(a), _p(b))


def crop(params, state, gmap, oob_val=0.0):
    p = OrcParams()
    for k, v in params.items():
        setattr(p, k, v)
    st = np.ascontiguousarray(state, np.float64)
    g = np.ascontiguousarray(gmap, np.float64)
    out = np.zeros((int(params["local_h"] * params["m_to_pix"]), int(params["local_w"] * params["m_to_pix"])), np.uint8)
    lib().orc_crop(C.byref(p), _p(st), _p(g), float(oob_val), _p(out))
    return out


def bench(params, ship_vertices, head, tail, packed, nenv, nsteps, nthreads, with_obs=True):
    """Time nenv oracle envs x nsteps env.step() on nthreads cores (OpenMP). Returns (env_steps, seconds)."""
    p = OrcParams()
    for k, v in params.items():
        setattr(p, k, v)
    sv = np.ascontiguousarray(ship_vertices, np.float64)
    hd = np.ascontiguousarray(head, np.float64)
    tl = np.ascontiguousarray(tail, np.float64)
    T, F, V = packed["verts"].shape[:3]
    n = C.c_long()
    sec = lib().orc_bench(C.byref(p), T, F, V, _p(packed["verts"]), _p(packed["counts"]), _p(packed["centres"]),
                          _p(packed["starts"]), _p(packed["nfloes"]), len(sv), _p(sv), _p(hd), _p(tl), int(nenv), int(nsteps),
                          int(nthreads), int(with_obs), C.byref(n))
    return n.value, sec


MAZE_INFO_KEYS = ["x", "y", "theta", "total_work", "work", "collision_reward", "scaled_collision_reward", "dist_increment_reward",
                  "trial_success", "boundary_violated", "wall_collision", "total_ke", "total_impulse", "n_post_solve",
                  "n_contact_pts", "n_first_contact"]


class OracleMaze:
    """Single-env oracle of maze-NAMO-v0 (maze_NAMO_env.py:325-485). `layout` = dict(centres [n,2], walls [w,4], start (x,y,th))."""

    def __init__(self, params, robot_vertices, wheel_vertices, obstacle_size, brute_force=False):
        self.L = lib()
        p = OrcParams()
        for k, v in params.items():
            setattr(p, k, v)
        p.brute_force = int(brute_force)
        self.params = dict(params)
        self.h = self.L.orc_create(C.byref(p))
        self.rv = np.ascontiguousarray(robot_vertices, np.float64)
        self.wv = np.ascontiguousarray(wheel_vertices, np.float64).reshape(-1, 4, 2)
        self.size = float(obstacle_size)
        self.obs_shape = (4, int(params["local_h"] * params["m_to_pix"]), int(params["local_w"] * params["m_to_pix"]))

    def __del__(self):
        try:
            if self.h:
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self, layout, observe=True):
        c = np.ascontiguousarray(layout["centres"], np.float64).reshape(-1, 2)
        w = np.ascontiguousarray(layout["walls"], np.float64).reshape(-1, 4)
        st = np.ascontiguousarray(layout["start"], np.float64)
        self.ns = self.L.orc_maze_reset(self.h, len(c), _p(c), self.size, _p(self.rv), len(self.rv), _p(self.wv), len(self.wv),
                                        _p(w), len(w), _p(st))
        obs = None
        if observe:
            obs = np.zeros(self.obs_shape, np.uint8)
            self.L.orc_maze_observe(self.h, _p(obs))
        return obs

    def step(self, action, observe=True):
        obs = np.zeros(self.obs_shape, np.uint8) if observe else None
        r, t = C.c_double(), C.c_int()
        info = np.zeros(len(MAZE_INFO_KEYS), np.float64)
        self.L.orc_maze_step(self.h, float(action), _p(obs) if observe else None, C.byref(r), C.byref(t), _p(info))
        return obs, r.value, bool(t.value), dict(zip(MAZE_INFO_KEYS, info.tolist()))

    def set_dist_map(self, m):
        self.L.orc_maze_set_dist_map(self.h, _p(np.ascontiguousarray(m, np.float64)))

    def shape_states(self):
        out = np.zeros((self.ns, 9), np.float64)
        self.L.orc_get_shape_states(self.h, _p(out))
        return out

    def maps(self):
        H, W = int(self.params["map_h"] * self.params["m_to_pix"]), int(self.params["map_w"] * self.params["m_to_pix"])
        a, b, c = np.zeros((H, W)), np.zeros((H, W)), np.zeros((H, W))
        self.L.orc_maze_maps(self.h, _p(a), _p(b), _p(c))
        return a, b, c

    def world_polys(self):
        out = np.zeros((self.ns, MAXV, 2), np.float64)
        cnt = np.zeros(self.ns, np.int32)
        self.L.orc_get_world_polys(self.h, _p(out), _p(cnt))
        return out, cnt


def nd_rotate(img, c, s, cval):
    img = np.ascontiguousarray(img, np.float64)
    out = np.zeros_like(img)
    lib().orc_nd_rotate(_p(img), img.shape[0], float(c), float(s), float(cval), _p(out))
    return out


def draw_polygon(r, c, shape=None):
    """Restated skimage.draw.polygon: (rr, cc) of the pixels inside the polygon with vertex rows r / columns c."""
    r = np.ascontiguousarray(r, np.float64); c = np.ascontiguousarray(c, np.float64)
    if shape is None:
        shape = (max(int(np.ceil(r.max())) + 2, 1), max(int(np.ceil(c.max())) + 2, 1))
    img = np.zeros((int(shape[0]), int(shape[1])), np.float64)
    lib().orc_draw_polygon(len(r), _p(r), _p(c), img.shape[0], img.shape[1], _p(img), 1.0)
    rr, cc = np.nonzero(img)
    return rr, cc


def cv_line(img, pt1, pt2, color):
    """Restated cv2.line (thickness 1, 8-connected) on a float64 image, in place."""
    assert img.dtype == np.float64 and img.flags["C_CONTIGUOUS"]
    lib().orc_cv_line(_p(img), img.shape[0], img.shape[1], int(pt1[0]), int(pt1[1]), int(pt2[0]), int(pt2[1]), float(color))
    return img
