"""ctypes binding of libbenchpush_hip.so (C ABI: include/benchpush_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable, loading / bp_create raise.
"""
import ctypes as C
import os

from .build import LIB_PATH

MAXV = 20
MAX_SHIP_VERTS = 32
MAX_WHEELS = 4
ENV_SHIP_ICE, ENV_MAZE, ENV_BOX = 0, 1, 2
BD_MAXBOX, BD_MAXWP = 24, 64
BD_INFO_KEYS = ["x", "y", "theta", "cumulative_distance", "cumulative_boxes", "cumulative_reward", "total_work", "ministeps", "inactivity",
                "robot_hit_obstacle", "substeps", "robot_distance", "boxes_distance", "num_waypoints", "num_boxes_left", "work"]
INFO_COUNT = 16
EPM_COUNT, EPM_RING = 6, 8
INFO_KEYS = ["x", "y", "theta", "total_work", "work", "collision_reward", "scaled_collision_reward", "dist_reward",
             "trial_success", "boundary_violated", "yaw_violated", "total_ke", "total_impulse", "n_post_solve",
             "n_contact_pts", "n_first_contact"]
ERRORS = {0: "BP_OK", -1: "BP_EINVAL", -2: "BP_ENOMEM", -3: "BP_EHIP", -4: "BP_ENODEVICE", -5: "BP_ESTATE", -6: "BP_ECAPACITY"}
EXPORTS = ["bp_abi_version", "bp_create", "bp_destroy", "bp_load_scenarios", "bp_load_maze", "bp_get_goal_map", "bp_reset", "bp_step", "bp_step_physics",
           "bp_observe", "bp_observe_global", "bp_set_resettle", "bp_sizeof_config", "bp_get_world_polys", "bp_get_body_state", "bp_get_low_dim_obs", "bp_costmap_update", "bp_nb_cap", "bp_obs_height",
           "bp_obs_width", "bp_get_num_bodies", "bp_check_errors", "bp_kernel_time_ms", "bp_enable_timing", "bp_get_step_cycles", "bp_set_step_cost_hint", "bp_sched_chunk", "bp_sched_resident", "bp_sched_warnings", "bp_get_clock_stamps", "bp_last_error",
           "bp_bd_create", "bp_bd_load", "bp_bd_sizeof_config", "bp_bd_get_maps", "bp_bd_get_state",
           "bp_get_episode_metrics", "bp_get_episode_history", "bp_start_uniform", "bp_debug_round2", "bp_debug_scramble_hints",
           "bp_copy_rows_masked", "bp_pair_mode", "bp_get_pair_stats", "bp_bd_get_stragglers",
           "bp_device_shared", "bp_launch_policy_query", "bp_bd_budget", "bp_get_cost_stats", "bp_bd_get_cycle_skips"]


class BpCostmapConfig(C.Structure):
    _fields_ = [("scale", C.c_double), ("m", C.c_int32), ("n", C.c_int32), ("alpha", C.c_double), ("ship_mass", C.c_double),
                ("horizon", C.c_double), ("margin", C.c_int32), ("pad_", C.c_int32)]


class BpConfig(C.Structure):
    _fields_ = [("dt", C.c_double), ("steps", C.c_int32), ("iterations", C.c_int32), ("persistence", C.c_int32),
                ("settle_steps", C.c_int32), ("damping_pow", C.c_double), ("bias_coef", C.c_double), ("slop", C.c_double),
                ("target_speed", C.c_double), ("max_yaw_rate", C.c_double), ("map_w", C.c_double), ("map_h", C.c_double),
                ("goal_y", C.c_double), ("m_to_pix", C.c_double), ("density", C.c_double), ("poly_radius", C.c_double),
                ("elasticity", C.c_double), ("friction", C.c_double), ("beta", C.c_double),
                ("boundary_penalty", C.c_double), ("terminal_reward", C.c_double), ("local_w", C.c_double),
                ("local_h", C.c_double), ("vshift", C.c_double), ("obs_range", C.c_double),
                ("num_ship_verts", C.c_int32), ("_pad", C.c_int32),
                ("ship_verts", (C.c_double * 2) * MAX_SHIP_VERTS), ("ship_head", C.c_double * 2),
                ("ship_tail", C.c_double * 2),
                ("env_kind", C.c_int32), ("num_wheels", C.c_int32), ("wheel_verts", ((C.c_double * 2) * 4) * MAX_WHEELS),
                ("goal_x", C.c_double), ("goal_reach", C.c_double), ("k_increment", C.c_double), ("wall_radius", C.c_double),
                ("obstacle_size", C.c_double),
                ("random_start", C.c_int32), ("_pad2", C.c_int32), ("start_x_range", C.c_double), ("start_seed", C.c_uint64),
                ("ship_mass", C.c_double)]


class BpBdConfig(C.Structure):
    _fields_ = [("ctrl_dt", C.c_double), ("steps", C.c_int32), ("iterations", C.c_int32), ("persistence", C.c_int32), ("settle_steps", C.c_int32),
                ("damping_pow", C.c_double), ("bias_coef", C.c_double), ("slop", C.c_double),
                ("room_length", C.c_double), ("room_width", C.c_double), ("recept_x", C.c_double), ("recept_y", C.c_double),
                ("recept_size", C.c_double), ("ppm", C.c_double), ("local_w", C.c_double), ("local_px", C.c_int32),
                ("use_correct_direction_reward", C.c_int32), ("robot_radius", C.c_double), ("robot_half_width", C.c_double),
                ("step_size", C.c_double), ("target_speed", C.c_double), ("partial_rewards_scale", C.c_double), ("goal_reward", C.c_double),
                ("collision_penalty", C.c_double), ("non_movement_penalty", C.c_double), ("correct_direction_reward_scale", C.c_double),
                ("ministep_size", C.c_double), ("sp_channel_scale", C.c_double), ("inactivity_cutoff", C.c_int32),
                ("invert_receptacle_map", C.c_int32), ("num_boxes", C.c_int32), ("step_limit", C.c_int32),
                ("action_type", C.c_int32), ("task", C.c_int32), ("omega_scale", C.c_double), ("v_scale", C.c_double), ("lfc", C.c_double),
                ("yaw_rate_step", C.c_double), ("t_max", C.c_int32), ("num_goal_points", C.c_int32), ("num_boundary_verts", C.c_int32),
                ("num_outer_verts", C.c_int32), ("boundary_penalty", C.c_double), ("box_cleared_reward", C.c_double),
                ("box_putback_penalty", C.c_double), ("truncation_penalty", C.c_double), ("terminal_reward", C.c_double),
                ("pushing_mult", C.c_double), ("distance_scale_max", C.c_double), ("boundary", (C.c_double * 2) * 8),
                ("outer_boundary", (C.c_double * 2) * 8), ("footprint_verts", (C.c_double * 2) * 4), ("goal_points", (C.c_double * 2) * 128),
                ("box_half", C.c_double), ("box_density", C.c_double), ("robot_verts", (C.c_double * 2) * 4),
                ("wheel_verts", ((C.c_double * 2) * 4) * 4), ("bumper_verts", (C.c_double * 2) * 4)]


class BpError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raises if it has not been built (python -m benchpush_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BpError("libbenchpush_hip.so not found at %s -- build it with `python -m benchpush_amd.build` "
                      "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    try:  # torch ships its own HIP runtime: load it first so that this library binds to the same copy (device memory and
        import torch  # noqa: F401   streams are shared with torch; two runtimes in one process do not see each other's devices)
    except ImportError:
        pass
    path = LIB_PATH
    if os.environ.get("BP_PROF") == "1":  # diagnostic build with in-kernel phase timers (tools/prof_phases.py)
        path = os.environ.get("BP_PROF_LIB", LIB_PATH.replace(".so", "_prof.so"))
    L = C.CDLL(path)
    vp = C.c_void_p
    L.bp_abi_version.restype = C.c_int32
    L.bp_sizeof_config.restype = C.c_int32
    if L.bp_sizeof_config() != C.sizeof(BpConfig):
        raise BpError("bp_config layout mismatch between _lib.BpConfig and the library")
    L.bp_create.argtypes = [C.POINTER(BpConfig), C.c_int32, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.bp_destroy.argtypes = [vp]
    L.bp_load_scenarios.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
    L.bp_load_maze.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, vp, vp]
    L.bp_get_goal_map.argtypes = [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.bp_reset.argtypes = [vp, vp, vp, vp, vp]
    L.bp_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.bp_step_physics.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.bp_observe.argtypes = [vp, vp, vp, vp]
    L.bp_observe_global.argtypes = [vp, vp, vp, vp]
    L.bp_get_world_polys.argtypes = [vp, vp, vp, vp]
    L.bp_get_body_state.argtypes = [vp, vp, vp]
    L.bp_get_low_dim_obs.argtypes = [vp, vp, vp]
    L.bp_get_episode_metrics.argtypes = [vp, vp, vp, vp]
    if hasattr(L, "bp_get_episode_history"):
        L.bp_get_episode_history.argtypes = [vp, vp, vp, vp, vp]
    L.bp_start_uniform.argtypes = [C.c_uint64, C.c_int64, C.c_int64]
    L.bp_start_uniform.restype = C.c_double
    L.bp_debug_round2.argtypes = [vp, vp, C.c_int32, vp]
    L.bp_copy_rows_masked.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, vp]
    if hasattr(L, "bp_debug_scramble_hints"):
        L.bp_debug_scramble_hints.argtypes = [vp, C.c_uint64, vp]
    L.bp_costmap_update.argtypes = [vp, C.POINTER(BpCostmapConfig), vp, C.c_double, vp, vp]
    for n in ("bp_nb_cap", "bp_obs_height", "bp_obs_width"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_int32
    L.bp_get_num_bodies.argtypes = [vp, vp]
    L.bp_check_errors.argtypes = [vp, vp]
    L.bp_kernel_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.bp_enable_timing.argtypes = [vp, C.c_int32]
    L.bp_get_step_cycles.argtypes = [vp, vp]
    L.bp_set_step_cost_hint.argtypes = [vp, vp]
    L.bp_sched_chunk.argtypes = [vp]
    L.bp_sched_resident.argtypes = [vp]
    L.bp_pair_mode.argtypes = [vp]
    L.bp_get_pair_stats.argtypes = [vp, vp]
    L.bp_bd_get_stragglers.argtypes = [vp, vp]
    L.bp_sched_chunk.restype = C.c_int32
    L.bp_sched_resident.restype = C.c_int32
    if hasattr(L, "bp_device_shared"):    # ABI 11 (tools/ab_libs.sh loads older builds for same-box comparisons)
        L.bp_device_shared.argtypes = [vp]
        L.bp_device_shared.restype = C.c_int32
        L.bp_bd_budget.argtypes = [vp]
        L.bp_bd_budget.restype = C.c_int32
        L.bp_launch_policy_query.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
        L.bp_get_cost_stats.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    if hasattr(L, "bp_bd_get_cycle_skips"):
        L.bp_bd_get_cycle_skips.argtypes = [vp, vp]
    if hasattr(L, "bp_get_clock_stamps"):
        L.bp_get_clock_stamps.argtypes = [vp, vp]
    if hasattr(L, "bp_sched_warnings"):   # absent from libraries of ABI 6 (tools/ab_bench.sh loads older builds for same-box comparisons)
        L.bp_sched_warnings.argtypes = [vp, vp]
    L.bp_last_error.argtypes = [vp]
    L.bp_last_error.restype = C.c_char_p
    L.bp_debug_trace.argtypes = [vp, vp, C.c_int32]
    L.bp_set_resettle.argtypes = [vp, C.c_int32]
    L.bp_debug_prof.argtypes = [vp, vp]
    L.bp_bd_sizeof_config.restype = C.c_int32
    if L.bp_bd_sizeof_config() != C.sizeof(BpBdConfig):
        raise BpError("bp_bd_config layout mismatch between _lib.BpBdConfig and the library")
    L.bp_bd_create.argtypes = [C.POINTER(BpBdConfig), C.c_int32, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.bp_bd_load.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    L.bp_bd_get_maps.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.bp_bd_get_state.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def check(L, h, rc, what):
    if rc != 0:
        msg = L.bp_last_error(h).decode() if h else ""
        raise BpError("%s failed: %s (%d) %s" % (what, ERRORS.get(rc, "?"), rc, msg))


def make_config(params, ship_vertices, head, tail, env_kind=ENV_SHIP_ICE, wheel_vertices=None, obstacle_size=0.0):
    cfg = BpConfig()
    fields = {f[0] for f in BpConfig._fields_}
    for k, v in params.items():
        if k in fields:
            setattr(cfg, k, v)
    cfg.env_kind = env_kind
    cfg.obstacle_size = float(obstacle_size)
    wheels = [] if wheel_vertices is None else wheel_vertices
    if len(wheels) > MAX_WHEELS:
        raise ValueError("too many wheels")
    cfg.num_wheels = len(wheels)
    for w, quad in enumerate(wheels):
        if len(quad) != 4:
            raise ValueError("wheel outlines must have 4 vertices")
        for i, (x, y) in enumerate(quad):
            cfg.wheel_verts[w][i][0] = float(x)
            cfg.wheel_verts[w][i][1] = float(y)
    sv = [list(map(float, v)) for v in ship_vertices]
    if len(sv) > MAX_SHIP_VERTS:
        raise ValueError("too many ship vertices")
    cfg.num_ship_verts = len(sv)
    for i, (x, y) in enumerate(sv):
        cfg.ship_verts[i][0] = x
        cfg.ship_verts[i][1] = y
    cfg.ship_head[0], cfg.ship_head[1] = float(head[0]), float(head[1])
    cfg.ship_tail[0], cfg.ship_tail[1] = float(tail[0]), float(tail[1])
    return cfg


def make_bd_config(phys, bd, cfg):
    """bp_bd_config from box_delivery_scenario.box_delivery_physics_params / box_delivery_params and the merged cfg."""
    c = BpBdConfig()
    fields = {f[0] for f in BpBdConfig._fields_}
    for src in (phys, bd):
        for k, v in src.items():
            if k in fields:
                setattr(c, k, v)
    c.ctrl_dt = float(phys["dt"])
    if bd.get("task", 0) == 1:      # area-clearing: boxes are squares of half-size obstacle_size, density sim.obstacle_density
        c.box_half = float(cfg.obstacle_size)
        c.box_density = float(cfg.sim.obstacle_density)
    else:
        c.box_half = float(cfg.boxes.box_size) / 2
        c.box_density = float(cfg.boxes.box_density)
    for i, (x, y) in enumerate(cfg.agent.vertices):
        c.robot_verts[i][0], c.robot_verts[i][1] = float(x), float(y)
    for w, quad in enumerate(cfg.agent.wheel_vertices):
        for i, (x, y) in enumerate(quad):
            c.wheel_verts[w][i][0], c.wheel_verts[w][i][1] = float(x), float(y)
    for i, (x, y) in enumerate(cfg.agent.front_bumper_vertices):
        c.bumper_verts[i][0], c.bumper_verts[i][1] = float(x), float(y)
    return c


def fill_area_geometry(c, boundary, outer_boundary, footprint, goal_points):
    """area-clearing geometry of bp_bd_config (convex boundary polygons with at most 8 vertices, at most 128 goal points)."""
    if len(boundary) > 8 or len(outer_boundary) > 8 or len(goal_points) > 128 or len(footprint) != 4:
        raise ValueError("area-clearing geometry exceeds the ABI capacities")
    c.num_boundary_verts, c.num_outer_verts, c.num_goal_points = len(boundary), len(outer_boundary), len(goal_points)
    for i, (x, y) in enumerate(boundary):
        c.boundary[i][0], c.boundary[i][1] = float(x), float(y)
    for i, (x, y) in enumerate(outer_boundary):
        c.outer_boundary[i][0], c.outer_boundary[i][1] = float(x), float(y)
    for i, (x, y) in enumerate(footprint):
        c.footprint_verts[i][0], c.footprint_verts[i][1] = float(x), float(y)
    for i, (x, y) in enumerate(goal_points):
        c.goal_points[i][0], c.goal_points[i][1] = float(x), float(y)
    return c
