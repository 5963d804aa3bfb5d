/*
 * benchpush_amd.h -- C ABI of the MI355X-native batched ship-ice environment (libbenchpush_hip.so).
 *
 * The reference (IvanIZ/BenchPush) is pure Python and has no FFI for this path: the hot path is
 * ShipIceEnv.step()/reset() calling pymunk.Space.step 400x per step plus pure-python rasters
 * (benchpush/environments/ship_ice_nav/ship_ice_env.py:223-355,378-409).  This ABI is what a ctypes
 * binding inside the reference's ShipIceEnv would bind instead; each entry point
 * cites the reference code it replaces.
 *
 * Conventions: every function returns 0 on success or a negative BP_E* code, never throws; the caller
 * owns all buffers passed in (PyTorch-ROCm allocations passed as raw device pointers); all device work
 * is enqueued on the caller's HIP stream (hipStream_t passed as void*) and nothing synchronises
 * except bp_load_scenarios and the bp_get_* / bp_check_errors host copies; a handle is bound to one
 * device and is not thread-safe.  Physics arithmetic is IEEE binary64 ("f64").
 */
#ifndef BENCHPUSH_AMD_H
#define BENCHPUSH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BP_ABI_VERSION 11
#define BP_MAXV 20          /* max hull vertices per shape (generate_polygon draws 10-20, polygon.py:53,72) */
#define BP_MAX_SHIP_VERTS 32
#define BP_OBS_C 4
#define BP_INFO_COUNT 16

enum {
    BP_OK = 0,
    BP_EINVAL = -1,      /* bad argument / unsupported configuration */
    BP_ENOMEM = -2,
    BP_EHIP = -3,        /* HIP runtime error, see bp_last_error */
    BP_ENODEVICE = -4,   /* no usable GPU: the product path refuses to run without one */
    BP_ESTATE = -5,      /* call order violated (e.g. step before load/reset) */
    BP_ECAPACITY = -6    /* an in-kernel capacity (neighbour list / arbiter slots) overflowed */
};

enum { BP_ENV_SHIP_ICE = 0, BP_ENV_MAZE = 1, BP_ENV_BOX = 2 };
#define BP_MAX_WHEELS 4

/* per-env error bits reported by bp_check_errors */
enum { BP_ERR_ADJ_OVERFLOW = 1, BP_ERR_ARB_OVERFLOW = 2, BP_ERR_LEVEL_OVERFLOW = 4,
       BP_ERR_SCHED_TIMEOUT = 8 /* a scheduler fault left more envs unfinished than the completion launch can finish (256): the step is incomplete;
                                   smaller faults are finished by the completion launch and counted by bp_sched_warnings (ABI 7) */ };

/* info[e][k] columns written by bp_step / bp_reset (ship_ice_env.py:337-345 plus the callback
 * bookkeeping of :150-180); state is raw (the host adapter applies round(.,2)). */
enum {
    BP_I_X = 0, BP_I_Y, BP_I_THETA, BP_I_TOTAL_WORK, BP_I_WORK, BP_I_COLL_REWARD, BP_I_SCALED_COLL,
    BP_I_DIST_REWARD, BP_I_SUCCESS, BP_I_BOUNDARY, BP_I_YAW, BP_I_KE, BP_I_IMPULSE, BP_I_NPOST,
    BP_I_NCONTACT, BP_I_NFIRST
};

/* Scalar configuration (ship_ice_nav/config.yaml + constants of ship_ice_env.py, see
 * benchpush_amd/config.py:ship_ice_physics_params for the file:line of each). */
typedef struct bp_config {
    double dt;               /* env step (s) */
    int32_t steps;           /* physics sub-steps per env step */
    int32_t iterations;      /* solver iterations */
    int32_t persistence;     /* Chipmunk collision_persistence */
    int32_t settle_steps;    /* sub-steps run by reset */
    double damping_pow;      /* pow(space.damping, dt/steps), in [0, 1] (ship_ice_env.py:120, maze_NAMO_env.py:148: space.damping = cfg.sim.damping).
                              * 0 (every shipped config) runs the specialised kernels; any other value selects a generic instantiation of the same
                              * sub-step (k_physics_step_damp: velocities scaled instead of cleared, moving list rebuilt from the velocity slots,
                              * no step scheduler) -- slower, and bodies once pushed keep moving, so the in-kernel capacities (BP_ECAPACITY) are
                              * reached sooner.  The same holds for box-delivery / area-clearing handles (bp_bd_config.damping_pow, box_delivery_env.py:204: k_bd_settle_damp /
                              * k_bd_physics_damp). */
    double bias_coef;        /* 1 - pow(collision_bias, dt/steps) */
    double slop;
    double target_speed;
    double max_yaw_rate;
    double map_w, map_h;
    double goal_y;
    double m_to_pix;
    double density;
    double poly_radius;
    double elasticity;
    double friction;
    double beta;
    double boundary_penalty;
    double terminal_reward;
    double local_w, local_h;
    double vshift;
    double obs_range;
    int32_t num_ship_verts;
    int32_t _pad;
    double ship_verts[BP_MAX_SHIP_VERTS][2];  /* cfg.ship.vertices */
    double ship_head[2], ship_tail[2];
    /* ---- maze-NAMO-v0 (benchpush/environments/maze_NAMO/maze_NAMO_env.py, config.yaml); ignored for ship-ice ---- */
    int32_t env_kind;        /* BP_ENV_SHIP_ICE | BP_ENV_MAZE; for the maze ship_verts holds cfg.robot.vertices and map_w/map_h
                                hold cfg.env.width/length, goal_y cfg.env.goal_y */
    int32_t num_wheels;      /* cfg.robot.wheel_vertices (4 quads) */
    double wheel_verts[BP_MAX_WHEELS][4][2];
    double goal_x;           /* cfg.env.goal_x */
    double goal_reach;       /* cfg.goal_radius + cfg.robot.min_r, maze_NAMO_env.py:528-536 */
    double k_increment;      /* 150, maze_NAMO_env.py:82 */
    double wall_radius;      /* 0.5, sim_utils.py:177 */
    double obstacle_size;    /* cfg.obstacle_size */
    /* ---- ship-ice: per-episode start pose and metric constants ---- */
    int32_t random_start;    /* cfg.random_start (ship_ice_env.py:201-203): start = (1 + u * (start_x_range - 1), 1.0, pi/2) with u drawn per
                                (env, episode) from a counter RNG instead of python's global `random` (bp_start_uniform below); every reset
                                then re-runs the settle sub-steps, because the settled field depends on where the ship sits */
    int32_t _pad2;
    double start_x_range;    /* cfg.start_x_range */
    uint64_t start_seed;     /* key of the counter RNG */
    double ship_mass;        /* cfg.ship.mass: m0 of ShipIceMetric.compute_effort_score (ship_ice_metric.py:62-69) */
} bp_config;

/* The uniform in [0, 1) that bp_reset draws for (global env id, episode index) when random_start is set: the top 53 bits of
 * splitmix64(seed ^ splitmix64(global_env_id * 2^32 + episode)) / 2^53.  Exposed so that callers (and the parity tests) can
 * reproduce start poses on the host. */
double bp_start_uniform(uint64_t seed, int64_t global_env_id, int64_t episode);

typedef struct bp_handle bp_handle;

/* Create a handle for `num_envs` environments on HIP device `device`.  `env_id_offset` is the global id
 * of local env 0 (multi-GPU sharding: rank r owns [r*E/R, (r+1)*E/R)); trial selection uses the global id.
 * Replaces ShipIceEnv.__init__ (ship_ice_env.py:37-106). */
int bp_create(const bp_config *cfg, int32_t num_envs, int64_t env_id_offset, int32_t device, bp_handle **out);
int bp_destroy(bp_handle *h);

/* Upload `num_trials` ice fields (host pointers).  verts[T][F][V][2] raw world-space polygon vertices as stored in
 * the reference's trial dicts ('vertices'), counts[T][F] vertex counts (0 = unused slot), centres[T][F][2]
 * ('centre'), starts[T][3] ship start pose, nfloes[T].  Performs the body/shape construction of
 * sim_utils.py:136-163 + ship.py:77-98 (convex hull, COG recentre, mass/moment from density) on the host and
 * copies SoA arrays to the device.  Replaces the pickle load of ship_ice_env.py:76-80 and init_ship_ice_env :186-216. */
int bp_load_scenarios(bp_handle *h, int32_t num_trials, int32_t F, int32_t V, const double *verts,
                      const int32_t *counts, const double *centres, const double *starts, const int32_t *nfloes);

/* maze-NAMO-v0 counterpart of bp_load_scenarios: `num_layouts` box layouts (host pointers), centres[T][nbox][2], the wall
 * segments walls[nwalls][4] = ax, ay, bx, by (construct_maze_walls, maze_NAMO_env.py:357-375) and the start poses start[num_layouts][3]
 * (the fixed pose of :241-245, or one draw of cfg.random_start per layout, :229-238).  Builds the
 * KINEMATIC robot (body + wheels, robot.py:77-118), the boxes (sim_utils.py:136-163), the static Segment(radius 0.5) walls
 * (sim_utils.py:174-181) and the BFS goal map (occupancy_map.py:435-485) on the host.  Replaces init_maze_NAMO_env (:221-269). */
int bp_load_maze(bp_handle *h, int32_t num_layouts, int32_t nbox, const double *centres, int32_t nwalls, const double *walls,
                 const double *start);
/* maze: goal map as the reference returns it in info['goal_dt'] (un-normalised wavefront distances), host double [grid_h][grid_w] */
int bp_get_goal_map(bp_handle *h, double *out_host, int32_t *grid_h, int32_t *grid_w);

/* reset() for the envs whose env_mask byte is non-zero (device pointer; NULL = all envs): next trial
 * ((global_env_id + episode_idx) % num_trials), new space, 1000 settle sub-steps, first observation.
 * obs: device uint8 [E][4][H][W] (rows of unmasked envs untouched); info: device double [E][BP_INFO_COUNT] or NULL.
 * Replaces ShipIceEnv.reset (ship_ice_env.py:223-249). */
int bp_reset(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, double *info, void *stream);

/* One env.step() for every env: actions device double [E] in [-1,1]; obs device uint8 [E][4][H][W]; reward device
 * double [E]; terminated / truncated device uint8 [E]; info device double [E][BP_INFO_COUNT] (may be NULL).
 * Replaces ShipIceEnv.step (ship_ice_env.py:261-355): 400 x pymunk.Space.step, work, reward, raster. */
int bp_step(bp_handle *h, const double *actions, uint8_t *obs, double *reward, uint8_t *terminated,
            uint8_t *truncated, double *info, void *stream);

/* reset() copies a per-trial template that was settled once at load time (the reference's reset is a pure function of
 * the trial when cfg.random_start is off); on != 0 re-runs the settle sub-steps in place instead (identical result). */
int bp_set_resettle(bp_handle *h, int32_t on);

/* Physics only / raster only halves of bp_step (profiling and tests). */
int bp_step_physics(bp_handle *h, const double *actions, double *reward, uint8_t *terminated, uint8_t *truncated,
                    double *info, void *stream);
int bp_observe(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, void *stream);

/* Planner observation of ship-ice (cfg.egocentric_obs: false, ship_ice_env.py:96-99,394-406): device uint8
 * [E][2][map_h/0.2][map_w/0.2] = [5x5 block-mean occupancy of every floe, ship footprint on the 0.2 m grid]. */
int bp_observe_global(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, void *stream);

/* Planner cost map of ship-ice: CostMap.__init__ / boundary_cost / update / populate_costmap of
 * benchpush/common/cost_map.py:27-126,284-287 evaluated on the environments' current obstacles (what the planners pass as
 * info['obs'], lattice.py:78-79).  out: device double [E][(int)(m*scale)][(int)(n*scale)].  ship_pos_y: device double [E] in
 * cost-map units (the planner passes ship_y*scale - max_ship_length/2) or NULL for 0; horizon <= 0 means horizon=None. */
typedef struct bp_costmap_config {
    double scale;       /* cells per metre (lattice_config.yaml:43) */
    int32_t m, n;       /* channel height / width in metres */
    double alpha;       /* collision cost weight */
    double ship_mass;
    double horizon;     /* metres ahead of ship_pos_y that are considered; <= 0: everything */
    int32_t margin;     /* boundary columns set to 1e10 */
    int32_t pad_;
} bp_costmap_config;
int bp_costmap_update(bp_handle *h, const bp_costmap_config *cfg, const double *ship_pos_y, double vs, double *out, void *stream);

/* info['obs'] (cost_map.py:275-281): world-space hull vertices of every shape, device double [E][nb_cap][BP_MAXV][2],
 * counts device int32 [E][nb_cap] (index 0 = ship, then floes in trial order). */
int bp_get_world_polys(bp_handle *h, double *out, int32_t *counts, void *stream);
/* body state, device double [E][nb_cap][9] = x, y, angle, vx, vy, w, v_bias.x, v_bias.y, w_bias */
int bp_get_body_state(bp_handle *h, double *out, void *stream);
/* low-dimensional observation (ship_ice_env.py:358-370): |centroid| of every floe, device double [E][nb_cap-1][2] */
int bp_get_low_dim_obs(bp_handle *h, double *out, void *stream);

/* Per-episode metrics accumulated on the device (ship-ice and maze handles; the maze uses MazeNamoMetric's L = wavefront length at the
 * start pixel, maze_namo_metric.py:68-75, and bp_config.ship_mass = cfg.robot.mass): what ShipIceMetric.update / reset keep per episode
 * (benchpush/common/metrics/ship_ice_metric.py:26-69, base_metric.py:12-16) -- summed reward, path length of the ship integrated from
 * the state rounded to 2 decimals like info['state'] (ship_ice_env.py:337-339), total_work, success, steps -- updated by every bp_step /
 * bp_reset.  When an env terminates (or is reset while its episode is still running: eps_complete by truncation) its row is written:
 *   rows   device double [E][BP_EPM_COUNT] = efficiency, effort, episode reward, success, episode length (steps), total_work
 *          of the env's most recently finished episode,
 *   counts device uint32 [E] episodes finished so far.
 * [E][6] is the block that crosses GPUs (one all-gather per evaluation batch). */
#define BP_EPM_COUNT 6
enum { BP_EPM_EFFICIENCY = 0, BP_EPM_EFFORT, BP_EPM_REWARD, BP_EPM_SUCCESS, BP_EPM_LENGTH, BP_EPM_TOTAL_WORK };
int bp_get_episode_metrics(bp_handle *h, double *rows, uint32_t *counts, void *stream);
/* The episode LISTS that BaseMetric keeps (base_metric.py:12-16: one entry per finished episode), without a host round trip per step:
 *   ring   device double [E][BP_EPM_RING][BP_EPM_COUNT]: episode n of an env (n = 0, 1, ...) is in slot n % BP_EPM_RING, so the last BP_EPM_RING
 *          finished episodes of every env are available in order,
 *   sums   device double [E][BP_EPM_COUNT]: the six row fields summed over ALL finished episodes of the env (the means of an evaluation are
 *          sums / counts; [E][7] with the count is the block that crosses GPUs),
 *   counts device uint32 [E] episodes finished so far.  Any pointer may be NULL. */
#define BP_EPM_RING 8
int bp_get_episode_history(bp_handle *h, double *ring, double *sums, uint32_t *counts, void *stream);
/* Rollout plumbing for vectorised callers (the reference's learners wrap the env in SB3's VecEnv, whose auto-reset hands the LAST observation of a
 * finished episode back as infos[i]['terminal_observation'], baselines/ship_ice_nav/ppo/policy.py:29-69): copies row r of src to row r of dst for every r with
 * mask[r] != 0 -- device pointers, rows of row_bytes bytes (a multiple of 4), nothing is read back, so the caller can save the observations of the envs that are
 * about to be reset without a host synchronisation.  Cost is proportional to the rows selected.  ABI 9. */
int bp_copy_rows_masked(bp_handle *h, const uint8_t *mask, const void *src, void *dst, int64_t rows, int64_t row_bytes, void *stream);
/* test hook: out[i] = the device's restatement of python's round(in[i], 2) (device doubles [n]) */
int bp_debug_round2(const double *in_dev, double *out_dev, int32_t n, void *stream);
/* test hook: overwrite every cached-plane hint word of the live envs with random valid contents (indices below the recorded vertex counts, random
 * flags); the following steps must produce bit-identical results -- the hints only steer how much of the plane search is skipped.
 * Returns the number of words rewritten (>= 0) or a negative BP_E*; synchronises the stream. */
int bp_debug_scramble_hints(bp_handle *h, uint64_t seed, void *stream);

int32_t bp_nb_cap(const bp_handle *h);       /* body slots per env (ship + max floes, padded) */
int32_t bp_obs_height(const bp_handle *h);
int32_t bp_obs_width(const bp_handle *h);
/* number of bodies (ship + floes) currently in each env: host int32 [E] (synchronises) */
int bp_get_num_bodies(bp_handle *h, int32_t *out_host);
/* Copies per-env error bits to host int32 [E] (may be NULL) and returns BP_ECAPACITY if any is set (synchronises). */
int bp_check_errors(bp_handle *h, int32_t *out_host);
/* kernel timing helper for bench.py: average duration (ms) of the physics kernel over launches since the last call,
 * measured with HIP events on the launch stream (synchronises). */
int bp_kernel_time_ms(bp_handle *h, double *physics_ms, double *raster_ms, int32_t *launches);
int bp_enable_timing(bp_handle *h, int32_t on);
/* Per-launch cost statistics of the bp_step calls since bp_enable_timing(h, 1) (ship-ice and maze handles, at most 1024 launches): out_host[k][0] = sum
 * over the envs of the shader cycles >> 8 their wavefronts spent in launch k, out_host[k][1] = the largest of them.  With the clock of the run they give
 * the two lower bounds of a launch -- all work packed into the device's wave slots, and the busy time of the heaviest env (bench.py roofline.ceiling).
 * Host uint64 [max_launches][2]; *launches = rows written (synchronises).  ABI 11. */
int bp_get_cost_stats(bp_handle *h, uint64_t *out_host, int32_t max_launches, int32_t *launches);
/* shader cycles >> 8 each env's wavefront spent in the last bp_step (the dispatch-order hint): host uint32 [E] (synchronises) */
int bp_get_step_cycles(bp_handle *h, uint32_t *out_host);
/* sub-steps per chunk of the preemptive step scheduler (k_physics_step_sched: envs are parked at chunk boundaries while another one is further
 * behind, and resumed by another workgroup; results are identical), 0 = one wavefront per env for the whole step.  Default 40 for ship-ice and
 * maze handles of up to 8192 envs; environment variable BP_SCHED=<chunk> (0 = off) overrides it at load time. */
int32_t bp_sched_chunk(bp_handle *h);
/* Resident wavefronts of the step scheduler (k_physics_step_schedl: one workgroup per wave slot of the device takes task after task itself instead of one
 * workgroup per task from the hardware dispatcher; results are identical): the number of resident workgroups of a scheduled launch, 0 = one workgroup per
 * task.  Default: 8 x the device's compute units for scheduled launches (k_physics_step_schedr in pairing launches); environment variables BP_SCHED_PERSIST=0 /
 * BP_PAIR_RESIDENT=0 turn it off.  (ABI 10)  A resident kernel holds every wave slot for the whole launch, which suits one process per GPU only: unless one of
 * the two variables is set, the handle checks every 64 launches whether another PROCESS has a resident handle on the same device (advisory lock on
 * $BP_LOCK_DIR|/tmp/benchpush_amd.resident.<pci bus id>.lock) and launches the dispatcher-driven kernels while that is so; the value returned is that of the
 * launches being issued now (0 while the device is shared).  bp_device_shared: 1 while the last check saw another process.  (ABI 11) */
int32_t bp_sched_resident(bp_handle *h);
int32_t bp_device_shared(bp_handle *h);
/* The load-time launch policy as a pure function (no GPU, no handle): regime boundaries are rounds of the device's wave slots (8 per compute unit), not env
 * counts -- pairing from 2.5 rounds, loose pairing limits from 3.5, no scheduler without pairing above 4 (5 120 / 7 168 / 8 192 envs on the 256 CUs of an
 * MI355X).  out8_host = wave slots, pair mode (0 / 2), tight limits (0 / 1), envs that start alone, sub-steps per scheduler chunk (0 = no scheduler), and
 * the three leave limits (active arbiters, work units per sub-step, work rate).  (ABI 11) */
int bp_launch_policy_query(int32_t num_envs, int32_t num_compute_units, int32_t can_pair, int32_t is_maze, int32_t *out8_host);
/* Two environments per wavefront (ship-ice handles with space.damping == 0 and at most 448 body slots (PP_NBCAP); lanes 0..31 one env, lanes 32..63 another, results
 * identical): 0 = off, 1 = fixed pairs of the dispatch order for the whole step (test kernel), 2 = inside the step scheduler: the heaviest envs of the
 * dispatch order start alone, the others in pairs, and an env that outgrows the half-wave capacities or turns heavy is parked at a sub-step boundary and
 * resumed in a wavefront of its own.  Environment variable BP_PAIR=<mode> selects it at load time.  ABI 9. */
int32_t bp_pair_mode(bp_handle *h);
/* Parameters and cumulative counters of the pairing (host int32 [16], synchronises): [0..7] = mode, envs of the dispatch order that start alone, and the
 * limits above which an env leaves its pair (arbiter lanes, velocity slots, moving bodies, active arbiters, warm arbiters x colours, work proxy per
 * sub-step); [8..15] = since load: paired first tasks, paired tasks formed from the queues, envs that finished their step in a pair, envs that left a pair
 * as heavy and carried on alone in the same wave slot, heavy envs queued, light envs queued (mates of split pairs and yields), -, -.  ABI 9. */
int bp_get_pair_stats(bp_handle *h, int32_t *out16_host);
/* Clock calibration for bench.py: the shader-clock counter (s_memtime) and the 100 MHz reference counter (s_memrealtime) stamped on the device right
 * after every physics launch of bp_step / bp_reset (ship-ice and maze handles) by one thread, filed under the XCD it ran on (the shader-clock counters
 * of different XCDs are not synchronised): out[x][0..1] = the latest pair taken on XCD x, zeros if none yet.  The clock the chip held between two
 * calls is (d out[x][0]) / (d out[x][1]) x 100 MHz for any x stamped before the first and before the second call (MI355X_MICROARCH.md, DVFS item 6).
 * Host uint64 [8][2] (synchronises).  ABI 8 (ABI 7: one pair, whatever the XCD). */
int bp_get_clock_stamps(bp_handle *h, uint64_t *out16_host);
/* Scheduler robustness counters since bp_load_*: out2_host[0] = launches in which the watchdog of the step scheduler fired (a workgroup gave up
 * waiting for a parked env -- a scheduler fault, never seen in practice), out2_host[1] = envs whose step the completion launch that follows every
 * scheduled launch had to finish.  Results are complete and identical either way; non-zero values are a warning, not an error (synchronises). */
int bp_sched_warnings(bp_handle *h, int32_t *out2_host);
/* overrides that hint for the next bp_step: host uint32 [E], larger = dispatched earlier (results never depend on the order; synchronises) */
int bp_set_step_cost_hint(bp_handle *h, const uint32_t *host_costs);

/* ---------------------------------------------------------------------------------------------------------------
 * box-delivery-v0 (benchpush/environments/box_delivery/box_delivery_env.py, config.yaml).  A handle made by bp_bd_create is
 * (task 0) is driven by the same bp_reset / bp_step / bp_observe / bp_get_body_state / bp_check_errors / bp_destroy entry points:
 *   actions  device double [E]: 'heading' action in [-1, 1] (box_delivery_env.py:706-723) or 'position' index into the local map;
 *            device double [E][2] = (linear, angular) speed for 'velocity' (:672-703), selected by bp_bd_config.action_type
 *   obs      device uint8 [E][local_px][local_px][4], channels last (box_delivery_env.py:1045-1059)
 *   info     device double [E][BP_INFO_COUNT]: BP_BD_I_* columns
 * Replaces BoxDeliveryEnv.step (:634-830: PositionController waypoints, execute_robot_path, step_simulation_until_still,
 * spfa-based box distances, rewards, receptacle removal) and generate_observation (:1045-1207).
 * --------------------------------------------------------------------------------------------------------------- */
enum {
    BP_BD_I_X = 0, BP_BD_I_Y, BP_BD_I_THETA, BP_BD_I_CUM_DIST, BP_BD_I_CUM_BOXES, BP_BD_I_CUM_REWARD, BP_BD_I_TOTAL_WORK,
    BP_BD_I_MINISTEPS, BP_BD_I_INACTIVITY, BP_BD_I_HIT, BP_BD_I_SUBSTEPS, BP_BD_I_ROBOT_DIST, BP_BD_I_BOXES_DIST, BP_BD_I_NWP,
    BP_BD_I_NALIVE, BP_BD_I_WORK
};
typedef struct bp_bd_config {
    double ctrl_dt;                    /* controller.dt (config.yaml:43); one sim step is ctrl_dt / steps */
    int32_t steps, iterations, persistence, settle_steps;   /* sim.steps, sim.iterations, Chipmunk persistence 3, 1000 (:284) */
    double damping_pow, bias_coef, slop;                     /* as in bp_config */
    double room_length, room_width;    /* env.room_length, env.room_width_small|large */
    double recept_x, recept_y, recept_size;                  /* get_receptacle_position_and_size (:322-324) */
    double ppm;                        /* local_map_pixel_width / local_map_width */
    double local_w;                    /* env.local_map_width */
    int32_t local_px;                  /* env.local_map_pixel_width */
    int32_t use_correct_direction_reward;
    double robot_radius, robot_half_width;                   /* :122-123 */
    double step_size, target_speed;    /* agent.step_size, controller.target_speed */
    double partial_rewards_scale, goal_reward, collision_penalty, non_movement_penalty, correct_direction_reward_scale;
    double ministep_size, sp_channel_scale;
    int32_t inactivity_cutoff, invert_receptacle_map, num_boxes, step_limit;
    int32_t action_type;               /* agent.action_type: 0 heading, 1 position, 2 velocity */
    int32_t task;                      /* 0 box-delivery-v0; 1 area-clearing-v0 (environments/area_clearing/area_clearing.py:611-778) */
    double omega_scale, v_scale, lfc;  /* apply_controller factors (3, 2 | 0.5, 5) and the DP look-ahead controller.Lfc (0 | 0.5) */
    /* area-clearing only: velocity action scale, sim.t_max, the reward constants of area_clearing.py:37-49, DISTANCE_SCALE_MAX */
    double yaw_rate_step;
    int32_t t_max, num_goal_points, num_boundary_verts, num_outer_verts;
    double boundary_penalty, box_cleared_reward, box_putback_penalty, truncation_penalty, terminal_reward, pushing_mult, distance_scale_max;
    double boundary[8][2], outer_boundary[8][2];   /* env_cfg.boundary / outer_boundary (convex polygons) */
    double footprint_verts[4][2];                  /* agent.footprint_vertices */
    double goal_points[128][2];                    /* _compute_boundary_goals (:225-264) */
    double box_half, box_density;      /* boxes.box_size / 2, boxes.box_density */
    double robot_verts[4][2];          /* agent.vertices */
    double wheel_verts[4][4][2];       /* agent.wheel_vertices */
    double bumper_verts[4][2];         /* agent.front_bumper_vertices */
} bp_bd_config;
/* area-clearing-v0 uses the same entry points with bp_bd_config.task = 1: heading/position/velocity actions, obs uint8
 * [E][224][224][4], info columns = x, y, theta, total_work, collision reward, diff_reward, box_completed_reward, box_count, ministeps,
 * robot_hit_obstacle, sim steps, robot distance, t, #waypoints, work, pushing reward.  bp_bd_load then takes no receptacle polygon;
 * static polygons (walls radius 0.1) get pymunk's default elasticity 0 and friction 0.99 (area_clearing.py:446-474).
 * Replaces AreaClearingEnv.step / reset / generate_observation (:563-778,927-1120). */
int bp_bd_create(const bp_bd_config *cfg, int32_t num_envs, int64_t env_id_offset, int32_t device, bp_handle **out);
/* `num_trials` episodes (host pointers): starts[T][3] robot start pose, boxes[T][nbox][3] = x, y, heading, and `nstatic` static
 * polygons per trial in generate_sim_bounds order (sim_utils.py:90-135): sverts[T][ns][4][2], scount[T][ns] (3 or 4 vertices),
 * spose[T][ns][3] body x, y, angle, srad[T][ns] shape radius, stype[T][ns] collision type (3 obstacle, 4 receptacle; exactly one
 * receptacle; scount 0 = unused padding slot when trials hold different numbers of columns).  Builds bodies/shapes (sim_utils.py:20-160), the configuration space, nearest-free-cell indices and the
 * receptacle distance map (box_delivery_env.py:1115-1175) on the host, settles every trial once.
 * Replaces init_box_delivery_sim / init_box_delivery_env (:193-292). */
int bp_bd_load(bp_handle *h, int32_t num_trials, int32_t nbox, const double *starts, const double *boxes, int32_t nstatic,
               const double *sverts, const int32_t *scount, const double *spose, const double *srad, const int32_t *stype);
int32_t bp_bd_sizeof_config(void);
/* tests: static rasters of trial t over the small-map window, host buffers (any may be NULL): dims[6] = H, W, SH, SW, si0, sj0;
 * cspace / cspace_thin uint8 [SH][SW]; edt uint16 [SH][SW][2]; recept float [SH][SW]; small_free uint8 [SH][SW] */
int bp_bd_get_maps(bp_handle *h, int32_t trial, int32_t *dims, uint8_t *cspace, uint8_t *cspace_thin, uint16_t *edt, float *recept,
                   uint8_t *small_free);
/* Two-pass step of box-delivery / area-clearing handles (ABI 9): out2_host[0] = env steps whose sim-step loop ran past the budget of the first pass and was
 * finished by the second one (beside the finish / map / observation kernels of all other envs), out2_host[1] = env steps whose execute_robot_path or
 * step_simulation_until_still loop ran into STEP_LIMIT (box_delivery_env.py:62,891-1023) -- cumulative since load (host uint32 [2], synchronises).
 * BP_BD_BUDGET=<sim steps> sets the budget of the first pass at load time (default 3000; 0 = one pass); bp_bd_budget returns the value the handle uses (ABI 11). */
int bp_bd_get_stragglers(bp_handle *h, uint32_t *out2_host);
int32_t bp_bd_budget(bp_handle *h);
/* Exact recurrences of execute_robot_path (box_delivery_env.py:891-988; ABI 11): while only the robot moves and no arbiter exists, a sim step is a function of the
 * robot's parts and the controller's loop variables; when that state is bit for bit the one of p sim steps ago (a robot that pushes against a wall until
 * STEP_LIMIT sits at a fixed point, p = 1, within a few hundred sim steps) whole periods are skipped -- results, sim-step counts and info columns are
 * those of the full loop.  out2_host[0] = recurrences found, out2_host[1] = sim steps they skipped, cumulative since load (host uint32 [2], synchronises).
 * The test runs in the second pass of the two-pass step (k_bd_physics_resume: the envs that ran out of the first pass's sim-step budget, BP_BD_BUDGET -- the kernel every
 * env runs stays lean); BP_BD_CYCLE=<n> at load time: only once a path has run n sim steps, 0: every sim step is run. */
int bp_bd_get_cycle_skips(bp_handle *h, uint32_t *out2_host);
/* tests: per-env box bookkeeping, host buffers: alive uint8 [E][24], waypoints double [E][64][3], nwp int32 [E] (synchronises) */
int bp_bd_get_state(bp_handle *h, uint8_t *alive, double *waypoints, int32_t *nwp);

const char *bp_last_error(const bp_handle *h);
int32_t bp_abi_version(void);
int32_t bp_sizeof_config(void);   /* sizeof(bp_config), so a binding can verify its struct layout */

#ifdef __cplusplus
}
#endif
#endif
