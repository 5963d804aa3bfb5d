// Device-side data layout and small math helpers shared by the kernels of libbenchpush_hip.so.
// gfx950 only: 64-wide wavefronts are assumed everywhere (one wavefront == one environment).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/benchpush_amd.h"

#define BP_KADJ 24   // Verlet neighbour slots per body
#define BP_ACAP 64   // arbiter (contact pair) slots per env: one per lane

typedef double2 d2;

enum { BODY_DYNAMIC = 0, BODY_KINEMATIC = 1, BODY_STATIC = 2 };
__device__ __forceinline__ int kind_ctype(int k) { return k & 0xFF; }
__device__ __forceinline__ int kind_group(int k) { return (k >> 8) & 0xFF; }
__device__ __forceinline__ int kind_btype(int k) { return (k >> 16) & 3; }
enum { ARB_FIRST = 0, ARB_NORMAL = 1, ARB_IGNORE = 2, ARB_CACHED = 3 };
#define ARB_FREE_KEY 0xFFFFFFFFu

struct DevParams {
    double dt_sub;
    int steps, iterations, persistence, settle_steps;
    double damping_pow, bias_coef, slop, target_speed, max_yaw_rate, map_w, map_h, goal_y, m_to_pix;
    double beta, boundary_penalty, terminal_reward, local_w, local_h, vshift, obs_range;
    double goal_x, goal_reach, k_increment;   // maze-NAMO-v0
    int env_kind;                              // BP_ENV_SHIP_ICE / BP_ENV_MAZE
    int nkin;                                  // bodies [0, nkin) are the parts of the kinematic agent (ship: 1, maze robot: 5)
    double skin;
    int nbcap, mvcap, num_envs, num_trials, num_ship_verts;
    long long env_offset;
    double ship_verts[BP_MAX_SHIP_VERTS][2];
    double ship_head[2], ship_tail[2];
    int obs_h, obs_w, grid_h, grid_w;
    int sq_chunk, sq_levels, sq_cap;           // scheduler: sub-steps per chunk, chunks per step, queue capacity per (XCD, level)
    int sq_cls;                                // issue-priority classes of the dispatch order: 0 = quarter / quarter / half -> 3 / 1 / 0, 1 = 1/16, 3/16, 1/4, 1/2 -> 3 / 2 / 1 / 0
    unsigned sq_ymask;                         // chunk boundaries (bit k: after k chunks) at which a running env may yield; all ones = every boundary
    int sq_dynprio;                            // > 0: issue priority re-set at every chunk boundary from the env's projected step length (per cent of *sq_thr for priority 3)
    int sq_parts, sq_part;                     // the scheduled launch as sq_parts kernels on as many streams; this kernel's index
    int sq_floor;                              // longest-remaining-first: lower bound of the estimated cost per sub-step left (wave cycles >> 8)
    int sq_lrpt, sq_bw, sq_hyst;               // longest-remaining-first scheduling: on / row width in wave cycles >> 8 / rows a waiting env must be ahead by
    int sq_hold;                               // 1: envs of the top priority class do not yield to envs that have not started yet (BP_SCHED_HOLD)
    int sq_mode;                               // 0 = scheduled launch, 1 = completion launch: workgroup b finishes env b if the scheduled launch left it unfinished
    int sq_debug;                              // test hook (BP_SCHED_DEBUG_DROP=1): env 1 is parked after its first chunk and never queued, the watchdog is short
    int dbg_paths;                             // test hook (BP_DEBUG_PATHS bit mask): 1 no candidate cache, 2 bound rounds through the sequential (flushing) loop,
                                               // 4 cached planes always through the support query, 8 manifold support vertices always through the support query
    // two environments per wavefront (bp_physics_pair.hpp): 0 off, 1 fixed pairs (2b, 2b + 1) for the whole step (k_physics_step_pair: parity tests),
    // 2 inside the preemptive scheduler (the pair_solo heaviest envs of the dispatch order start alone, the others in pairs)
    int pair_mode, pair_solo;
    int pp_max_keys, pp_max_slots, pp_max_mv, pp_max_act, pp_max_work;   // when a half leaves its pair (pair_should_leave)
    int pp_snake, pp_heavy_only;               // pairing order heaviest-with-lightest; waves yield only to waiting HEAVY envs
    int pp_rate;                               // an env whose mean work proxy per sub-step (S.costp) is above this is parked as heavy: it carries on alone
    int random_start;                          // ship-ice: per-episode start x from the counter RNG (ship_ice_env.py:201-203)
    double start_x_range, ship_mass;
    unsigned long long start_seed;
};

// All device arrays of one handle.  "sc_" = scenario (per trial, read-only), others = per-env state.
struct DevPtrs {
    // scenarios [T][nbcap]...
    const int *sc_nb;        // [T] bodies in trial (ship + kept floes)
    const int *sc_nv;        // [T][nbcap] hull vertex count (0 = unused)
    const d2 *sc_lv;         // [T][nbcap][MAXV] local hull vertices
    const d2 *sc_ln;         // [T][nbcap][MAXV] local plane normals
    const double4 *sc_mass;  // [T][nbcap] m_inv, i_inv, cog.x, cog.y
    const double4 *sc_pose;  // [T][nbcap] x, y, angle, -
    const double4 *sc_prop;  // [T][nbcap] shape radius, elasticity, friction, -
    const int *sc_kind;      // [T][nbcap] collision_type | group << 8 | body_type << 16 (group != 0: parts of one body)
    const double *dist_map;  // maze: normalised BFS goal map [grid_h][grid_w]
    const unsigned char *wall_map; // maze: wall raster [grid_h][grid_w]
    const double *goal_raw;  // maze: un-normalised wavefront map (info['goal_dt']) [grid_h][grid_w]
    const double *maze_obs_map; // maze: dist_map with the wall raster in the sign bit (wall cells: -1.0), one load per cell for k_observe_maze
    // env state
    int *e_trial, *e_episode, *e_nb, *e_err;
    int *e_flags;            // [E] maze: bit0 wall_collision (sticky), bit1 prev_dist valid
    double *e_prevdist;      // [E] maze: previous goal-map value
    unsigned *e_stamp;
    double *e_currdt, *e_total_work, *e_ke, *e_imp;
    unsigned *e_cnt;         // [E][4] n_post_solve, n_contact_pts, n_first_contact, -
    unsigned *e_cost;        // [E] wave cycles (>>8) the env's last step took: dispatch-order hint only
    const int *order;        // [E] env handled by workgroup b (heaviest first), or null = identity
    // preemptive step scheduler (k_physics_step_sched): per XCD and level (= chunks of the step already done) a queue of waiting envs
    int *sq_items;           // [16][SQ_MAXLEV][sq_cap] env ids, -1 = not yet written (rows 0..7: XCD x, envs that run alone; 8..15: XCD x - 8, envs that were light when parked)
    int *sq_ctr;             // [16][SQ_MAXLEV + 2][2]: (head, tail) per level; row SQ_MAXLEV = (finished, total) of the XCD
    unsigned *sq_carry;      // [E][4] step-local state across chunks: yaw_violated, boundary_violated, work proxy, wave cycles >> 8
    unsigned char *sq_moved; // [E][nbcap] shape moved in an earlier chunk of this step
    int *sq_done;            // [E] the env's step is complete (cleared by k_sched_init)
    int *sq_lev;             // [E] chunks completed when the env was last parked
    unsigned *sq_thr;        // [1] cost (256-cycle units) of the previous step's env at rank E / 256 from the top: k_make_order; 0xFFFFFFFF before the first step
    int *sq_sub;             // [E] sub-steps of the current step completed when the env was last parked (0: not parked in this step); cleared by k_sched_init
    int *sq_pairstat;        // [8] cumulative: paired first tasks, paired tasks taken from the queues, halves that finished their step in a pair, halves that left as
                             //     heavy (carried on alone in the same slot), heavy halves queued, light halves queued (split mates + yields), tasks declined at load, -
    int *sq_rescue;          // [1 + SQ_RESCUE] count and ids of the envs the scheduled launch left unfinished (k_sched_scan)
    int *sq_warn;            // [2] scheduler watchdog events, envs finished by the completion launch (cumulative; bp_sched_warnings)
    d2 *pxy;                 // [E][nbcap] position of COG
    double *ang;             // [E][nbcap]
    d2 *rot;                 // [E][nbcap] cos, sin
    d2 *velv, *velw, *velb;  // [E][nbcap] (vx,vy) (w,w_bias) (vbx,vby)
    d2 *wv, *wn, *pv;        // [E][nbcap][MAXV] world verts / normals / previous-step world verts
    double4 *bb, *fat;       // [E][nbcap] l,b,r,t
    unsigned short *adj;     // [E][nbcap][KADJ]
    unsigned char *adjn;     // [E][nbcap]
    unsigned long long *hint; // [E][nbcap][KADJ] cached plane-search winners of the pair behind each neighbour slot (HW_* below)
    // persisted arbiter slots [E][ACAP]
    unsigned *a_key, *a_stamp, *a_sc, *a_h0, *a_h1;
    double *a_d;             // [E][ACAP][14] jn0 jt0 jn1 jt1 nx ny r1x0 r1y0 r2x0 r2y0 r1x1 r1y1 r2x1 r2y1
    // result of the last step per env (read by k_episode_metrics) and the on-device episode metrics
    double *e_lastrew;       // [E]
    int *e_lastflag;         // [E] bit0 terminated, bit1 trial_success
    double *m_acc;           // [E][8] episode reward, path length l0, previous rounded x, y, L, steps, total_work, success
    double *m_rows;          // [E][BP_EPM_COUNT] most recently finished episode
    double *m_ring;          // [E][BP_EPM_RING][BP_EPM_COUNT] the last BP_EPM_RING finished episodes (episode n in slot n % BP_EPM_RING)
    double *m_sum;           // [E][BP_EPM_COUNT] row fields summed over all finished episodes
    unsigned *m_count;       // [E] finished episodes
    unsigned char *m_open;   // [E] an episode is running (reset seen, not yet terminated)
    unsigned long long *clk; // [8][2] per XCD: shader-clock counter (s_memtime) and 100 MHz reference (s_memrealtime) stamped after a physics launch by a thread of that XCD
    // debug
    double *dbg;             // optional [substeps][nbcap][3] pose trace of env dbg_env
    int dbg_env;
    unsigned long long *prof; // optional [E][24] phase cycle counters (BP_PROF builds)
};

__device__ __forceinline__ d2 mk2(double x, double y) { d2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ double vdot(d2 a, d2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double vcross(d2 a, d2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ d2 vsub(d2 a, d2 b) { return mk2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ d2 vadd(d2 a, d2 b) { return mk2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ d2 vneg(d2 a) { return mk2(-a.x, -a.y); }
__device__ __forceinline__ d2 vmul(d2 a, double s) { return mk2(a.x * s, a.y * s); }
__device__ __forceinline__ d2 vperp(d2 a) { return mk2(-a.y, a.x); }
__device__ __forceinline__ d2 vrotate(d2 a, d2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double vlen(d2 a) { return __builtin_sqrt(vdot(a, a)); }
__device__ __forceinline__ d2 vlerp(d2 a, d2 b, double t) { return vadd(vmul(a, 1.0 - t), vmul(b, t)); }
__device__ __forceinline__ double clamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }
__device__ __forceinline__ double fclampd(double f, double lo, double hi) { return fmin(fmax(f, lo), hi); }

// Hint word of a neighbour-list entry (64 bits): the planes that won the two sides of the pair's plane search when it was last run, their support
// vertices on the other shape, and the two vertex counts (for the cyclic neighbours of a support vertex).  A cached plane is re-evaluated exactly:
// its separation is a lower bound of the pair's maximum, so above the radii the pair is rejected at once; every other plane is first bounded from
// above with a few vertices and searched only if the bound reaches the cached value.  0 = nothing cached.
#define HW_PLANE_A(h) ((int)((h) & 31ull))
#define HW_PLANE_B(h) ((int)(((h) >> 5) & 31ull))
#define HW_VERT_A(h) ((int)(((h) >> 10) & 31ull))   // support vertex (on B) of the cached plane of A
#define HW_VERT_B(h) ((int)(((h) >> 15) & 31ull))   // support vertex (on A) of the cached plane of B
#define HW_NV_A(h) ((int)(((h) >> 20) & 31ull))
#define HW_NV_B(h) ((int)(((h) >> 25) & 31ull))
#define HW_HAS_A (1ull << 32)
#define HW_HAS_B (1ull << 33)
#define HW_PRIM_B (1ull << 34)   // the side whose plane had the larger separation (the one tested first) is B
#define HW_BOTH (1ull << 35)     // the pair got past the cached-plane test last time: both cached planes are evaluated up front
// A minimum of d . v over a convex polygon's vertices found at vertex j is exact as soon as both cyclic neighbours of j exceed it by this margin:
// the true sequence is unimodal and the computed dot products are within ~1e-13 of the true ones (coordinates below 64 m), so every other vertex
// is then larger too and j is the unique first minimum.
#define BP_SUPPORT_MARGIN 1e-10
// narrow phase, parallel facing edges: how much higher above the winning plane the stand-in support vertex may lie (the CPU checker uses the same constant)
#define BP_TIE_TOL 1e-9
#ifndef BP_QCAP
#define BP_QCAP 96          // support queries per batch (LDS)
#endif

#define BP_CC_IDX_BITS 14   // candidate cache word (LdsCtx::cc): bits per body index; bp_load_* refuse nb_cap >= 1 << BP_CC_IDX_BITS
#define BP_EVCAP 32         // box-delivery: pre_solve events per sub-step
#define BP_MBOX 16          // manifold mailbox entries per hand-over batch
#ifndef BP_NSLOT
#define BP_NSLOT 96         // velocity slots per env (bodies with a non-zero velocity or an arbiter)
#endif
#define BP_PROFN 64         // diagnostic build: phase timers / trip counters per env
#define BP_SNAP_ROWS 9      // box-delivery recurrence snapshot: one row per kinematic part of the robot (at most 8) + one for the controller
#define BP_SNAP_COLS 12

// LDS map of the physics kernels (one wavefront = one env), byte offsets from the start of dynamic LDS.  Used by the kernels (carve_lds) and by the
// host (launch size), so the two cannot drift apart.
struct LdsMap {
    unsigned sv, sw, sb, sp, ag, tf, q_dir, q_c, r_val, q_meta, q_aux, r_idx, pt_a, pt_thr, cc, cc_hw, res_smA, res_smB, res_iA, res_iB, res_jA, res_jB, mvs, owner, colmask, mvo, sbody, mv,
        slot_of, rf, ev_d, ev_key, ctl, snap, prof, total;
};
__host__ __device__ inline LdsMap bp_lds_map(const int nbcap, const int mvcap, const bool box, const bool prof)
{
    LdsMap m;
    unsigned p = 0;
    m.sv = p; p += 16u * (BP_NSLOT + 1);    // + 1: scratch slot BP_NSLOT
    m.sw = p; p += 16u * (BP_NSLOT + 1);
    m.sb = p; p += 16u * (BP_NSLOT + 1);
    m.sp = p; p += 16u * (BP_NSLOT + 1);    // position of the body that holds the slot (LDS copy of pxy)
    m.ag = p; p += 32u;                     // agent (body 0): (angle, -), (cos, sin)
    // narrow-phase scratch; the transforms of the integrate phase ([64][2] d2 = 2 KB <= q_dir + q_c) and, later, the manifold mailbox
    // (BP_MBOX x 96 B) reuse the query buffers from q_dir on
    m.tf = p;
    m.q_dir = p; p += 16u * BP_QCAP;
    m.q_c = p; p += 8u * BP_QCAP;
    m.r_val = p; p += 8u * BP_QCAP;
    m.q_meta = p; p += 4u * BP_QCAP;
    m.q_aux = p; p += 4u * BP_QCAP;
    m.r_idx = p; p += 4u * BP_QCAP;
    m.pt_a = p; p += 16u * 64;
    m.pt_thr = p; p += 16u * 64;
    m.cc = p; p += 8u * 64;                   // candidate cache of the first candidate round (persists across sub-steps)
    m.cc_hw = p; p += 8u * 64;
    m.res_smA = p; p += 8u * 64;              // res_smA .. res_jB are contiguous: the AABB keys of the transform phase ([64][4] u64) alias them
    m.res_smB = p; p += 8u * 64;
    m.res_iA = p; p += 4u * 64;
    m.res_iB = p; p += 4u * 64;
    m.res_jA = p; p += 4u * 64;
    m.res_jB = p; p += 4u * 64;
    m.mvs = p; p += 4u * (unsigned)nbcap;
    m.owner = p; p += 2u * (BP_NSLOT + 2);    // per velocity slot
    m.colmask = p; p += 2u * (BP_NSLOT + 2);
    m.mvo = p; p += 4u * (BP_NSLOT + 2);      // per velocity slot: stamp of the sub-step whose moving list the body has joined
    m.sbody = p; p += 2u * (BP_NSLOT + 2);    // per velocity slot: the body that holds it (read by the damping != 0 instantiation only)
    m.mv = p; p += 2u * (unsigned)mvcap;
    m.slot_of = p; p += (unsigned)nbcap;
    m.rf = p; p += 64u;
    p = (p + 15u) & ~15u;
    m.ev_d = p; if (box) p += 16u * 3 * BP_EVCAP;
    m.ev_key = p; if (box) p += 4u * BP_EVCAP;
    p = (p + 7u) & ~7u;
    m.ctl = p; if (box) p += 8u * 4;           // box-delivery: wave-uniform doubles of the path controller, kept out of the VGPR file across the sim step
    m.snap = p; if (box) p += 8u * BP_SNAP_ROWS * BP_SNAP_COLS;   // box-delivery: snapshot of the robot's state for the recurrence test of execute_robot_path
    m.prof = p; if (prof) p += 8u * BP_PROFN;
    m.total = (p + 15u) & ~15u;
    return m;
}

#define SQ_MAXLEV 16
#define BP_DBL_MIN 2.2250738585072014e-308
#define BP_PI 3.14159265358979323846
#define BP_INF (__builtin_inf())

// Deterministic sin/cos: Cody-Waite reduction by pi/2 + fdlibm kernel polynomials.  Same operation order as the CPU
// oracle so that results are bit-identical (stands in for libm's sin/cos inside Chipmunk's cpvforangle).
__device__ __forceinline__ void bp_sincos(double x, double &sn, double &cs)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = __builtin_rint(x * invpio2);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y0 = r - w;
    // second reduction stage (arguments within 2^-17 of a multiple of pi/2: cancellation): rare, so the wave branches around it as a whole instead of
    // computing it for every call and selecting (the per-lane result is the same either way)
    const bool deep = __builtin_fabs(y0) < __builtin_fabs(x) * 7.62939453125e-06;
    if (__builtin_expect(__ballot(deep) != 0ull, 0)) {
        const double t = r;
        const double w2 = fn * pio2_2;
        const double r2 = t - w2;
        const double w3 = fn * pio2_2t - ((t - r2) - w2);
        if (deep) { r = r2; w = w3; y0 = r2 - w3; }
    }
    double y1 = (r - y0) - w;
    int q = (int)((long long)fn & 3);
    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double kc = 1.0 - (0.5 * z - (z * rc - y0 * y1));
    // quadrant: q = 0 (ks, kc), 1 (kc, -ks), 2 (-ks, -kc), 3 (-kc, ks) -- two selects and two sign-bit flips (a negation is exactly a flip of the sign bit)
    const bool odd = (q & 1) != 0;
    const double s0 = odd ? kc : ks, c0 = odd ? ks : kc;
    const int sflip = (q & 2) ? (int)0x80000000 : 0, cflip = ((q + 1) & 2) ? (int)0x80000000 : 0;
    sn = __hiloint2double(__double2hiint(s0) ^ sflip, __double2loint(s0));
    cs = __hiloint2double(__double2hiint(c0) ^ cflip, __double2loint(c0));
}

// counter RNG of the per-episode start pose (include/benchpush_amd.h: bp_start_uniform)
__host__ __device__ inline unsigned long long bp_splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double bp_start_u01(unsigned long long seed, long long gid, long long episode)
{
    const unsigned long long k = bp_splitmix64(seed ^ bp_splitmix64(((unsigned long long)gid << 32) + (unsigned long long)episode));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

// wave-wide helpers (64 lanes) ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc_below(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// ---- cross-lane reductions on DPP (row = 16 lanes) ------------------------------------------------------------------
// min / max are exact and order-independent, so any reduction tree gives bit-identical results.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// all 32 lanes of each half-wave receive the max / min of their half
__device__ __forceinline__ double half_max(double v)
{
    v = fmax(v, dpp_mov_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov_f64<0x141>(v));  // row_half_mirror
    v = fmax(v, dpp_mov_f64<0x140>(v));  // row_mirror
    v = fmax(v, __shfl_xor(v, 16));
    return v;
}
__device__ __forceinline__ double half_min(double v)
{
    v = fmin(v, dpp_mov_f64<0xB1>(v));
    v = fmin(v, dpp_mov_f64<0x4E>(v));
    v = fmin(v, dpp_mov_f64<0x141>(v));
    v = fmin(v, dpp_mov_f64<0x140>(v));
    v = fmin(v, __shfl_xor(v, 16));
    return v;
}
// maxima of the two half waves (lanes 0..31 -> lo, 32..63 -> hi), wave-uniform; max is exact, so the reduction order is irrelevant
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void halves_max_f64(double v, double &lo, double &hi)
{
    v = fmax(v, dpp_mov_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_mov_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_mov_f64<0x141>(v));  // row_half_mirror
    v = fmax(v, dpp_mov_f64<0x140>(v));  // row_mirror: every lane holds the maximum of its row of 16
    lo = fmax(readlane_f64(v, 0), readlane_f64(v, 16));
    hi = fmax(readlane_f64(v, 32), readlane_f64(v, 48));
}
// 8-lane groups (lanes 8g .. 8g+7): every lane receives the group's minimum
__device__ __forceinline__ double oct_min_f64(double v)
{
    double p;
    p = dpp_mov_f64<0xB1>(v); v = (p < v) ? p : v;    // quad_perm [1,0,3,2]
    p = dpp_mov_f64<0x4E>(v); v = (p < v) ? p : v;    // quad_perm [2,3,0,1]
    p = dpp_mov_f64<0x141>(v); v = (p < v) ? p : v;   // row_half_mirror: lane i <-> 7 - i of each half row
    return v;
}
__device__ __forceinline__ int oct_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    return v;
}
// order-preserving bijection binary64 -> u64 (for LDS integer atomics); -0 must be folded into +0 by the caller
__device__ __forceinline__ unsigned long long f64_key(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __builtin_bit_cast(double, b);
}
// LDS-only ordering point for a single-wave workgroup: the LDS operations of one wave execute in issue order, so a lane's ds_read
// issued after another lane's ds_write / atomic sees it without any wait; only the compiler must be kept from moving accesses
// across the point (it still inserts the s_waitcnt each register use needs).  Cross-lane communication through *global* memory
// needs __syncthreads() (vmcnt drain) instead.
__device__ __forceinline__ void lds_sync() { asm volatile("" ::: "memory"); }
