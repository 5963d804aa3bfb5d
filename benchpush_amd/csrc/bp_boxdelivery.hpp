// box-delivery-v0: BoxDeliveryEnv.step (box_delivery_env.py:634-830) as gfx950 kernels.
//   k_bd_plan     heading action -> spatial action -> waypoints        position_controller.py:56-181, box_delivery_env.py:706-726
//   k_bd_physics  execute_robot_path + step_simulation_until_still     box_delivery_env.py:891-1023 (substep<BP_ENV_BOX> per sim step)
//   k_bd_finish   box distances (spfa per box), rewards, removal, work box_delivery_env.py:736-823
//   k_bd_observe  uint8 [224][224][4] observation                       box_delivery_env.py:1045-1207
// One 64-lane wavefront per environment for plan / physics / finish; scalar control code is executed redundantly by all lanes.
// spfa.spfa (third-party, source absent) is restated as its least fixed point: exact Dijkstra over unit-width distance buckets
// (every edge is >= 1, so a bucket is final when it is reached), float32 relaxations through integer atomicMin, parents by the
// local rule "first neighbour, in spfa's direction order, whose distance + edge equals mine".
#pragma once
#include "bp_kernels.hpp"

#define BD_MAXWP 64
#define BD_MAXBOX 24
#define BD_QCAP 3072
#define BD_PATHCAP 1024
#define BD_INF_BITS 0x7F800000u

struct BdParams {
    double room_length, room_width, recept_x, recept_y, recept_size, ppm, local_w, robot_radius, step_size, target_speed, ctrl_dt;
    double partial_rewards_scale, goal_reward, collision_penalty, non_movement_penalty, correct_direction_reward_scale, ministep_size;
    double sp_channel_scale;
    int local_px, use_correct_direction_reward, inactivity_cutoff, num_boxes, step_limit;
    int H, W, SH, SW, si0, sj0;     // padded room, small-map window and its origin in the padded room
    int nbox, first_box;            // box slots [first_box, first_box + nbox)
    int nrecept;                    // receptacle polygons (not physics slots)
    int action_type;                // 0 heading, 1 position (index into the local map), 2 velocity (actions [E][2])
    // task switch: 0 box-delivery-v0, 1 area-clearing-v0 (benchpush/environments/area_clearing/area_clearing.py)
    int task;
    double omega_scale, v_scale, lfc;   // apply_controller factors (box-delivery 3, 2; area-clearing 0.5, 5) and DP look-ahead Lfc
    double yaw_rate_step;               // area-clearing velocity actions
    int t_max, ngoal, nbd, nob;
    double boundary_penalty, box_cleared_reward, box_putback_penalty, truncation_penalty, terminal_reward, pushing_mult;
    double bd_poly[8][2], ob_poly[8][2], footprint[4][2];
    float recept_outside;               // channel-3 map outside the small-map window: box-delivery 0; area-clearing 1, and 2 within
    int out_r;                          // out_r pixels of the window (dilated outer walls: the whole window border is wall, checked at load)
    // Two-pass step: pass 0 runs every env for at most `budget` sim steps (0 = no limit), pass 1 resumes the envs that were not done.
    // The envs that finish in pass 0 -- all but the handful that run into the reference's 10 001-step loops -- go through the finish / robot-map / observe
    // kernels while pass 1 is still running on another stream; `sel_want` tells a tail kernel which of the two groups it serves (-1: every env).
    int budget, pass, sel_want;
    int resume_sel;                     // second pass: 0 = every unfinished env, 1 = those that stopped inside execute_robot_path, 2 = the others (until-still loop)
    int cycle_skip;                     // > 0: the second pass of the two-pass step looks for exact recurrences of the robot's state in execute_robot_path once a path has run this many sim steps, and skips whole periods (bd_physics_body<DAMP, CYC>; BP_BD_CYCLE=0 turns it off)
};
struct BdPtrs {
    // per map (trial -> map index): window rasters
    const int *map_of_trial;        // [T]
    const unsigned *free_bits;      // [M][SH*SW/32] configuration_space
    const unsigned *thin_bits;      // [M][SH*SW/32] configuration_space_thin
    const unsigned short *edt;      // [M][SH*SW][2] closest_cspace_indices (window coordinates)
    const float *recept;            // [M][SH*SW] create_global_shortest_path_to_receptacle_map (window)
    const unsigned char *small_free;// [M][SH*SW] small_obstacle_map (1 free, 0 wall)
    const d2 *recept_poly;          // [M][4] receptacle world vertices (hull order) + planes
    const d2 *recept_n;             // [M][4]
    const unsigned char *robot_chan;// [local_px*local_px] robot_state_channel * 255
    // per env
    unsigned char *alive;           // [E][BD_MAXBOX]
    unsigned char *order;           // [E][BD_MAXBOX] self.boxes list order
    int *nalive, *nprev;            // [E]
    double *boxdist;                // [E][BD_MAXBOX]
    d2 *boxpos;                     // [E][BD_MAXBOX] box position the cached distance was computed for
    d2 *prev;                       // [E][BD_MAXBOX][4] prev_boxes by list position
    double *cum;                    // [E][4] cumulative_distance, cumulative_reward, -, -
    int *cnt;                       // [E][4] inactivity, cumulative_boxes, -, -
    double *wp;                     // [E][BD_MAXWP][3]
    int *nwp;                       // [E]
    double *stepf;                  // [E][8] robot_distance, ix, iy, ih, hit, substeps, -, -
    unsigned char *unfin;           // [E] 1 = the env's sim-step loop hit the budget of pass 0 (its loop state is in rs_i / rs_d) -- written by pass 0 for every env
    int *rs_i;                      // [E][16] phase, wi, path0, done_turning, dp_valid, sp_one, sim_steps, kcount, have_prev, still_done, total_sub, robot_hit, cycles >> 8
    double *rs_d;                   // [E][4 + 2 * 64] the controller's doubles (L.ctl) and the until-still loop's previous positions (one d2 per lane)
    unsigned *straggler;            // [4] cumulative: envs resumed by pass 1, envs whose loops ran into STEP_LIMIT, recurrences found in execute_robot_path, sim steps they skipped
    float *dist;                    // [E][SH*SW] spfa scratch
    float *rmap;                    // [E][SH*SW] spfa map from the robot (observation channel 2)
    const d2 *goals;                // [ngoal] area-clearing goal points
    unsigned char *cleared;         // [E][BD_MAXBOX] area-clearing box_clearance_statuses
};

// ---- deterministic libm replacements (fdlibm s_atan.c / e_atan2.c restated, fixed operation order, no FMA contraction) --------
__device__ __forceinline__ double bd_atan(double x)
{
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    const unsigned hx = (unsigned)__double2hiint(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    int id;
    if (ix >= 0x44100000u) return neg ? -(atanhi[3] + atanlo[3]) : (atanhi[3] + atanlo[3]);
    if (ix < 0x3fdc0000u) {
        if (ix < 0x3e200000u) return x;
        id = -1;
    } else {
        x = __builtin_fabs(x);
        if (ix < 0x3ff30000u) {
            if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else {
            if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
            else { id = 3; x = -1.0 / x; }
        }
    }
    double z = x * x;
    const double w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const double hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
    const double lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    z = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -z : z;
}
__device__ __forceinline__ double bd_atan2(double y, double x)
{
    const double pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    const unsigned hx = (unsigned)__double2hiint(x), hy = (unsigned)__double2hiint(y);
    const unsigned lx = (unsigned)__double2loint(x), ly = (unsigned)__double2loint(y);
    const unsigned ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (x == 1.0) return bd_atan(y);
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);
    if ((iy | ly) == 0) { if (m < 2) return y; return m == 2 ? pi : -pi; }
    if ((ix | lx) == 0) return (hy >> 31) ? -pi_o_2 : pi_o_2;
    const int k = ((int)iy - (int)ix) >> 20;
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;
    else if ((hx >> 31) && k < -60) z = 0.0;
    else z = bd_atan(__builtin_fabs(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
__device__ __forceinline__ double bd_pymod(double a, double b)
{
    double r = __builtin_fabs(a);
    while (r >= b) {
        double t = b;
        while (t + t <= r) t = t + t;
        r = r - t;
    }
    if (a < 0) r = -r;
    if (r != 0.0 && r < 0) r = r + b;
    return r;
}
__device__ __forceinline__ double bd_restrict(double h) { return bd_pymod(h + BP_PI, 2 * BP_PI) - BP_PI; }
__device__ __forceinline__ double bd_hdiff(double h1, double h2) { return bd_restrict(h1 - h2); }
__device__ __forceinline__ double bd_dist2(double ax, double ay, double bx, double by)
{
    const double dx = ax - bx, dy = ay - by;
    return __builtin_sqrt(dx * dx + dy * dy);
}

// ---- grid helpers -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bd_pos_to_win(const BdParams &B, double x, double y, int &wi, int &wj)
{
    // position_to_pixel_indices (box_delivery_env.py:1325-1330), then into the small-map window (positions inside the room
    // always fall inside it)
    long long i = (long long)__builtin_floor((double)B.H / 2 - y * B.ppm), j = (long long)__builtin_floor((double)B.W / 2 + x * B.ppm);
    i = i < 0 ? 0 : (i > B.H - 1 ? B.H - 1 : i);
    j = j < 0 ? 0 : (j > B.W - 1 ? B.W - 1 : j);
    int a = (int)i - B.si0, b = (int)j - B.sj0;
    a = a < 0 ? 0 : (a > B.SH - 1 ? B.SH - 1 : a);
    b = b < 0 ? 0 : (b > B.SW - 1 ? B.SW - 1 : b);
    wi = a; wj = b;
}
__device__ __forceinline__ void bd_win_to_pos(const BdParams &B, int wi, int wj, double &x, double &y)
{
    x = ((double)(wj + B.sj0) - (double)B.W / 2) / B.ppm;
    y = ((double)B.H / 2 - (double)(wi + B.si0)) / B.ppm;
}
__device__ __forceinline__ bool bd_bit(const unsigned *bits, int idx) { return (bits[idx >> 5] >> (idx & 31)) & 1u; }

__constant__ int BD_DI[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
__constant__ int BD_DJ[8] = {-1, 0, 1, 1, 1, 0, -1, -1};

struct BdLds {
    unsigned *freeb, *thinb;      // window bit rasters of this env's map
    unsigned short *q;            // [3][BD_QCAP]
    int *qn;                      // [4]
    unsigned short *pi, *pj;      // [BD_PATHCAP] dense path (window coordinates), later the kept waypoints
    unsigned char *keep;          // [BD_PATHCAP]
    unsigned short *stk;          // [BD_PATHCAP][2]
    double *wpx, *wpy;            // [BD_MAXWP] waypoints of the last shortest_path call
};
__device__ __forceinline__ size_t bd_lds_bytes(const BdParams &B)
{
    const size_t words = (size_t)((B.SH * B.SW + 31) / 32);
    return words * 8 + (size_t)3 * BD_QCAP * 2 + 16 + (size_t)BD_PATHCAP * 2 * 2 + BD_PATHCAP + (size_t)BD_PATHCAP * 4 + (size_t)BD_MAXWP * 16 + 64;
}
__device__ __forceinline__ void bd_carve(const BdParams &B, char *p, BdLds &L)
{
    const size_t words = (size_t)((B.SH * B.SW + 31) / 32);
    L.wpx = (double *)p; p += BD_MAXWP * 8;
    L.wpy = (double *)p; p += BD_MAXWP * 8;
    L.freeb = (unsigned *)p; p += words * 4;
    L.thinb = (unsigned *)p; p += words * 4;
    L.qn = (int *)p; p += 16;
    L.q = (unsigned short *)p; p += (size_t)3 * BD_QCAP * 2;
    L.pi = (unsigned short *)p; p += BD_PATHCAP * 2;
    L.pj = (unsigned short *)p; p += BD_PATHCAP * 2;
    L.stk = (unsigned short *)p; p += BD_PATHCAP * 4;
    L.keep = (unsigned char *)p; p += BD_PATHCAP;
}
__device__ __forceinline__ float bd_ld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// spfa from window cell `src` over the free cells; stops once `target` (>= 0) is settled.  dist: global [SH*SW], +inf = unreached.
__device__ __forceinline__ void bd_spfa(const BdParams &B, const BdLds &L, float *dist, int src, int target, int &err)
{
    const int lane = lane_id();
    const int NW = B.SH * B.SW;
    unsigned *du = (unsigned *)dist;
    for (int i = lane; i < NW; i += 64) du[i] = BD_INF_BITS;
    if (lane < 3) L.qn[lane] = 0;
    __syncthreads();
    if (lane == 0) { du[src] = 0u; L.q[0] = (unsigned short)src; L.qn[0] = 1; }
    __syncthreads();
    const float SQ2 = __builtin_sqrtf(2.0f);
    int empty_run = 0;
    for (int b = 0; b < 8192; b++) {
        const int qi = b % 3;
        const int n = min(L.qn[qi], BD_QCAP);
        if (n == 0) { if (++empty_run >= 3) break; continue; }
        empty_run = 0;
        unsigned short *qq = L.q + qi * BD_QCAP;
        for (int base = 0; base < n; base += 64) {
            const int idx = base + lane;
            bool act = idx < n;
            const int v = act ? (int)qq[idx] : 0;
            const float d = bd_ld(dist + v);
            act = act && ((int)d == b);
            const int vi = v / B.SW, vj = v - vi * B.SW;
            float old[8]; int nidx[8]; bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int ni = vi + BD_DI[k], nj = vj + BD_DJ[k];
                const bool inb = ni >= 0 && nj >= 0 && ni < B.SH && nj < B.SW;
                nidx[k] = inb ? ni * B.SW + nj : 0;
                ok[k] = act && inb && bd_bit(L.freeb, nidx[k]);
                old[k] = bd_ld(dist + nidx[k]);
            }
            // all eight relaxations are issued before any result is used: one L2 round trip instead of eight (the result does not
            // depend on the order: distances are the least fixed point, parents follow a local rule afterwards)
            unsigned prevb[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float nd = d + ((k & 1) ? 1.0f : SQ2);
                ok[k] = ok[k] && nd < old[k];
                prevb[k] = 0u;
                if (ok[k]) prevb[k] = atomicMin(du + nidx[k], __float_as_uint(nd));
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float nd = d + ((k & 1) ? 1.0f : SQ2);
                if (ok[k] && __float_as_uint(nd) < prevb[k]) {
                    const int tq = ((int)nd) % 3;
                    const int pos = atomicAdd(&L.qn[tq], 1);
                    if (pos < BD_QCAP) L.q[tq * BD_QCAP + pos] = (unsigned short)nidx[k];
                    else err |= BP_ERR_ARB_OVERFLOW;
                }
            }
        }
        __syncthreads();
        if (lane == 0) L.qn[qi] = 0;
        __syncthreads();
        if (target >= 0) {
            const float dt = bd_ld(dist + target);
            if (__float_as_uint(dt) != BD_INF_BITS && (int)dt <= b) break;
        }
    }
    __syncthreads();
}

// skimage.draw.line(r0, c0, r1, c1) membership in closed form: does the line cross a cell whose bit is 0?
__device__ __forceinline__ bool bd_line_blocked(const BdParams &B, const unsigned *bits, int r0, int c0, int r1, int c1)
{
    const int lane = lane_id();
    int dr = abs(r1 - r0), dc = abs(c1 - c0);
    int sc = (c1 - c0) > 0 ? 1 : -1, sr = (r1 - r0) > 0 ? 1 : -1;
    const bool steep = dr > dc;
    int maj0 = c0, min0 = r0;
    if (steep) { maj0 = r0; min0 = c0; int t = dc; dc = dr; dr = t; t = sc; sc = sr; sr = t; }
    bool blocked = false;
    for (int base = 0; base <= dc; base += 64) {
        const int i = base + lane;
        bool bl = false;
        if (i <= dc) {
            int mj, mn;
            if (i == dc) { mj = steep ? r1 : c1; mn = steep ? c1 : r1; }
            else { mj = maj0 + sc * i; mn = min0 + sr * (int)(((long long)2 * dr * i + dc) / ((long long)2 * dc)); }
            const int r = steep ? mj : mn, c = steep ? mn : mj;
            bl = !bd_bit(bits, r * B.SW + c);
        }
        if (ballot(bl)) { blocked = true; break; }
    }
    return blocked;
}

// shortest_path (box_delivery_env.py:1209-1264): waypoints into wp[][2] (global or local memory), returns their number.
// Window coordinates are used on the grid; Douglas-Peucker works on padded-room pixel coordinates like the reference.
__device__ __forceinline__ int bd_shortest_path(const BdParams &B, const BdPtrs &Q, const BdLds &L, int map, float *dist,
                                                double sx, double sy, double tx, double ty, bool check_straight, int &err)
{
    const int lane = lane_id();
    int si, sj, ti, tj;
    bd_pos_to_win(B, sx, sy, si, sj);
    bd_pos_to_win(B, tx, ty, ti, tj);
    __syncthreads(); // earlier readers of the waypoint buffer are done
    if (check_straight && !bd_line_blocked(B, L.thinb, si, sj, ti, tj)) {
        if (lane == 0) { L.wpx[0] = sx; L.wpy[0] = sy; L.wpx[1] = tx; L.wpy[1] = ty; }
        __syncthreads();
        return 2;
    }
    const unsigned short *edt = Q.edt + (size_t)map * B.SH * B.SW * 2;
    { const int a = edt[(si * B.SW + sj) * 2], b = edt[(si * B.SW + sj) * 2 + 1]; si = a; sj = b; }
    { const int a = edt[(ti * B.SW + tj) * 2], b = edt[(ti * B.SW + tj) * 2 + 1]; ti = a; tj = b; }
    const int src = si * B.SW + sj, tgt = ti * B.SW + tj;
    bd_spfa(B, L, dist, src, tgt, err);
    // dense path target -> source through the local parent rule
    int n = 0;
    {
        int ci = ti, cj = tj;
        if (lane == 0) { L.pi[0] = (unsigned short)ci; L.pj[0] = (unsigned short)cj; }
        n = 1;
        const float SQ2 = __builtin_sqrtf(2.0f);
        while (!(ci == si && cj == sj)) {
            const float dcur = bd_ld(dist + ci * B.SW + cj);
            bool okk = false;
            int ui = 0, uj = 0;
            if (lane < 8 && __float_as_uint(dcur) != BD_INF_BITS) {
                ui = ci - BD_DI[lane]; uj = cj - BD_DJ[lane];
                if (ui >= 0 && uj >= 0 && ui < B.SH && uj < B.SW && bd_bit(L.freeb, ui * B.SW + uj)) {
                    const float dn = bd_ld(dist + ui * B.SW + uj);
                    okk = (float)(dn + ((lane & 1) ? 1.0f : SQ2)) == dcur;
                }
            }
            const unsigned long long m = ballot(okk);
            if (!m) break;
            const int k = __ffsll((long long)m) - 1;
            ci = __shfl(ui, k); cj = __shfl(uj, k);
            if (n >= BD_PATHCAP) { err |= BP_ERR_ARB_OVERFLOW; break; }
            if (lane == 0) { L.pi[n] = (unsigned short)ci; L.pj[n] = (unsigned short)cj; }
            n++;
        }
    }
    __syncthreads();
    // approximate_polygon(coords, tolerance=1): Douglas-Peucker with an explicit stack
    for (int i = lane; i < n; i += 64) L.keep[i] = (i == 0 || i == n - 1) ? 1 : 0;
    __syncthreads();
    {
        int sp = 0;
        if (lane == 0) { L.stk[0] = 0; L.stk[1] = (unsigned short)(n - 1); }
        sp = 1;
        __syncthreads();
        while (sp > 0) {
            sp--;
            const int start = L.stk[sp * 2], end = L.stk[sp * 2 + 1];
            const long long r0 = (long long)L.pi[start] + B.si0, c0 = (long long)L.pj[start] + B.sj0;
            const long long r1 = (long long)L.pi[end] + B.si0, c1 = (long long)L.pj[end] + B.sj0;
            const long long dr = r1 - r0, dc = c1 - c0;
            const double ang = -bd_atan2((double)dr, (double)dc);
            double sa, ca;
            bp_sincos(ang, sa, ca);
            const double seg_dist = (double)c0 * sa + (double)r0 * ca;
            double best = -1.0; int arg = 0x7FFFFFFF; bool any = false;
            for (int k = start + 1 + lane; k < end; k += 64) {
                const long long rk = (long long)L.pi[k] + B.si0, ck = (long long)L.pj[k] + B.sj0;
                const long long dr0 = rk - r0, dc0 = ck - c0, dr1 = rk - r1, dc1 = ck - c1;
                const long long pl0 = dr0 * dr + dc0 * dc, pl1 = -dr1 * dr - dc1 * dc;
                double d;
                if (pl0 > 0 && pl1 > 0) d = __builtin_fabs(((double)rk * ca + (double)ck * sa) - seg_dist);
                else d = fmin(__builtin_sqrt((double)(dc0 * dc0 + dr0 * dr0)), __builtin_sqrt((double)(dc1 * dc1 + dr1 * dr1)));
                if (d > 1.0) any = true;
                if (d > best) { best = d; arg = k; }
            }
            // first maximum: largest d, smallest index among equals
            double wb = best;
            for (int o = 32; o >= 1; o >>= 1) wb = fmax(wb, __shfl_xor(wb, o));
            int wa = (best == wb && arg != 0x7FFFFFFF) ? arg : 0x7FFFFFFF;
            for (int o = 32; o >= 1; o >>= 1) wa = min(wa, __shfl_xor(wa, o));
            if (ballot(any)) {
                if (lane == 0) {
                    L.stk[sp * 2] = (unsigned short)wa; L.stk[sp * 2 + 1] = (unsigned short)end;
                    L.stk[sp * 2 + 2] = (unsigned short)start; L.stk[sp * 2 + 3] = (unsigned short)wa;
                    L.keep[wa] = 1;
                }
                sp += 2;
            }
            __syncthreads();
        }
    }
    // compact the kept points (order preserved)
    int m = 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const bool kp = k < n && L.keep[k];
        const unsigned short a = k < n ? L.pi[k] : 0, b = k < n ? L.pj[k] : 0;
        const unsigned long long mk = ballot(kp);
        __syncthreads();
        if (kp) { const int pos = m + popc_below(mk, lane); L.pi[pos] = a; L.pj[pos] = b; }
        m += __popcll(mk);
        __syncthreads();
    }
    // remove unnecessary waypoints: new_coords kept in place behind a write cursor q (q <= k always)
    int q = 1;
    for (int k = 1; k < m - 1; k++) {
        const bool bl = bd_line_blocked(B, L.freeb, L.pi[q - 1], L.pj[q - 1], L.pi[k + 1], L.pj[k + 1]);
        if (bl) {
            const unsigned short a = L.pi[k], b = L.pj[k];
            __syncthreads();
            if (lane == 0) { L.pi[q] = a; L.pj[q] = b; }
            q++;
            __syncthreads();
        }
    }
    if (m > 1) {
        const unsigned short a = L.pi[m - 1], b = L.pj[m - 1];
        __syncthreads();
        if (lane == 0) { L.pi[q] = a; L.pj[q] = b; }
        q++;
        __syncthreads();
    }
    if (q < 2) {
        if (lane == 0) { L.wpx[0] = sx; L.wpy[0] = sy; L.wpx[1] = tx; L.wpy[1] = ty; }
        __syncthreads();
        return 2;
    }
    if (q > BD_MAXWP) { q = BD_MAXWP; err |= BP_ERR_ARB_OVERFLOW; }
    for (int k = lane; k < q; k += 64) {
        double x, y;
        bd_win_to_pos(B, L.pi[q - 1 - k], L.pj[q - 1 - k], x, y);
        if (k == 0) { x = sx; y = sy; }
        if (k == q - 1) { x = tx; y = ty; }
        L.wpx[k] = x; L.wpy[k] = y;
    }
    __syncthreads();
    return q;
}

__device__ __forceinline__ double bd_path_distance(const BdParams &B, const BdPtrs &Q, const BdLds &L, int map, float *dist,
                                                   double sx, double sy, double tx, double ty, int &err)
{
    const int n = bd_shortest_path(B, Q, L, map, dist, sx, sy, tx, ty, false, err);
    double sum = 0.0;
    for (int i = 1; i < n; i++) sum += bd_dist2(L.wpx[i - 1], L.wpy[i - 1], L.wpx[i], L.wpy[i]);
    return sum;
}

__device__ __forceinline__ void bd_load_bits(const BdParams &B, const BdPtrs &Q, const BdLds &L, int map)
{
    const int words = (B.SH * B.SW + 31) / 32;
    const unsigned *f = Q.free_bits + (size_t)map * words, *t = Q.thin_bits + (size_t)map * words;
    for (int i = lane_id(); i < words; i += 64) { L.freeb[i] = f[i]; L.thinb[i] = t[i]; }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bd_plan: one wave per env
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bd_plan(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q,
                                                const double *__restrict__ actions)
{
    const int env = blockIdx.x;
    const int lane = lane_id();
    BdLds L;
    bd_carve(B, (char *)bp_smem, L);
    const int map = Q.map_of_trial[D.e_trial[env]];
    bd_load_bits(B, Q, L, map);
    int err = 0;
    const size_t eb = (size_t)env * P.nbcap;
    const double ix = D.pxy[eb].x, iy = D.pxy[eb].y, ih = bd_restrict(D.ang[eb]);
    if (B.action_type == 2) { // velocity control needs no plan: remember the start pose and the two speeds
        if (lane == 0) {
            double *sf = Q.stepf + (size_t)env * 8;
            sf[1] = ix; sf[2] = iy; sf[3] = ih; sf[6] = actions[2 * env]; sf[7] = actions[2 * env + 1];
            Q.nwp[env] = 0;
        }
        return;
    }
    // heading action -> pixel of the local map (box_delivery_env.py:706-723), in binary64; a position action is the index itself
    const double angle = (actions[env] + 1) * BP_PI + BP_PI / 2;
    double sa, ca;
    bp_sincos(angle, sa, ca);
    const double x_movement = B.step_size * ca, y_movement = B.step_size * sa;
    int x_pixel = (int)((double)B.local_px / 2 + x_movement * B.ppm);
    int y_pixel = (int)((double)B.local_px / 2 - y_movement * B.ppm);
    if (B.action_type == 1) { const long long idx = (long long)actions[env]; y_pixel = (int)(idx / B.local_px); x_pixel = (int)(idx % B.local_px); }
    // get_waypoints_to_spatial_action (position_controller.py:56-123)
    const double xm = -B.local_w / 2 + (double)x_pixel / B.ppm;
    const double ym = B.local_w / 2 - (double)y_pixel / B.ppm;
    const double sld = __builtin_sqrt(xm * xm + ym * ym);
    const double turn = bd_atan2(-xm, ym);
    const double slh = bd_restrict(ih + turn);
    double sh_, ch_;
    bp_sincos(slh, sh_, ch_);
    double tx = ix + sld * ch_, ty = iy + sld * sh_;
    const double dfx = tx - ix, dfy = ty - iy;
    double ratio_x = 1, ratio_y = 1;
    const double sgx = (double)((tx > 0) - (tx < 0)), sgy = (double)((ty > 0) - (ty < 0));
    const double bound_x = sgx * B.room_length / 2, bound_y = sgy * B.room_width / 2;
    if (__builtin_fabs(tx) > __builtin_fabs(bound_x)) ratio_x = (bound_x - ix) / (tx - ix);
    if (__builtin_fabs(ty) > __builtin_fabs(bound_y)) ratio_y = (bound_y - iy) / (ty - iy);
    const double ratio = ratio_y < ratio_x ? ratio_y : ratio_x;
    tx = ix + ratio * dfx; ty = iy + ratio * dfy;
    const int nwp = bd_shortest_path(B, Q, L, map, Q.dist + (size_t)env * B.SH * B.SW, ix, iy, tx, ty, true, err);
    const double dte = bd_dist2(L.wpx[nwp - 2], L.wpy[nwp - 2], L.wpx[nwp - 1], L.wpy[nwp - 1]);
    const double signed_dist = dte - B.robot_radius;
    const bool backing = nwp > 2 && signed_dist < 0; // "avoid awkward backing up": waypoint[-2] := waypoint[-1]
    double *o = Q.wp + (size_t)env * BD_MAXWP * 3;
    for (int i = lane; i < nwp; i += 64) {
        double x = L.wpx[i], y = L.wpy[i];
        if (backing && i == nwp - 2) { x = L.wpx[nwp - 1]; y = L.wpy[nwp - 1]; }
        double hd = 0.0; // waypoint 0 has no heading (None)
        if (i >= 1) {
            // headings come from the unmodified list, except [-2] which is recomputed from the moved waypoint
            const double px = L.wpx[i - 1], py = L.wpy[i - 1];
            hd = bd_restrict(bd_atan2(y - py, x - px));
        }
        o[3 * i] = x; o[3 * i + 1] = y; o[3 * i + 2] = hd;
    }
    const unsigned long long em = ballot(err != 0);
    if (lane == 0) {
        Q.nwp[env] = nwp;
        double *sf = Q.stepf + (size_t)env * 8;
        sf[1] = ix; sf[2] = iy; sf[3] = ih;
        if (em) atomicOr(&D.e_err[env], BP_ERR_ARB_OVERFLOW);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bd_physics: execute_robot_path + step_simulation_until_still, one wave per env (persistent over all sim steps)
// ---------------------------------------------------------------------------------------------------------------------
// cpPolyShapePointQuery + cpSpacePointQuery(max_distance 0): p strictly inside the (rounded) convex shape
__device__ __forceinline__ bool bd_point_in_shape(const d2 *wv, const d2 *wn, int n, double r, double4 bb, d2 p)
{
    if (!(bb.x <= p.x && p.x <= bb.z && bb.y <= p.y && p.y <= bb.w)) return false;
    d2 v0 = wv[n - 1];
    double minDist = BP_INF;
    bool outside = false;
    for (int i = 0; i < n; i++) {
        const d2 v1 = wv[i];
        outside = outside || (vdot(wn[i], vsub(p, v1)) > 0.0);
        const d2 delta = vsub(v0, v1);
        const double t = clamp01(vdot(delta, vsub(p, v1)) / vdot(delta, delta));
        const d2 closest = vadd(v1, vmul(delta, t));
        const double d = vlen(vsub(p, closest));
        if (d < minDist) minDist = d;
        v0 = v1;
    }
    const double dist = outside ? minDist : -minDist;
    return (dist - r) < 0.0;
}
// body.local_to_world(v) for hull vertex q of body i from its current pose (cpTransformPoint)
__device__ __forceinline__ d2 bd_local_to_world(const EnvCtx &E, int i, int q)
{
    const d2 p = gE(E.pxy, i), r = gE(E.rot, i);
    const double4 ms = gE(E.mass, i);
    const double tx = p.x - (ms.z * r.x - ms.w * r.y), ty = p.y - (ms.z * r.y + ms.w * r.x);
    const d2 lv = gE(E.lv, i * BP_MAXV + q);
    return mk2((r.x * lv.x + (-r.y) * lv.y) + tx, (r.y * lv.x + r.x * lv.y) + ty);
}

// ---- exact recurrences of execute_robot_path (bd_physics_body) -------------------------------------------------------------------------------------------
// The robot's state -- per kinematic part: position, angle, (cos, sin), velocity, (w, w_bias), bias velocity, velocity slot; once: the controller's doubles and
// the packed loop / sub-step words handed in -- is compared with the snapshot (store == 0: returns 1 if every value is bit for bit the snapshot's) or written
// to it (store != 0).  Only the kernel of the second pass of the two-pass step holds this code (bd_physics_body<DAMP, CYC = true>, k_bd_physics_resume): with it
// inline the sim step of k_bd_physics spilled 53 VGPRs instead of 11 and ran 5 % slower for every env, as a non-inlined function 41.
__device__ __forceinline__ int bd_cycle_visit(unsigned long long *snap, const d2 *sp, const d2 *sv, const d2 *sw, const d2 *sb, const unsigned char *slot_of,
                                                        const double *ctl, const d2 *rot, const double *ang, const int nkin, const int store,
                                                        const unsigned long long w5, const unsigned long long w6, const unsigned long long w7, const unsigned long long w8,
                                                        const unsigned long long w9, const double ke, const double imp, const double curr_dt)
{
    const int lane = lane_id();
    bool same = true;
    lds_sync();
    if (lane <= nkin) {
        unsigned long long *row = snap + lane * BP_SNAP_COLS;
        auto item = [&](const int k, const unsigned long long v) { if (store) row[k] = v; else same = same && (row[k] == v); };
        auto itemd = [&](const int k, const double v) { item(k, __builtin_bit_cast(unsigned long long, v)); };
        if (lane < nkin) {          // part `lane` of the kinematic robot (bodies [0, nkin) hold the velocity slots [0, nkin))
            const d2 p_ = sp[lane]; itemd(0, p_.x); itemd(1, p_.y);
            itemd(2, ang[lane]);
            const d2 r_ = rot[lane]; itemd(3, r_.x); itemd(4, r_.y);
            const d2 v_ = sv[lane]; itemd(5, v_.x); itemd(6, v_.y);
            const d2 w_ = sw[lane]; itemd(7, w_.x); itemd(8, w_.y);
            const d2 b_ = sb[lane]; itemd(9, b_.x); itemd(10, b_.y);
            item(11, (unsigned long long)slot_of[lane]);
        } else {                    // the controller and whatever else of the sub-step state a sim step can read or change
            itemd(0, ctl[0]); itemd(1, ctl[1]); itemd(2, ctl[3]); itemd(3, ke); itemd(4, imp);
            item(5, w5); item(6, w6); item(7, w7); item(8, w8); item(9, w9);
            itemd(10, curr_dt);
        }
    }
    lds_sync();
    return (ballot(!same) == 0ull) ? 1 : 0;
}
// the counters of `skip` skipped iterations that live in LDS / HBM: the "moved in the sub-step that just ended" stamps of the robot's parts, the advanced
// path length (TargetCourse.advance of every skipped iteration, added one by one so that it rounds like the loop's own additions) and the statistics
__device__ __forceinline__ void bd_cycle_skip(unsigned *mvs, double *ctl, const int nkin, const unsigned stamp, const int skip, const double target_speed,
                                                        const double ctrl_dt, unsigned *stats)
{
    const int lane = lane_id();
    lds_sync();
    if (lane < nkin) mvs[lane] = stamp;
    if (lane == 0) {
        double al = ctl[2];
        for (int k = 0; k < skip; k++) al = al + target_speed * ctrl_dt;
        ctl[2] = al;
        if (stats != nullptr) { atomicAdd(&stats[2], 1u); atomicAdd(&stats[3], (unsigned)skip); }
    }
    lds_sync();
}
#ifndef BP_BD_WAVES
#define BP_BD_WAVES 2
#endif
// DAMP: space.damping != 0 (box_delivery_env.py:204, area_clearing.py: `space.damping = cfg.sim.damping`; no shipped config) -> substep<BP_ENV_BOX, true>
template <bool DAMP, bool CYC = false>
__device__ __forceinline__ void bd_physics_body(const DevParams &P, const DevPtrs &D, const BdParams &B, const BdPtrs &Q)
{
    const int env = (D.order != nullptr) ? D.order[blockIdx.x] : (int)blockIdx.x;
    const bool resume = B.pass == 1;
    if (resume && Q.unfin[env] == 0) return;       // pass 1 serves only the envs that pass 0 left unfinished
    // ... in two launches: the envs that ran out of budget inside execute_robot_path (saved phase 0 = PH_PATH) go to the kernel that holds the recurrence test,
    // the ones inside step_simulation_until_still -- the reference's own 10 001-step loops, which never recur -- stay on the lean kernel
    if (resume && B.resume_sel != 0 && ((Q.rs_i[(size_t)env * 16] == 0) != (B.resume_sel == 1))) return;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const int lane = lane_id();
    LdsCtx L;
    carve_lds<BP_ENV_BOX>(P, L);
    const int trial = D.e_trial[env];
    EnvCtx E;
    E.nb = D.e_nb[env];
    env_ctx(P, D, env, trial, E);
    ArbReg A;
    SubState S;
#ifdef BP_PROF
    if (lane < BP_PROFN) L.prof[lane] = 0ull;
    lds_sync();
    const unsigned long long _t_kernel0 = __builtin_amdgcn_s_memtime();
    unsigned long long _t_ctrl = 0;
#endif
    init_regs(A, S);
    load_state_a<BP_ENV_BOX>(P, D, E, L, A, S, env);
    load_state_b<BP_ENV_BOX>(P, D, E, L, A, S, env);
    const int map = Q.map_of_trial[trial];
    const double *wp = Q.wp + (size_t)env * BD_MAXWP * 3;
    const int nwp = Q.nwp[env];
    const double *sf = Q.stepf + (size_t)env * 8;
    // box-delivery clears robot_hit_obstacle at the start of a step (:640), area-clearing at its end (area_clearing.py:776)
    S.robot_hit = (B.task == 1) ? (int)sf[4] : 0;
    // Wave-uniform doubles of the path controller live in LDS (L.ctl), not in VGPRs: the kernel sits at the 256-VGPR line and spilled them to scratch around
    // every sim step; they are touched a handful of times per sim step.  ctl[0] prev_heading_diff, [1] path length, [2] advanced length, [3] robot_distance.
    if (lane < 4) L.ctl[lane] = resume ? Q.rs_d[(size_t)env * 132 + lane] : 0.0;
    lds_sync();
    unsigned total_sub = 0;
    // One loop with a single substep call site (the physics is one large inlined function); the phase selects what happens
    // before and after each sim step:
    //   PATH   execute_robot_path (box_delivery_env.py:891-988 / area_clearing.py:800-901)
    //   VEL    box-delivery velocity control (box_delivery_env.py:672-703): re-aim every sim step, stop at the first boundary hit
    //   FIXED  area-clearing: `steps` sim steps with the last commanded twist (area_clearing.py:691-693)
    //   STILL  box-delivery step_simulation_until_still (box_delivery_env.py:990-1023)
    enum { PH_PATH = 0, PH_VEL, PH_FIXED, PH_STILL, PH_DONE };
    int phase = PH_PATH;
    // path-execution state.  Wave-uniform doubles live in VGPRs, so only what cannot be re-read is kept across the sim step:
    // waypoints / set-point candidates are re-read from the waypoint list (scalar loads), the pose from the body arrays.
    int wi = 1, path0 = 0;
    bool done_turning = false, dp_valid = false, sp_one = false;
    int sim_steps = 0, kcount = 0;
    // until-still state
    const int nalive = Q.nalive[env];
    const unsigned char *order = Q.order + (size_t)env * BD_MAXBOX;
    d2 prevp = mk2(0.0, 0.0); // lane q < nalive: box order[q]; lane nalive: robot
    bool have_prev = false, still_done = false;
    const int first_static = B.first_box + B.nbox;
    const int nstat = E.nb - first_static;
    bool first_after_resume = false, hit_limit = false;   // hit_limit: one of the two loops ran into STEP_LIMIT in this env step (statistics only)
    // Exact recurrence of execute_robot_path (kind-(i) stragglers: a robot that pushes against a wall until STEP_LIMIT, box_delivery_env.py:891-988).  While
    // only the robot moves and no arbiter exists, the next sim step is a function of the robot's parts (pose, velocities), the controller's loop variables
    // and nothing else -- boxes lie still and touch nothing, walls never move, cached planes / neighbour lists / the candidate cache only steer how the
    // same results are reached.  If that state is bit for bit the one of p sim steps ago, the loop is periodic from here on, and whole periods can be
    // skipped: only counters advance (sim steps, stamps, the advanced path length `al`, which is re-added step by step so that it rounds as the loop's
    // own additions do, and which decides nothing once the set point has switched).  Brent's scheme finds any period with one snapshot (a power-of-two
    // sim step ago, L.snap) and one comparison per sim step; the robots of the traces in profiles/r03_box/straggler_period.txt sit at a fixed point
    // (p = 1) within a few hundred sim steps.  The last iteration of the loop always runs for real: it is the one that leaves.
    // Where: in the second pass of the two-pass step only (CYC, k_bd_physics_resume).  The comparison costs ~3 % of a sim step and, inline in the one kernel
    // every env runs, 42 more spilled VGPRs (-5 % for everybody); the envs that need it are exactly those that run out of the first pass's budget.
    int cyc_pow = 1, cyc_lam = 0;
    bool cyc_have = false;
    unsigned cyc_costp = 0u;
    unsigned cycles_before = 0u;
    if (resume) {
        // the loop state of pass 0 (the bodies, arbiters and velocities came back through load_state, as after a park of the step scheduler)
        const int *ri = Q.rs_i + (size_t)env * 16;
        phase = ri[0]; wi = ri[1]; path0 = ri[2]; done_turning = ri[3] != 0; dp_valid = ri[4] != 0; sp_one = ri[5] != 0;
        sim_steps = ri[6]; kcount = ri[7]; have_prev = ri[8] != 0; still_done = ri[9] != 0; total_sub = (unsigned)ri[10]; S.robot_hit = ri[11];
        cycles_before = (unsigned)ri[12]; hit_limit = ri[13] != 0;
        const double *rd = Q.rs_d + (size_t)env * 132 + 4 + 2 * lane;
        prevp = mk2(rd[0], rd[1]);
        first_after_resume = true;   // the "moved in the last sim step" stamps are gone: the first stuck-box test looks at every box (the test is a pure function of the pose)
    } else
    if (B.action_type == 2) {
        if (B.task == 1) {
            // area-clearing velocity control (area_clearing.py:660-667): set once; the common sim steps follow
            const d2 r = L.ag[1];
            const double sv = B.target_speed * sf[6];
            if (lane < P.nkin) {
                L.sw[lane] = mk2(B.yaw_rate_step * sf[7] / 2, L.sw[lane].y);
                L.sv[lane] = mk2(r.x * sv + -r.y * 0.0, r.y * sv + r.x * 0.0);
            }
            __syncthreads();
            phase = PH_FIXED;
        } else phase = PH_VEL;
    }
    while (phase != PH_DONE) {
        // ---------------- before the sim step ----------------
        if (phase == PH_PATH) {
#ifdef BP_PROF
            const unsigned long long _tc0 = __builtin_amdgcn_s_memtime();
#endif
            const double prevx = L.sp[0].x, prevy = L.sp[0].y, prevh = bd_restrict(L.ag[0].x); // pose left by the last sim step
            const double hd = bd_hdiff(prevh, wp[3 * wi + 2]);
            if (!(__builtin_fabs(hd) > 15 * (BP_PI / 180.0) && __builtin_fabs(hd - L.ctl[0]) > 0.001)) done_turning = true;
            lds_sync();
            if (lane == 0) L.ctl[0] = hd; // prev_heading_diff of the next iteration (only read above)
            const double cx0 = wp[3 * path0], cy0 = wp[3 * path0 + 1], cx1 = wp[3 * path0 + 3], cy1 = wp[3 * path0 + 4];
            if (!dp_valid) { // DP(...) -> TargetCourse.init_setpoint (dp.py:67-88)
                const double dx = cx1 - cx0, dy = cy1 - cy0;
                const double plen0 = __builtin_sqrt(dx * dx + dy * dy);
                const double d0 = bd_dist2(prevx, prevy, cx0, cy0), d1 = bd_dist2(prevx, prevy, cx1, cy1);
                bool one = d1 < d0;
                // look-ahead (dp.py:78-83): only ever advances from point 0 to point 1; never runs with Lfc == 0
                if (!one && B.lfc > d0) one = true;
                if (lane == 0) { L.ctl[1] = plen0; L.ctl[2] = plen0; }
                lds_sync();
                sp_one = one;
                dp_valid = true;
            }
            const double spx = sp_one ? cx1 : cx0, spy = sp_one ? cy1 : cy0;
            // ideal_control (dp.py:217-248)
            const double theta_d = bd_atan2(spy - prevy, spx - prevx);
            double theta_e = theta_d - prevh;
            // the two sincos of the controller (theta_e here, the heading for the goal velocity below) are independent: lane 1 evaluates the second one
            // while the other lanes evaluate the first -- one pass of the polynomial instead of two
            double sn2, cs2;
            bp_sincos(lane == 1 ? prevh : theta_e, sn2, cs2);
            const double se = __shfl(sn2, 0), ce = __shfl(cs2, 0);
            const double sy_ = __shfl(sn2, 1), cy_ = __shfl(cs2, 1);
            theta_e = bd_atan2(se, ce);
            double omega = 1.0 * theta_e;
            omega = omega / B.ctrl_dt;
            const double gvx = cy_ * B.target_speed + -sy_ * 0.0, gvy = sy_ * B.target_speed + cy_ * 0.0;
            const double al = L.ctl[2] + B.target_speed * B.ctrl_dt;  // TargetCourse.advance
            sp_one = L.ctl[1] < al;
            lds_sync();
            if (lane == 0) L.ctl[2] = al;
            // apply_controller (box_delivery_env.py:887-889 / area_clearing.py:903-906)
            if (lane < P.nkin) {
                L.sw[lane] = mk2(omega * B.omega_scale, L.sw[lane].y);
                L.sv[lane] = done_turning ? mk2(gvx * B.v_scale, gvy * B.v_scale) : mk2((gvx * 0) * B.v_scale, (gvy * 0) * B.v_scale);
            }
            lds_sync();   // velocity slots live in LDS and the workgroup is one wavefront: no drain of the global stores of the last sim step
#ifdef BP_PROF
            _t_ctrl += __builtin_amdgcn_s_memtime() - _tc0;
#endif
        } else if (phase == PH_VEL) {
            const d2 r = L.ag[1]; // (cos, sin) of body.angle, refreshed by the previous sim step
            double lin = sf[6];
            const double angv = sf[7];
            if (__builtin_fabs(lin) >= B.target_speed) lin = B.target_speed * (double)((lin > 0) - (lin < 0));
            if (lane < P.nkin) {
                L.sw[lane] = mk2(angv, L.sw[lane].y);
                L.sv[lane] = mk2(r.x * lin + -r.y * 0.0, r.y * lin + r.x * 0.0);
            }
            __syncthreads();
        } else if (phase == PH_STILL) {
            // boxes with a vertex strictly inside an obstacle shape are moved to the nearest free cell (box_delivery_env.py:995-1004).  The test is a pure
            // function of the box's pose (the obstacle shapes never move), and a box that failed it was relocated, i.e. it moved: a box that did not move in
            // the last sim step (L.mvs, the stamp of the sub-step that last integrated it) would repeat the answer "free" of the previous iteration and is
            // skipped; the first iteration tests every box.  Of the (box vertex, obstacle shape) items of the boxes that are left, those whose vertex lies
            // outside the shape's AABB -- nearly all -- are dropped by a first pass that costs a comparison; the point query runs on the compacted rest.
            unsigned long long stuck = 0ull;
            const bool cand = lane < nalive && (!have_prev || first_after_resume || L.mvs[B.first_box + order[lane]] == S.stamp);
            first_after_resume = false;
            const unsigned long long candm = ballot(cand);
            if (cand) L.rf[popc_below(candm, lane)] = (unsigned char)lane;   // L.rf: scratch of the integrate phase, free between sim steps
            lds_sync();
            const int items = __popcll(candm) * 4 * nstat;
            unsigned *surv = L.q_meta;   // [BP_QCAP] item ids that passed the AABB test (narrow-phase scratch, free between sim steps)
            int nsurv = 0;
            auto run_queries = [&]() {   // full point query on the compacted items
                for (int b2 = 0; b2 < nsurv; b2 += 64) {
                    const int k = b2 + lane;
                    bool hit = false;
                    int q = 0;
                    if (k < nsurv) {
                        const int it = (int)surv[k];
                        const int c = it / (4 * nstat);
                        q = L.rf[c];
                        const int rem = it - c * 4 * nstat, vi = rem / nstat, s2 = first_static + (rem - vi * nstat);
                        const int body = B.first_box + order[q];
                        hit = bd_point_in_shape(E.wv + s2 * BP_MAXV, E.wn + s2 * BP_MAXV, gE(E.nv, s2), gE(E.prop, s2).x, gE(E.bb, s2), bd_local_to_world(E, body, vi));
                    }
                    unsigned long long hm = ballot(hit);
                    while (hm) { const int l = __ffsll((long long)hm) - 1; hm &= hm - 1; stuck |= 1ull << __shfl(q, l); }
                }
                nsurv = 0;
            };
            for (int base = 0; base < items; base += 64) {
                const int it = base + lane;
                bool pre = false;
                if (it < items) {
                    const int c = it / (4 * nstat);
                    const int q = L.rf[c];
                    const int rem = it - c * 4 * nstat, vi = rem / nstat, s2 = first_static + (rem - vi * nstat);
                    const int body = B.first_box + order[q];
                    if (vi < gE(E.nv, body)) {
                        const d2 p = bd_local_to_world(E, body, vi);
                        const double4 bb = gE(E.bb, s2);
                        pre = bb.x <= p.x && p.x <= bb.z && bb.y <= p.y && p.y <= bb.w;   // the first test of bd_point_in_shape
                    }
                }
                const unsigned long long pm = ballot(pre);
                static_assert(BP_QCAP >= 64 && BD_MAXBOX <= 64, "a flush leaves room for one wave's survivors in q_meta[BP_QCAP]; L.rf[64] is indexed by box rank");
                static_assert(BP_NSLOT + 2 <= 255, "slot_of keeps 255 as its no-slot sentinel");
                if (pm) {
                    if (nsurv + __popcll(pm) > BP_QCAP) { lds_sync(); run_queries(); lds_sync(); }
                    if (pre) surv[nsurv + popc_below(pm, lane)] = (unsigned)it;
                    nsurv += __popcll(pm);
                }
            }
            lds_sync();
            if (nsurv) run_queries();
            while (stuck) {
                const int q = __ffsll((long long)stuck) - 1;
                stuck &= stuck - 1;
                const int body = B.first_box + order[q];
                const d2 p = gE(E.pxy, body);
                int wi_, wj_;
                bd_pos_to_win(B, p.x, p.y, wi_, wj_);
                const unsigned short *edt = Q.edt + ((size_t)map * B.SH * B.SW + (size_t)wi_ * B.SW + wj_) * 2;
                double nx, ny;
                bd_win_to_pos(B, edt[0], edt[1], nx, ny);
                __syncthreads();
                if (lane == 0) {
                    gE(E.pxy, body) = mk2(nx, ny);
                    const int sl = L.slot_of[body];
                    if (sl != 255) { L.sv[sl] = mk2(0.0, 0.0); L.sp[sl] = mk2(nx, ny); }
                }
                // make sure the body is re-cached by the next sub-step
                bool present = false;
                for (int k0 = 0; k0 < S.nmv; k0 += 64) present = present || (ballot(k0 + lane < S.nmv && L.mv[k0 + lane] == (unsigned short)body) != 0);
                if (!present && S.nmv < P.mvcap) { if (lane == 0) L.mv[S.nmv] = (unsigned short)body; S.nmv++; S.cc_ok = 0; }
                __syncthreads();
            }
            d2 cur = mk2(0.0, 0.0);
            if (lane < nalive) cur = gE(E.pxy, B.first_box + order[lane]);
            else if (lane == nalive) cur = L.sp[0];
            if (have_prev) {
                // python loop with break: the comparison is pure, so "any" gives the same answer
                const bool moved = lane <= nalive && bd_dist2(prevp.x, prevp.y, cur.x, cur.y) > 0.005;
                still_done = ballot(moved) == 0;
            }
            prevp = cur; have_prev = true;
        }
        // ---------------- the sim step ----------------
        substep<BP_ENV_BOX, DAMP>(P, E, L, A, S, P.dt_sub, false);
        if (BP_UNLIKELY2(BP_TRACE_ON(D) && env == D.dbg_env && total_sub < 10100u)) { // bp_debug_trace: (x, y, angle) of every body after each sim step
            for (int i = lane; i < E.nb; i += 64) {
                double *o = D.dbg + ((size_t)total_sub * P.nbcap + i) * 3;
                o[0] = gE(E.pxy, i).x; o[1] = gE(E.pxy, i).y; o[2] = gE(E.ang, i);
            }
        }
        total_sub++;
        // ---------------- after the sim step ----------------
        const int after_move = (B.task == 1) ? PH_FIXED : PH_STILL;
        if (phase == PH_PATH) {
            const double px = L.sp[0].x, py = L.sp[0].y, ph = bd_restrict(L.ag[0].x);
            const double pwx = wp[3 * (wi - 1)], pwy = wp[3 * (wi - 1) + 1]; // robot_prev_waypoint_position
            const double wpx_ = wp[3 * wi], wpy_ = wp[3 * wi + 1], wph_ = wp[3 * wi + 2];
            bool leave = false;
            if (bd_dist2(pwx, pwy, px, py) > 0.05 && S.robot_hit) leave = true;
            else {
                if (bd_dist2(px, py, wpx_, wpy_) < 0.6 && __builtin_fabs(ph - wph_) < 10 * (BP_PI / 180.0)) {
                    { const double rd = L.ctl[3] + bd_dist2(pwx, pwy, px, py); lds_sync(); if (lane == 0) L.ctl[3] = rd; lds_sync(); }
                    if (wi == nwp - 1) leave = true;
                    else { wi++; done_turning = false; dp_valid = false; path0++; }
                }
                if (!leave) { sim_steps++; if (sim_steps > B.step_limit) { leave = true; hit_limit = true; } }
            }
            if (leave) { phase = after_move; sim_steps = 0; kcount = 0; }
            else if (CYC && B.cycle_skip > 0 && sim_steps >= B.cycle_skip && !BP_TRACE_ON(D) && P.nkin < BP_SNAP_ROWS) {
                // (CYC: the kernel of the second pass, which only the envs that ran out of the first pass's sim-step budget reach -- the 99th percentile of a
                //  whole env step is ~1 100 sim steps, the budget 3 000)
                const bool quiet = S.nmv == P.nkin && ballot(A.key != ARB_FREE_KEY) == 0ull && sp_one && dp_valid;
                if (!quiet) { cyc_have = false; cyc_pow = 1; cyc_lam = 0; }
                else {
                    const unsigned long long w5 = (unsigned long long)(done_turning ? 1u : 0u) | ((unsigned long long)(unsigned)S.robot_hit << 1) | ((unsigned long long)(unsigned)S.wall_flag << 2) |
                                                  ((unsigned long long)(unsigned)wi << 8) | ((unsigned long long)(unsigned)path0 << 20) | ((unsigned long long)(unsigned)S.nslots << 32) |
                                                  ((unsigned long long)(unsigned)S.nlevels << 44);
                    const unsigned long long w7 = (unsigned long long)S.n_post | ((unsigned long long)S.n_contact << 32);
                    const unsigned long long w8 = (unsigned long long)S.n_first | ((unsigned long long)(unsigned)S.err << 32);
                    const unsigned long long w9 = (unsigned long long)(unsigned)S.yaw_violated | ((unsigned long long)(unsigned)S.boundary_violated << 1) | ((unsigned long long)(unsigned)S.quiescent << 2);
                    const int hit = cyc_have ? bd_cycle_visit(L.snap, L.sp, L.sv, L.sw, L.sb, L.slot_of, L.ctl, E.rot, E.ang, P.nkin, 0, w5, S.prev_amask, w7, w8, w9, S.total_ke, S.total_imp, S.curr_dt) : 0;
                    const int iters_left = B.step_limit + 1 - sim_steps;      // iterations of the loop still to come; the last one leaves
                    if (hit) {
                        const int period = cyc_lam;
                        const int skip = ((iters_left - 1) / period) * period;
                        if (skip > 0) {
                            sim_steps += skip; total_sub += (unsigned)skip; S.stamp += (unsigned)skip;
                            S.costp += (unsigned)skip * ((S.costp - cyc_costp) / (unsigned)period);   // work proxy (dispatch heuristics only)
                            bd_cycle_skip(L.mvs, L.ctl, P.nkin, S.stamp, skip, B.target_speed, B.ctrl_dt, Q.straggler);
                        }
                        cyc_have = false; cyc_pow = 1; cyc_lam = 0;
                    } else {
                        if (!cyc_have || cyc_lam == cyc_pow) {   // Brent: the snapshot moves to this sim step, the search window doubles
                            bd_cycle_visit(L.snap, L.sp, L.sv, L.sw, L.sb, L.slot_of, L.ctl, E.rot, E.ang, P.nkin, 1, w5, S.prev_amask, w7, w8, w9, S.total_ke, S.total_imp, S.curr_dt);
                            if (cyc_have) cyc_pow *= 2;
                            cyc_have = true; cyc_lam = 0; cyc_costp = S.costp;
                        }
                        cyc_lam++;
                    }
                }
            }
        } else if (phase == PH_VEL) {
            kcount++;
            if (S.robot_hit || kcount >= P.steps) {
                { const double rd = bd_dist2(sf[1], sf[2], L.sp[0].x, L.sp[0].y); lds_sync(); if (lane == 0) L.ctl[3] = rd; lds_sync(); }
                phase = after_move; sim_steps = 0; kcount = 0;
            }
        } else if (phase == PH_FIXED) {
            kcount++;
            if (kcount >= P.steps) phase = PH_DONE;
        } else { // PH_STILL
            sim_steps++;
            if (!still_done && sim_steps > B.step_limit) hit_limit = true;
            if (still_done || sim_steps > B.step_limit) phase = PH_DONE;
        }
        if (B.pass == 0 && B.budget > 0 && (int)total_sub >= B.budget && phase != PH_DONE) break;   // out of budget: pass 1 carries on from here
    }
    const bool unfinished = phase != PH_DONE;
    __syncthreads();
    store_state(P, D, L, A, env);
    if (B.pass == 0 && lane == 0) Q.unfin[env] = unfinished ? 1 : 0;
    if (unfinished) {
        int *ri = Q.rs_i + (size_t)env * 16;
        double *rd = Q.rs_d + (size_t)env * 132;
        if (lane == 0) {
            ri[0] = phase; ri[1] = wi; ri[2] = path0; ri[3] = done_turning; ri[4] = dp_valid; ri[5] = sp_one; ri[6] = sim_steps; ri[7] = kcount;
            ri[8] = have_prev; ri[9] = still_done; ri[10] = (int)total_sub; ri[11] = S.robot_hit;
            ri[12] = (int)((__builtin_amdgcn_s_memtime() - t_begin) >> 8); ri[13] = hit_limit;
            D.e_stamp[env] = S.stamp; D.e_currdt[env] = S.curr_dt;
            if (Q.straggler != nullptr) atomicAdd(&Q.straggler[0], 1u);
        }
        if (lane < 4) rd[lane] = L.ctl[lane];
        rd[4 + 2 * lane] = prevp.x; rd[4 + 2 * lane + 1] = prevp.y;
        const int err_u = (ballot((S.err & BP_ERR_ADJ_OVERFLOW) != 0) ? BP_ERR_ADJ_OVERFLOW : 0) | (ballot((S.err & BP_ERR_ARB_OVERFLOW) != 0) ? BP_ERR_ARB_OVERFLOW : 0) |
                          (ballot((S.err & BP_ERR_LEVEL_OVERFLOW) != 0) ? BP_ERR_LEVEL_OVERFLOW : 0);
        if (lane == 0 && err_u) atomicOr(&D.e_err[env], err_u);
        return;
    }
#ifdef BP_PROF
    if (D.prof != nullptr) {
        if (lane == 0) { L.prof[23] = __builtin_amdgcn_s_memtime() - _t_kernel0; L.prof[22] = total_sub; L.prof[15] = _t_ctrl; }
        lds_sync();
        if (lane < BP_PROFN) D.prof[(size_t)env * BP_PROFN + lane] = L.prof[lane];
    }
#endif
    const int err_any = (ballot((S.err & BP_ERR_ADJ_OVERFLOW) != 0) ? BP_ERR_ADJ_OVERFLOW : 0) |
                        (ballot((S.err & BP_ERR_ARB_OVERFLOW) != 0) ? BP_ERR_ARB_OVERFLOW : 0) |
                        (ballot((S.err & BP_ERR_LEVEL_OVERFLOW) != 0) ? BP_ERR_LEVEL_OVERFLOW : 0);
    if (lane == 0) {
        D.e_stamp[env] = S.stamp; D.e_currdt[env] = S.curr_dt;
        D.e_cost[env] = (unsigned)((__builtin_amdgcn_s_memtime() - t_begin) >> 8) + cycles_before;
        if (err_any) atomicOr(&D.e_err[env], err_any);
        if (Q.straggler != nullptr && hit_limit) atomicAdd(&Q.straggler[1], 1u);
        double *o = Q.stepf + (size_t)env * 8;
        o[0] = L.ctl[3]; o[4] = (double)S.robot_hit; o[5] = (double)total_sub;
    }
}
__global__ __launch_bounds__(64, BP_BD_WAVES) void k_bd_physics(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q)
{
    bd_physics_body<false>(P, D, B, Q);
}
// the second pass of the two-pass step: the same body with the recurrence test of execute_robot_path compiled in
__global__ __launch_bounds__(64, BP_BD_WAVES) void k_bd_physics_resume(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q)
{
    bd_physics_body<false, true>(P, D, B, Q);
}
__global__ __launch_bounds__(64, BP_BD_WAVES) void k_bd_physics_damp(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q)
{
    bd_physics_body<true>(P, D, B, Q);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bd_finish: box distances, rewards, receptacle test + removal, work, termination, robot spfa map.
// init != 0: episode start (after the settle): distances, prev_boxes, counters and the robot map only.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bd_finish(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q, const int init,
                                                  const int tmpl, double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                  unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    const int env = tmpl ? P.num_envs + (int)blockIdx.x : (int)blockIdx.x;
    if (!tmpl && !init && B.sel_want >= 0 && (int)Q.unfin[env] != B.sel_want) return;   // two-pass step: the other group's env
    const int lane = lane_id();
    BdLds L;
    bd_carve(B, (char *)bp_smem, L);
    const int trial = D.e_trial[env];
    const int map = Q.map_of_trial[trial];
    bd_load_bits(B, Q, L, map);
    EnvCtx E;
    E.nb = D.e_nb[env];
    env_ctx(P, D, env, trial, E);
    const size_t eb = (size_t)env * P.nbcap;
    float *dist = Q.dist + (size_t)env * B.SH * B.SW;
    unsigned char *alive = Q.alive + (size_t)env * BD_MAXBOX, *order = Q.order + (size_t)env * BD_MAXBOX;
    double *boxdist = Q.boxdist + (size_t)env * BD_MAXBOX;
    d2 *prev = Q.prev + (size_t)env * BD_MAXBOX * 4;
    int err = 0;
    if (init) {
        if (lane < BD_MAXBOX) { alive[lane] = lane < B.nbox ? 1 : 0; order[lane] = (unsigned char)lane; }
        if (lane == 0) {
            Q.nalive[env] = B.nbox; Q.nprev[env] = B.nbox;
            Q.cum[env * 4 + 0] = 0.0; Q.cum[env * 4 + 1] = 0.0; Q.cnt[env * 4 + 0] = 0; Q.cnt[env * 4 + 1] = 0;
            D.e_total_work[env] = 0.0;
        }
        __syncthreads();
    }
    int nalive = init ? B.nbox : Q.nalive[env];
    const double *sf = Q.stepf + (size_t)env * 8;
    double robot_reward = 0.0, boxes_distance = 0.0;
    int robot_boxes = 0;
    unsigned long long remove_mask = 0ull; // by list position
    const d2 *rp = Q.recept_poly + (size_t)map * 4, *rn = Q.recept_n + (size_t)map * 4;
    double4 rbb;
    rbb.x = fmin(fmin(rp[0].x, rp[1].x), fmin(rp[2].x, rp[3].x)); rbb.z = fmax(fmax(rp[0].x, rp[1].x), fmax(rp[2].x, rp[3].x));
    rbb.y = fmin(fmin(rp[0].y, rp[1].y), fmin(rp[2].y, rp[3].y)); rbb.w = fmax(fmax(rp[0].y, rp[1].y), fmax(rp[2].y, rp[3].y));
    d2 *boxpos = Q.boxpos + (size_t)env * BD_MAXBOX;
    for (int q = 0; q < nalive; q++) {
        const int k = order[q], body = B.first_box + k;
        const d2 p = gE(E.pxy, body);
        // shortest_path_distance is a pure function of (position, layout): a box that did not move keeps its distance
        const d2 lastp = boxpos[k];
        const bool same = !init && lastp.x == p.x && lastp.y == p.y;
        const double fin = same ? boxdist[k] : bd_path_distance(B, Q, L, map, dist, p.x, p.y, B.recept_x, B.recept_y, err);
        if (!init) {
            double moved = boxdist[k] - fin;
            boxes_distance += __builtin_fabs(moved);
            if (B.use_correct_direction_reward && moved > 0) moved *= B.correct_direction_reward_scale;
            robot_reward += B.partial_rewards_scale * moved;
            bool inside = true;
            for (int vi = 0; vi < gE(E.nv, body); vi++) inside = inside && bd_point_in_shape(rp, rn, 4, 0.0, rbb, bd_local_to_world(E, body, vi));
            if (inside) { remove_mask |= 1ull << q; robot_boxes += 1; robot_reward += B.goal_reward; }
        }
        __syncthreads();
        if (lane == 0) { boxdist[k] = fin; boxpos[k] = p; }
    }
    int inactivity = init ? 0 : Q.cnt[env * 4 + 0];
    if (remove_mask) {
        inactivity = 0;
        // space.remove(box.body, box): park the slot (empty AABB, zero velocity) and drop its cached arbiters
        for (int q = 0; q < nalive; q++) {
            if (!((remove_mask >> q) & 1ull)) continue;
            const int k = order[q], body = B.first_box + k;
            if (lane == 0) {
                alive[k] = 0;
                double4 e4; e4.x = BP_INF; e4.y = BP_INF; e4.z = -BP_INF; e4.w = -BP_INF;
                D.bb[eb + body] = e4; D.fat[eb + body] = e4;
                D.velv[eb + body] = mk2(0.0, 0.0); D.velw[eb + body] = mk2(0.0, 0.0); D.velb[eb + body] = mk2(0.0, 0.0);
            }
            const size_t ab = (size_t)env * BP_ACAP + lane;
            const unsigned key = D.a_key[ab];
            if (key != ARB_FREE_KEY && ((int)(key >> 16) == body || (int)(key & 0xFFFFu) == body)) D.a_key[ab] = ARB_FREE_KEY;
        }
        __syncthreads();
        if (lane == 0) {
            int w = 0;
            for (int q = 0; q < nalive; q++) if (!((remove_mask >> q) & 1ull)) order[w++] = order[q];
        }
        nalive -= __popcll(remove_mask);
        __syncthreads();
    }
    // the robot's spfa map (observation channel 2) is k_bd_robot_map's, launched after this kernel
    const unsigned long long em = ballot(err != 0);
    if (init) {
        for (int q = lane; q < B.nbox * 4; q += 64) prev[q] = bd_local_to_world(E, B.first_box + q / 4, q & 3);
        if (lane == 0 && em) atomicOr(&D.e_err[env], BP_ERR_ARB_OVERFLOW);
        return;
    }
    // ---- rewards / stats (box_delivery_env.py:769-823) ----
    const int hit = (int)sf[4];
    const double robot_distance = sf[0];
    if (hit) robot_reward -= B.collision_penalty;
    const double rh = bd_restrict(D.ang[eb]);
    const double turn_angle = bd_hdiff(sf[3], rh);
    if (robot_distance < 0.05 && __builtin_fabs(turn_angle) < 0.05 * (BP_PI / 180.0)) robot_reward -= B.non_movement_penalty;
    // work: total_work_done(prev_boxes, updated_boxes) zips the lists by position (metrics.py:96-113)
    double work = 0.0;
    {
        const int nprev = Q.nprev[env];
        const int n = nprev < nalive ? nprev : nalive;
        double contrib = 0.0;
        d2 nowv[4];
        if (lane < nalive) {
            const int body = B.first_box + order[lane];
            for (int i = 0; i < 4; i++) nowv[i] = bd_local_to_world(E, body, i);
            if (lane < n) {
                d2 pv[4];
                for (int i = 0; i < 4; i++) pv[i] = prev[lane * 4 + i];
                const double area = poly_area_seq(pv, 4);
                const d2 ca = poly_centroid_seq(pv, 4), cb = poly_centroid_seq(nowv, 4);
                contrib = __builtin_sqrt((ca.x - cb.x) * (ca.x - cb.x) + (ca.y - cb.y) * (ca.y - cb.y)) * area;
            }
        }
        for (int q = 0; q < n; q++) work += __shfl(contrib, q);
        __syncthreads();
        if (lane < nalive) for (int i = 0; i < 4; i++) prev[lane * 4 + i] = nowv[i];
    }
    if (lane == 0) {
        const double cumd = Q.cum[env * 4 + 0] + robot_distance, cumr = Q.cum[env * 4 + 1] + robot_reward;
        const int cumb = Q.cnt[env * 4 + 1] + robot_boxes;
        const double tw = D.e_total_work[env] + work;
        if (robot_boxes == 0) inactivity += 1;
        int term = 0, trunc = 0;
        if (cumb == B.num_boxes) term = 1;
        if (inactivity >= B.inactivity_cutoff) { term = 1; trunc = 1; }
        Q.cum[env * 4 + 0] = cumd; Q.cum[env * 4 + 1] = cumr; Q.cnt[env * 4 + 0] = inactivity; Q.cnt[env * 4 + 1] = cumb;
        D.e_total_work[env] = tw;
        Q.nalive[env] = nalive; Q.nprev[env] = nalive;
        if (reward) reward[env] = robot_reward;
        if (terminated) terminated[env] = (unsigned char)term;
        if (truncated) truncated[env] = (unsigned char)trunc;
        if (info) {
            double *o = info + (size_t)env * BP_INFO_COUNT;
            const d2 rpos = D.pxy[eb];
            o[0] = rpos.x; o[1] = rpos.y; o[2] = D.ang[eb]; o[3] = cumd; o[4] = (double)cumb; o[5] = cumr; o[6] = tw;
            o[7] = robot_distance / B.ministep_size; o[8] = (double)inactivity; o[9] = (double)hit; o[10] = sf[5]; o[11] = robot_distance;
            o[12] = boxes_distance; o[13] = (double)Q.nwp[env]; o[14] = (double)nalive; o[15] = work;
        }
        if (em) atomicOr(&D.e_err[env], BP_ERR_ARB_OVERFLOW);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// area-clearing-v0 tail of step (area_clearing.py:695-778): completion test, rewards, work, robot spfa map
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ac_orient(const double (*p)[2], int n)
{
    double a = 0.0;
    for (int i = 0; i < n; i++) { const int j = (i + 1) % n; a += p[i][0] * p[j][1] - p[j][0] * p[i][1]; }
    return a > 0 ? 1 : -1;
}
// shapely Polygon.intersects for two convex polygons: no edge of either is a strictly separating line
__device__ __forceinline__ bool ac_sep_axis(const double (*a)[2], int na, const double (*b)[2], int nb)
{
    const int o = ac_orient(a, na);
    for (int i = 0; i < na; i++) {
        const int j = (i + 1) % na;
        bool all_out = true;
        for (int k = 0; k < nb && all_out; k++) {
            const double cr = (a[j][0] - a[i][0]) * (b[k][1] - a[i][1]) - (a[j][1] - a[i][1]) * (b[k][0] - a[i][0]);
            if (!(cr * o < 0)) all_out = false;
        }
        if (all_out) return true;
    }
    return false;
}
__device__ __forceinline__ bool ac_intersects(const double (*a)[2], int na, const double (*b)[2], int nb)
{
    return !(ac_sep_axis(a, na, b, nb) || ac_sep_axis(b, nb, a, na));
}

__global__ __launch_bounds__(64) void k_ac_finish(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q, const int init,
                                                  const int tmpl, double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                  unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    const int env = tmpl ? P.num_envs + (int)blockIdx.x : (int)blockIdx.x;
    if (!tmpl && !init && B.sel_want >= 0 && (int)Q.unfin[env] != B.sel_want) return;   // two-pass step: the other group's env
    const int lane = lane_id();
    BdLds L;
    bd_carve(B, (char *)bp_smem, L);
    const int trial = D.e_trial[env];
    const int map = Q.map_of_trial[trial];
    bd_load_bits(B, Q, L, map);
    EnvCtx E;
    E.nb = D.e_nb[env];
    env_ctx(P, D, env, trial, E);
    const size_t eb = (size_t)env * P.nbcap;
    float *dist = Q.dist + (size_t)env * B.SH * B.SW;
    d2 *prev = Q.prev + (size_t)env * BD_MAXBOX * 4;
    unsigned char *cleared = Q.cleared + (size_t)env * BD_MAXBOX;
    double *sf = Q.stepf + (size_t)env * 8;
    int err = 0;
    // lane k < nbox: box k (the list never shrinks in this task)
    double nowv[4][2], prv[4][2];
    const bool act = lane < B.nbox;
    if (act) {
        for (int i = 0; i < 4; i++) {
            const d2 w = bd_local_to_world(E, B.first_box + lane, i);
            nowv[i][0] = w.x; nowv[i][1] = w.y;
            const d2 pv = prev[lane * 4 + i];
            prv[i][0] = pv.x; prv[i][1] = pv.y;
        }
    }
    double rwd = 0.0, work = 0.0, diff_reward = 0.0, box_reward = 0.0, pushing_reward = 0.0;
    int num_completed = 0, term = 0, trunc = 0, tcount = init ? 0 : Q.cnt[env * 4 + 3];
    const int hit = init ? 0 : (int)sf[4];
    if (!init) {
        const bool inter = act && ac_intersects(B.bd_poly, B.nbd, nowv, 4);
        num_completed = __popcll(ballot(act && !inter));
        double contrib_diff = 0.0, contrib_work = 0.0;
        if (act) {
            d2 pa[4], pb[4];
            for (int i = 0; i < 4; i++) { pa[i] = mk2(prv[i][0], prv[i][1]); pb[i] = mk2(nowv[i][0], nowv[i][1]); }
            const d2 ca = poly_centroid_seq(pa, 4), cb = poly_centroid_seq(pb, 4);
            if (ac_intersects(B.bd_poly, B.nbd, prv, 4)) {   // obs_to_goal_difference (metrics.py:73-94)
                double min_a = BP_INF, min_b = BP_INF;
                for (int g = 0; g < B.ngoal; g++) {
                    const d2 gp = Q.goals[g];
                    const double da = bd_dist2(ca.x, ca.y, gp.x, gp.y), db = bd_dist2(cb.x, cb.y, gp.x, gp.y);
                    if (da < min_a) min_a = da;
                    if (db < min_b) min_b = db;
                }
                contrib_diff = min_a - min_b;
            }
            const double area = poly_area_seq(pa, 4);
            contrib_work = __builtin_sqrt((ca.x - cb.x) * (ca.x - cb.x) + (ca.y - cb.y) * (ca.y - cb.y)) * area;
        }
        const unsigned long long dm = ballot(act && ac_intersects(B.bd_poly, B.nbd, prv, 4));
        for (int k = 0; k < B.nbox; k++) {    // python loop order; boxes outside the boundary are skipped (not added as 0)
            if ((dm >> k) & 1ull) diff_reward += __shfl(contrib_diff, k);
            work += __shfl(contrib_work, k);
        }
        pushing_reward = diff_reward * B.pushing_mult;
        const int cleared_count = Q.cnt[env * 4 + 2];
        tcount += 1; // self.t += 1 at the top of step()
        const double dcount = __builtin_fabs((double)(num_completed - cleared_count));
        if (num_completed > cleared_count) { box_reward = dcount * B.box_cleared_reward; tcount = 0; }
        else box_reward = dcount * B.box_putback_penalty;
        const double collision_penalty = hit ? B.boundary_penalty : 0;
        const double nonmovement_penalty = 0;
        term = num_completed == B.nbox;
        rwd = box_reward + collision_penalty + pushing_reward + nonmovement_penalty;
        trunc = tcount >= B.t_max;
        if (trunc) rwd += B.truncation_penalty;
        else if (term) rwd += B.terminal_reward;
        __syncthreads();
        if (act) cleared[lane] = inter ? 0 : 1;
    } else if (lane < BD_MAXBOX) {
        cleared[lane] = 0;
        Q.alive[(size_t)env * BD_MAXBOX + lane] = lane < B.nbox ? 1 : 0;
        Q.order[(size_t)env * BD_MAXBOX + lane] = (unsigned char)lane;
    }
    if (act) for (int i = 0; i < 4; i++) prev[lane * 4 + i] = mk2(nowv[i][0], nowv[i][1]);
    // the robot's spfa map (channel 2) is k_bd_robot_map's, launched after this kernel
    const unsigned long long em = ballot(err != 0);
    if (lane == 0) {
        if (em) atomicOr(&D.e_err[env], BP_ERR_ARB_OVERFLOW);
        if (init) {
            Q.nalive[env] = B.nbox; Q.nprev[env] = B.nbox;
            for (int q = 0; q < 4; q++) { Q.cum[env * 4 + q] = 0.0; Q.cnt[env * 4 + q] = 0; }
            D.e_total_work[env] = 0.0;
            sf[4] = 0.0;
        } else {
            const double tw = D.e_total_work[env] + work;
            D.e_total_work[env] = tw;
            Q.cnt[env * 4 + 2] = num_completed; Q.cnt[env * 4 + 3] = tcount;
            sf[4] = 0.0; // self.robot_hit_obstacle = False at the end of step (area_clearing.py:776)
            if (reward) reward[env] = rwd;
            if (terminated) terminated[env] = (unsigned char)term;
            if (truncated) truncated[env] = (unsigned char)trunc;
            if (info) {
                double *o = info + (size_t)env * BP_INFO_COUNT;
                const d2 rpos = D.pxy[eb];
                o[0] = rpos.x; o[1] = rpos.y; o[2] = D.ang[eb]; o[3] = tw; o[4] = -work; o[5] = diff_reward; o[6] = box_reward;
                o[7] = (double)num_completed; o[8] = (B.action_type == 2) ? 1.0 : sf[0] / 2.5; o[9] = (double)hit; o[10] = sf[5]; o[11] = sf[0];
                o[12] = (double)tcount; o[13] = (double)Q.nwp[env]; o[14] = work; o[15] = pushing_reward;
            }
        }
    }
}

// reset(): box-delivery extras of the settled template -> env (k_reset_copy moves the physics state)
// ---------------------------------------------------------------------------------------------------------------------
// k_bd_robot_map: the spfa map from the robot for observation channel 2 (create_global_shortest_path_map, box_delivery_env.py:1131-1138 with the channel
// scale, area_clearing.py:1046-1053 without), BDR_THREADS threads per env.  The same bucketed Dijkstra as bd_spfa -- the labels are the least fixed point of
// the float32 relaxations, so the order in which a bucket's cells are relaxed does not matter -- with the cells of a bucket spread over four wavefronts;
// the tentative distances live in the output array itself (Q.rmap) and are scaled in place afterwards.  Only the free-cell raster and the queues need LDS
// (26 KB), so six workgroups = 24 wavefronts share a CU, against four single-wave workgroups when this ran at the end of k_bd_finish / k_ac_finish.
// ---------------------------------------------------------------------------------------------------------------------
#define BDR_THREADS 256
__global__ __launch_bounds__(BDR_THREADS) void k_bd_robot_map(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q, const int tmpl)
{
    const int env = tmpl ? P.num_envs + (int)blockIdx.x : (int)blockIdx.x;
    if (!tmpl && B.sel_want >= 0 && (int)Q.unfin[env] != B.sel_want) return;   // two-pass step: the other group's env
    const int tid = threadIdx.x;
    const int NW = B.SH * B.SW, words = (NW + 31) / 32;
    unsigned *freeb = (unsigned *)bp_smem;                               // [words]
    unsigned short *q = (unsigned short *)(freeb + ((words + 3) & ~3));  // [3][BD_QCAP]
    __shared__ int qn[4];
    __shared__ int s_err;
    const int map = Q.map_of_trial[D.e_trial[env]];
    const unsigned *f = Q.free_bits + (size_t)map * words;
    for (int i = tid; i < words; i += BDR_THREADS) freeb[i] = f[i];
    float *dist = Q.rmap + (size_t)env * NW;
    unsigned *du = (unsigned *)dist;
    for (int i = tid; i < NW; i += BDR_THREADS) du[i] = BD_INF_BITS;
    if (tid < 3) qn[tid] = 0;
    if (tid == 0) s_err = 0;
    const d2 rpos = D.pxy[(size_t)env * P.nbcap];
    int wi, wj;
    bd_pos_to_win(B, rpos.x, rpos.y, wi, wj);
    const unsigned short *edt = Q.edt + ((size_t)map * NW + (size_t)wi * B.SW + wj) * 2;
    const int src = (int)edt[0] * B.SW + (int)edt[1];
    __syncthreads();
    if (tid == 0) { du[src] = 0u; q[0] = (unsigned short)src; qn[0] = 1; }
    __syncthreads();
    const float SQ2 = __builtin_sqrtf(2.0f);
    int empty_run = 0;
    for (int b = 0; b < 8192; b++) {
        const int qi = b % 3;
        const int n = min(qn[qi], BD_QCAP);
        // An empty bucket takes a barrier too: a wave that ran ahead into bucket b + 1 would push into qn[(b + 3) % 3] = qn[qi] while a slower wave has
        // not read it yet, and that wave would then take the other branch (buckets can be empty between non-empty ones: a diagonal-only passage
        // gives distances k * sqrt(2), which skip bucket 3).  After the barrier every wave has made the same decision from the same value.
        if (n == 0) { __syncthreads(); if (++empty_run >= 3) break; continue; }
        empty_run = 0;
        const unsigned short *qq = q + qi * BD_QCAP;
        for (int base = 0; base < n; base += BDR_THREADS) {
            const int idx = base + tid;
            bool act = idx < n;
            const int v = act ? (int)qq[idx] : 0;
            const float d = bd_ld(dist + v);
            act = act && ((int)d == b);
            const int vi = v / B.SW, vj = v - vi * B.SW;
            float old[8]; int nidx[8]; bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int ni = vi + BD_DI[k], nj = vj + BD_DJ[k];
                const bool inb = ni >= 0 && nj >= 0 && ni < B.SH && nj < B.SW;
                nidx[k] = inb ? ni * B.SW + nj : 0;
                ok[k] = act && inb && bd_bit(freeb, nidx[k]);
                old[k] = bd_ld(dist + nidx[k]);
            }
            unsigned prevb[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float nd = d + ((k & 1) ? 1.0f : SQ2);
                ok[k] = ok[k] && nd < old[k];
                prevb[k] = 0u;
                if (ok[k]) prevb[k] = atomicMin(du + nidx[k], __float_as_uint(nd));
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float nd = d + ((k & 1) ? 1.0f : SQ2);
                if (ok[k] && __float_as_uint(nd) < prevb[k]) {
                    const int tq = ((int)nd) % 3;
                    const int pos = atomicAdd(&qn[tq], 1);
                    if (pos < BD_QCAP) q[tq * BD_QCAP + pos] = (unsigned short)nidx[k];
                    else s_err = 1;
                }
            }
        }
        __syncthreads();
        if (tid == 0) qn[qi] = 0;
        __syncthreads();
    }
    __syncthreads();
    // distances -> channel values, in place (float32 arithmetic of the reference; unreached cells 0)
    const float ppm32 = (float)B.ppm, scale32 = (float)B.sp_channel_scale;
    const double div2 = (__builtin_sqrt(2.0) * (double)B.local_px) / B.ppm;
    for (int i = tid; i < NW; i += BDR_THREADS) {
        const float d = bd_ld(dist + i);
        float v = (__float_as_uint(d) == BD_INF_BITS) ? 0.0f : d;
        v = v / ppm32;
        v = (float)((double)v / div2);
        if (B.task == 0) v = v * scale32;
        dist[i] = v;
    }
    if (tid == 0 && s_err) atomicOr(&D.e_err[env], BP_ERR_ARB_OVERFLOW);
}

__global__ __launch_bounds__(256) void k_bd_reset_copy(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q,
                                                       const unsigned char *__restrict__ mask, double *__restrict__ info)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int trial = D.e_trial[env]; // already advanced by k_reset_copy
    const size_t se = (size_t)P.num_envs + trial;
    copy_span(Q.alive + (size_t)env * BD_MAXBOX, Q.alive + se * BD_MAXBOX, (size_t)BD_MAXBOX, tid, nt);
    copy_span(Q.order + (size_t)env * BD_MAXBOX, Q.order + se * BD_MAXBOX, (size_t)BD_MAXBOX, tid, nt);
    copy_span(Q.boxdist + (size_t)env * BD_MAXBOX, Q.boxdist + se * BD_MAXBOX, (size_t)BD_MAXBOX, tid, nt);
    copy_span(Q.boxpos + (size_t)env * BD_MAXBOX, Q.boxpos + se * BD_MAXBOX, (size_t)BD_MAXBOX, tid, nt);
    copy_span(Q.prev + (size_t)env * BD_MAXBOX * 4, Q.prev + se * BD_MAXBOX * 4, (size_t)BD_MAXBOX * 4, tid, nt);
    copy_span(Q.rmap + (size_t)env * B.SH * B.SW, Q.rmap + se * B.SH * B.SW, (size_t)B.SH * B.SW, tid, nt);
    if (Q.cleared != nullptr) copy_span(Q.cleared + (size_t)env * BD_MAXBOX, Q.cleared + se * BD_MAXBOX, (size_t)BD_MAXBOX, tid, nt);
    if (tid == 0) {
        Q.nalive[env] = Q.nalive[se]; Q.nprev[env] = Q.nprev[se];
        for (int q = 0; q < 4; q++) { Q.cum[env * 4 + q] = 0.0; Q.cnt[env * 4 + q] = 0; }
        Q.stepf[(size_t)env * 8 + 4] = 0.0;
        if (info) {
            double *o = info + (size_t)env * BP_INFO_COUNT;
            const size_t eb = (size_t)env * P.nbcap;
            for (int q = 0; q < BP_INFO_COUNT; q++) o[q] = 0.0;
            o[0] = D.pxy[eb].x; o[1] = D.pxy[eb].y; o[2] = D.ang[eb]; o[14] = (double)Q.nalive[se];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bd_observe: generate_observation (box_delivery_env.py:1045-1207), uint8 [lp][lp][4] channels-last, 256 threads per env.
// The overhead map is rasterised into an LDS byte image of the small-map window (class k stands for k/8: 0 wall/outside,
// 1 floor, 3 receptacle, 4 box, 6 robot; cv2.fillPoly = 8-connected outline + scanline fill, later polygons overwrite);
// every output pixel then evaluates scipy's order-0 rotate (nearest sample of the axis-aligned crop) for all four channels.
// ---------------------------------------------------------------------------------------------------------------------
#define BDO_THREADS 256
__device__ __forceinline__ void bd_fill_poly_lds(unsigned char *img, int H, int W, const long long *px, const long long *py, int n,
                                                 unsigned char code, int tid, int nt)
{
    for (int e = 0; e < n; e++) { // outline: edge (v[e-1], v[e])
        const int j = (e + n - 1) % n;
        const LineSpec s = make_line(W, H, px[j], py[j], px[e], py[e]);
        if (!s.valid) continue;
        const long long dmaj = s.vert ? s.dy : s.dx, dmin = s.vert ? s.dx : s.dy;
        for (long long t = tid; t <= dmaj; t += nt) {
            const long long mt = dmaj == 0 ? 0 : (2 * dmin * t + dmaj - 1) / (2 * dmaj);
            const long long x = s.vert ? s.x0 + mt : s.x0 + t, y = s.vert ? s.y0 + s.sy * t : s.y0 + s.sy * mt;
            if (x >= 0 && x < W && y >= 0 && y < H) img[y * W + x] = code;
        }
    }
    long long ymin = py[0], ymax = py[0];
    for (int i = 1; i < n; i++) { ymin = py[i] < ymin ? py[i] : ymin; ymax = py[i] > ymax ? py[i] : ymax; }
    if (ymax > H) ymax = H;
    for (long long y = ymin + tid; y < ymax; y += nt) {
        if (y < 0) continue;
        long long xs[4]; int cnt = 0;
        for (int i = 0; i < n && cnt < 4; i++) {
            const int j = (i + n - 1) % n;
            const long long x0 = px[j] << 16, x1 = px[i] << 16, y0 = py[j], y1 = py[i];
            if (y0 == y1) continue;
            const long long edx = (x1 - x0) / (y1 - y0);
            long long ex, ey0, ey1;
            if (y0 < y1) { ey0 = y0; ey1 = y1; ex = x0; } else { ey0 = y1; ey1 = y0; ex = x1; }
            if (y < ey0 || y >= ey1) continue;
            xs[cnt++] = ex + (y - ey0) * edx;
        }
        for (int a = 1; a < cnt; a++) { const long long v = xs[a]; int b = a - 1; while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; } xs[b + 1] = v; }
        for (int a = 0; a + 1 < cnt; a += 2) {
            long long xl = (xs[a] + 65535) >> 16, xr = xs[a + 1] >> 16;
            if (xl < W && xr >= 0) {
                if (xl < 0) xl = 0;
                if (xr >= W) xr = W - 1;
                for (long long x = xl; x <= xr; x++) img[y * W + x] = code;
            }
        }
    }
}

__global__ __launch_bounds__(BDO_THREADS) void k_bd_observe(const DevParams P, const DevPtrs D, const BdParams B, const BdPtrs Q,
                                                            const unsigned char *__restrict__ mask, unsigned char *__restrict__ obs)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    if (mask == nullptr && B.sel_want >= 0 && (int)Q.unfin[env] != B.sel_want) return;   // two-pass step: the other group's env
    const int tid = threadIdx.x;
    unsigned char *img = (unsigned char *)bp_smem;
    __shared__ long long spx[4], spy[4];
    __shared__ float red2[BDO_THREADS / 64], red3[BDO_THREADS / 64];
    const int trial = D.e_trial[env];
    const int map = Q.map_of_trial[trial];
    EnvCtx E;
    E.nb = D.e_nb[env];
    env_ctx(P, D, env, trial, E);
    const int NW = B.SH * B.SW;
    const unsigned char *sfree = Q.small_free + (size_t)map * NW;
    for (int i = tid; i < NW; i += BDO_THREADS) img[i] = sfree[i] ? 1 : 0;
    __syncthreads();
    const int off = (int)(B.local_w * B.ppm / 2) + 10;
    const int nalive = Q.nalive[env];
    const unsigned char *order = Q.order + (size_t)env * BD_MAXBOX;
    const d2 *rp = Q.recept_poly + (size_t)map * 4;
    const int npoly = (B.task == 1) ? 4 + B.nbox + 1 : 1 + nalive + 1;
    for (int pidx = 0; pidx < npoly; pidx++) {
        unsigned char code;
        if (B.task == 1) {
            // area-clearing (area_clearing.py:968-1026): 4 goal-area rectangles, every box (cleared ones in another class), robot footprint
            code = pidx < 4 ? 3 : (pidx < 4 + B.nbox ? (Q.cleared[(size_t)env * BD_MAXBOX + (pidx - 4)] ? 7 : 4) : 5);
            if (tid < 4) {
                d2 w;
                if (pidx < 4) {
                    const double il = __builtin_fabs(B.bd_poly[0][0]) * 2, iw = __builtin_fabs(B.bd_poly[0][1]) * 2;
                    const double th = __builtin_fabs(B.ob_poly[0][0]) - __builtin_fabs(B.bd_poly[0][0]);
                    double x, y, l, wd;
                    if (pidx == 0) { x = -il / 2 - th / 2; y = 0; l = th; wd = iw; }
                    else if (pidx == 1) { x = il / 2 + th / 2; y = 0; l = th; wd = iw; }
                    else if (pidx == 2) { x = 0; y = -iw / 2 - th / 2; l = il + 2 * th; wd = th; }
                    else { x = 0; y = iw / 2 + th / 2; l = il + 2 * th; wd = th; }
                    w = mk2((tid == 0 || tid == 3) ? x - l / 2 : x + l / 2, (tid < 2) ? y - wd / 2 : y + wd / 2);
                } else if (pidx < 4 + B.nbox) w = bd_local_to_world(E, B.first_box + (pidx - 4), tid);
                else {
                    const d2 p = gE(E.pxy, 0), r = gE(E.rot, 0);
                    const double fx = B.footprint[tid][0], fy = B.footprint[tid][1];
                    w = mk2((r.x * fx + (-r.y) * fy) + p.x, (r.y * fx + r.x * fy) + p.y);
                }
                const double vx = w.x * B.ppm, vy = w.y * B.ppm;
                long long ixp = (long long)(int)vx, iyp = (long long)(int)vy;
                ixp += off; iyp += off;
                iyp = B.SH - iyp;
                spx[tid] = ixp; spy[tid] = iyp;
            }
        } else {
            code = pidx == 0 ? 3 : (pidx <= nalive ? 4 : 6);
            if (tid < 4) {
                d2 w;
                if (pidx == 0) w = rp[tid];
                else if (pidx <= nalive) w = bd_local_to_world(E, B.first_box + order[pidx - 1], tid);
                else w = bd_local_to_world(E, 0, tid);
                const double vx = w.x * B.ppm, vy = w.y * B.ppm;
                long long ixp = (long long)(int)vx, iyp = (long long)(int)vy; // astype(np.int32)
                ixp += off; iyp += off;
                iyp = B.SH - iyp;
                spx[tid] = ixp; spy[tid] = iyp;
            }
        }
        __syncthreads();
        long long px[4], py[4];
        for (int i = 0; i < 4; i++) { px[i] = spx[i]; py[i] = spy[i]; }
        bd_fill_poly_lds(img, B.SH, B.SW, px, py, 4, code, tid, BDO_THREADS);
        __syncthreads();
    }
    // ---- get_local_map geometry (box_delivery_env.py:1078-1096) ----
    const size_t eb = (size_t)env * P.nbcap;
    const double rx = D.pxy[eb].x, ry = D.pxy[eb].y, rh = D.ang[eb];
    const int lp = B.local_px;
    const int cwid_full = (int)(__builtin_ceil(((double)lp * __builtin_sqrt(2.0)) / 2) * 2);
    const int pi_ = (int)__builtin_floor(-ry * B.ppm + (double)B.H / 2), pj_ = (int)__builtin_floor(rx * B.ppm + (double)B.W / 2);
    int i0 = pi_ - cwid_full / 2, i1 = pi_ + cwid_full / 2, j0 = pj_ - cwid_full / 2, j1 = pj_ + cwid_full / 2;
    i0 = i0 < 0 ? 0 : i0; j0 = j0 < 0 ? 0 : j0; i1 = i1 > B.H ? B.H : i1; j1 = j1 > B.W ? B.W : j1;
    int ch = i1 - i0, cw = j1 - j0;
    ch = ch < 0 ? 0 : ch; cw = cw < 0 ? 0 : cw;
    double s, c;
    bp_sincos(BP_PI / 2 - rh, s, c);
    int oh, ow;
    {
        const double b0[4] = {c * 0 + s * 0, c * 0 + s * cw, c * ch + s * 0, c * ch + s * cw};
        const double b1[4] = {-s * 0 + c * 0, -s * 0 + c * cw, -s * ch + c * 0, -s * ch + c * cw};
        double mn0 = b0[0], mx0 = b0[0], mn1 = b1[0], mx1 = b1[0];
        for (int k = 1; k < 4; k++) { mn0 = b0[k] < mn0 ? b0[k] : mn0; mx0 = b0[k] > mx0 ? b0[k] : mx0; mn1 = b1[k] < mn1 ? b1[k] : mn1; mx1 = b1[k] > mx1 ? b1[k] : mx1; }
        oh = (int)((mx0 - mn0) + 0.5); ow = (int)((mx1 - mn1) + 0.5);
    }
    const int a0 = oh / 2 - lp / 2, b0_ = ow / 2 - lp / 2;
    const double oc0 = ((double)oh - 1) / 2, oc1 = ((double)ow - 1) / 2, ic0 = ((double)ch - 1) / 2, ic1 = ((double)cw - 1) / 2;
    const double off0 = ic0 - (c * oc0 + s * oc1), off1 = ic1 - (-s * oc0 + c * oc1);
    const float *rmap = Q.rmap + (size_t)env * NW, *rcp = Q.recept + (size_t)map * NW;
    float mn2 = BP_INF, mn3 = BP_INF;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            for (int o = 32; o >= 1; o >>= 1) { mn2 = fminf(mn2, __shfl_xor(mn2, o)); mn3 = fminf(mn3, __shfl_xor(mn3, o)); }
            if ((tid & 63) == 0) { red2[tid >> 6] = mn2; red3[tid >> 6] = mn3; }
            __syncthreads();
            mn2 = red2[0]; mn3 = red3[0];
            for (int k = 1; k < BDO_THREADS / 64; k++) { mn2 = fminf(mn2, red2[k]); mn3 = fminf(mn3, red3[k]); }
        }
        for (int pix = tid; pix < lp * lp; pix += BDO_THREADS) {
            const int i = pix / lp, j = pix - i * lp;
            const int oi = a0 + i, oj = b0_ + j;
            int code = 0;
            float v2 = 0.0f, v3 = 0.0f;
            bool sampled = false, inwin = false;
            int wi_ = 0, wj_ = 0;
            if (oi >= 0 && oi < oh && oj >= 0 && oj < ow) {
                double c0 = 0.0, c1 = 0.0;
                c0 += (double)oi * c; c0 += (double)oj * s; c0 += off0;
                c1 += (double)oi * -s; c1 += (double)oj * c; c1 += off1;
                if (!(c0 < 0 || c0 > ch - 1 || c1 < 0 || c1 > cw - 1)) {
                    const long long s0 = (long long)__builtin_floor(c0 + 0.5), s1 = (long long)__builtin_floor(c1 + 0.5);
                    const int wi = (int)(i0 + s0) - B.si0, wj = (int)(j0 + s1) - B.sj0;
                    sampled = true; wi_ = wi; wj_ = wj;
                    if (wi >= 0 && wi < B.SH && wj >= 0 && wj < B.SW) {
                        const int w = wi * B.SW + wj;
                        inwin = true;
                        code = img[w]; v2 = rmap[w]; v3 = rcp[w];
                    }
                }
            }
            if (sampled && !inwin) { // the static map outside the small-map window (still inside the padded room)
                v3 = B.recept_outside;
                if (B.task == 1) {
                    const int gi = wi_ + B.si0 - B.si0, gj = wj_ + B.sj0 - B.sj0; // window coordinates of the sample
                    const int dy = gi < 0 ? -gi : (gi > B.SH - 1 ? gi - (B.SH - 1) : 0), dx = gj < 0 ? -gj : (gj > B.SW - 1 ? gj - (B.SW - 1) : 0);
                    if (dx * dx + dy * dy <= B.out_r * B.out_r) v3 = 2.0f;
                }
            }
            if (pass == 0) { mn2 = fminf(mn2, v2); mn3 = fminf(mn3, v3); }
            else {
                const unsigned c0u = code == 1 ? 31u : code == 3 ? 95u : code == 4 ? 127u : code == 5 ? 159u : code == 6 ? 191u : code == 7 ? 223u : 0u;
                const unsigned c1u = Q.robot_chan[pix];
                const unsigned c2u = (unsigned)((int)((v2 - mn2) * 255.0f) & 0xFF); // numpy astype(uint8): truncate, wrap mod 256
                const unsigned c3u = (unsigned)((int)((v3 - mn3) * 255.0f) & 0xFF);
                ((unsigned *)obs)[(size_t)env * lp * lp + pix] = c0u | (c1u << 8) | (c2u << 16) | (c3u << 24);
            }
        }
    }
}
