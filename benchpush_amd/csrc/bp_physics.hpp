// substep(): one pymunk.Space.step (Chipmunk2D 7.0.3 cpSpaceStep; reference call sites ship_ice_env.py:219,281,
// maze_NAMO_env.py:268,415) for one environment, executed wave-synchronously by the 64 lanes of its wavefront:
//   1  position integrate + world vertices + AABB of the bodies that move      (cpBodyUpdatePosition, cpPolyShapeCacheData)
//   2  Verlet neighbour-list refresh for bodies that left their fat AABB          (any exact broadphase == cpBBTree)
//   3  candidate pairs of moving bodies -> AABB test -> cached-plane early out    (cpSpaceCollideShapes / QueryReject)
//   4a plane separations of the surviving pairs, one (pair, plane) item per lane  (cpCollide: GJK/EPA closest features)
//   4b normal + Chipmunk support edges + ContactPoints, one pair per lane
//   4c manifolds handed to the arbiter slots (one per lane): impulse carry-over    (cpArbiterUpdate, cpSpaceArbiterSetFilter)
//   5  prestep; warm set; velocity integrate (damping 0); warm start; sequential impulses colour by colour so that the
//      result equals the sequential sweep in ascending (colour, shapeA, shapeB) order; exact fixed-point early exit
//   6  agent x obstacle bookkeeping (ship_ice_env.py:155-173), agent rules, next moving list
// Bodies that do not move and arbiters none of whose bodies moved produce exactly the results of the previous
// sub-step, so they are carried over instead of recomputed (exactness argument in DESIGN.md section 4).
#pragma once
#include "bp_device.hpp"
#include <type_traits>

// Optional in-kernel phase timers (diagnostic build only: -DBP_PROF).  Stamps go to D.prof, which nothing else reads.
#ifdef BP_PROF
#define PROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime();
#define PROF_ACC(slot) { unsigned long long _n = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) L.prof[slot] += _n - _pt; _pt = _n; }
#define PROF_CNT(slot, v) { const unsigned long long _v = (unsigned long long)(v); if (lane_id() == 0) L.prof[slot] += _v; }
#define PROF_MAX(slot, v) { const unsigned long long _v = (unsigned long long)(v); if (lane_id() == 0 && _v > L.prof[slot]) L.prof[slot] = _v; }
#else
#define PROF_DECL
#define PROF_ACC(slot)
#define PROF_CNT(slot, v)
#define PROF_MAX(slot, v)
#endif


// Wave-level branches that are rarely taken (neighbour-list refresh, first use of a velocity slot, new arbiter, recolouring, fixed-point exit,
// debug / quiescent paths): telling the compiler keeps them out of line.  Block placement matters here: marking the second contact of the solver
// pass *likely* costs 8 %, marking these unlikely gains 1.5 % (same-box A/B, tools/experiments/README.md).
#define BP_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define BP_UNLIKELY2(x) __builtin_expect(!!(x), 0)
// Diagnostic build (-DBP_DEBUG_PATHS, libbenchpush_hip_dbgpaths.so): BP_DEBUG_PATHS=<mask> in the environment forces the fallbacks that the fast paths of the
// narrow phase normally shadow (DevParams::dbg_paths).  In the product build the tests cost nothing: a run-time test of the mask measured -2 %.
// bp_debug_trace (per-sub-step poses of one env) lives in the same twin: the product kernels do not test the trace pointer every sub-step.
#ifdef BP_DEBUG_PATHS
#define BP_DBGP(bit) ((P.dbg_paths & (bit)) != 0)
#define BP_TRACE_ON(D) ((D).dbg != nullptr)
#else
#define BP_DBGP(bit) false
#define BP_TRACE_ON(D) false
#endif
struct ArbReg {
    unsigned key, stamp, h0, h1;
    int state, count, level, rank;
    double jn0, jt0, jn1, jt1;
    d2 n, r1_0, r2_0, r1_1, r2_1;
    double ma, ia, mb, ib;
    double e, u;               // elasticity / friction products of the two shapes (cpArbiterUpdate)
    int slotA, slotB;          // velocity slots of the two bodies
};

struct EnvCtx {
    int nb;
    const int *nv;
    const d2 *lv, *ln;
    const double4 *mass;
    const double4 *prop;
    const int *kind;
    d2 *pxy, *rot, *wv, *wn, *pv;
    double *ang;
    double4 *bb, *fat;
    unsigned short *adj;
    unsigned char *adjn;
    unsigned long long *hint;
};

// Element idx of a per-env array: the env's base pointer is wave-uniform (an SGPR pair) and the arrays of one env are far below 4 GB, so the byte offset is formed
// in 32 bits and the access takes the scalar-base form (global_load v, v_offset, s[base]) -- one VALU instruction per address instead of a sign extension and a
// 64-bit shift-and-add, and one VGPR per live address instead of two.  Indices are non-negative.
template <typename T> __device__ __forceinline__ T &gE(T *const base, const int idx) { return *(T *)((char *)base + (size_t)((unsigned)idx * (unsigned)sizeof(T))); }

struct LdsCtx {
    d2 *sv, *sw, *sb;          // [BP_NSLOT] (vx,vy) (w,w_bias) (vbx,vby) of the bodies that hold a velocity slot
    d2 *ag;                    // agent (body 0): (angle, -), (cos, sin), kept current by the integrate phase
    d2 *sp;                    // [BP_NSLOT] their positions (copy of E.pxy, kept current by whoever moves the body)
    unsigned char *slot_of;    // [nbcap] velocity slot of a body, 255 = none (velocity is exactly zero)
    unsigned *mvs;             // [nbcap] stamp of the sub-step in which the body last moved
    unsigned short *owner;     // [BP_NSLOT] per velocity slot: scratch of the warm-set closure and of the moving-list arbitration
    unsigned short *colmask;   // [BP_NSLOT] colours already used at the body of a slot (solve-order colouring)
    d2 *tf;                    // [64][2] (cos, sin) (tx, ty) of the moving bodies of the current chunk
    unsigned short *mv;        // [P.mvcap] moving-body list
    unsigned *mvo;             // [BP_NSLOT] stamp of the sub-step whose (next) moving list the slot's body has joined
    unsigned short *sbody;     // [BP_NSLOT] body that holds the slot (substep<KIND, DAMP = true> rebuilds the moving list from the slots)
    unsigned char *rf;         // [64] refresh flags of the current chunk
    // narrow phase (per candidate round, indexed by survivor rank): best plane separation of side A / B as order-preserving keys, its plane
    // index and support vertex
    unsigned long long *res_smA, *res_smB; // [64]
    unsigned *res_iA, *res_iB, *res_jA, *res_jB; // [64]
    // support queries (support_queries below): direction, (body | vertex count << 16), result value / first index; q_aux / q_c carry the plane
    // a query belongs to (pair rank | side << 8 | plane << 16) and dot(fn, fp) of that plane
    d2 *q_dir;                 // [BP_QCAP]
    double *q_c, *r_val;       // [BP_QCAP]
    unsigned *q_meta, *q_aux, *r_idx; // [BP_QCAP]
    // pair table of a candidate round: what the plane-bound rounds need of each surviving pair
    uint4 *pt_a;               // [64] sa | sb << 16, nA | nB << 8 | hA << 16 | hB << 24, evaluated sides | jA << 8 | jB << 16, -
    d2 *pt_thr;                // [64] separation of the cached plane of side A / B (-inf: none)
    // candidate cache: what the lanes of the FIRST candidate round found out about their (moving body, neighbour slot) -- body indices, slot, vertex
    // counts, the static part of the pair filter -- and the slot's hint word, valid while the moving list and the neighbour lists stay as they are
    unsigned long long *cc, *cc_hw; // [64]
    d2 *mbox;                  // [BP_MBOX][6] manifold mailbox (aliases the query buffers)
    // box-delivery only (substep<BP_ENV_BOX>): (1,3)/(2,3) pre_solve calls of the current sub-step
    unsigned *ev_key;          // [BP_EVCAP] shapeA << 16 | shapeB
    d2 *ev_d;                  // [BP_EVCAP][3] normal, r1, r2 of contact 0
    double *ctl;               // [4] box-delivery path controller: prev_heading_diff, path length, advanced length, robot_distance (k_bd_physics)
    unsigned long long *snap;  // [BP_SNAP_ROWS][BP_SNAP_COLS] box-delivery: the robot's state at an earlier sim step of execute_robot_path (recurrence test, k_bd_physics)
#ifdef BP_PROF
    unsigned long long *prof;  // [BP_PROFN] phase cycle counters and trip counts of this run (diagnostic build)
#endif
};

struct SubState {
    unsigned stamp;
    double curr_dt;
    int nmv;
    int nslots;
    unsigned long long prev_amask;
    int nlevels;
    double total_ke, total_imp;
    unsigned n_post, n_contact, n_first;
    int err;
    unsigned costp;            // work proxy of the env step: sum over sub-steps of 16 + 2 * active arbiters + 4 * warm arbiters * colours
    int yaw_violated, boundary_violated;
    int wall_flag;             // maze: robot body touched a wall (pre_solve of the (1,3) handler)
    int cc_ok, cc_kmax;        // candidate cache valid (wave-uniform), neighbour stride it was built with
    double ecoef_e, ecoef;     // last elasticity product seen by the post-solve bookkeeping and its (1 - e) / (1 + e)
    int quiescent;             // set by substep(): nothing moves and no arbiter is warm -> later sub-steps are no-ops
    unsigned ship_post, ship_contacts; // per-sub-step bookkeeping increments of the (cold) ship arbiters
    int nev;                   // box-delivery: recorded pre_solve events of this sub-step
    int robot_hit;             // box-delivery: robot_hit_obstacle (box_delivery_env.py:208-210)
    unsigned long long evmask; // box-delivery: bodies (index < 64) whose position a pre_solve changed in this sub-step
};

// Velocity slot of `body` (wave-uniform call): allocate a zeroed one on first use.  Slot 0 is the ship.
__device__ __forceinline__ int slot_get(const LdsCtx &L, SubState &S, const d2 *pxy, int body)
{
    int s = L.slot_of[body];
    if (BP_UNLIKELY(s == 255)) {
        s = S.nslots;
        if (s >= BP_NSLOT) { S.err |= BP_ERR_ARB_OVERFLOW; s = BP_NSLOT - 1; }
        else S.nslots = s + 1;
        if (lane_id() == 0) {
            L.slot_of[body] = (unsigned char)s; L.sbody[s] = (unsigned short)body;
            L.sv[s] = mk2(0.0, 0.0); L.sw[s] = mk2(0.0, 0.0); L.sb[s] = mk2(0.0, 0.0);
            L.sp[s] = pxy[body];
        }
        lds_sync();
    }
    return s;
}

struct Manifold { int count; d2 n; d2 p1_0, p2_0, p1_1, p2_1; unsigned h0, h1; };

__device__ __forceinline__ bool bb_overlap(double4 a, double4 b)
{
    return (a.x <= b.z && b.x <= a.z && a.y <= b.w && b.y <= a.w);
}

// Rebuild the neighbour list of body i around its current AABB (uniform call).
__device__ __forceinline__ void refresh_body(const DevParams &P, const EnvCtx &E, int i, int &err)
{
    const int lane = lane_id();
    const double4 b = gE(E.bb, i);
    double4 nf;
    nf.x = b.x - P.skin; nf.y = b.y - P.skin; nf.z = b.z + P.skin; nf.w = b.w + P.skin;
    if (lane == 0) gE(E.fat, i) = nf;
    __syncthreads();
    int cnt = 0;
    for (int base = 0; base < E.nb; base += 64) {
        const int j = base + lane;
        const bool valid = (j < E.nb) && (j != i);
        const double4 fj = valid ? gE(E.fat, j) : nf;
        const bool ov = valid && bb_overlap(nf, fj);
        const unsigned long long m = ballot(ov);
        const int pos = cnt + popc_below(m, lane);
        if (ov && pos < BP_KADJ) { gE(E.adj, i * BP_KADJ + pos) = (unsigned short)j; gE(E.hint, i * BP_KADJ + pos) = 0; }
        cnt += __popcll(m);
        if (ov && kind_btype(gE(E.kind, j)) != BODY_STATIC) { // static shapes never move: their own lists are never read
            int nj = gE(E.adjn, j);
            bool found = false;
            for (int s2 = 0; s2 < nj; s2++) found = found || (gE(E.adj, j * BP_KADJ + s2) == (unsigned short)i);
            if (!found) {
                if (nj >= BP_KADJ) { // purge entries of j that no longer fat-overlap j
                    int w = 0;
                    for (int s2 = 0; s2 < nj; s2++) {
                        const int k = gE(E.adj, j * BP_KADJ + s2);
                        const double4 fk = (k == i) ? nf : gE(E.fat, k);
                        if (bb_overlap(fj, fk)) {
                            gE(E.adj, j * BP_KADJ + w) = (unsigned short)k;
                            gE(E.hint, j * BP_KADJ + w) = gE(E.hint, j * BP_KADJ + s2);
                            w++;
                        }
                    }
                    nj = w;
                }
                if (nj < BP_KADJ) {
                    gE(E.adj, j * BP_KADJ + nj) = (unsigned short)i;
                    gE(E.hint, j * BP_KADJ + nj) = 0;
                    gE(E.adjn, j) = (unsigned char)(nj + 1);
                } else {
                    gE(E.adjn, j) = (unsigned char)nj;
                    err |= BP_ERR_ADJ_OVERFLOW;
                }
            }
        }
    }
    if (cnt > BP_KADJ) { err |= BP_ERR_ADJ_OVERFLOW; cnt = BP_KADJ; }
    if (lane == 0) gE(E.adjn, i) = (unsigned char)cnt;
    __syncthreads();
}

// world vertices/normals/AABB of body i from its pose (one lane per vertex within a 32-lane group)
__device__ __forceinline__ void world_from_pose(const DevParams &P, const EnvCtx &E, int i, bool active, int q, double4 t,
                                                double4 &outbb)
{
    const int n = active ? gE(E.nv, i) : 0;
    const bool valid = active && (q < n);
    double vx = 0, vy = 0;
    if (valid) {
        const d2 lv = gE(E.lv, i * BP_MAXV + q), ln = gE(E.ln, i * BP_MAXV + q);
        const double c = t.x, s = t.y;
        vx = (c * lv.x + (-s) * lv.y) + t.z;
        vy = (s * lv.x + c * lv.y) + t.w;
        const double nx = c * ln.x + (-s) * ln.y;
        const double ny = s * ln.x + c * ln.y;
        gE(E.wv, i * BP_MAXV + q) = mk2(vx, vy);
        gE(E.wn, i * BP_MAXV + q) = mk2(nx, ny);
    }
    const double l = half_min(valid ? vx : BP_INF);
    const double r = half_max(valid ? vx : -BP_INF);
    const double bo = half_min(valid ? vy : BP_INF);
    const double tp = half_max(valid ? vy : -BP_INF);
    const double rad = gE(E.prop, i).x;
    outbb.x = l - rad; outbb.y = bo - rad; outbb.z = r + rad; outbb.w = tp + rad;
}

__device__ __forceinline__ void apply_contact_impulses(const ArbReg &A, int c, d2 &va, double &wa, d2 &vb, double &wb, d2 j)
{
    const d2 r1 = c ? A.r1_1 : A.r1_0, r2 = c ? A.r2_1 : A.r2_0;
    const d2 jn = vneg(j);
    va = vadd(va, vmul(jn, A.ma));
    wa += A.ia * vcross(r1, jn);
    vb = vadd(vb, vmul(j, A.mb));
    wb += A.ib * vcross(r2, j);
}

// Support queries: for each of the nq records in L.q_dir / L.q_meta (direction d, body | vertex count << 16) the minimum of d . v over the body's
// world vertices and the FIRST vertex index that attains it -- what the sequential loop "for q: if (d . v[q] < mn) { mn = ...; jm = q; }" leaves in
// (mn, jm).  Eight lanes share a query (vertex q goes to lane q mod 8; each lane scans its vertices in ascending order with the same strict '<'),
// the group's minimum comes from three DPP steps, and among the lanes that hold it the one with the lowest index writes (value, index) to L.r_val /
// L.r_idx: the products, the sums and the comparisons are those of the sequential loop, so the result is bit-identical.  Wave-uniform call.
template <int VL>
__device__ __forceinline__ void support_queries(const EnvCtx &E, const LdsCtx &L, const int nq)
{
    const int lane = lane_id();
    const int l8 = lane & 7, g = lane >> 3;
    for (int base = 0; base < nq; base += 8) {
        const int k = base + g;
        const bool act = k < nq;
        const d2 dir = L.q_dir[act ? k : 0];
        const unsigned meta = L.q_meta[act ? k : 0];
        const int body = (int)(meta & 0xFFFFu), nv = (int)(meta >> 16);
        double best = BP_INF;
        int bi = 255;
#pragma unroll
        for (int t = 0; t < (VL + 7) / 8; t++) {
            const int q = l8 + 8 * t;
            const bool ok = act && (q < nv);
            const d2 v = gE(E.wv, body * BP_MAXV + (ok ? q : 0));
            const double d = vdot(dir, v);
            if (ok && d < best) { best = d; bi = q; }
        }
        const double m = oct_min_f64(best);
        const int mi = oct_min_i32((best == m) ? bi : 255);
        if (act && best == m && bi == mi) { L.r_val[k] = best; L.r_idx[k] = (unsigned)bi; }
    }
}

// One sub-step.  ship_rules: agent rules applied after the sub-step -- 0 none (reset / settle), 1 the yaw + boundary rules of ShipIceEnv.step
// (ship_ice_env.py:284-290), 2 the boundary rule alone (MazeNAMO.step, maze_NAMO_env.py:417-419: a maze handle served by a generic KIND 0 instantiation
// -- damping != 0, hulls above 8 vertices -- must not get the ship's yaw clamp).
// KIND == BP_ENV_BOX adds box-delivery's collision handlers (box_delivery_env.py:208-229,294-311); other values compile them out.
// DAMP: space.damping != 0 (ship_ice_env.py:120, maze_NAMO_env.py:148: `space.damping = cfg.sim.damping`; every shipped config sets 0).  cpBodyUpdateVelocity
// then multiplies the velocities of the dynamic bodies by damping^dt instead of clearing them, so a body keeps moving after its contacts are gone: the
// velocity integrate scales every velocity slot, an arbiter is warm as soon as one of its bodies has a velocity, and the next moving list is rebuilt from the
// velocity slots (L.sbody) instead of the active arbiters' bodies.  Everything else -- carried-over contacts of unmoved bodies, cold arbiters, the quiescent
// shortcut -- holds as stated, because each rests on "did not move" / "zero velocity", not on how a velocity came to be zero.
template <int KIND, bool DAMP = false>
__device__ __forceinline__ void substep(const DevParams &P, const EnvCtx &E, const LdsCtx &L, ArbReg &A, SubState &S,
                                        const double dt, const int ship_rules)
{
    const int lane = lane_id();
    // vertex loops run to the largest hull of the environment family (box-delivery: quads and triangles only; maze: the
    // 8-vertex robot outline, checked at load -- larger outlines use the generic instantiation)
    constexpr int VL = (KIND == BP_ENV_BOX) ? 4 : (KIND == BP_ENV_MAZE) ? 8 : BP_MAXV;
    if (KIND == BP_ENV_BOX) { S.nev = 0; S.evmask = 0ull; }
    S.stamp += 1u;
    const unsigned now = S.stamp;
    const double prev_dt = S.curr_dt;
    S.curr_dt = dt;
    if (A.key != ARB_FREE_KEY && A.stamp == now - 1u) A.state = ARB_NORMAL;
    PROF_DECL
    PROF_CNT(16, S.nmv)

    // ---- 1. integrate positions of the moving bodies; world geometry; AABBs ----------------------------------
    for (int k0 = 0; k0 < S.nmv; k0 += 64) {
        const int k = k0 + lane;
        // everything the lane's body needs in this phase is requested together: pose, mass / centre of gravity, radius, fat AABB, vertex count
        double rad = 0.0;
        double4 fatb; fatb.x = fatb.y = -BP_INF; fatb.z = fatb.w = BP_INF;
        if (k < S.nmv) {
            const int i = L.mv[k];
            const int sl = L.slot_of[i];
            d2 v = mk2(0.0, 0.0), w2 = mk2(0.0, 0.0), vb = mk2(0.0, 0.0);
            if (sl != 255) { v = L.sv[sl]; w2 = L.sw[sl]; vb = L.sb[sl]; }
            d2 p = gE(E.pxy, i);
            if (sl != 255) p = L.sp[sl];
            const double a = gE(E.ang, i);
            d2 r = gE(E.rot, i);
            const double4 ms = gE(E.mass, i);
            rad = gE(E.prop, i).x;
            fatb = gE(E.fat, i);
            L.rf[lane] = (unsigned char)gE(E.nv, i);
            p.x = p.x + (v.x + vb.x) * dt;
            p.y = p.y + (v.y + vb.y) * dt;
            const double a2 = a + (w2.x + w2.y) * dt;
            if (a2 != a) { double sn, cs; bp_sincos(a2, sn, cs); r = mk2(cs, sn); }
            gE(E.pxy, i) = p; gE(E.ang, i) = a2; gE(E.rot, i) = r;
            if (i == 0) { L.ag[0] = mk2(a2, 0.0); L.ag[1] = r; }
            if (sl != 255) { L.sb[sl] = mk2(0.0, 0.0); L.sw[sl].y = 0.0; L.sp[sl] = p; }   // w itself is unchanged: only the bias half is cleared (8-byte store)
            double4 t;
            t.x = r.x; t.y = r.y;
            t.z = p.x - (ms.z * r.x - ms.w * r.y);
            t.w = p.y - (ms.z * r.y + ms.w * r.x);
            L.tf[2 * lane] = mk2(t.x, t.y);
            L.tf[2 * lane + 1] = mk2(t.z, t.w);
            L.mvs[i] = now;
        }
        const int cnt = min(64, S.nmv - k0);
        // world vertices / normals, one (body, vertex) item per lane; AABB through LDS atomic min/max on order-preserving
        // keys (min and max are exact, so the reduction order is irrelevant).  bbk aliases the narrow-phase scratch.
        unsigned long long *bbk = L.res_smA; // [64][4] = min x, max x, min y, max y
        if (lane < cnt) { bbk[lane * 4 + 0] = ~0ull; bbk[lane * 4 + 1] = 0ull; bbk[lane * 4 + 2] = ~0ull; bbk[lane * 4 + 3] = 0ull; }
        lds_sync();
        PROF_ACC(33)
        for (int t0 = 0; t0 < cnt * VL; t0 += 64) {
            const int t = t0 + lane;
            const int kk = t / VL, q = t - kk * VL;
            if (kk < cnt) {
                const int i = L.mv[k0 + kk];
                if (q < (int)L.rf[kk]) {
                    const d2 t0_ = L.tf[2 * kk], t1_ = L.tf[2 * kk + 1];
                    const double c = t0_.x, s = t0_.y;
                    const d2 lv = gE(E.lv, i * BP_MAXV + q), ln = gE(E.ln, i * BP_MAXV + q);
                    const double vx = (c * lv.x + (-s) * lv.y) + t1_.x;
                    const double vy = (s * lv.x + c * lv.y) + t1_.y;
                    const double nx = c * ln.x + (-s) * ln.y;
                    const double ny = s * ln.x + c * ln.y;
                    gE(E.wv, i * BP_MAXV + q) = mk2(vx, vy);
                    gE(E.wn, i * BP_MAXV + q) = mk2(nx, ny);
                    const unsigned long long kx = f64_key(vx), ky = f64_key(vy);
                    atomicMin(&bbk[kk * 4 + 0], kx); atomicMax(&bbk[kk * 4 + 1], kx);
                    atomicMin(&bbk[kk * 4 + 2], ky); atomicMax(&bbk[kk * 4 + 3], ky);
                }
            }
        }
        lds_sync();
        PROF_ACC(34)
        bool leftfat = false;
        if (lane < cnt) {
            const int i = L.mv[k0 + lane];
            double4 nbb;
            nbb.x = key_f64(bbk[lane * 4 + 0]) - rad; nbb.y = key_f64(bbk[lane * 4 + 2]) - rad;
            nbb.z = key_f64(bbk[lane * 4 + 1]) + rad; nbb.w = key_f64(bbk[lane * 4 + 3]) + rad;
            gE(E.bb, i) = nbb;
            leftfat = !(nbb.x >= fatb.x && nbb.y >= fatb.y && nbb.z <= fatb.z && nbb.w <= fatb.w);
        }
        PROF_ACC(0)
        // ---- 2. Verlet refresh --------------------------------------------------------------------------------
        unsigned long long rm = ballot(leftfat);
        if (BP_UNLIKELY(rm != 0)) { S.cc_ok = 0; __syncthreads(); } // refresh_body reads the AABBs other lanes have just stored; the lists change
        while (BP_UNLIKELY(rm != 0)) {
            const int kk = __ffsll((long long)rm) - 1;
            rm &= rm - 1;
            refresh_body(P, E, L.mv[k0 + kk], S.err);
            PROF_CNT(17, 1)
        }
        PROF_ACC(1)
    }
    __syncthreads();
    PROF_ACC(35)

    // ---- 3./4. candidate pairs of moving bodies ----------------------------------------------------------------
    int kmax = S.cc_kmax; // largest neighbour count among the moving bodies (wave-uniform, found with 5 ballots per chunk)
    if (!S.cc_ok) {
        kmax = 0;
        for (int k0 = 0; k0 < S.nmv; k0 += 64) {
            const int k = k0 + lane;
            const int cnt = (k < S.nmv) ? (int)gE(E.adjn, L.mv[k]) : 0;
            int m = 0;
            for (int bit = 16; bit >= 1; bit >>= 1) { if (ballot(cnt >= (m | bit))) m |= bit; }
            kmax = max(kmax, m);
        }
        S.cc_kmax = kmax;
    }
    const int ncand_slots = S.nmv * kmax;
    for (int base = 0; base < ncand_slots; base += 64) {
        // What a lane knows about its (moving body, neighbour slot) before any geometry: which pair it is, the vertex counts, and whether the pair
        // is evaluated at all (list bounds, "evaluated from the other body's list", shapes of one body, two infinite masses).  None of it changes
        // while the moving list and the neighbour lists stay the same, so the first round keeps it in LDS (with the slot's hint word, which this
        // lane alone rewrites) and the next sub-step starts from there: one LDS read and ONE global round trip instead of two dependent ones.
        int i, j, s, nA_h, nB_h;
        bool valid, flagonly;
        unsigned long long hw;
        const bool cached = S.cc_ok && base == 0 && !BP_DBGP(1);
        if (cached) {
            const unsigned long long c = L.cc[lane];
            hw = L.cc_hw[lane];
            static_assert(2 * BP_CC_IDX_BITS == 28, "cc word layout: i | j << BP_CC_IDX_BITS | slot << 28 ...");
            i = (int)(c & ((1u << BP_CC_IDX_BITS) - 1u)); j = (int)((c >> BP_CC_IDX_BITS) & ((1u << BP_CC_IDX_BITS) - 1u)); s = (int)((c >> 28) & 31u);
            nA_h = (int)((c >> 33) & 31u); nB_h = (int)((c >> 38) & 31u);
            valid = ((c >> 43) & 1u) != 0; flagonly = ((c >> 44) & 1u) != 0;
        } else {
            const int idx = base + lane;
            const int k = idx / kmax;
            s = idx - k * kmax;
            // round trip 1: neighbour id / count / hint word; 2 (below): kinds, masses, vertex counts
            const bool inlist = k < S.nmv;
            i = inlist ? (int)L.mv[k] : 0;
            const int sc = min(s, BP_KADJ - 1);
            const int adjn_i = gE(E.adjn, i);
            j = gE(E.adj, i * BP_KADJ + sc) < E.nb ? (int)gE(E.adj, i * BP_KADJ + sc) : 0;
            hw = gE(E.hint, i * BP_KADJ + sc);
            valid = inlist && (s < adjn_i);
            if (valid && L.mvs[j] == now && j < i) valid = false; // pair is evaluated from j's list
            const int ki = gE(E.kind, i), kj = gE(E.kind, j);
            const double mi = gE(E.mass, i).x, mj = gE(E.mass, j).x;
            nA_h = gE(E.nv, min(i, j)); nB_h = gE(E.nv, max(i, j));
            flagonly = false; // two infinite-mass shapes: evaluated only for the (1,3) robot x wall handler, never solved
            if (valid) {
                if (kind_group(ki) != 0 && kind_group(ki) == kind_group(kj)) valid = false; // shapes of one body
                else if (mi == 0.0 && mj == 0.0) {
                    const int ci = kind_ctype(ki), cj = kind_ctype(kj);
                    flagonly = (ci == 1 && cj == 3) || (ci == 3 && cj == 1);
                    valid = flagonly;
                }
            }
            s = min(s, 31);
            if (base == 0) {
                L.cc[lane] = (unsigned long long)(unsigned)i | ((unsigned long long)(unsigned)j << BP_CC_IDX_BITS) | ((unsigned long long)(unsigned)s << 28) |
                             ((unsigned long long)(unsigned)nA_h << 33) | ((unsigned long long)(unsigned)nB_h << 38) |
                             ((unsigned long long)(valid ? 1u : 0u) << 43) | ((unsigned long long)(flagonly ? 1u : 0u) << 44);
                L.cc_hw[lane] = hw;
            }
        }
        const int sc = min(s, BP_KADJ - 1);
        const int sa = min(i, j), sb = max(i, j);
        const double4 bbi = gE(E.bb, i);
        const double4 bbj = gE(E.bb, j);
        const double radA = gE(E.prop, sa).x, radB = gE(E.prop, sb).x;
        const double rsum = radA + radB;
        // the cached planes (hint word) of both sides travel with this round trip too, and so do the cached support vertex of each with its two
        // cyclic neighbours: plane and vertex indices only need to be valid addresses here
        const int hA = HW_PLANE_A(hw) < BP_MAXV ? HW_PLANE_A(hw) : 0, hB = HW_PLANE_B(hw) < BP_MAXV ? HW_PLANE_B(hw) : 0;
        const int cnA = HW_NV_A(hw), cnB = HW_NV_B(hw);
        const int jA0 = HW_VERT_A(hw) < BP_MAXV ? HW_VERT_A(hw) : 0, jB0 = HW_VERT_B(hw) < BP_MAXV ? HW_VERT_B(hw) : 0;
        const int jAm = (jA0 == 0) ? max(min(cnB, BP_MAXV) - 1, 0) : jA0 - 1, jAp = (jA0 + 1 >= cnB) ? 0 : jA0 + 1;   // on B
        const int jBm = (jB0 == 0) ? max(min(cnA, BP_MAXV) - 1, 0) : jB0 - 1, jBp = (jB0 + 1 >= cnA) ? 0 : jB0 + 1;   // on A
        const d2 fnA = gE(E.wn, sa * BP_MAXV + hA), fpA = gE(E.wv, sa * BP_MAXV + hA);
        const d2 fnB = gE(E.wn, sb * BP_MAXV + hB), fpB = gE(E.wv, sb * BP_MAXV + hB);
        const d2 vAm = gE(E.wv, sb * BP_MAXV + jAm), vA0 = gE(E.wv, sb * BP_MAXV + jA0), vAp = gE(E.wv, sb * BP_MAXV + jAp);
        const d2 vBm = gE(E.wv, sa * BP_MAXV + jBm), vB0 = gE(E.wv, sa * BP_MAXV + jB0), vBp = gE(E.wv, sa * BP_MAXV + jBp);
        if (valid) valid = bb_overlap(bbi, bbj);
        PROF_ACC(27)
        PROF_CNT(24, 1)
        PROF_CNT(36, __popcll(ballot(valid)))
        if (ballot(valid) == 0) { PROF_ACC(2) continue; }
        // ---- 4a. cached planes: the plane that won a side of this pair's plane search last time is evaluated exactly (its minimum over the other
        //      shape's vertices, found by eight lanes per plane).  Any plane's separation is a lower bound of the pair's maximum, so a cached plane
        //      that clears the radii rejects the pair (cpCollide would find no contact).  Pairs that were rejected last time test only the plane
        //      that rejected them; pairs that got through test the cached plane of both sides.
        const bool evA = valid && (hw & HW_HAS_A) && ((hw & HW_BOTH) || !(hw & HW_PRIM_B));
        const bool evB = valid && (hw & HW_HAS_B) && ((hw & HW_BOTH) || (hw & HW_PRIM_B));
        double sepAc = -BP_INF, sepBc = -BP_INF;
        int jAc = jA0, jBc = jB0;
        const double cA = vdot(fnA, fpA), cB = vdot(fnB, fpB);
        bool qryA = evA, qryB = evB; // sides whose cached support vertex does not pass the local test: searched by support_queries
        {   // the cached support vertex against its two neighbours (BP_SUPPORT_MARGIN): exact minimum and first index without a search
            const double dm = vdot(fnA, vAm), d0 = vdot(fnA, vA0), dp = vdot(fnA, vAp);
            if (evA && cnB == nB_h && cnB >= 2 && d0 + BP_SUPPORT_MARGIN <= dm && d0 + BP_SUPPORT_MARGIN <= dp && !BP_DBGP(4)) { sepAc = (d0 - cA) + 0.0; qryA = false; }
        }
        {
            const double dm = vdot(fnB, vBm), d0 = vdot(fnB, vB0), dp = vdot(fnB, vBp);
            if (evB && cnA == nA_h && cnA >= 2 && d0 + BP_SUPPORT_MARGIN <= dm && d0 + BP_SUPPORT_MARGIN <= dp && !BP_DBGP(4)) { sepBc = (d0 - cB) + 0.0; qryB = false; }
        }
        {
            const unsigned long long mqA = ballot(qryA), mqB = ballot(qryB);
            const int nqA = __popcll(mqA), nq1 = nqA + __popcll(mqB);
            PROF_CNT(37, nq1)
            if (BP_UNLIKELY(nq1)) {
                const int slA = popc_below(mqA, lane), slB = nqA + popc_below(mqB, lane);
                for (int q0 = 0; q0 < nq1; q0 += BP_QCAP) { // one batch unless more than BP_QCAP planes need the search in this round
                    const bool inA = qryA && slA >= q0 && slA < q0 + BP_QCAP, inB = qryB && slB >= q0 && slB < q0 + BP_QCAP;
                    if (inA) { L.q_dir[slA - q0] = fnA; L.q_meta[slA - q0] = (unsigned)sb | ((unsigned)nB_h << 16); }
                    if (inB) { L.q_dir[slB - q0] = fnB; L.q_meta[slB - q0] = (unsigned)sa | ((unsigned)nA_h << 16); }
                    lds_sync();
                    support_queries<VL>(E, L, min(nq1 - q0, BP_QCAP));
                    lds_sync();
                    if (inA) { sepAc = (L.r_val[slA - q0] - cA) + 0.0; jAc = (int)L.r_idx[slA - q0]; }
                    if (inB) { sepBc = (L.r_val[slB - q0] - cB) + 0.0; jBc = (int)L.r_idx[slB - q0]; }
                    lds_sync();
                }
            }
        }
        if (valid && (sepAc > rsum || sepBc > rsum)) {
            valid = false;
            if (hw & HW_BOTH) {
                const unsigned long long nh = (hw & ~(HW_BOTH | HW_PRIM_B)) | ((sepAc > rsum) ? 0ull : HW_PRIM_B);
                gE(E.hint, i * BP_KADJ + s) = nh;
                if (base == 0) L.cc_hw[lane] = nh;
            }
        }
        const unsigned long long cm = ballot(valid);
        PROF_ACC(28)
        PROF_CNT(18, __popcll(cm))
        if (cm == 0) continue;
        PROF_CNT(25, 1)
        // ---- 4a'. every other plane of the surviving pairs: one round per pair, lanes 0..31 = planes of A, 32..63 = planes of B.  A plane's separation
        //      is a minimum over the other shape's vertices, so its minimum over THREE of them (the support vertex of the side's cached plane and its two
        //      neighbours) bounds it from above; only planes whose bound reaches the cached plane's exact value can win the side (ties included) and are searched exactly.
        const int nc = __popcll(cm);
        const int myr = popc_below(cm, lane); // rank of this lane's pair among the survivors
        const int nA_l = valid ? nA_h : 0, nB_l = valid ? nB_h : 0;
        if (valid) {
            uint4 pa;
            pa.x = (unsigned)sa | ((unsigned)sb << 16);
            pa.y = (unsigned)nA_l | ((unsigned)nB_l << 8) | ((unsigned)hA << 16) | ((unsigned)hB << 24);
            pa.z = (evA ? 1u : 0u) | (evB ? 2u : 0u) | ((unsigned)jAc << 8) | ((unsigned)jBc << 16);
            pa.w = 0u;
            L.pt_a[myr] = pa;
            L.pt_thr[myr] = mk2(sepAc, sepBc);
            L.res_smA[myr] = evA ? f64_key(sepAc) : 0ull; L.res_iA[myr] = evA ? (unsigned)hA : 0xFFFFFFFFu; L.res_jA[myr] = (unsigned)jAc;
            L.res_smB[myr] = evB ? f64_key(sepBc) : 0ull; L.res_iB[myr] = evB ? (unsigned)hB : 0xFFFFFFFFu; L.res_jB[myr] = (unsigned)jBc;
        }
        lds_sync();
        {
            int nq = 0, g0 = 0; // queries collected, first pair rank of the current group
            const int side = lane >> 5, f = lane & 31;
            // One bound round: lanes 0..31 = planes of A, 32..63 = planes of B of the pair with rank rr.
            auto round_addr = [&](const uint4 pa, int &pbody, int &qbody, int &np, int &nqv, int &hX, bool &evX, int &jc, int &jm, int &jp) {
                const int psa = (int)(pa.x & 0xFFFFu), psb = (int)(pa.x >> 16);
                const int pna = (int)(pa.y & 0xFFu), pnb = (int)((pa.y >> 8) & 0xFFu);
                pbody = side ? psb : psa; qbody = side ? psa : psb;
                np = side ? pnb : pna; nqv = side ? pna : pnb;
                hX = (int)((pa.y >> (side ? 24 : 16)) & 0xFFu);
                evX = ((pa.z >> side) & 1u) != 0;
                jc = (int)((pa.z >> (side ? 16 : 8)) & 0xFFu);
                jm = (jc == 0) ? max(nqv, 1) - 1 : jc - 1; jp = (jc + 1 >= nqv) ? 0 : jc + 1;   // cyclic neighbours of the support vertex
            };
            // Fast path: the rounds are taken two at a time so that their loads travel together; the surviving planes go straight into the query
            // buffer.  If they do not all fit (rare: pairs without cached planes), the sequential loop below redoes the rounds group by group.
            int rr_start = nc;
            {
                bool fits = !BP_DBGP(2);
                for (int r0 = 0; r0 < nc && fits; r0 += 2) {
                    const bool two = r0 + 1 < nc;
                    const uint4 pa0 = L.pt_a[r0], pa1 = L.pt_a[two ? r0 + 1 : r0];
                    const d2 thr0 = L.pt_thr[r0], thr1 = L.pt_thr[two ? r0 + 1 : r0];
                    int pb0, qb0, np0, nqv0, hX0, jc0, jm0, jp0, pb1, qb1, np1, nqv1, hX1, jc1, jm1, jp1; bool ev0, ev1;
                    round_addr(pa0, pb0, qb0, np0, nqv0, hX0, ev0, jc0, jm0, jp0);
                    round_addr(pa1, pb1, qb1, np1, nqv1, hX1, ev1, jc1, jm1, jp1);
                    const int fc0 = (f < np0) ? f : 0, fc1 = (f < np1) ? f : 0;
                    const d2 fn0 = gE(E.wn, pb0 * BP_MAXV + fc0), fp0 = gE(E.wv, pb0 * BP_MAXV + fc0), vb0 = gE(E.wv, qb0 * BP_MAXV + jc0), vm0 = gE(E.wv, qb0 * BP_MAXV + jm0), vp0 = gE(E.wv, qb0 * BP_MAXV + jp0);
                    const d2 fn1 = gE(E.wn, pb1 * BP_MAXV + fc1), fp1 = gE(E.wv, pb1 * BP_MAXV + fc1), vb1 = gE(E.wv, qb1 * BP_MAXV + jc1), vm1 = gE(E.wv, qb1 * BP_MAXV + jm1), vp1 = gE(E.wv, qb1 * BP_MAXV + jp1);
                    {
                        const double th = side ? thr0.y : thr0.x;
                        const bool pv = (f < np0) && !(ev0 && f == hX0);
                        const double c = vdot(fn0, fp0);
                        const double bound = (fmin(vdot(fn0, vb0), fmin(vdot(fn0, vm0), vdot(fn0, vp0))) - c) + 0.0;
                        const bool surv = pv && (!ev0 || bound >= th);
                        const unsigned long long sm = ballot(surv);
                        const int sl = nq + popc_below(sm, lane);
                        if (surv && sl < BP_QCAP) {
                            L.q_dir[sl] = fn0; L.q_meta[sl] = (unsigned)qb0 | ((unsigned)nqv0 << 16);
                            L.q_aux[sl] = (unsigned)r0 | ((unsigned)side << 8) | ((unsigned)f << 16);
                            L.q_c[sl] = c;
                        }
                        nq += __popcll(sm);
                    }
                    {
                        const double th = side ? thr1.y : thr1.x;
                        const bool pv = two && (f < np1) && !(ev1 && f == hX1);
                        const double c = vdot(fn1, fp1);
                        const double bound = (fmin(vdot(fn1, vb1), fmin(vdot(fn1, vm1), vdot(fn1, vp1))) - c) + 0.0;
                        const bool surv = pv && (!ev1 || bound >= th);
                        const unsigned long long sm = ballot(surv);
                        const int sl = nq + popc_below(sm, lane);
                        if (surv && sl < BP_QCAP) {
                            L.q_dir[sl] = fn1; L.q_meta[sl] = (unsigned)qb1 | ((unsigned)nqv1 << 16);
                            L.q_aux[sl] = (unsigned)(r0 + 1) | ((unsigned)side << 8) | ((unsigned)f << 16);
                            L.q_c[sl] = c;
                        }
                        nq += __popcll(sm);
                    }
                    fits = nq <= BP_QCAP;
                }
                if (!fits) { nq = 0; rr_start = 0; }
                lds_sync();
            }
            for (int rr = rr_start; rr <= nc; rr++) {
                // the queries collected so far are searched when the next round might not fit, and after the last pair
                if (nq > 0 && (rr == nc || nq + 64 > BP_QCAP)) {
                    PROF_ACC(29)
                    support_queries<VL>(E, L, nq);
                    lds_sync();
                    PROF_CNT(26, 1)
                    PROF_CNT(38, nq)
                    // separation of every searched plane -> per (pair, side) maximum, lowest plane index on ties, its support vertex
                    for (int s0 = 0; s0 < nq; s0 += 64) {
                        const int sl = s0 + lane;
                        if (sl < nq) {
                            const unsigned aux = L.q_aux[sl];
                            const int r = (int)(aux & 0xFFu);
                            const double sp = (L.r_val[sl] - L.q_c[sl]) + 0.0; // "+ 0.0": -0 and +0 share one key
                            const unsigned long long key = f64_key(sp);
                            L.q_c[sl] = __builtin_bit_cast(double, key);
                            atomicMax((aux & 0x100u) ? &L.res_smB[r] : &L.res_smA[r], key);
                        }
                    }
                    lds_sync();
                    if (valid && myr >= g0 && myr < rr) { // a cached plane that has been beaten gives up its index
                        if (evA && f64_key(sepAc) != L.res_smA[myr]) L.res_iA[myr] = 0xFFFFFFFFu;
                        if (evB && f64_key(sepBc) != L.res_smB[myr]) L.res_iB[myr] = 0xFFFFFFFFu;
                    }
                    lds_sync();
                    for (int s0 = 0; s0 < nq; s0 += 64) {
                        const int sl = s0 + lane;
                        if (sl < nq) {
                            const unsigned aux = L.q_aux[sl];
                            const int r = (int)(aux & 0xFFu);
                            const unsigned long long key = __builtin_bit_cast(unsigned long long, L.q_c[sl]);
                            if (key == ((aux & 0x100u) ? L.res_smB[r] : L.res_smA[r])) atomicMin((aux & 0x100u) ? &L.res_iB[r] : &L.res_iA[r], aux >> 16);
                        }
                    }
                    lds_sync();
                    for (int s0 = 0; s0 < nq; s0 += 64) {
                        const int sl = s0 + lane;
                        if (sl < nq) {
                            const unsigned aux = L.q_aux[sl];
                            const int r = (int)(aux & 0xFFu);
                            const unsigned long long key = __builtin_bit_cast(unsigned long long, L.q_c[sl]);
                            const bool onB = (aux & 0x100u) != 0;
                            if (key == (onB ? L.res_smB[r] : L.res_smA[r]) && (aux >> 16) == (onB ? L.res_iB[r] : L.res_iA[r])) {
                                if (onB) L.res_jB[r] = L.r_idx[sl]; else L.res_jA[r] = L.r_idx[sl];
                            }
                        }
                    }
                    lds_sync();
                    nq = 0; g0 = rr;
                    PROF_ACC(30)
                }
                if (rr == nc) break;
                const uint4 pa = L.pt_a[rr];
                const d2 thr = L.pt_thr[rr];
                int pbody, qbody, np, nqv, hX, jc, jm, jp; bool evX;
                round_addr(pa, pbody, qbody, np, nqv, hX, evX, jc, jm, jp);
                const double th = side ? thr.y : thr.x;
                const bool pv = (f < np) && !(evX && f == hX);
                const int fc = (f < np) ? f : 0;
                const d2 fn = gE(E.wn, pbody * BP_MAXV + fc), fp = gE(E.wv, pbody * BP_MAXV + fc);
                const d2 vb = gE(E.wv, qbody * BP_MAXV + jc), vm = gE(E.wv, qbody * BP_MAXV + jm), vp = gE(E.wv, qbody * BP_MAXV + jp);
                const double c = vdot(fn, fp);
                const double bound = (fmin(vdot(fn, vb), fmin(vdot(fn, vm), vdot(fn, vp))) - c) + 0.0;
                const bool surv = pv && (!evX || bound >= th);
                const unsigned long long sm = ballot(surv);
                if (surv) {
                    const int sl = nq + popc_below(sm, lane);
                    L.q_dir[sl] = fn; L.q_meta[sl] = (unsigned)qbody | ((unsigned)nqv << 16);
                    L.q_aux[sl] = (unsigned)rr | ((unsigned)side << 8) | ((unsigned)f << 16);
                    L.q_c[sl] = c;
                }
                nq += __popcll(sm);
                lds_sync();
            }
        }
        PROF_ACC(3)
        // ---- 4b. closest features -> normal -> Chipmunk ContactPoints, one pair per lane ------------------------------
        Manifold M;
        M.count = 0; M.h0 = M.h1 = 0; M.n = mk2(0, 0);
        M.p1_0 = M.p2_0 = M.p1_1 = M.p2_1 = mk2(0, 0);
        bool touching = false;
        int src = 2; // where the normal comes from: 0 = plane iA of A, 1 = plane iB of B (negated), 2 = a vertex pair
        d2 n = mk2(0, 0);
        int iA = 0, iB = 0, jA = 0, jB = 0;
        int i1A = 0, i1B = 0;              // support vertices of the contact (PolySupportPointIndex)
        bool needA = false, needB = false; // ... that a support query has to find
        const int nA = nA_l, nB = nB_l;
        const d2 *Av = E.wv + sa * BP_MAXV, *Bv = E.wv + sb * BP_MAXV;   // (for the rare tie_partner scan; everything else goes through gE)
        const int oA = sa * BP_MAXV, oB = sb * BP_MAXV;
        if (valid) {
            const double sA = key_f64(L.res_smA[myr]), sB = key_f64(L.res_smB[myr]);
            iA = (int)L.res_iA[myr]; iB = (int)L.res_iB[myr]; jA = (int)L.res_jA[myr]; jB = (int)L.res_jB[myr];
            const bool useA = (sA >= sB);
            const double smax = useA ? sA : sB;
            touching = true;
            // every vertex / normal the three cases below can need is fetched up front: one round trip instead of a dependent chain
            const int iA0 = (iA == 0) ? nA - 1 : iA - 1, iB0 = (iB == 0) ? nB - 1 : iB - 1;
            const d2 nAi = gE(E.wn, oA + iA), nBi = gE(E.wn, oB + iB);
            const d2 aA = gE(E.wv, oA + iA0), bA = gE(E.wv, oA + iA), qA = gE(E.wv, oB + jA);
            const d2 aB = gE(E.wv, oB + iB0), bB = gE(E.wv, oB + iB), qB = gE(E.wv, oA + jB);
            // the outer neighbours of the two winning edges: they certify the support vertex of the shape that owns the normal (below)
            const int iA0m = (iA0 == 0) ? nA - 1 : iA0 - 1, iAp = (iA + 1 >= nA) ? 0 : iA + 1;
            const int iB0m = (iB0 == 0) ? nB - 1 : iB0 - 1, iBp = (iB + 1 >= nB) ? 0 : iB + 1;
            const d2 oA0 = gE(E.wv, oA + iA0m), oA1 = gE(E.wv, oA + iAp), oB0 = gE(E.wv, oB + iB0m), oB1 = gE(E.wv, oB + iBp);
            if (smax > rsum) touching = false;
            else if (smax <= 0.0) { n = useA ? nAi : vneg(nBi); src = useA ? 0 : 1; }
            else {
                const d2 eA = vsub(bA, aA);
                const double uA = vdot(vsub(qA, aA), eA), eeA = vdot(eA, eA);
                const bool spanA = !(uA < 0.0) && !(uA > eeA);
                const d2 eB = vsub(bB, aB);
                const double uB = vdot(vsub(qB, aB), eB), eeB = vdot(eB, eB);
                const bool spanB = !(uB < 0.0) && !(uB > eeB);
                // Facing edges that are parallel to rounding and overlap only partly: the first-minimum support vertex of a side can be the end of the other
                // shape's facing edge that lies beyond this edge while its neighbour lies over it (narrow-phase hardening) -- the closest
                // features are then the two edges, not a vertex pair.  The neighbour towards the span (both hulls counter-clockwise: along the facing side the
                // tangential coordinate falls with the index) stands in if it lies over the edge and at most BP_TIE_TOL higher above the plane.  Rare: the
                // neighbour is fetched only here.
                auto tie_partner = [&](const d2 aP, const d2 eP, const double eeP, const d2 nP, const d2 *Qv, const int nQ, const int j, const d2 q0, const double u) -> bool {
                    const int jn = (u < 0.0) ? ((j == 0) ? nQ - 1 : j - 1) : ((j + 1 >= nQ) ? 0 : j + 1);
                    const d2 q1 = Qv[jn];
                    const double u1 = vdot(vsub(q1, aP), eP);
                    return !(u1 < 0.0) && !(u1 > eeP) && (vdot(nP, q1) - vdot(nP, q0) <= BP_TIE_TOL);
                };
                if (useA) {
                    if (spanA) { n = nAi; src = 0; }
                    else if (sB > 0.0 && spanB) { n = vneg(nBi); src = 1; }
                    else if (tie_partner(aA, eA, eeA, nAi, Bv, nB, jA, qA, uA)) { n = nAi; src = 0; }
                    else if (sB > 0.0 && tie_partner(aB, eB, eeB, nBi, Av, nA, jB, qB, uB)) { n = vneg(nBi); src = 1; }
                    else {
                        const d2 pp = vsub(qA, (uA < 0.0) ? aA : bA);
                        const double dl = vlen(pp);
                        if (dl > rsum) touching = false;
                        n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                    }
                } else {
                    if (spanB) { n = vneg(nBi); src = 1; }
                    else if (sA > 0.0 && spanA) { n = nAi; src = 0; }
                    else if (tie_partner(aB, eB, eeB, nBi, Av, nA, jB, qB, uB)) { n = vneg(nBi); src = 1; }
                    else if (sA > 0.0 && tie_partner(aA, eA, eeA, nAi, Bv, nB, jA, qA, uA)) { n = nAi; src = 0; }
                    else {
                        const d2 pp = vsub((uB < 0.0) ? aB : bB, qB);
                        const double dl = vlen(pp);
                        if (dl > rsum) touching = false;
                        n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                    }
                }
            }
            // Support vertices (PolySupportPointIndex: first maximum of v . n over A, of v . -n over B).  With the normal of plane iA, v . -n over B is the
            // negated sequence whose first minimum the plane search has already found (jA); likewise jB for a normal from B.  The shape that owns the
            // normal: the maximum lies on the winning edge, i.e. at one of its two end vertices (equal in exact arithmetic); when both outer neighbours
            // stay BP_SUPPORT_MARGIN below, no other vertex of the convex hull can reach it, and the first maximum of the sequential scan is the larger of
            // the two, the lower index on a tie.  Otherwise (and for vertex-vertex normals) a support query finds it.
            i1A = jB; i1B = jA;
            needA = touching && src != 1; needB = touching && src != 0;
            if (touching && src == 0) {
                const double c0 = vdot(aA, n), c1 = vdot(bA, n), o0 = vdot(oA0, n), o1 = vdot(oA1, n);
                const double cm = (c0 > c1) ? c0 : c1, om = (o0 > o1) ? o0 : o1;
                if ((nA == 2 || cm >= om + BP_SUPPORT_MARGIN) && !BP_DBGP(8)) { i1A = (c0 > c1) ? iA0 : (c1 > c0) ? iA : min(iA0, iA); needA = false; }
            }
            if (touching && src == 1) {
                const d2 nn = vneg(n);
                const double c0 = vdot(aB, nn), c1 = vdot(bB, nn), o0 = vdot(oB0, nn), o1 = vdot(oB1, nn);
                const double cm = (c0 > c1) ? c0 : c1, om = (o0 > o1) ? o0 : o1;
                if ((nB == 2 || cm >= om + BP_SUPPORT_MARGIN) && !BP_DBGP(8)) { i1B = (c0 > c1) ? iB0 : (c1 > c0) ? iB : min(iB0, iB); needB = false; }
            }
            // the winners of both sides are the next sub-step's cached planes
            const unsigned long long nh = (unsigned long long)((unsigned)iA | ((unsigned)iB << 5) | ((unsigned)jA << 10) | ((unsigned)jB << 15) |
                                                               ((unsigned)nA << 20) | ((unsigned)nB << 25)) |
                                          HW_HAS_A | HW_HAS_B | (useA ? 0ull : HW_PRIM_B) | ((smax > rsum) ? 0ull : HW_BOTH);
            gE(E.hint, i * BP_KADJ + s) = nh;
            if (base == 0) L.cc_hw[lane] = nh;
        }
        PROF_ACC(31)
        // support queries for the support vertices that could not be certified: direction -n over A (n over B); the first minimum is the first maximum wanted
        {
            const unsigned long long mA = ballot(needA), mB = ballot(needB);
            const int nqa = __popcll(mA), nq2 = nqa + __popcll(mB);
            PROF_CNT(39, nq2)
            const int slA = popc_below(mA, lane), slB = nqa + popc_below(mB, lane);
            for (int q0 = 0; q0 < nq2; q0 += BP_QCAP) {
                const bool inA = needA && slA >= q0 && slA < q0 + BP_QCAP, inB = needB && slB >= q0 && slB < q0 + BP_QCAP;
                if (inA) { L.q_dir[slA - q0] = vneg(n); L.q_meta[slA - q0] = (unsigned)sa | ((unsigned)nA << 16); }
                if (inB) { L.q_dir[slB - q0] = n; L.q_meta[slB - q0] = (unsigned)sb | ((unsigned)nB << 16); }
                lds_sync();
                support_queries<VL>(E, L, min(nq2 - q0, BP_QCAP));
                lds_sync();
                if (inA) i1A = (int)L.r_idx[slA - q0];
                if (inB) i1B = (int)L.r_idx[slB - q0];
                lds_sync();
            }
        }
        PROF_ACC(32)
        if (touching) {
            const d2 nn = vneg(n);
            d2 e1a, e1b, e2a, e2b;
            int e1ia, e1ib, e2ia, e2ib;
            {   // both candidate edges of each shape are fetched together, then selected
                const int a0 = (i1A == 0) ? nA - 1 : i1A - 1, a2 = (i1A + 1 == nA) ? 0 : i1A + 1;
                const int b0 = (i1B == 0) ? nB - 1 : i1B - 1, b2 = (i1B + 1 == nB) ? 0 : i1B + 1;
                const d2 nA1 = gE(E.wn, oA + i1A), nA2 = gE(E.wn, oA + a2), vA0 = gE(E.wv, oA + a0), vA1 = gE(E.wv, oA + i1A), vA2 = gE(E.wv, oA + a2);
                const d2 nB1 = gE(E.wn, oB + i1B), nB2 = gE(E.wn, oB + b2), vB0 = gE(E.wv, oB + b0), vB1 = gE(E.wv, oB + i1B), vB2 = gE(E.wv, oB + b2);
                const bool fa = vdot(n, nA1) > vdot(n, nA2);
                e1a = fa ? vA0 : vA1; e1ia = fa ? a0 : i1A; e1b = fa ? vA1 : vA2; e1ib = fa ? i1A : a2;
                const bool fb = vdot(nn, nB1) > vdot(nn, nB2);
                e2a = fb ? vB0 : vB1; e2ia = fb ? b0 : i1B; e2b = fb ? vB1 : vB2; e2ib = fb ? i1B : b2;
            }
            const double r1 = radA, r2 = radB;
            const double d_e1_a = vcross(e1a, n), d_e1_b = vcross(e1b, n);
            const double d_e2_a = vcross(e2a, n), d_e2_b = vcross(e2b, n);
            const double e1_denom = 1.0 / (d_e1_b - d_e1_a + BP_DBL_MIN);
            const double e2_denom = 1.0 / (d_e2_b - d_e2_a + BP_DBL_MIN);
            M.n = n;
            {
                const d2 p1 = vadd(vmul(n, r1), vlerp(e1a, e1b, clamp01((d_e2_b - d_e1_a) * e1_denom)));
                const d2 p2 = vadd(vmul(n, -r2), vlerp(e2a, e2b, clamp01((d_e1_a - d_e2_a) * e2_denom)));
                const double dist = vdot(vsub(p2, p1), n);
                if (dist <= 0.0) { M.p1_0 = p1; M.p2_0 = p2; M.h0 = ((unsigned)e1ia << 8) | (unsigned)e2ib; M.count = 1; }
            }
            {
                const d2 p1 = vadd(vmul(n, r1), vlerp(e1a, e1b, clamp01((d_e2_a - d_e1_a) * e1_denom)));
                const d2 p2 = vadd(vmul(n, -r2), vlerp(e2a, e2b, clamp01((d_e1_b - d_e2_a) * e2_denom)));
                const double dist = vdot(vsub(p2, p1), n);
                if (dist <= 0.0) {
                    const unsigned h = ((unsigned)e1ib << 8) | (unsigned)e2ia;
                    if (M.count == 0) { M.p1_0 = p1; M.p2_0 = p2; M.h0 = h; M.count = 1; }
                    else { M.p1_1 = p1; M.p2_1 = p2; M.h1 = h; M.count = 2; }
                }
            }
        }
        PROF_ACC(10)
        if (KIND == BP_ENV_BOX) {
            // (1,3) robot x boundary and (2,3) box x boundary pre_solve: remember normal and contact 0 (r1, r2 relative to the
            // bodies' positions at collision time); they run after the collision phase in ascending key order
            const bool ev = valid && M.count > 0 && (flagonly || (kind_ctype(gE(E.kind, sa)) == 2 && kind_ctype(gE(E.kind, sb)) == 3));
            const unsigned long long em = ballot(ev);
            if (em) {
                const int pos = S.nev + popc_below(em, lane);
                if (ev && pos < BP_EVCAP) {
                    L.ev_key[pos] = ((unsigned)sa << 16) | (unsigned)sb;
                    L.ev_d[pos * 3 + 0] = M.n;
                    L.ev_d[pos * 3 + 1] = vsub(M.p1_0, gE(E.pxy, sa));
                    L.ev_d[pos * 3 + 2] = vsub(M.p2_0, gE(E.pxy, sb));
                }
                S.nev += __popcll(em);
                if (S.nev > BP_EVCAP) { S.nev = BP_EVCAP; S.err |= BP_ERR_ARB_OVERFLOW; }
            }
        }
        // ---- 4c. cpArbiterUpdate: hand each manifold to the lane that owns the pair's arbiter slot ---------------------
        if (ballot(valid && flagonly && M.count > 0)) S.wall_flag = 1;
        const unsigned long long dm = ballot(valid && !flagonly && M.count > 0);
        const int drank = popc_below(dm, lane);
        const int ndel = __popcll(dm);
        lds_sync(); // the mailbox aliases the plane-search scratch: all reads of it are done
        for (int dbase = 0; dbase < ndel; dbase += BP_MBOX) {
            const bool mine = valid && !flagonly && M.count > 0 && drank >= dbase && drank < dbase + BP_MBOX;
            if (mine) {
                d2 *mb = L.mbox + (drank - dbase) * 6;
                mb[0] = M.n; mb[1] = M.p1_0; mb[2] = M.p2_0; mb[3] = M.p1_1; mb[4] = M.p2_1;
                mb[5] = mk2(__hiloint2double((int)M.h0, M.count), __hiloint2double((int)M.h1, 0));
            }
            lds_sync();
            int my_mb = -1;
            bool fresh = false;
            unsigned long long m = dm;
            int dr = 0;
            const unsigned keyv = ((unsigned)sa << 16) | (unsigned)sb;
            while (m) { // every arbiter lane looks for its pair among the delivered ones; a pair nobody owns gets a free slot (rare)
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (dr >= dbase && dr < dbase + BP_MBOX) {
                    const unsigned key = (unsigned)__builtin_amdgcn_readlane((int)keyv, l);
                    const bool own = (A.key == key);
                    if (BP_UNLIKELY2(!ballot(own))) {
                        const unsigned long long om = ballot(A.key == ARB_FREE_KEY);
                        if (!om) S.err |= BP_ERR_ARB_OVERFLOW;
                        else {
                            const int owner = __ffsll((long long)om) - 1;
                            const int s1 = slot_get(L, S, E.pxy, (int)(key >> 16)), s2 = slot_get(L, S, E.pxy, (int)(key & 0xFFFFu));
                            if (lane == owner) { my_mb = dr - dbase; fresh = true; A.key = key; A.slotA = s1; A.slotB = s2; }
                        }
                    } else if (own) my_mb = dr - dbase;
                }
                dr++;
            }
            if (my_mb >= 0) {
                const d2 *mb = L.mbox + my_mb * 6;
                const d2 mn_ = mb[0], mp10 = mb[1], mp20 = mb[2], mp11 = mb[3], mp21 = mb[4], mh = mb[5];
                const d2 pa = L.sp[A.slotA], pbp = L.sp[A.slotB];
                const unsigned mh0 = (unsigned)__double2hiint(mh.x), mh1 = (unsigned)__double2hiint(mh.y);
                const int mcount = __double2loint(mh.x);
                if (BP_UNLIKELY2(fresh)) { // masses and material products of a pair stay with its arbiter
                    A.state = ARB_FIRST; A.count = 0; A.h0 = A.h1 = 0; A.jn0 = A.jt0 = A.jn1 = A.jt1 = 0.0;
                    const int usa = (int)(A.key >> 16), usb = (int)(A.key & 0xFFFFu);
                    const double4 m1 = gE(E.mass, usa), m2 = gE(E.mass, usb);
                    A.ma = m1.x; A.ia = m1.y; A.mb = m2.x; A.ib = m2.y;
                    const double4 q1 = gE(E.prop, usa), q2 = gE(E.prop, usb);
                    A.e = q1.y * q2.y; A.u = q1.z * q2.z;
                }
                double njn0 = 0.0, njt0 = 0.0, njn1 = 0.0, njt1 = 0.0;
                if (A.count > 0 && A.h0 == mh0) { njn0 = A.jn0; njt0 = A.jt0; }
                if (A.count > 1 && A.h1 == mh0) { njn0 = A.jn1; njt0 = A.jt1; }
                if (mcount > 1) {
                    if (A.count > 0 && A.h0 == mh1) { njn1 = A.jn0; njt1 = A.jt0; }
                    if (A.count > 1 && A.h1 == mh1) { njn1 = A.jn1; njt1 = A.jt1; }
                }
                A.jn0 = njn0; A.jt0 = njt0; A.jn1 = njn1; A.jt1 = njt1;
                A.h0 = mh0; A.h1 = mh1;
                A.r1_0 = vsub(mp10, pa); A.r2_0 = vsub(mp20, pbp);
                A.r1_1 = vsub(mp11, pa); A.r2_1 = vsub(mp21, pbp);
                A.count = mcount;
                A.n = mn_;
                if (A.state == ARB_CACHED) A.state = ARB_FIRST;
                A.stamp = now;
            }
            lds_sync();
        }
    }
    S.cc_ok = 1;   // the first candidate round has (re)built its cache; the moving-list rebuild and the neighbour-list refresh invalidate it
    PROF_ACC(4)
    if (KIND == BP_ENV_BOX) {
        // box-delivery: most sim steps of a robot that drives through free space have nothing to solve -- no arbiter exists (none was delivered in this sub-step,
        // none is still cached from an earlier one), no pre_solve event was recorded and only the robot's own slots are in the moving list.  Steps 5-7 then
        // reduce to what is written here: an empty active set (the colouring cache follows it), the work proxy, no ship bookkeeping, and a moving list
        // that comes out as it is (the robot's slots lead the list and are re-listed every sub-step) -- about a quarter of such a sim step's instructions.
        if (S.nev == 0 && S.nmv == P.nkin && ballot(A.key != ARB_FREE_KEY) == 0) {
            if (S.prev_amask != 0ull) { S.prev_amask = 0ull; S.nlevels = 0; }
            S.costp += 16u;
            S.ship_post = 0u; S.ship_contacts = 0u;
            S.quiescent = 0;
            PROF_ACC(9)
            return;
        }
    }
    // arbiters whose bodies did not move keep last sub-step's contacts
    if (A.key != ARB_FREE_KEY && A.stamp == now - 1u) {
        const int a = (int)(A.key >> 16), b = (int)(A.key & 0xFFFFu);
        if (L.mvs[a] != now && L.mvs[b] != now) A.stamp = now;
    }
    if (KIND == BP_ENV_BOX && S.nev > 0) {
        // prevent_boundary_intersection (box_delivery_env.py:294-311), ascending key order; every lane does the same arithmetic
        __syncthreads();
        unsigned last = 0u;
        for (int e = 0; e < S.nev; e++) {
            unsigned best = 0xFFFFFFFFu; int bi = 0;
            for (int q = 0; q < S.nev; q++) { const unsigned k = L.ev_key[q]; if ((e == 0 || k > last) && k < best) { best = k; bi = q; } }
            last = best;
            const int a = (int)(best >> 16), b = (int)(best & 0xFFFFu);
            const d2 n = L.ev_d[bi * 3 + 0], r1 = L.ev_d[bi * 3 + 1], r2 = L.ev_d[bi * 3 + 2];
            const int sl = L.slot_of[a];
            const d2 v = (sl != 255) ? L.sv[sl] : mk2(0.0, 0.0);
            const double f = 2 * (v.x * n.x + v.y * n.y);
            const d2 refl = mk2(v.x - n.x * f, v.y - n.y * f);
            const d2 nv = mk2(refl.x * 0.5, refl.y * 0.5);
            const d2 pa = gE(E.pxy, a), pb = gE(E.pxy, b);
            const double depth = vdot(vsub(vadd(pb, r2), vadd(pa, r1)), n);
            const d2 pn = mk2(pa.x + n.x * depth, pa.y + n.y * depth);
            const bool robot = a < P.nkin;
            if (robot) S.robot_hit = depth < 0;
            __syncthreads(); // all lanes have read pa before it is overwritten
            if (robot) {
                if (lane < P.nkin) { gE(E.pxy, lane) = pn; L.sp[lane] = pn; L.sv[lane] = nv; }
                if (pn.x != pa.x || pn.y != pa.y) S.evmask |= 1ull;
            } else {
                if (lane == 0) { gE(E.pxy, a) = pn; if (sl != 255) { L.sv[sl] = nv; L.sp[sl] = pn; } }
                if (pn.x != pa.x || pn.y != pa.y) S.evmask |= 1ull << a;
            }
            __syncthreads();
        }
    }
    // ---- 5. cpSpaceArbiterSetFilter ---------------------------------------------------------------------------
    if (A.key != ARB_FREE_KEY) {
        const unsigned ticks = now - A.stamp;
        if (ticks >= 1u && A.state != ARB_CACHED) A.state = ARB_CACHED;
        if (ticks >= (unsigned)P.persistence) A.key = ARB_FREE_KEY;
    }
    const bool active = (A.key != ARB_FREE_KEY) && (A.stamp == now);
    const unsigned long long amask = ballot(active);
    const int ba = (int)(A.key >> 16), bbi = (int)(A.key & 0xFFFFu);

    PROF_ACC(5)
    PROF_CNT(19, __popcll(amask))
    PROF_MAX(11, __popcll(ballot(A.key != ARB_FREE_KEY)))
    PROF_MAX(12, __popcll(amask))
    PROF_MAX(14, S.nslots)
    PROF_MAX(15, S.nmv)
    PROF_CNT(20, S.nlevels)
    // ---- 6a. prestep (cpArbiterPreStep) -----------------------------------------------------------------------
    // per-sub-step terms of the lane's arbiter: defined here for every lane, so that nothing keeps them alive across the collision phase
    double nMass0 = 0.0, tMass0 = 0.0, bias0 = 0.0, bounce0 = 0.0, jBias0 = 0.0;
    double nMass1 = 0.0, tMass1 = 0.0, bias1 = 0.0, bounce1 = 0.0, jBias1 = 0.0;
    if (active) {
        const d2 pa = L.sp[A.slotA], pb = L.sp[A.slotB];
        const d2 va = L.sv[A.slotA], vb = L.sv[A.slotB];
        const double wa = L.sw[A.slotA].x, wb = L.sw[A.slotB].x;
        const d2 n = A.n;
        const d2 body_delta = vsub(pb, pa);
        const d2 t = vperp(n);
        {
            const double rcn1 = vcross(A.r1_0, n), rcn2 = vcross(A.r2_0, n);
            nMass0 = 1.0 / ((A.ma + A.ia * rcn1 * rcn1) + (A.mb + A.ib * rcn2 * rcn2));
            const double rct1 = vcross(A.r1_0, t), rct2 = vcross(A.r2_0, t);
            tMass0 = 1.0 / ((A.ma + A.ia * rct1 * rct1) + (A.mb + A.ib * rct2 * rct2));
            const double dist = vdot(vadd(vsub(A.r2_0, A.r1_0), body_delta), n);
            bias0 = -P.bias_coef * fmin(0.0, dist + P.slop);   // divided by dt below when it is not a zero (a signed zero / dt is that zero)
            jBias0 = 0.0;
            const d2 v1 = vadd(va, vmul(vperp(A.r1_0), wa));
            const d2 v2 = vadd(vb, vmul(vperp(A.r2_0), wb));
            bounce0 = vdot(vsub(v2, v1), n) * A.e;
        }
        if (A.count > 1) {
            const double rcn1 = vcross(A.r1_1, n), rcn2 = vcross(A.r2_1, n);
            nMass1 = 1.0 / ((A.ma + A.ia * rcn1 * rcn1) + (A.mb + A.ib * rcn2 * rcn2));
            const double rct1 = vcross(A.r1_1, t), rct2 = vcross(A.r2_1, t);
            tMass1 = 1.0 / ((A.ma + A.ia * rct1 * rct1) + (A.mb + A.ib * rct2 * rct2));
            const double dist = vdot(vadd(vsub(A.r2_1, A.r1_1), body_delta), n);
            bias1 = -P.bias_coef * fmin(0.0, dist + P.slop);
            jBias1 = 0.0;
            const d2 v1 = vadd(va, vmul(vperp(A.r1_1), wa));
            const d2 v2 = vadd(vb, vmul(vperp(A.r2_1), wb));
            bounce1 = vdot(vsub(v2, v1), n) * A.e;
        }
    }
    // the bias velocity -bias_coef * min(0, dist + slop) / dt is a signed zero unless a contact is deeper than the slop: the two divisions run only then
    if (ballot(active && (bias0 != 0.0 || bias1 != 0.0))) { bias0 = bias0 / dt; bias1 = bias1 / dt; }
    // ---- warm set: arbiters that can produce a non-zero impulse this sub-step ------------------------------------
    // Seeds: a kinematic body that moves, a cached impulse, a bias or a bounce term; closed under "shares a dynamic
    // body".  Every other arbiter provably keeps all its impulses at exactly 0 and is skipped.
    bool warm = false;
    if (active) {
        warm = (A.jn0 != 0.0) || (A.jt0 != 0.0) || (bias0 != 0.0) || (bounce0 != 0.0);
        if (A.count > 1) warm = warm || (A.jn1 != 0.0) || (A.jt1 != 0.0) || (bias1 != 0.0) || (bounce1 != 0.0);
        // a velocity that survives the velocity integrate: a kinematic body's always does, a dynamic body's only with damping != 0 (a product that
        // underflows to zero there makes the arbiter warm for nothing, which is exact: warm arbiters run the full arithmetic)
        if (DAMP || A.ma == 0.0) { const d2 v = L.sv[A.slotA]; warm = warm || (v.x != 0.0) || (v.y != 0.0) || (L.sw[A.slotA].x != 0.0); }
        if (DAMP || A.mb == 0.0) { const d2 v = L.sv[A.slotB]; warm = warm || (v.x != 0.0) || (v.y != 0.0) || (L.sw[A.slotB].x != 0.0); }
        if (A.ma != 0.0) L.owner[A.slotA] = 0;
        if (A.mb != 0.0) L.owner[A.slotB] = 0;
    }
    unsigned long long wmask = ballot(warm);
    if (wmask != 0 && wmask != amask) {
        lds_sync();
        for (;;) {
            if (warm) { if (A.ma != 0.0) L.owner[A.slotA] = 1; if (A.mb != 0.0) L.owner[A.slotB] = 1; }
            lds_sync();
            if (active && !warm) warm = (A.ma != 0.0 && L.owner[A.slotA] != 0) || (A.mb != 0.0 && L.owner[A.slotB] != 0);
            const unsigned long long nm = ballot(warm);
            PROF_CNT(40, 1)
            if (nm == wmask) break;
            wmask = nm;
        }
    }
    const bool any_bias = ballot(warm && ((bias0 != 0.0) || (A.count > 1 && bias1 != 0.0))) != 0;
    // ---- solve order: greedy colouring of the active set in ascending key order (cached while the set is unchanged);
    //      arbiters of one colour share no dynamic body, so a colour runs in parallel; order = (colour, key) ----------
    if (BP_UNLIKELY(amask != S.prev_amask)) {
        PROF_CNT(41, 1)
        int rank = 0;
        unsigned long long m = amask;
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)A.key, l);
            rank += (k < A.key) ? 1 : 0;
        }
        A.rank = rank;
        lds_sync();
        if (active) { L.colmask[A.slotA] = 0; L.colmask[A.slotB] = 0; }
        lds_sync();
        const int nact = __popcll(amask);
        int nlev = 0;
        for (int r = 0; r < nact; r++) {
            const unsigned long long rm = ballot(active && A.rank == r);
            const int l = __ffsll((long long)rm) - 1;
            const bool adyn = __builtin_amdgcn_readlane(__double2hiint(A.ma), l) != 0 || __builtin_amdgcn_readlane(__double2loint(A.ma), l) != 0;
            const bool bdyn = __builtin_amdgcn_readlane(__double2hiint(A.mb), l) != 0 || __builtin_amdgcn_readlane(__double2loint(A.mb), l) != 0;
            const int a = __builtin_amdgcn_readlane(A.slotA, l), b = __builtin_amdgcn_readlane(A.slotB, l); // the bodies' slots
            const unsigned ua = adyn ? (unsigned)L.colmask[a] : 0u, ub = bdyn ? (unsigned)L.colmask[b] : 0u;
            const unsigned used = ua | ub;
            int c = __ffs(~used) - 1;
            if (c > 15) { c = 15; S.err |= BP_ERR_LEVEL_OVERFLOW; }
            if (adyn) L.colmask[a] = (unsigned short)(ua | (1u << c));
            if (bdyn) L.colmask[b] = (unsigned short)(ub | (1u << c));
            if (lane == l) A.level = c + 1;
            nlev = max(nlev, c + 1);
        }
        S.nlevels = nlev;
        S.prev_amask = amask;
    }
    PROF_CNT(21, __popcll(wmask))
    PROF_MAX(13, __popcll(wmask))
    S.costp += 16u + 2u * (unsigned)__popcll(amask) + 4u * (unsigned)(__popcll(wmask) * S.nlevels);
    lds_sync();
    // ---- 6b. velocity integrate: damping^dt == 0, no gravity/forces -> dynamic bodies' v, w := +0 ---------------
    if (DAMP) {
        // cpBodyUpdateVelocity with damping != 0: v = v * damping^dt + (g + f / m) * dt, w = w * damping^dt + t / I * dt with g = f = t = 0, i.e. "+ 0.0"
        // (a negative zero becomes +0, like the oracle's).  Every dynamic body with a non-zero velocity holds a velocity slot; bodies of slots at or above
        // nkin are dynamic or static (zero velocity: unchanged by the arithmetic).
        for (int s0 = 0; s0 < S.nslots; s0 += 64) {
            const int sl = s0 + lane;
            if (sl >= P.nkin && sl < S.nslots) {
                const d2 v = L.sv[sl], w2 = L.sw[sl];
                L.sv[sl] = mk2(v.x * P.damping_pow + 0.0, v.y * P.damping_pow + 0.0);
                L.sw[sl] = mk2(w2.x * P.damping_pow + 0.0, w2.y);
            }
        }
    } else
    for (int k0 = 0; k0 < S.nmv; k0 += 64) {
        const int k = k0 + lane;
        if (k < S.nmv) {
            const int i = L.mv[k];
            const int sl = L.slot_of[i];
            // bodies of the moving list are parts of the kinematic agent (indices below nkin) or dynamic; a static shape in the list (first sub-step
            // of a new space) has zero velocity already
            if (sl != 255 && i >= P.nkin) { L.sv[sl] = mk2(0.0, 0.0); L.sw[sl].x = 0.0; }
        }
    }
    lds_sync();
    PROF_ACC(6)
    // ---- 6c. warm start (cpArbiterApplyCachedImpulse) -----------------------------------------------------------
    const double dt_coef = (prev_dt == 0.0) ? 0.0 : (prev_dt == dt) ? 1.0 : dt / prev_dt; // x / x == 1 exactly: no division in the steady case
    const int nlevels = wmask ? S.nlevels : 0;
    unsigned lvlmask = 0; // colours that hold at least one warm arbiter
    for (int lvl = 1; lvl <= nlevels; lvl++) if (ballot(warm && A.level == lvl)) lvlmask |= 1u << lvl;
    for (unsigned lm = lvlmask; lm; lm &= lm - 1u) {
        const int lvl = __ffs((int)lm) - 1;
        if (warm && A.level == lvl && A.state != ARB_FIRST) {
            d2 va = L.sv[A.slotA], vb = L.sv[A.slotB];
            double wa = L.sw[A.slotA].x, wb = L.sw[A.slotB].x;   // the bias halves are not touched by the warm start: 8-byte accesses
            {
                const d2 j = vmul(vrotate(A.n, mk2(A.jn0, A.jt0)), dt_coef);
                apply_contact_impulses(A, 0, va, wa, vb, wb, j);
            }
            if (A.count > 1) {
                const d2 j = vmul(vrotate(A.n, mk2(A.jn1, A.jt1)), dt_coef);
                apply_contact_impulses(A, 1, va, wa, vb, wb, j);
            }
            if (A.ma != 0.0) { L.sv[A.slotA] = va; L.sw[A.slotA].x = wa; }
            if (A.mb != 0.0) { L.sv[A.slotB] = vb; L.sw[A.slotB].x = wb; }
        }
        lds_sync();
    }
    PROF_ACC(7)
    // ---- 6d. sequential impulses (cpArbiterApplyImpulse) ------------------------------------------------------
    // two copies of the loop: with and without bias terms (wave-uniform, fixed for the sub-step) -> no per-contact uniform branches
    // Write-back slots of a solver pass: the (unchanged) velocity of an infinite-mass body goes to the scratch slot BP_NSLOT instead of
    // being predicated away -- two EXEC-mask branches less per pass (+2.8 % env-steps/s, same-box A/B).  Not for box-delivery, whose
    // kernel sits at the 256-VGPR limit of two waves per SIMD.
    constexpr bool SCRATCH_WB = (KIND != BP_ENV_BOX);
    const int wA = (!SCRATCH_WB || A.ma != 0.0) ? A.slotA : BP_NSLOT, wB = (!SCRATCH_WB || A.mb != 0.0) ? A.slotB : BP_NSLOT;
    auto iterate = [&](auto bias_tag) {
    constexpr bool AB = decltype(bias_tag)::value;
    // the three parts of a colour pass for the lane's arbiter: velocities of its two bodies from their slots, cpArbiterApplyImpulse for its one or two contacts,
    // velocities back to the slots (an infinite-mass body's to the scratch slot)
    auto gather = [&](d2 &va, d2 &vb, d2 &wa2, d2 &wb2, d2 &vba, d2 &vbb) {
        va = L.sv[A.slotA]; vb = L.sv[A.slotB];
        wa2 = mk2(0.0, 0.0); wb2 = mk2(0.0, 0.0); vba = mk2(0.0, 0.0); vbb = mk2(0.0, 0.0);
        if (AB) { wa2 = L.sw[A.slotA]; wb2 = L.sw[A.slotB]; vba = L.sb[A.slotA]; vbb = L.sb[A.slotB]; }
        else { wa2.x = L.sw[A.slotA].x; wb2.x = L.sw[A.slotB].x; }   // without bias terms only the angular velocity itself is read and written back: 8-byte LDS accesses
    };
    auto scatter = [&](const d2 va, const d2 vb, const d2 wa2, const d2 wb2, const d2 vba, const d2 vbb) {
        if (SCRATCH_WB) {
            L.sv[wA] = va; if (AB) { L.sw[wA] = wa2; L.sb[wA] = vba; } else L.sw[wA].x = wa2.x;
            L.sv[wB] = vb; if (AB) { L.sw[wB] = wb2; L.sb[wB] = vbb; } else L.sw[wB].x = wb2.x;
        } else {
            if (A.ma != 0.0) { L.sv[A.slotA] = va; if (AB) { L.sw[A.slotA] = wa2; L.sb[A.slotA] = vba; } else L.sw[A.slotA].x = wa2.x; }
            if (A.mb != 0.0) { L.sv[A.slotB] = vb; if (AB) { L.sw[A.slotB] = wb2; L.sb[A.slotB] = vbb; } else L.sw[A.slotB].x = wb2.x; }
        }
    };
    // "did any accumulated impulse change in this iteration": the differences jnAcc - jnOld, jtAcc - jtOld (and of the bias impulse) exist anyway;
    // a difference of finite doubles is zero exactly when they are equal, and a sum of magnitudes is zero exactly when every term is (no
    // cancellation, no underflow: binary64 keeps subnormals; a NaN stays non-zero), so their absolute values are accumulated into one per-lane
    // double -- two additions per contact instead of the four integer operations of the bit-pattern form, and no compare-and-merge of lane masks
    auto contacts = [&](d2 &va, d2 &vb, d2 &wa2, d2 &wb2, d2 &vba, d2 &vbb, double &chg) {
        const d2 n = A.n;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c == 0 || A.count > 1) { // an arbiter always has its first contact
                const d2 r1 = c ? A.r1_1 : A.r1_0, r2 = c ? A.r2_1 : A.r2_0;
                const double nMass = c ? nMass1 : nMass0, tMass = c ? tMass1 : tMass0;
                const double bias = c ? bias1 : bias0, bounce = c ? bounce1 : bounce0;
                const d2 v1 = vadd(va, vmul(vperp(r1), wa2.x));
                const d2 v2 = vadd(vb, vmul(vperp(r2), wb2.x));
                const d2 vr = vsub(v2, v1);
                const double vrn = vdot(vr, n);
                const double vrt = vdot(vr, vperp(n));
                const double jbnOld = c ? jBias1 : jBias0;
                double jBias = jbnOld;
                if (AB) { // with no bias term anywhere every bias impulse stays exactly 0
                    const d2 vb1 = vadd(vba, vmul(vperp(r1), wa2.y));
                    const d2 vb2 = vadd(vbb, vmul(vperp(r2), wb2.y));
                    const double vbn = vdot(vsub(vb2, vb1), n);
                    const double jbn = (bias - vbn) * nMass;
                    jBias = fmax(jbnOld + jbn, 0.0);
                }
                const double jn = -(bounce + vrn) * nMass;
                const double jnOld = c ? A.jn1 : A.jn0;
                const double jnAcc = fmax(jnOld + jn, 0.0);
                const double jtMax = A.u * jnAcc;
                const double jt = -vrt * tMass;
                const double jtOld = c ? A.jt1 : A.jt0;
                const double jtAcc = fclampd(jtOld + jt, -jtMax, jtMax);
                if (c) { jBias1 = jBias; A.jn1 = jnAcc; A.jt1 = jtAcc; }
                else   { jBias0 = jBias; A.jn0 = jnAcc; A.jt0 = jtAcc; }
                if (AB) {
                    const double djb = jBias - jbnOld;
                    chg += __builtin_fabs(djb);
                    const d2 jb = vmul(n, djb);
                    const d2 jbneg = vneg(jb);
                    vba = vadd(vba, vmul(jbneg, A.ma));
                    wa2.y += A.ia * vcross(r1, jbneg);
                    vbb = vadd(vbb, vmul(jb, A.mb));
                    wb2.y += A.ib * vcross(r2, jb);
                }
                const double djn = jnAcc - jnOld, djt = jtAcc - jtOld;
                chg += __builtin_fabs(djn); chg += __builtin_fabs(djt);   // two adds with |.| as a source modifier; off the pass's dependency chain
                const d2 j = vrotate(n, mk2(djn, djt));
                apply_contact_impulses(A, c, va, wa2.x, vb, wb2.x, j);
            }
        }
    };
    // ---- one warm colour: the warm arbiters share no dynamic body, so nobody else reads or writes the velocities a lane works on between its passes --
    // they stay in registers for the ten iterations (one gather, one scatter; a third of a pass is its LDS round trip, profiles/r04_floor).  An
    // infinite-mass body's velocity is re-read unchanged by every pass of the loop below, while here the (zero) impulse is added to the lane's copy:
    // x + (+-0) == x bit for bit unless a component of x is a negative zero, so lanes check their infinite-mass sides once and the wave takes
    // the slot loop if any such component exists (a kinematic body commanded with -0.0).
    // (no-bias copy only, and not in the box-delivery instantiation: the extra live registers of the other copies push those kernels over the 256-VGPR line)
    if (!AB && KIND != BP_ENV_BOX && (lvlmask & (lvlmask - 1u)) == 0u && lvlmask != 0u) {
        d2 va = mk2(0.0, 0.0), vb = va, wa2 = va, wb2 = va, vba = va, vbb = va;
        if (warm) gather(va, vb, wa2, wb2, vba, vbb);
        auto negzero = [](double x) { return (((unsigned)__double2hiint(x) ^ 0x80000000u) | (unsigned)__double2loint(x)) == 0u; };
        bool nz = false;
        if (warm && A.ma == 0.0) nz = negzero(va.x) || negzero(va.y) || negzero(wa2.x) || (AB && (negzero(wa2.y) || negzero(vba.x) || negzero(vba.y)));
        if (warm && A.mb == 0.0) nz = nz || negzero(vb.x) || negzero(vb.y) || negzero(wb2.x) || (AB && (negzero(wb2.y) || negzero(vbb.x) || negzero(vbb.y)));
        if (!ballot(nz)) {
            for (int it = 0; it < P.iterations; it++) {
                PROF_CNT(42, 1)
                PROF_CNT(43, 1)
                double chg = 0.0;
                if (warm) contacts(va, vb, wa2, wb2, vba, vbb, chg);
                if (BP_UNLIKELY2(!ballot(warm && chg != 0.0))) break;
            }
            if (warm) scatter(va, vb, wa2, wb2, vba, vbb);
            lds_sync();
            return;
        }
    }
    for (int it = 0; it < P.iterations; it++) {
        PROF_CNT(42, 1)
        double chg = 0.0;
        for (unsigned lm = lvlmask; lm; lm &= lm - 1u) { // colours that hold a warm arbiter, ascending
            const int lvl = __ffs((int)lm) - 1;
            PROF_CNT(43, 1)
            PROF_CNT(44, ballot(warm && A.level == lvl && A.count > 1) ? 1 : 0)
            if (warm && A.level == lvl) {
                d2 va, vb, wa2, wb2, vba, vbb;
                gather(va, vb, wa2, wb2, vba, vbb);
                contacts(va, vb, wa2, wb2, vba, vbb, chg);
                scatter(va, vb, wa2, wb2, vba, vbb);
            }
            lds_sync();
        }
        // an iteration that changed no accumulated impulse applied only zero impulses: the state is a fixed point and
        // the remaining iterations would repeat it exactly
        if (BP_UNLIKELY2(!ballot(warm && chg != 0.0))) break;
    }
    };
    if (any_bias) iterate(std::true_type{}); else iterate(std::false_type{});
    PROF_ACC(8)
    // ---- 7. post-solve bookkeeping for ship(0) x floe arbiters, ascending key order ------------------------------
    {
        const bool shiparb = active && ba == 0;
        const unsigned long long sm = ballot(shiparb);
        S.ship_post = (unsigned)__popcll(sm);
        S.ship_contacts = (unsigned)__popcll(sm) + (unsigned)__popcll(ballot(shiparb && A.count > 1));
        if (sm) {
            // integer bookkeeping is order-free; cold arbiters add exactly +0 to the float sums
            S.n_post += (unsigned)__popcll(sm);
            S.n_contact += (unsigned)__popcll(sm) + (unsigned)__popcll(ballot(shiparb && A.count > 1));
            S.n_first += (unsigned)__popcll(ballot(shiparb && A.state == ARB_FIRST));
            const bool ws = shiparb && warm;
            const unsigned long long wsm = ballot(ws);
            if (wsm) {
                // (1 - e) / (1 + e): one division per distinct elasticity product instead of one per sub-step (all ship x floe arbiters share it)
                const double e_first = readlane_f64(A.e, __ffsll((long long)wsm) - 1);
                double eCoef;
                if (!ballot(ws && A.e != e_first)) {
                    if (e_first != S.ecoef_e) { S.ecoef_e = e_first; S.ecoef = (1 - e_first) / (1 + e_first); }
                    eCoef = S.ecoef;
                } else eCoef = (1 - A.e) / (1 + A.e);
                double ke = 0.0;
                d2 js = mk2(0.0, 0.0);
                if (ws) {
                    ke += eCoef * A.jn0 * A.jn0 / nMass0 + A.jt0 * A.jt0 / tMass0;
                    js = vadd(js, vrotate(A.n, mk2(A.jn0, A.jt0)));
                    if (A.count > 1) {
                        ke += eCoef * A.jn1 * A.jn1 / nMass1 + A.jt1 * A.jt1 / tMass1;
                        js = vadd(js, vrotate(A.n, mk2(A.jn1, A.jt1)));
                    }
                }
                const double imp = vlen(js);
                const int ns = __popcll(sm);
                for (int r = 0; r < ns; r++) { // ship arbiters have the smallest keys of the active set: ranks 0..ns-1
                    const unsigned long long rm = ballot(ws && A.rank == r);
                    if (!rm) continue;
                    const int l = __ffsll((long long)rm) - 1;
                    S.total_ke += readlane_f64(ke, l);
                    S.total_imp += readlane_f64(imp, l);
                }
            }
        }
    }
    // ---- agent rules applied after every sub-step: yaw limits + channel boundary (ship_ice_env.py:284-290),
    //      boundary only for the maze robot (maze_NAMO_env.py:417-419) ------------------------------------------------
    if (ship_rules) {
        const double a0 = L.ag[0].x;
        const double x0 = L.sp[0].x;
        if (KIND == BP_ENV_SHIP_ICE && ship_rules == 1 && (a0 <= 0.0 || a0 >= BP_PI)) {
            if (lane < P.nkin) L.sw[lane] = mk2(0.0, L.sw[lane].y);
            S.yaw_violated = 1;
        }
        if (x0 < 0.0 || x0 > P.map_w) S.boundary_violated = 1;
    }
    lds_sync();
    // ---- next sub-step's moving list: bodies of active arbiters with a non-zero velocity, plus the ship ----------
    {
        bool wantA = false, wantB = false;
        if (active) {
            if (A.ma != 0.0) {
                const d2 v = L.sv[A.slotA], w2 = L.sw[A.slotA], vb = L.sb[A.slotA];
                wantA = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
            }
            if (A.mb != 0.0) {
                const d2 v = L.sv[A.slotB], w2 = L.sw[A.slotB], vb = L.sb[A.slotB];
                wantB = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
            }
            if (KIND == BP_ENV_BOX) { // a body moved by a pre_solve push-out must be re-cached next sub-step (it has an arbiter)
                if (A.ma != 0.0 && ((S.evmask >> ba) & 1ull)) wantA = true;
                if (A.mb != 0.0 && ((S.evmask >> bbi) & 1ull)) wantB = true;
            }
        }
        const d2 v0 = L.sv[0], w0 = L.sw[0];
        // every part of the kinematic agent; the box-delivery robot is re-cached every sub-step (its controller rewrites the
        // velocity between sub-steps; re-evaluating an unmoved body reproduces the carried-over result exactly)
        const int shipmv = (KIND == BP_ENV_BOX || v0.x != 0.0 || v0.y != 0.0 || w0.x != 0.0) ? P.nkin : 0;
        // the list is rewritten in place; if it comes out as it was, the candidate cache of the first candidate round stays valid
        bool differs = false;
        if (lane < shipmv) { differs = L.mv[lane] != (unsigned short)lane; L.mv[lane] = (unsigned short)lane; }
        int nA_ = 0;
        unsigned long long mB = 0ull;
        if (DAMP) {
            // damping != 0: a body keeps its velocity without any arbiter -- every velocity slot whose body still moves joins the list, in slot order
            // (the order of the list only permutes independent work: pairs are keyed by body, arbiters are solved in (colour, key) order)
            int pos0 = shipmv;
            for (int s0 = 0; s0 < S.nslots; s0 += 64) {
                const int sl = s0 + lane;
                bool want = false;
                unsigned short body = 0;
                if (sl >= P.nkin && sl < S.nslots) {
                    body = L.sbody[sl];
                    const d2 v = L.sv[sl], w2 = L.sw[sl], vb = L.sb[sl];
                    want = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
                    if (KIND == BP_ENV_BOX && ((S.evmask >> body) & 1ull)) want = true;   // moved by a pre_solve push-out: re-cached next sub-step
                }
                const unsigned long long m = ballot(want);
                if (want) {
                    const int pos = pos0 + popc_below(m, lane);
                    if (pos < P.mvcap) { differs = differs || L.mv[pos] != body; L.mv[pos] = body; }
                }
                pos0 += __popcll(m);
            }
            if (pos0 > P.mvcap) { S.err |= BP_ERR_ARB_OVERFLOW; pos0 = P.mvcap; }
            nA_ = pos0 - shipmv;
        } else {
        // a body that several arbiters want joins the list once: the first claim of this sub-step's stamp wins (one LDS atomic per side)
        const bool gotA = wantA && atomicMax(&L.mvo[A.slotA], now) < now;
        const bool gotB = wantB && atomicMax(&L.mvo[A.slotB], now) < now;
        const unsigned long long mA = ballot(gotA);
        mB = ballot(gotB);
        nA_ = __popcll(mA);
        if (gotA) { const int pos = shipmv + popc_below(mA, lane); differs = differs || L.mv[pos] != (unsigned short)ba; L.mv[pos] = (unsigned short)ba; }
        if (gotB) { const int pos = shipmv + nA_ + popc_below(mB, lane); differs = differs || L.mv[pos] != (unsigned short)bbi; L.mv[pos] = (unsigned short)bbi; }
        }
        const int newn = shipmv + nA_ + __popcll(mB);
        if (newn != S.nmv || ballot(differs)) S.cc_ok = 0;
        S.nmv = newn;
    }
    S.quiescent = (S.nmv == 0) && (wmask == 0);
    lds_sync();
    PROF_ACC(9)
}
