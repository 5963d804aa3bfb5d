// Kernels: k_physics (env.step()/reset() physics + work/reward) and k_observe (4x150x150 u8 raster).
#pragma once
#include "bp_physics.hpp"

enum { MODE_STEP = 0, MODE_RESET = 1 };

extern __shared__ double2 bp_smem[];

// numpy restatements with sequential sums (geometry/polygon.py:25-41)
__device__ __forceinline__ double poly_area_seq(const d2 *v, int n)
{
    double d1 = 0.0, d2_ = 0.0;
    for (int i = 0; i < n; i++) {
        const int p = (i - 1 + n) % n;
        d1 += v[i].x * v[p].y;
        d2_ += v[i].y * v[p].x;
    }
    return 0.5 * __builtin_fabs(d1 - d2_);
}
__device__ __forceinline__ d2 poly_centroid_seq(const d2 *v, int n)
{
    const double A = poly_area_seq(v, n);
    double sx = 0.0, sy = 0.0;
    for (int i = 0; i < n; i++) {
        const int p = (i - 1 + n) % n;
        const double u = v[i].x * v[p].y - v[p].x * v[i].y;
        sx += (v[i].x + v[p].x) * u;
        sy += (v[i].y + v[p].y) * u;
    }
    const double f = 1.0 / (6.0 * A);
    return mk2(__builtin_fabs(f * sx), __builtin_fabs(f * sy));
}

#include "bp_physics_pair.hpp"

#ifdef BP_PROF
#define BP_PROF_ON true
#else
#define BP_PROF_ON false
#endif
template <int KIND>
__device__ __forceinline__ void carve_lds(const DevParams &P, LdsCtx &L)
{
    const LdsMap m = bp_lds_map(P.nbcap, P.mvcap, KIND == BP_ENV_BOX, BP_PROF_ON);
    char *b = (char *)bp_smem;
    L.sv = (d2 *)(b + m.sv); L.sw = (d2 *)(b + m.sw); L.sb = (d2 *)(b + m.sb); L.sp = (d2 *)(b + m.sp); L.ag = (d2 *)(b + m.ag);
    L.tf = (d2 *)(b + m.tf);
    L.q_dir = (d2 *)(b + m.q_dir); L.q_c = (double *)(b + m.q_c); L.r_val = (double *)(b + m.r_val);
    L.q_meta = (unsigned *)(b + m.q_meta); L.q_aux = (unsigned *)(b + m.q_aux); L.r_idx = (unsigned *)(b + m.r_idx);
    L.pt_a = (uint4 *)(b + m.pt_a); L.pt_thr = (d2 *)(b + m.pt_thr);
    L.cc = (unsigned long long *)(b + m.cc); L.cc_hw = (unsigned long long *)(b + m.cc_hw);
    L.mbox = (d2 *)(b + m.q_dir);   // BP_MBOX * 96 B = 1536 B = the q_dir array
    L.res_smA = (unsigned long long *)(b + m.res_smA); L.res_smB = (unsigned long long *)(b + m.res_smB);
    L.res_iA = (unsigned *)(b + m.res_iA); L.res_iB = (unsigned *)(b + m.res_iB);
    L.res_jA = (unsigned *)(b + m.res_jA); L.res_jB = (unsigned *)(b + m.res_jB);
    L.mvs = (unsigned *)(b + m.mvs);
    L.owner = (unsigned short *)(b + m.owner); L.colmask = (unsigned short *)(b + m.colmask);
    L.mv = (unsigned short *)(b + m.mv); L.mvo = (unsigned *)(b + m.mvo); L.sbody = (unsigned short *)(b + m.sbody);
    L.slot_of = (unsigned char *)(b + m.slot_of); L.rf = (unsigned char *)(b + m.rf);
    L.ev_key = nullptr; L.ev_d = nullptr; L.ctl = nullptr;
    L.snap = nullptr;
    if (KIND == BP_ENV_BOX) { L.ev_d = (d2 *)(b + m.ev_d); L.ev_key = (unsigned *)(b + m.ev_key); L.ctl = (double *)(b + m.ctl); L.snap = (unsigned long long *)(b + m.snap); }
#ifdef BP_PROF
    L.prof = (unsigned long long *)(b + m.prof);
#endif
}

__device__ __forceinline__ void env_ctx(const DevParams &P, const DevPtrs &D, int env, int trial, EnvCtx &E)
{
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap, tb = (size_t)trial * nbcap;
    E.nv = D.sc_nv + tb;
    E.lv = D.sc_lv + tb * BP_MAXV;
    E.ln = D.sc_ln + tb * BP_MAXV;
    E.mass = D.sc_mass + tb;
    E.prop = D.sc_prop + tb;
    E.kind = D.sc_kind + tb;
    E.pxy = D.pxy + eb; E.rot = D.rot + eb; E.ang = D.ang + eb;
    E.wv = D.wv + eb * BP_MAXV; E.wn = D.wn + eb * BP_MAXV; E.pv = D.pv + eb * BP_MAXV;
    E.bb = D.bb + eb; E.fat = D.fat + eb;
    E.adj = D.adj + eb * BP_KADJ; E.adjn = D.adjn + eb; E.hint = D.hint + eb * BP_KADJ;
}

__device__ __forceinline__ void init_regs(ArbReg &A, SubState &S)
{
    S.costp = 0u;
    S.cc_ok = 0; S.cc_kmax = 0;
    S.ecoef_e = -1.0; S.ecoef = 0.0; S.quiescent = 0; S.ship_post = 0; S.ship_contacts = 0; S.wall_flag = 0; S.nev = 0; S.robot_hit = 0; S.evmask = 0ull;
    A.e = 0.0; A.u = 0.0;
    S.err = 0; S.yaw_violated = 0; S.boundary_violated = 0; S.prev_amask = 0; S.nlevels = 0;
    A.level = 0; A.rank = 0;
    A.ma = A.ia = A.mb = A.ib = 0.0;

}

// persistent per-env state -> LDS / registers (first half: before the agent's control is written)
template <int KIND>
__device__ __forceinline__ void load_state_a(const DevParams &P, const DevPtrs &D, const EnvCtx &E, const LdsCtx &L, ArbReg &A,
                                             SubState &S, int env)
{
    const int lane = lane_id();
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap;
    // ---- load persistent state ----
    for (int base = 0; base < nbcap; base += 64) {
        const int i = base + lane;
        if (i < nbcap) { L.mvs[i] = 0u; L.slot_of[i] = (i < P.nkin) ? (unsigned char)i : 255; }
    }
    for (int i = lane; i < BP_NSLOT + 2; i += 64) { L.mvo[i] = 0u; L.sbody[i] = (unsigned short)i; }   // slots [0, nkin) belong to bodies [0, nkin)
    if (lane < P.nkin) { L.sv[lane] = D.velv[eb + lane]; L.sw[lane] = D.velw[eb + lane]; L.sb[lane] = D.velb[eb + lane]; L.sp[lane] = gE(E.pxy, lane); }
    if (lane == 0) { L.ag[0] = mk2(gE(E.ang, 0), 0.0); L.ag[1] = gE(E.rot, 0); }
    S.nslots = P.nkin;
    S.wall_flag = (P.env_kind == BP_ENV_MAZE) ? (D.e_flags[env] & 1) : 0;
    const size_t ab = (size_t)env * BP_ACAP + lane;
    A.key = D.a_key[ab]; A.stamp = D.a_stamp[ab];
    { const unsigned sc = D.a_sc[ab]; A.state = (int)(sc & 0xFF); A.count = (int)(sc >> 8); }
    A.h0 = D.a_h0[ab]; A.h1 = D.a_h1[ab];
    const double *ad = D.a_d + ab * 14;
    A.jn0 = ad[0]; A.jt0 = ad[1]; A.jn1 = ad[2]; A.jt1 = ad[3];
    A.n = mk2(ad[4], ad[5]);
    A.r1_0 = mk2(ad[6], ad[7]); A.r2_0 = mk2(ad[8], ad[9]); A.r1_1 = mk2(ad[10], ad[11]); A.r2_1 = mk2(ad[12], ad[13]);
    A.slotA = A.slotB = 0;
    if (A.key != ARB_FREE_KEY) {
        const double4 m1 = gE(E.mass, A.key >> 16), m2 = gE(E.mass, A.key & 0xFFFFu);
        A.ma = m1.x; A.ia = m1.y; A.mb = m2.x; A.ib = m2.y;
        const double4 q1 = gE(E.prop, A.key >> 16), q2 = gE(E.prop, A.key & 0xFFFFu);
        A.e = q1.y * q2.y; A.u = q1.z * q2.z;
    }
    S.stamp = D.e_stamp[env]; S.curr_dt = D.e_currdt[env];
    S.total_ke = D.e_ke[env]; S.total_imp = D.e_imp[env];
    S.n_post = D.e_cnt[env * 4 + 0]; S.n_contact = D.e_cnt[env * 4 + 1]; S.n_first = D.e_cnt[env * 4 + 2];
    __syncthreads();
}
// second half: moving list, velocity slots of moving bodies and of the persisted arbiters
template <int KIND>
__device__ __forceinline__ void load_state_b(const DevParams &P, const DevPtrs &D, const EnvCtx &E, const LdsCtx &L, ArbReg &A,
                                             SubState &S, int env)
{
    const int lane = lane_id();
    const size_t eb = (size_t)env * P.nbcap;
    // moving list: every body with a non-zero velocity gets a velocity slot (the ship owns slot 0)
    int n = 0;
    for (int base = 0; base < E.nb; base += 64) {
        const int i = base + lane;
        bool mvg = false;
        d2 v = mk2(0.0, 0.0), w2 = v, vb = v;
        if (i < E.nb) {
            if (i < P.nkin) { v = L.sv[i]; w2 = L.sw[i]; vb = L.sb[i]; }
            else { v = D.velv[eb + i]; w2 = D.velw[eb + i]; vb = D.velb[eb + i]; }
            mvg = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
            if (KIND == BP_ENV_BOX && i < P.nkin) mvg = true; // the robot is re-cached every sub-step (see substep)
        }
        const unsigned long long m = ballot(mvg);
        const unsigned long long ms = ballot(mvg && i >= P.nkin);
        if (mvg) { const int pos = n + popc_below(m, lane); if (pos < P.mvcap) L.mv[pos] = (unsigned short)i; }
        if (mvg && i >= P.nkin) {
            int sl = S.nslots + popc_below(ms, lane);
            if (sl >= BP_NSLOT) { S.err |= BP_ERR_ARB_OVERFLOW; sl = BP_NSLOT - 1; }
            L.slot_of[i] = (unsigned char)sl; L.sbody[sl] = (unsigned short)i;
            L.sv[sl] = v; L.sw[sl] = w2; L.sb[sl] = vb; L.sp[sl] = gE(E.pxy, i);
        }
        n += __popcll(m);
        S.nslots = min(S.nslots + __popcll(ms), BP_NSLOT);
    }
    if (n > P.mvcap) { S.err |= BP_ERR_ARB_OVERFLOW; n = P.mvcap; }
    S.nmv = n;
    lds_sync();
    // velocity slots for the bodies of the persisted arbiters
    {
        unsigned long long m = ballot(A.key != ARB_FREE_KEY);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const unsigned key = (unsigned)__builtin_amdgcn_readlane((int)A.key, l);
            const int s1 = slot_get(L, S, E.pxy, (int)(key >> 16)), s2 = slot_get(L, S, E.pxy, (int)(key & 0xFFFFu));
            if (lane == l) { A.slotA = s1; A.slotB = s2; }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void store_state(const DevParams &P, const DevPtrs &D, const LdsCtx &L, const ArbReg &A, int env)
{
    const int lane = lane_id();
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap;
    // ---- write back persistent state ----
    for (int base = 0; base < nbcap; base += 64) {
        const int i = base + lane;
        if (i < nbcap) {
            const int sl = L.slot_of[i];
            const d2 z = mk2(0.0, 0.0);
            D.velv[eb + i] = (sl != 255) ? L.sv[sl] : z; D.velw[eb + i] = (sl != 255) ? L.sw[sl] : z; D.velb[eb + i] = (sl != 255) ? L.sb[sl] : z;
        }
    }
    {
        const size_t ab = (size_t)env * BP_ACAP + lane;
        D.a_key[ab] = A.key; D.a_stamp[ab] = A.stamp; D.a_sc[ab] = (unsigned)A.state | ((unsigned)A.count << 8);
        D.a_h0[ab] = A.h0; D.a_h1[ab] = A.h1;
        double *ad = D.a_d + ab * 14;
        ad[0] = A.jn0; ad[1] = A.jt0; ad[2] = A.jn1; ad[3] = A.jt1; ad[4] = A.n.x; ad[5] = A.n.y;
        ad[6] = A.r1_0.x; ad[7] = A.r1_0.y; ad[8] = A.r2_0.x; ad[9] = A.r2_0.y;
        ad[10] = A.r1_1.x; ad[11] = A.r1_1.y; ad[12] = A.r2_1.x; ad[13] = A.r2_1.y;
    }
}

// ---- queues of the preemptive step scheduler (k_physics_step_sched below) -----------------------------------------------------------------
// D.sq_ctr: per XCD x SQ_MAXLEV + 2 rows of two ints: rows 0 .. SQ_MAXLEV-1 = (head, tail) of the queue of envs that have completed that many
// chunks; row SQ_MAXLEV of XCD x = (envs waiting in any of x's queues, -); XCD 0 additionally keeps the launch-wide counters in row SQ_MAXLEV + 1 =
// (envs finished, first chunks started).  D.sq_items: [16][SQ_MAXLEV][sq_cap] rings of env | priority class << 24, -1 = empty (claimed but not yet written, or taken).
__device__ __forceinline__ int sq_xcc_id()
{
#if defined(__gfx942__) || defined(__gfx950__)
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7u);
#else
    return 0; // no XCC id register: one queue set (the library is built for gfx950; this keeps other targets assembling)
#endif
}
__device__ __forceinline__ int sq_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int *sq_row(const DevPtrs &D, const int x, const int row) { return D.sq_ctr + ((size_t)x * (SQ_MAXLEV + 2) + row) * 2; }
__device__ __forceinline__ int *sq_finished(const DevPtrs &D) { return sq_row(D, 0, SQ_MAXLEV + 1); }
__device__ __forceinline__ int *sq_started(const DevPtrs &D) { return sq_row(D, 0, SQ_MAXLEV + 1) + 1; }
__device__ __forceinline__ int *sq_waiting(const DevPtrs &D, const int x) { return sq_row(D, x, SQ_MAXLEV); }
__device__ __forceinline__ int *sq_abort(const DevPtrs &D) { return sq_row(D, 1, SQ_MAXLEV + 1); }   // set by the watchdog: every poller leaves
// Queue rows: SQ_NX = 16 = 8 XCDs x 2 kinds.  Kind 0 (rows 0..7) holds envs that run in a wavefront of their own, kind 1 (rows 8..15, used with
// P.pair_mode == 2) envs that were light when they were parked: a workgroup that takes one of those takes a second one along and runs the two in one wavefront.
#define SQ_NX 16
// lane 0: take a waiting env of row xk that has completed exactly l chunks; -1 = none
__device__ __forceinline__ int sq_pop_level(const DevParams &P, const DevPtrs &D, const int xk, const int l)
{
    int *ctr = sq_row(D, xk, l);
    int h = sq_ld(ctr);
    while (h < sq_ld(ctr + 1)) {
        const int got = atomicCAS(ctr, h, h + 1);
        if (got == h) {
            // rows are rings of P.sq_cap entries: an env waits in at most one row at a time, so a row never holds more than num_envs = sq_cap live
            // entries however often the same env is queued at the same level (pairing: a mate that leaves at an arbitrary sub-step, a pair declined
            // at load, a paired wave that yields mid-chunk; longest-remaining-first keys repeat too).  The popper TAKES the entry (exchange with -1),
            // the pusher waits for an empty one: entries of a row are interchangeable, so two poppers that meet on one slot a lap apart are harmless.
            int *slot = D.sq_items + ((size_t)xk * SQ_MAXLEV + l) * P.sq_cap + (int)((unsigned)h % (unsigned)P.sq_cap);
            int e;
            while ((e = atomicExch(slot, -1)) < 0) __builtin_amdgcn_s_sleep(1); // the pusher has taken the index, the id is on its way
            atomicSub(sq_waiting(D, xk), 1);
            return e;
        }
        h = got;
    }
    return -1;
}
// lane 0: take the waiting env of XCD x that has completed the fewest chunks (either kind; kind 0 first on a tie); -1 = nothing waiting
__device__ __forceinline__ int sq_pop(const DevParams &P, const DevPtrs &D, const int x, int &lev, int &kind)
{
    const bool any0 = sq_ld(sq_waiting(D, x)) > 0, any1 = P.pair_mode == 2 && sq_ld(sq_waiting(D, x + 8)) > 0;
    if (!any0 && !any1) return -1;
    const int nrows = P.sq_lrpt ? SQ_MAXLEV : P.sq_levels;
    for (int l = 0; l < nrows; l++) {
        if (any0) { const int e = sq_pop_level(P, D, x, l); if (e >= 0) { lev = l; kind = 0; return e; } }
        if (any1) { const int e = sq_pop_level(P, D, x + 8, l); if (e >= 0) { lev = l; kind = 1; return e; } }
    }
    return -1;
}
__device__ __forceinline__ void sq_push(const DevParams &P, const DevPtrs &D, const int xk, const int lev, const int item)
{
    const int idx = atomicAdd(sq_row(D, xk, lev) + 1, 1);
    int *slot = D.sq_items + ((size_t)xk * SQ_MAXLEV + lev) * P.sq_cap + (int)((unsigned)idx % (unsigned)P.sq_cap);
    // ring entry of a lap ago: empty unless its popper has claimed the index and not yet taken the id (see sq_pop_level) -- bounded wait, never past the row
    for (int spin = 0; atomicCAS(slot, -1, item) != -1; spin++) {
        if (spin > (1 << 22)) { atomicOr(&D.e_err[0], BP_ERR_SCHED_TIMEOUT); return; }   // cannot happen by the invariant above; never write anywhere else
        __builtin_amdgcn_s_sleep(1);
    }
    atomicAdd(sq_waiting(D, xk), 1);
}
// lane 0: is some env behind one that has completed `lev` chunks?  Envs whose first chunk has not been dispatched yet are behind everybody.
// Longest-remaining-first keys (P.sq_lrpt): a queue row is not "chunks completed" but 15 - (estimated remaining run time of the env's step / P.sq_bw), so the
// ascending scan of sq_pop takes the env with the MOST work left, and a running wave yields only to a waiting env that has at least P.sq_hyst rows more left
// than itself (or to a not-yet-started env whose previous step cost more than that).  The estimate is the env's own pace in this step: cycles so far / sub-steps
// so far x sub-steps left.  With least-advanced-first every env advances at the same pace and the launch ends when the heaviest env does, which then has
// waited its fair share (3-5 ms of 17: tools/sched_trace.py); with longest-remaining-first the heavy envs run through and the light ones fill the slots.
__device__ __forceinline__ int sq_lrpt_key(const DevParams &P, const unsigned long long rem_units) { return 15 - (int)min(15ull, rem_units / (unsigned long long)P.sq_bw); }
__device__ __forceinline__ bool sq_someone_heavier(const DevParams &P, const DevPtrs &D, const int x, const int my_key, const unsigned long long my_rem)
{
    bool any = false;
    const int started = sq_ld(sq_started(D));
    if (started < P.num_envs) {   // the next first task of the dispatch order (heaviest first): what its env cost in the previous step
        if (D.order == nullptr) any = true;
        else any = (unsigned long long)D.e_cost[D.order[started]] > my_rem + (unsigned long long)P.sq_bw * (unsigned)P.sq_hyst;
    }
    const int upto = my_key - P.sq_hyst;   // rows 0 .. upto hold envs with at least sq_hyst rows more left
    if (!any && sq_ld(sq_waiting(D, x)) > 0)
        for (int l = 0; l <= upto; l++) { const int *ctr = sq_row(D, x, l); any = any || (sq_ld(ctr) < sq_ld(ctr + 1)); }
    return any;
}
// `hold`: the caller keeps its slot against envs that merely have not started yet (it still yields to envs that wait in the queues)
__device__ __forceinline__ bool sq_someone_behind(const DevParams &P, const DevPtrs &D, const int x, const int lev, const bool hold = false)
{
    bool any = !hold && sq_ld(sq_started(D)) < P.num_envs;
    if (!any && sq_ld(sq_waiting(D, x)) > 0)
        for (int l = 0; l < lev; l++) { const int *ctr = sq_row(D, x, l); any = any || (sq_ld(ctr) < sq_ld(ctr + 1)); }
    // (P.pp_heavy_only: light envs that wait are background work for the slots that free up -- nobody parks itself for them)
    if (!any && P.pair_mode == 2 && !P.pp_heavy_only && sq_ld(sq_waiting(D, x + 8)) > 0)
        for (int l = 0; l < lev; l++) { const int *ctr = sq_row(D, x + 8, l); any = any || (sq_ld(ctr) < sq_ld(ctr + 1)); }
    return any;
}
// CHUNKED (k_physics_step_sched): the call resumes env `c_env`'s step after `c_lev` chunks of P.sq_chunk sub-steps and runs until the step is complete
// (returns true) or, at a chunk boundary, an env that has completed fewer chunks is waiting in XCD c_x's queues (returns false, *c_lev_out = chunks
// completed): an env that is behind everybody else -- a heavy one -- is never parked.  Parking goes through the same store / load as a step boundary,
// the step-local flags through D.sq_carry.
template <int mode, int KIND, bool CHUNKED = false, bool DAMP = false>
__device__ __forceinline__ bool physics_body(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions,
                                             const unsigned char *__restrict__ mask, double *__restrict__ reward,
                                             unsigned char *__restrict__ terminated, unsigned char *__restrict__ truncated,
                                             double *__restrict__ info, const int tmpl, const int boff = 0, const int c_env = 0, const int c_lev = 0,
                                             const int c_x = 0, int *c_lev_out = nullptr, const bool c_noyield = false, int *c_light_out = nullptr,
                                             const bool c_hold = false)
{
    // MODE_RESET with tmpl != 0 settles the per-trial reset templates: state slot num_envs + t holds trial t
    // MODE_STEP: workgroup b steps the env at position boff + b of the dispatch order
    const int env = CHUNKED ? c_env : (mode == MODE_RESET) ? (tmpl ? P.num_envs + (int)blockIdx.x : (int)blockIdx.x)
                                         : (D.order != nullptr ? D.order[blockIdx.x + boff] : (int)blockIdx.x + boff);
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#ifdef BP_SCHED_TRACE
    const unsigned long long _tt_a = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = lane_id();
#if 1
    // heaviest-first dispatch: the first workgroups carry the envs that set the launch time -> issue priority over their SIMD mates
    if (!CHUNKED && mode == MODE_STEP && D.order != nullptr) {
        if (blockIdx.x < gridDim.x / 4) __builtin_amdgcn_s_setprio(3);
        else if (blockIdx.x < gridDim.x / 2) __builtin_amdgcn_s_setprio(1);
    }
#endif
    if (mode == MODE_RESET && !tmpl && mask != nullptr && mask[env] == 0) return true;
    const int nbcap = P.nbcap;
    // CHUNKED: sub-steps of this step the env had completed when it was parked (0 = the step starts here).  A park of the scheduler leaves a multiple of
    // P.sq_chunk; an env that left a paired wave (bp_physics_pair.hpp) any sub-step.  c_lev only ranks the env in the queues.
    const int c_sub = CHUNKED ? D.sq_sub[env] : 0;

    // ---- carve LDS ----
    LdsCtx L;
    carve_lds<KIND>(P, L);

    // ---- env context ----
    const size_t eb = (size_t)env * nbcap;
    int trial, episode;
    if (mode == MODE_RESET) {
        episode = D.e_episode[env] + 1; // first reset: -1 -> 0 (ship_ice_env.py:226-229)
        trial = tmpl ? (int)blockIdx.x : (int)(((long long)P.env_offset + env + episode) % P.num_trials);
    } else {
        episode = D.e_episode[env];
        trial = D.e_trial[env];
    }
    const size_t tb = (size_t)trial * nbcap;
    EnvCtx E;
    E.nb = (mode == MODE_RESET) ? D.sc_nb[trial] : D.e_nb[env];
    env_ctx(P, D, env, trial, E);

    ArbReg A;
    SubState S;
#ifdef BP_PROF
    if (lane < BP_PROFN) L.prof[lane] = 0ull;
    lds_sync();
    const unsigned long long _t_kernel0 = __builtin_amdgcn_s_memtime();
#endif
    init_regs(A, S);
    if (mode == MODE_RESET) {
        // ---- new space + bodies from the trial (ship_ice_env.py:109-216) ----
        for (int base = 0; base < nbcap; base += 64) {
            const int i = base + lane;
            if (i < nbcap) {
                L.mvs[i] = 0u;
                if (i < BP_NSLOT + 2) { L.mvo[i] = 0u; L.sbody[i] = (unsigned short)i; }
                L.slot_of[i] = (i < P.nkin) ? (unsigned char)i : 255;
                if (i < P.nkin) { L.sv[i] = mk2(0.0, 0.0); L.sw[i] = mk2(0.0, 0.0); L.sb[i] = mk2(0.0, 0.0); }
                if (i < E.nb) {
                    double4 ps = D.sc_pose[tb + i];
                    if (P.random_start && i == 0 && !tmpl) { // ship_ice_env.py:201-203 with a counter RNG keyed (env, episode)
                        const double u = bp_start_u01(P.start_seed, P.env_offset + env, episode);
                        ps.x = 1 + u * (P.start_x_range - 1); ps.y = 1.0; ps.z = BP_PI / 2;
                    }
                    double sn, cs;
                    bp_sincos(ps.z, sn, cs);
                    gE(E.pxy, i) = mk2(ps.x, ps.y); gE(E.ang, i) = ps.z; gE(E.rot, i) = mk2(cs, sn);
                    if (i < P.nkin) L.sp[i] = mk2(ps.x, ps.y);
                    if (i == 0) { L.ag[0] = mk2(ps.z, 0.0); L.ag[1] = mk2(cs, sn); }
                    gE(E.adjn, i) = 0;
                }
            }
        }
        __syncthreads();
        for (int g0 = 0; g0 < E.nb; g0 += 2) {
            const int i = g0 + (lane >> 5);
            const bool act = i < E.nb;
            double4 t; t.x = 1; t.y = 0; t.z = 0; t.w = 0;
            if (act) {
                const d2 p = gE(E.pxy, i), r = gE(E.rot, i);
                const double4 ms = gE(E.mass, i);
                t.x = r.x; t.y = r.y;
                t.z = p.x - (ms.z * r.x - ms.w * r.y);
                t.w = p.y - (ms.z * r.y + ms.w * r.x);
            }
            double4 nbb;
            world_from_pose(P, E, act ? i : 0, act, lane & 31, t, nbb);
            if (act && (lane & 31) == 0) {
                gE(E.bb, i) = nbb;
                double4 nf;
                nf.x = nbb.x - P.skin; nf.y = nbb.y - P.skin; nf.z = nbb.z + P.skin; nf.w = nbb.w + P.skin;
                gE(E.fat, i) = nf;
            }
        }
        __syncthreads();
        // neighbour lists for every body (all fat boxes are final here)
        for (int i = 0; i < E.nb; i++) {
            if (kind_btype(gE(E.kind, i)) == BODY_STATIC) { if (lane == 0) gE(E.adjn, i) = 0; continue; } // never moves: list unused
            const double4 fi = gE(E.fat, i);
            int cnt = 0;
            for (int base = 0; base < E.nb; base += 64) {
                const int j = base + lane;
                const bool ov = (j < E.nb) && (j != i) && bb_overlap(fi, gE(E.fat, j));
                const unsigned long long m = ballot(ov);
                const int pos = cnt + popc_below(m, lane);
                if (ov && pos < BP_KADJ) { gE(E.adj, i * BP_KADJ + pos) = (unsigned short)j; gE(E.hint, i * BP_KADJ + pos) = 0; }
                cnt += __popcll(m);
            }
            if (cnt > BP_KADJ) { S.err |= BP_ERR_ADJ_OVERFLOW; cnt = BP_KADJ; }
            if (lane == 0) gE(E.adjn, i) = (unsigned char)cnt;
        }
        A.key = ARB_FREE_KEY; A.stamp = 0; A.state = ARB_FIRST; A.count = 0; A.h0 = A.h1 = 0;
        A.jn0 = A.jt0 = A.jn1 = A.jt1 = 0.0;
        A.n = mk2(0, 0); A.r1_0 = A.r2_0 = A.r1_1 = A.r2_1 = mk2(0, 0);
        // first sub-step of a new space: every shape is new to the broadphase, so every pair gets tested once
        for (int base = 0; base < E.nb; base += 64) {
            const int i = base + lane;
            if (i < E.nb) L.mv[i] = (unsigned short)i;
        }
        S.stamp = 0; S.curr_dt = 0.0; S.nmv = E.nb; S.nslots = P.nkin;
        A.slotA = A.slotB = 0;
        S.total_ke = 0.0; S.total_imp = 0.0; S.n_post = S.n_contact = S.n_first = 0;
        __syncthreads();
    } else {
        load_state_a<KIND>(P, D, E, L, A, S, env);
        if (CHUNKED && c_sub > 0) {
            const unsigned *cy = D.sq_carry + (size_t)env * 4;
            S.yaw_violated = (int)(cy[0] & 1u); S.boundary_violated = (int)(cy[1] & 1u); S.costp = cy[2];
            S.wall_flag = (int)((cy[0] >> 1) & 1u);
        }
        // ship control (ship_ice_env.py:265-274): set once per env step
        if ((!CHUNKED || c_sub == 0) && lane < P.nkin) { // every part of the kinematic agent carries the same velocity
            const double act = actions[env] * P.max_yaw_rate;
            const d2 r = gE(E.rot, 0);
            L.sv[lane] = mk2(r.x * P.target_speed + -r.y * 0.0, r.y * P.target_speed + r.x * 0.0);
            L.sw[lane] = mk2(act, L.sw[lane].y);
        }
        __syncthreads();
        load_state_b<KIND>(P, D, E, L, A, S, env);
    }

#ifdef BP_SCHED_TRACE   // where a task's time goes: resume (everything before the first sub-step), the first sub-step after a resume, parking -- summed into D.prof[1..6]
    const unsigned long long _tt_b = __builtin_amdgcn_s_memrealtime();
    unsigned long long _tt_c = 0ull;
#endif
    const unsigned stamp_start = S.stamp;
    const unsigned costp_resume = S.costp;
    const int nsub = (mode == MODE_RESET) ? P.settle_steps : P.steps;
    const int it_first = CHUNKED ? c_sub : 0;
    const unsigned carry_units = (CHUNKED && c_sub > 0) ? D.sq_carry[(size_t)env * 4 + 3] : 0u;   // wave cycles >> 8 of the step's earlier runs
    // P.sq_dynprio: issue priority by the env's own pace.  An env that runs on at a chunk boundary is behind the others, and the envs whose step projects
    // longest are the chain the launch ends with: they get the SIMD ahead of the wave they share it with (an env alone on its SIMD runs 1.2-1.4 x faster than
    // beside a second wave: tools/sched_trace.py with 512 envs).  Priority 3 from P.sq_dynprio per cent of the previous step's 99.6th-percentile env cost
    // (k_make_order), 2 from 5/6 and 1 from 2/3 of that; the projection is cycles so far / sub-steps so far x sub-steps of the step, at the start of a step
    // the env's previous cost.  Replaces the three static classes of the dispatch order for the launch (they are what a task starts with otherwise).
    auto pace_prio = [&](const unsigned long long el, const int done) {
        const unsigned long long proj = done > 0 ? el * (unsigned)nsub / (unsigned)done : el;
        const unsigned long long t3 = D.sq_thr != nullptr ? (unsigned long long)D.sq_thr[0] * (unsigned)P.sq_dynprio / 100ull : ~0ull >> 8;
        if (proj >= t3) __builtin_amdgcn_s_setprio(3);
        else if (proj * 6 >= t3 * 5) __builtin_amdgcn_s_setprio(2);
        else if (proj * 3 >= t3 * 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    };
    if (CHUNKED && P.sq_dynprio) pace_prio(c_sub > 0 ? carry_units : (D.order != nullptr ? D.e_cost[env] : 0u), c_sub);
    int c_it_parked = 0;
    bool step_done = true;
    int to_boundary = CHUNKED ? P.sq_chunk : 0x7FFFFFFF;   // sub-steps until the next chunk boundary
    if (CHUNKED && c_sub == 0) { // nothing has moved in this step yet
        unsigned char *mvd_ = D.sq_moved + (size_t)env * nbcap;
        for (int i = lane; i < nbcap; i += 64) mvd_[i] = 0;
    }
#ifdef BP_PRED
    unsigned pr_c200 = 0, pr_c100 = 0, pr_c50 = 0, pr_c10 = 0;
#endif
    for (int it = it_first; it < nsub; it++) {
        if (CHUNKED) {
            if (to_boundary == 0) { // chunk boundary: yield to an env that is further behind, otherwise carry on without a context switch
                int yield = 0, key = it / P.sq_chunk;
                // (P.sq_ymask: the chunk boundaries at which an env may yield, bit k = after k chunks -- an even finish needs fine turns only towards the end of the step)
                if (lane == 0 && !c_noyield) {
                    if (!((P.sq_ymask >> key) & 1u)) yield = 0;
                    else if (P.sq_lrpt) {
                        const unsigned long long el = (unsigned long long)carry_units + ((__builtin_amdgcn_s_memtime() - t_begin) >> 8);
                        // (no env is taken for lighter than P.sq_floor per sub-step left: an env's pace so far says little about a cluster it has yet to run into)
                        const unsigned long long rem = max(el * (unsigned)(nsub - it) / (unsigned)max(it, 1), (unsigned long long)P.sq_floor * (unsigned)(nsub - it));
                        key = sq_lrpt_key(P, rem);
                        yield = sq_someone_heavier(P, D, c_x, key, rem) ? 1 : 0;
                    } else yield = sq_someone_behind(P, D, c_x, key, c_hold) ? 1 : 0;
                    if (P.sq_debug && c_env == 1 && it == P.sq_chunk) yield = 1;   // (fault-injection hook of the diagnostic twin: env 1 parks at its first boundary whatever the mask says)
                }
                if (__builtin_amdgcn_readfirstlane(yield)) { step_done = false; *c_lev_out = __builtin_amdgcn_readfirstlane(key); c_it_parked = it; break; }
                if (P.sq_dynprio) pace_prio((unsigned long long)carry_units + ((__builtin_amdgcn_s_memtime() - t_begin) >> 8), it);
                to_boundary = P.sq_chunk;
            }
            to_boundary--;
        }
#ifdef BP_PRED
        if (it == nsub - 200) pr_c200 = S.costp;
        if (it == nsub - 100) pr_c100 = S.costp;
        if (it == nsub - 50) pr_c50 = S.costp;
        if (it == nsub - 10) pr_c10 = S.costp;
#endif
        // agent rules: a ship-ice handle gets the yaw + boundary rules, a maze handle the boundary rule alone -- also when a generic KIND 0 kernel
        // serves it (k_physics_step_damp, k_physics_step with hulls above 8 vertices); the scheduled ship kernel is only ever launched for ship-ice handles
        substep<KIND, DAMP>(P, E, L, A, S, P.dt_sub, (mode != MODE_STEP) ? 0 : (KIND == BP_ENV_SHIP_ICE && (CHUNKED || P.env_kind == BP_ENV_SHIP_ICE)) ? 1 : 2);
#ifdef BP_SCHED_TRACE
        if (it == it_first) _tt_c = __builtin_amdgcn_s_memrealtime();
#endif
        if (BP_UNLIKELY2(S.quiescent && !BP_TRACE_ON(D))) {
            // Nothing moves and no arbiter can produce an impulse: every remaining sub-step leaves all positions,
            // velocities and impulses untouched.  Apply their only effects in closed form: the stamp advances, active
            // arbiters are re-stamped (FIRST -> NORMAL), cached ones age out after `persistence` sub-steps, and the
            // post-solve callback keeps counting the (cold) ship arbiters.
            const unsigned k = (unsigned)(nsub - 1 - it);
            if (k > 0) {
                const unsigned now = S.stamp;
                if (A.key != ARB_FREE_KEY) {
                    if (A.stamp == now) { A.stamp = now + k; A.state = ARB_NORMAL; }
                    else if ((now + k) - A.stamp >= (unsigned)P.persistence) A.key = ARB_FREE_KEY;
                }
                S.stamp = now + k;
                S.n_post += k * S.ship_post;
                S.n_contact += k * S.ship_contacts;
            }
            step_done = true;
            break;
        }
        if (BP_UNLIKELY2(BP_TRACE_ON(D) && env == D.dbg_env)) {
            for (int base = 0; base < E.nb; base += 64) {
                const int i = base + lane;
                if (i < E.nb) {
                    double *o = D.dbg + ((size_t)it * nbcap + i) * 3;
                    o[0] = gE(E.pxy, i).x; o[1] = gE(E.pxy, i).y; o[2] = gE(E.ang, i);
                }
            }
        }
    }

#ifdef BP_SCHED_TRACE
    const unsigned long long _tt_d = __builtin_amdgcn_s_memrealtime();
#endif
    if (CHUNKED && !step_done) {
        // ---- end of a chunk: park the env exactly as at a step boundary; the step-local flags and the shapes that have moved so far go along
        unsigned char *mvd_ = D.sq_moved + (size_t)env * nbcap;
        for (int i = lane; i < E.nb; i += 64) if (L.mvs[i] > stamp_start) mvd_[i] = 1;
        __syncthreads();
        store_state(P, D, L, A, env);
        if (KIND == BP_ENV_SHIP_ICE && P.pair_mode == 2 && c_light_out != nullptr) {
            // is the env light enough to carry on in half a wavefront (bp_physics_pair.hpp)?  Arbiter lanes and moving bodies well inside the half-wave, and
            // a mean work proxy per sub-step of this run below the pairing limit
            const int keys = __popcll(ballot(A.key != ARB_FREE_KEY)), act = __popcll(S.prev_amask);
            const unsigned rate = (S.costp - costp_resume) / (unsigned)max(c_it_parked - it_first, 1);
            *c_light_out = (keys <= P.pp_max_keys - 6 && S.nmv <= P.pp_max_mv / 2 && act <= P.pp_max_act && rate <= (unsigned)P.pp_rate) ? 1 : 0;
        }
        const int err_c = (ballot((S.err & BP_ERR_ADJ_OVERFLOW) != 0) ? BP_ERR_ADJ_OVERFLOW : 0) |
                          (ballot((S.err & BP_ERR_ARB_OVERFLOW) != 0) ? BP_ERR_ARB_OVERFLOW : 0) |
                          (ballot((S.err & BP_ERR_LEVEL_OVERFLOW) != 0) ? BP_ERR_LEVEL_OVERFLOW : 0);
        if (lane == 0) {
            D.e_stamp[env] = S.stamp; D.e_currdt[env] = S.curr_dt;
            D.e_ke[env] = S.total_ke; D.e_imp[env] = S.total_imp;
            D.e_cnt[env * 4 + 0] = S.n_post; D.e_cnt[env * 4 + 1] = S.n_contact; D.e_cnt[env * 4 + 2] = S.n_first;
            unsigned *cy = D.sq_carry + (size_t)env * 4;
            cy[0] = (unsigned)S.yaw_violated | ((unsigned)S.wall_flag << 1); cy[1] = (unsigned)S.boundary_violated; cy[2] = S.costp;
            cy[3] = (c_sub == 0 ? 0u : cy[3]) + (unsigned)((__builtin_amdgcn_s_memtime() - t_begin) >> 8);
            D.sq_sub[env] = c_it_parked;
            if (err_c) atomicOr(&D.e_err[env], err_c);
        }
#ifdef BP_PROF
        if (D.prof != nullptr) { // a step that is run in several chunks accumulates (the tool clears the buffer before each step)
            if (lane == 0) L.prof[23] = __builtin_amdgcn_s_memtime() - _t_kernel0;
            lds_sync();
            if (lane < BP_PROFN) D.prof[(size_t)env * BP_PROFN + lane] += L.prof[lane];
        }
#endif
#ifdef BP_SCHED_TRACE
        __syncthreads();
        if (lane == 0 && D.prof != nullptr) {
            const unsigned long long _tt_e = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&D.prof[3], _tt_e - _tt_d); atomicAdd(&D.prof[7], 1ull);                                   // parks: time from the last sub-step to the end of the stores
            if (c_sub > 0) { atomicAdd(&D.prof[1], _tt_b - _tt_a); atomicAdd(&D.prof[2], _tt_c - _tt_b); atomicAdd(&D.prof[4], 1ull); }   // resumes: load, first sub-step
            atomicAdd(&D.prof[5], (unsigned long long)(c_it_parked - it_first)); atomicAdd(&D.prof[6], _tt_d - _tt_b);   // sub-steps of the run and their time
        }
#endif
        return false;
    }
    // ---- end of step: work / reward / termination (ship_ice_env.py:291-345) or reset snapshot ----
    double work = 0.0;
    if (mode == MODE_STEP) {
        for (int base = 0; base < E.nb; base += 64) {
            const int i = base + lane;
            // shapes that did not move contribute exactly +0, so the test only saves work; across chunks the flag comes from D.sq_moved
            const bool mvd = (i < E.nb) && ((L.mvs[i] > stamp_start) || (CHUNKED && c_sub > 0 && D.sq_moved[(size_t)env * nbcap + i] != 0)) &&
                             (kind_ctype(gE(E.kind, i)) == 2); // floes / boxes only
            double contrib = 0.0;
            if (mvd) {
                const int n = gE(E.nv, i);
                d2 *prev = E.pv + (size_t)i * BP_MAXV;
                const d2 *nowv = E.wv + (size_t)i * BP_MAXV;
                const double area = poly_area_seq(prev, n);
                const d2 ca = poly_centroid_seq(prev, n);
                const d2 cb = poly_centroid_seq(nowv, n);
                const double d = __builtin_sqrt((ca.x - cb.x) * (ca.x - cb.x) + (ca.y - cb.y) * (ca.y - cb.y));
                contrib = d * area;
                for (int q = 0; q < n; q++) prev[q] = nowv[q];
            }
            unsigned long long m = ballot(mvd);
            while (m) { // ascending floe order, like the python loop
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                work += __shfl(contrib, l);
            }
        }
    } else {
        for (int base = 0; base < E.nb * BP_MAXV; base += 64) {
            const int q = base + lane;
            if (q < E.nb * BP_MAXV) gE(E.pv, q) = gE(E.wv, q);
        }
    }
    __syncthreads();

    store_state(P, D, L, A, env);
#ifdef BP_PRED
    if (D.prof != nullptr && lane == 0 && mode == MODE_STEP) { // candidate predictors of the next step's cost (tools/cost_predictability.py)
        unsigned long long *o = D.prof + (size_t)env * BP_PROFN;
        o[0] = S.costp; o[1] = S.costp - pr_c200; o[2] = S.costp - pr_c100; o[3] = S.costp - pr_c50; o[4] = S.costp - pr_c10;
        o[5] = (unsigned long long)S.nmv; o[6] = (unsigned long long)__popcll(S.prev_amask); o[7] = (unsigned long long)S.nslots;
        o[8] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
#ifdef BP_PROF
    if (D.prof != nullptr) {
        if (lane == 0) L.prof[23] = __builtin_amdgcn_s_memtime() - _t_kernel0;
        lds_sync();
        if (lane < BP_PROFN) D.prof[(size_t)env * BP_PROFN + lane] += L.prof[lane];
    }
#endif
    const int err_any = (ballot((S.err & BP_ERR_ADJ_OVERFLOW) != 0) ? BP_ERR_ADJ_OVERFLOW : 0) |
                        (ballot((S.err & BP_ERR_ARB_OVERFLOW) != 0) ? BP_ERR_ARB_OVERFLOW : 0) |
                        (ballot((S.err & BP_ERR_LEVEL_OVERFLOW) != 0) ? BP_ERR_LEVEL_OVERFLOW : 0);
    if (lane == 0) {
        const d2 sp = gE(E.pxy, 0);
        const double sa = gE(E.ang, 0);
        D.e_stamp[env] = S.stamp; D.e_currdt[env] = S.curr_dt;
        if (mode == MODE_STEP) D.e_cost[env] = (unsigned)((__builtin_amdgcn_s_memtime() - t_begin) >> 8) + ((CHUNKED && c_sub > 0) ? D.sq_carry[(size_t)env * 4 + 3] : 0u);
        D.e_ke[env] = S.total_ke; D.e_imp[env] = S.total_imp;
        D.e_cnt[env * 4 + 0] = S.n_post; D.e_cnt[env * 4 + 1] = S.n_contact; D.e_cnt[env * 4 + 2] = S.n_first;
        if (err_any) atomicOr(&D.e_err[env], err_any);
        double total_work;
        if (mode == MODE_RESET) {
            D.e_trial[env] = trial; D.e_episode[env] = episode; D.e_nb[env] = E.nb;
            total_work = 0.0;
            D.e_total_work[env] = 0.0;
            D.e_flags[env] = 0; D.e_prevdist[env] = 0.0; // reset() clears wall_collision after the settle (maze_NAMO_env.py:338)
            if (info) {
                double *o = info + (size_t)env * BP_INFO_COUNT;
                for (int k = 0; k < BP_INFO_COUNT; k++) o[k] = 0.0;
                o[BP_I_X] = sp.x; o[BP_I_Y] = sp.y; o[BP_I_THETA] = sa;
                o[BP_I_KE] = S.total_ke; o[BP_I_IMPULSE] = S.total_imp;
                o[BP_I_NPOST] = (double)S.n_post; o[BP_I_NCONTACT] = (double)S.n_contact; o[BP_I_NFIRST] = (double)S.n_first;
            }
        } else if (P.env_kind == BP_ENV_MAZE) {
            // MazeNAMO.step tail (maze_NAMO_env.py:421-474)
            total_work = D.e_total_work[env] + work;
            D.e_total_work[env] = total_work;
            const double gdx = sp.x - P.goal_x, gdy = sp.y - P.goal_y;
            const double gd = __builtin_sqrt(gdx * gdx + gdy * gdy);
            const int goal = gd <= P.goal_reach;
            const int wall = S.wall_flag;
            const int term = goal || wall;
            int px = (int)(sp.x * P.m_to_pix), py = (int)(sp.y * P.m_to_pix);
            px = px < 0 ? 0 : (px > P.grid_w - 1 ? P.grid_w - 1 : px);
            py = py < 0 ? 0 : (py > P.grid_h - 1 ? P.grid_h - 1 : py);
            const double dist_value = D.dist_map[(size_t)py * P.grid_w + px];
            int fl = D.e_flags[env];
            double dinc = 0.0;
            if (sp.x != P.goal_x || sp.y != P.goal_y) {
                if (fl & 2) dinc = (D.e_prevdist[env] - dist_value) * P.k_increment;
                D.e_prevdist[env] = dist_value;
                fl |= 2;
            }
            D.e_flags[env] = (fl & ~1) | (wall ? 1 : 0);
            const double coll = -work;
            double rwd = P.beta * coll + dinc;
            if (S.boundary_violated || wall) rwd += P.boundary_penalty;
            int success = 0;
            if (term && !wall) { rwd += P.terminal_reward; success = 1; }
            if (reward) reward[env] = rwd;
            if (terminated) terminated[env] = (unsigned char)term;
            if (truncated) truncated[env] = 0;
            D.e_lastrew[env] = rwd; D.e_lastflag[env] = term | (success << 1);
            if (info) {
                double *o = info + (size_t)env * BP_INFO_COUNT;
                o[BP_I_X] = sp.x; o[BP_I_Y] = sp.y; o[BP_I_THETA] = sa; o[BP_I_TOTAL_WORK] = total_work; o[BP_I_WORK] = work;
                o[BP_I_COLL_REWARD] = coll; o[BP_I_SCALED_COLL] = coll * P.beta; o[BP_I_DIST_REWARD] = dinc;
                o[BP_I_SUCCESS] = success; o[BP_I_BOUNDARY] = S.boundary_violated; o[BP_I_YAW] = wall;
                o[BP_I_KE] = S.total_ke; o[BP_I_IMPULSE] = S.total_imp;
                o[BP_I_NPOST] = (double)S.n_post; o[BP_I_NCONTACT] = (double)S.n_contact; o[BP_I_NFIRST] = (double)S.n_first;
            }
        } else {
            total_work = D.e_total_work[env] + work;
            D.e_total_work[env] = total_work;
            int boundary_terminal = 0;
            if (sp.x < 0.0 && __builtin_fabs(sp.x - 0.0) >= 0.0) boundary_terminal = 1;
            if (sp.x > P.map_w && __builtin_fabs(sp.x - P.map_w) >= 0.0) boundary_terminal = 1;
            int term = 0;
            if (sp.y >= P.goal_y) term = 1;
            else if (boundary_terminal) term = 1;
            double dist_reward = 0.0;
            if (sp.y < P.goal_y) {
                const d2 r = gE(E.rot, 0);
                dist_reward = 1.0 * (r.x * 0.0 + r.y * 1.0);
            }
            const double coll = -work;
            double rwd = P.beta * coll + dist_reward;
            if (S.yaw_violated) rwd += 0.0;
            if (S.boundary_violated) rwd += P.boundary_penalty;
            int success = 0;
            if (term && !boundary_terminal) { rwd += P.terminal_reward; success = 1; }
            if (reward) reward[env] = rwd;
            if (terminated) terminated[env] = (unsigned char)term;
            if (truncated) truncated[env] = 0;
            D.e_lastrew[env] = rwd; D.e_lastflag[env] = term | (success << 1);
            if (info) {
                double *o = info + (size_t)env * BP_INFO_COUNT;
                o[BP_I_X] = sp.x; o[BP_I_Y] = sp.y; o[BP_I_THETA] = sa; o[BP_I_TOTAL_WORK] = total_work; o[BP_I_WORK] = work;
                o[BP_I_COLL_REWARD] = coll; o[BP_I_SCALED_COLL] = coll * P.beta; o[BP_I_DIST_REWARD] = dist_reward;
                o[BP_I_SUCCESS] = success; o[BP_I_BOUNDARY] = S.boundary_violated; o[BP_I_YAW] = S.yaw_violated;
                o[BP_I_KE] = S.total_ke; o[BP_I_IMPULSE] = S.total_imp;
                o[BP_I_NPOST] = (double)S.n_post; o[BP_I_NCONTACT] = (double)S.n_contact; o[BP_I_NFIRST] = (double)S.n_first;
            }
        }
    }
    return true;
}

// env.step(): 400 sub-steps + work / reward / termination
__global__ __launch_bounds__(64, 2) void k_physics_step(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                     double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                     unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    physics_body<MODE_STEP, 0>(P, D, actions, nullptr, reward, terminated, truncated, info, 0);
}
// Two environments per wavefront, fixed pairs for the whole step (P.pair_mode == 1; BP_PAIR=1): workgroup b steps the envs at positions 2b and 2b + 1 of the
// dispatch order.  No env leaves its pair here: the half-wave capacities are hard limits (per-env error bits as in the solo kernel).  The parity tests run
// the paired sub-step through this kernel; the product path is the scheduler below with P.pair_mode == 2.
__global__ __launch_bounds__(64, 2) void k_physics_step_pair(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                          double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                          unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    // P.pair_solo != 0 (experiment switch of this test kernel): heaviest with lightest (positions b and N - 1 - b of the dispatch order) instead of neighbours
    const int p0 = P.pair_solo ? (int)blockIdx.x : 2 * (int)blockIdx.x, p1 = P.pair_solo ? P.num_envs - 1 - (int)blockIdx.x : p0 + 1;
    const int e0 = D.order != nullptr ? D.order[p0] : p0;
    const int e1 = (p1 < P.num_envs && p1 != p0) ? (D.order != nullptr ? D.order[p1] : p1) : -1;
    PairLimits Q;
    Q.max_keys = Q.max_slots = Q.max_mv = Q.max_act = Q.max_work = 0x7FFFFFFF;
    Q.gc_slots = PP_NSLOT - 12; Q.max_rate = 0x7FFFFFFF;
    int it, score, heavy;
    pair_task<false>(P, D, actions, reward, terminated, truncated, info, e0, e1, Q, it, score, heavy, [](const int) { return false; });
}
// ---- preemptive step scheduler ------------------------------------------------------------------------------------------------------
// A launch ends with its last env, and which envs will be heavy in a step is only half predictable from the previous one: with the static
// heaviest-first order an env that turns out heavy after starting in the second round of the 2 048 wave slots sets the launch time (measured:
// 22.9 ms, 19.2 ms when the order is that of the step's own costs, tools/oracle_order.py).  Here a step is cut into chunks of P.sq_chunk sub-steps;
// at a chunk boundary a wave parks its env if another one is further behind, and a new workgroup takes the waiting env that has completed the
// fewest chunks: every env advances at about the same sub-step pace, a heavy env is always the one furthest behind -- it is never parked and keeps
// its slot from the first sub-step to the last -- and nothing needs to be predicted.  A parked env goes through the same store / load as at a step
// boundary, so the results are those of k_physics_step bit for bit.  Queues are per XCD (an env's arrays stay in one L2) and per level.
__global__ void k_sched_init(const DevParams P, const DevPtrs D)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    for (int i = tid; i < SQ_NX * SQ_MAXLEV * P.sq_cap; i += nthr) D.sq_items[i] = -1;
    for (int i = tid; i < SQ_NX * (SQ_MAXLEV + 2) * 2; i += nthr) D.sq_ctr[i] = 0;
    for (int i = tid; i < P.num_envs; i += nthr) { D.sq_done[i] = 0; D.sq_lev[i] = 0; D.sq_sub[i] = 0; }
}
// After the scheduled launch: the envs whose step is not complete (none, unless the scheduler's watchdog fired) are listed for the completion launch;
// more than SQ_RESCUE of them is reported as BP_ERR_SCHED_TIMEOUT (the step is then incomplete).
#define SQ_RESCUE 256
__global__ __launch_bounds__(256) void k_sched_scan(const DevParams P, const DevPtrs D)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    if (sq_ld(sq_finished(D)) < P.num_envs)
        for (int e = threadIdx.x; e < P.num_envs; e += blockDim.x)
            if (D.sq_done[e] == 0) { const int pos = atomicAdd(&cnt, 1); if (pos < SQ_RESCUE) D.sq_rescue[1 + pos] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        D.sq_rescue[0] = min(cnt, SQ_RESCUE);
        if (cnt > SQ_RESCUE) atomicOr(&D.e_err[0], BP_ERR_SCHED_TIMEOUT);
    }
}
// what a scheduler workgroup knows about itself for the -DBP_SCHED_TRACE record of its task
struct SchedTraceCtx { unsigned long long trw; int idle, first, home; };
// A paired task of the scheduler: envs pe0 / pe1 (pe1 = -1: an idle half) resumed in one wavefront until they finish, yield or split.  Returns -1, or -- a heavy env
// that left its pair carries on alone in this wave slot -- its queue item (env | priority class << 24) | chunks completed << 26.
template <int KIND>
__device__ __forceinline__ int sched_paired_block(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions, double *__restrict__ reward,
                                                  unsigned char *__restrict__ terminated, unsigned char *__restrict__ truncated, double *__restrict__ info,
                                                  const int pe0, const int pe1, const int x, const bool first)
{
    const int lane = lane_id();
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the envs' arrays as the waves that parked them left them
        PairLimits Q;
        Q.max_keys = P.pp_max_keys; Q.max_slots = P.pp_max_slots; Q.max_mv = P.pp_max_mv; Q.max_act = P.pp_max_act; Q.max_work = P.pp_max_work;
        Q.gc_slots = P.pp_max_slots - 6; Q.max_rate = P.pp_rate;
        int it_half = 0, score_half = 0, heavy_half = 0;
        auto behind = [&](const int level) -> bool {
            int y = 0;
            if (lane == 0) y = (((P.sq_ymask >> level) & 1u) && sq_someone_behind(P, D, x, level)) ? 1 : 0;   // (P.sq_ymask: see physics_body)
            return __builtin_amdgcn_readfirstlane(y) != 0;
        };
        const int st_half = pair_task<true>(P, D, actions, reward, terminated, truncated, info, pe0, pe1, Q, it_half, score_half, heavy_half, behind);
        const int st0 = __builtin_amdgcn_readlane(st_half, 0), st1 = __builtin_amdgcn_readlane(st_half, 32);
        const int it0 = __builtin_amdgcn_readlane(it_half, 0), it1 = __builtin_amdgcn_readlane(it_half, 32);
        const int sc0 = __builtin_amdgcn_readlane(score_half, 0), sc1 = __builtin_amdgcn_readlane(score_half, 32);
        const int hv0 = __builtin_amdgcn_readlane(heavy_half, 0), hv1 = __builtin_amdgcn_readlane(heavy_half, 32);
        pair_gsync();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int nfin = (st0 == 1 ? 1 : 0) + (st1 == 1 ? 1 : 0);
        if (lane == 0) {
            if (st0 == 1) D.sq_done[pe0] = 1;
            if (st1 == 1) D.sq_done[pe1] = 1;
            if (nfin) atomicAdd(sq_finished(D), nfin);
        }
        const bool pk0 = st0 == 2, pk1 = st1 == 2;
        if (lane == 0 && D.sq_pairstat != nullptr) {
            atomicAdd(&D.sq_pairstat[first ? 0 : 1], 1);
            if (nfin) atomicAdd(&D.sq_pairstat[2], nfin);
        }
        if (!pk0 && !pk1) return -1;
        // a heavy env carries on here, alone (the heavier of two); whatever else was parked goes to the queue of its kind and level
        const bool c0 = pk0 && hv0 && (!(pk1 && hv1) || sc0 >= sc1), c1 = !c0 && pk1 && hv1;
        if (lane == 0) {
            if (pk0 && !c0) { D.sq_lev[pe0] = it0 / P.sq_chunk; sq_push(P, D, x + (hv0 ? 0 : 8), it0 / P.sq_chunk, pe0); }
            if (pk1 && !c1) { D.sq_lev[pe1] = it1 / P.sq_chunk; sq_push(P, D, x + (hv1 ? 0 : 8), it1 / P.sq_chunk, pe1); }
            if (D.sq_pairstat != nullptr) {
                if (c0 || c1) atomicAdd(&D.sq_pairstat[3], 1);
                const int qh = ((pk0 && !c0 && hv0) ? 1 : 0) + ((pk1 && !c1 && hv1) ? 1 : 0), ql = ((pk0 && !c0 && !hv0) ? 1 : 0) + ((pk1 && !c1 && !hv1) ? 1 : 0);
                if (qh) atomicAdd(&D.sq_pairstat[4], qh);
                if (ql) atomicAdd(&D.sq_pairstat[5], ql);
            }
        }
        if (!c0 && !c1) return -1;
        __syncthreads();
        return ((c0 ? pe0 : pe1) | (3 << 24)) | (((c0 ? it0 : it1) / P.sq_chunk) << 26);   // top issue priority: it left its pair because it is heavy
    }
}
// A solo task of the scheduler: env `item` resumed after `lev` chunks until it finishes or yields at a chunk boundary (then it goes back to the queue of its kind).
template <int KIND>
__device__ __forceinline__ void sched_solo_block(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions, double *__restrict__ reward,
                                                 unsigned char *__restrict__ terminated, unsigned char *__restrict__ truncated, double *__restrict__ info,
                                                 const int item, const int lev, const int x, const bool completion, const SchedTraceCtx &T)
{
    const int lane = lane_id();
    const int env = item & 0xFFFFFF;
    if ((item >> 24) == 3) __builtin_amdgcn_s_setprio(3);
    else if ((item >> 24) == 1) __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the env's arrays as the wave that parked it left them
#ifdef BP_SCHED_TRACE   // diagnostic build (tools/sched_trace.py): every task's (env, levels, XCD, start, end) in the 100 MHz reference clock, into D.prof
    const unsigned long long _tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    int lev_out = lev + 1, light_out = 0;
    // P.sq_hold: the envs of the top issue-priority class -- the heaviest quarter of the dispatch order, and envs that left a pair as heavy -- keep their slot
    // while other envs merely have not started yet: their chain is what the launch waits for at the end, and a first chunk that waits costs it a round
    const bool done = physics_body<MODE_STEP, KIND, true>(P, D, actions, nullptr, reward, terminated, truncated, info, 0, 0, env, lev, x, &lev_out, completion,
                                                          &light_out, P.sq_hold != 0 && (item >> 24) == 3);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifdef BP_SCHED_TRACE
    if (lane == 0 && D.prof != nullptr) {
        const unsigned long long _tr1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long idx = atomicAdd(&D.prof[0], 1ull);
        unsigned long long *o = D.prof + 8 + 4 * idx;
        o[0] = (unsigned long long)(unsigned)env | ((unsigned long long)(unsigned)lev << 32) | ((unsigned long long)(unsigned)(done ? 255 : lev_out) << 40) | ((unsigned long long)(unsigned)T.home << 48) |
               ((unsigned long long)(unsigned)x << 52) | ((unsigned long long)(T.first ? 1u : 0u) << 56);
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));   // wave slot, SIMD, CU, SE of this wave: which slot the task ran in
        o[1] = _tr0; o[2] = _tr1;
        o[3] = ((_tr0 - T.trw) << 32) | ((unsigned long long)(hwid & 0xFFFFu) << 16) | (unsigned long long)(unsigned)min(T.idle, 65535);   // workgroup start -> task start, slot id, empty polls
    }
#endif
    if (lane == 0) {
        if (done) { D.sq_done[env] = 1; if (!completion) atomicAdd(sq_finished(D), 1); }
        else {
            D.sq_lev[env] = lev_out;
            if (!(P.sq_debug && env == 1 && lev_out == 1)) sq_push(P, D, x + ((P.pair_mode == 2 && !completion && light_out) ? 8 : 0), lev_out, item);   // test hook: the item is lost
        }
    }
}
// One workgroup per (env, chunk) task: the hardware dispatcher is the persistent loop, and the step code is compiled as in k_physics_step.
// The first num_envs workgroups start the envs in the heaviest-first order without touching a queue (first chunks come before everything else under
// the least-advanced-first rule anyway, and workgroups are dispatched in index order); every later workgroup takes the least-advanced waiting env of
// its XCD, waits if there is none yet, helps another XCD after a few empty polls, and leaves without work only when every env has finished (most of
// the grid does: only parked envs need a second workgroup).  An env's home XCD is the one its first chunk ran on.
// ROLE 0: the scheduler with one env per wavefront only (no paired code in the kernel).  ROLE 1 (k_physics_step_schedp): the paired first tasks, every
// workgroup that starts from the queues, and the solo continuation of split pairs.  The two roles are two kernels because they cannot share one register
// allocation without loss: with the paired path inlined beside it the solo step body picked up ~100 scratch accesses and ran 6 % slower (230 k against
// 245 k env-steps/s with pairing switched off, same box) -- and the solo body is the chain of the heaviest envs.  A pairing launch runs both kernels side by
// side on two streams: ROLE 0 with exactly P.pair_solo workgroups (the envs that start alone, on the lean code), ROLE 1 with the rest.
template <int KIND, int ROLE = 0>
__device__ __forceinline__ bool sched_body(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions,
                                           double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                           unsigned char *__restrict__ truncated, double *__restrict__ info, const int bid_in = -1)
{
    // bid_in >= 0 (sched_resident): the position in the dispatch order that this pass of a resident workgroup serves (>= the number of first tasks: it serves
    // the queues); the return value tells it to leave (every env has finished, or the watchdog has fired)
    const int lane = lane_id();
    const int home = sq_xcc_id();
#ifdef BP_SCHED_TRACE
    const unsigned long long _trw = __builtin_amdgcn_s_memrealtime();   // the workgroup's own start
    int _tr_idle = 0;
#endif
    int item = -1, lev = 0, x = home;
    int pe0 = -1, pe1 = -1;          // a paired task: two envs for one wavefront
    const bool completion = P.sq_mode == 1;
    // Two environments per wavefront (P.pair_mode == 2, ship-ice).  Light envs advance two to a wavefront, heavy ones alone, and an env changes sides
    // at any boundary: the first P.pair_solo positions of the dispatch order start alone, the others in pairs (neighbours of the order: similar envs share
    // trip counts) -- with every env paired the 2 048 wave slots hold 4 096 envs from the first cycle; a paired wave whose env turns heavy (or outgrows the
    // half-wave) parks both envs at that sub-step boundary, carries on with the heavy one alone in the same slot and queues its mate; paired and solo waves
    // alike yield at chunk boundaries to envs that are further behind; and an env that is light when it is parked waits in a queue of its own kind, from
    // which a starting workgroup takes two at a time.
    const bool pairing = ROLE == 1 && KIND == BP_ENV_SHIP_ICE && P.pair_mode == 2 && !completion;
    const int npairs = pairing ? (P.num_envs - P.pair_solo + 1) / 2 : 0;
    // workgroups that start envs without touching a queue: ROLE 1 numbers its workgroups from P.pair_solo on (the first P.pair_solo positions of the dispatch
    // order are the ROLE 0 kernel's, launched beside it)
    // (P.sq_parts > 1, ROLE 0: the launch is split into that many kernels on as many streams -- hardware queues dispatch their workgroups in order, and a
    // queue whose next workgroup is bound for a full shader engine leaves free slots elsewhere idle; part k starts the envs at positions k, k + parts, ...
    // of the dispatch order with its first num_envs / parts workgroups)
    const int nfirst_part = (ROLE == 0 && P.sq_parts > 1) ? (P.num_envs - P.sq_part + P.sq_parts - 1) / P.sq_parts : 0;
    const int bid = (bid_in >= 0) ? bid_in : (ROLE == 1) ? (int)blockIdx.x + P.pair_solo
                  : (P.sq_parts > 1) ? (((int)blockIdx.x < nfirst_part) ? (int)blockIdx.x * P.sq_parts + P.sq_part : P.num_envs + (int)blockIdx.x)
                  : (int)blockIdx.x;
    const int nfirst = pairing ? P.pair_solo + npairs : P.num_envs;
    if (completion) {
        // Completion launch (always follows the scheduled one and k_sched_scan, SQ_RESCUE workgroups): workgroup b takes the b-th env of the list of
        // unfinished envs and leaves at once if the list is shorter -- the normal case: it is empty.  After a scheduler fault (watchdog) such an env's
        // state is that of its last chunk boundary: the step is resumed there and run to its end.
        if ((int)blockIdx.x >= D.sq_rescue[0]) return true;
        item = D.sq_rescue[1 + blockIdx.x]; lev = D.sq_lev[item];
        if (lane == 0) atomicAdd(&D.sq_warn[1], 1);
    } else if (pairing && bid >= P.pair_solo && bid < nfirst) {
        // neighbours of the dispatch order (similar envs share trip counts), or -- P.pp_snake -- the heaviest with the lightest: a heavy env declines its
        // pair at once and carries on alone from the first cycle, and the mate that has to wait for a wave slot is then one with slack
        const int p0 = P.pp_snake ? bid : P.pair_solo + 2 * (bid - P.pair_solo), p1 = P.pp_snake ? P.num_envs - 1 - (bid - P.pair_solo) : p0 + 1;
        pe0 = D.order != nullptr ? D.order[p0] : p0;
        pe1 = (p1 < P.num_envs && p1 != p0) ? (D.order != nullptr ? D.order[p1] : p1) : -1;
        if (lane == 0) atomicAdd(sq_started(D), pe1 >= 0 ? 2 : 1);
    } else if (bid < nfirst) {
        const int pos = bid;
        // issue-priority class of the env for the whole step: the heaviest quarter of the predicted order 3, the next quarter 1
        const int cls = (pos < P.num_envs / 4) ? 3 : (pos < P.num_envs / 2) ? 1 : 0;
        item = (D.order != nullptr ? D.order[pos] : pos) | (cls << 24);
        if (lane == 0) atomicAdd(sq_started(D), 1);
    } else {
        int kind = 0, mate = -1;
        if (lane == 0) {
            const int limit = P.sq_debug ? 64 : (1 << 20);
            for (int idle = 0;; idle++) {
                item = sq_pop(P, D, home, lev, kind);
                if (item < 0 && (idle & 3) == 3)
                    for (int o = 1; o < 8 && item < 0; o++) { const int y = (home + o) & 7; item = sq_pop(P, D, y, lev, kind); if (item >= 0) x = y; }
#ifdef BP_SCHED_TRACE
                _tr_idle = idle;
#endif
                if (item >= 0 || sq_ld(sq_finished(D)) >= P.num_envs || sq_ld(sq_abort(D)) != 0) break;
                // watchdog: ~20 s of empty polls can only mean a scheduler fault -- raise the abort flag (every poller leaves on it) and leave rather
                // than hold the GPU; the completion launch finishes the envs that are still parked
                if (idle > limit) { if (atomicExch(sq_abort(D), 1) == 0) atomicAdd(&D.sq_warn[0], 1); break; }
                for (int q = 0; q < 4; q++) __builtin_amdgcn_s_sleep(127);
            }
            // a light env takes a second light one of the same XCD along: the least advanced that waits, from its own level upwards
            if (ROLE == 1 && item >= 0 && kind == 1 && sq_ld(sq_waiting(D, x + 8)) > 0)
                for (int l = lev; l < P.sq_levels && mate < 0; l++) mate = sq_pop_level(P, D, x + 8, l);
        }
        item = __builtin_amdgcn_readfirstlane(item); lev = __builtin_amdgcn_readfirstlane(lev); x = __builtin_amdgcn_readfirstlane(x);
        mate = __builtin_amdgcn_readfirstlane(mate);
        if (item < 0) return true;
        if (mate >= 0) { pe0 = item & 0xFFFFFF; pe1 = mate & 0xFFFFFF; item = -1; }
    }
    SchedTraceCtx T;
    T.trw = 0ull; T.idle = 0; T.first = bid < nfirst ? 1 : 0; T.home = home;
#ifdef BP_SCHED_TRACE
    T.trw = _trw; T.idle = _tr_idle;
#endif
    if (ROLE == 1 && pe0 >= 0) {
        const int r = sched_paired_block<KIND>(P, D, actions, reward, terminated, truncated, info, pe0, pe1, x, bid < nfirst);
        if (r < 0) return false;
        item = r & 0x3FFFFFF; lev = (int)((unsigned)r >> 26);
    }
    sched_solo_block<KIND>(P, D, actions, reward, terminated, truncated, info, item, lev, x, completion, T);
    return false;
}
// ---- the same scheduler with resident wavefronts (k_physics_step_schedl*, the default of a scheduled launch without pairing) -----------------------------------
// What the hardware dispatcher costs the launch above (tools/micro/wg_turnover.hip, tools/sched_trace.py): workgroup i goes to XCD i % 8, and inside an XCD the
// workgroups go round-robin to its four shader engines IN ORDER -- the dispatcher waits while the engine whose turn it is has no free wave slot, whatever is free
// in the other three (a model that reproduces the microbenchmark's launch times to three digits).  With tasks that end one by one at random times that leaves
// ~6 % of the slot-time empty: mean 90 us, p99 0.9 ms between a task's end and the next workgroup's start in the same slot.  Here one workgroup per wave slot
// stays for the whole launch and takes task after task itself (1.1 us between tasks): positions of the dispatch order from a counter while there are first
// tasks, then the least-advanced waiting env of its XCD -- sched_body with the position handed in.
// Every pass of the loop reads the launch constants from device memory (k_store_params) through a pointer the compiler cannot follow out of the loop: as
// kernel arguments they are loop invariants that get hoisted into ~200 SGPRs living across the whole step body -- the register allocation that sank the
// first resident version (127 k against 193 k env-steps/s, 1 332 spill reloads).  Read this way they are scalar loads next to their uses: 296 v_readlane
// and 118 spilled SGPRs against 877 / 268 in k_physics_step_sched.  (The same constants by pointer WITHOUT the loop: -14 %; DevParams by value inside the loop: -1 %.)
__device__ __forceinline__ int *sq_nextpos(const DevPtrs &D) { return sq_row(D, 2, SQ_MAXLEV + 1); }
template <int KIND, int ROLE>
__device__ __forceinline__ void sched_resident(const DevParams *Pg, const DevPtrs *Dg, const double *__restrict__ actions,
                                               double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                               unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    for (;;) {
        unsigned long long pa = (unsigned long long)Pg, da = (unsigned long long)Dg;
        asm volatile("" : "+s"(pa), "+s"(da));
        const DevParams &P = *(const DevParams *)(const __attribute__((address_space(4))) DevParams *)pa;
        const DevPtrs &D = *(const DevPtrs *)(const __attribute__((address_space(4))) DevPtrs *)da;
        __builtin_amdgcn_s_setprio(0);
        const bool pairing = ROLE == 1 && KIND == BP_ENV_SHIP_ICE && P.pair_mode == 2;
        const int nfirst = pairing ? P.pair_solo + (P.num_envs - P.pair_solo + 1) / 2 : P.num_envs;
        int bid = nfirst;
        if (lane_id() == 0 && sq_ld(sq_nextpos(D)) < nfirst) bid = min(atomicAdd(sq_nextpos(D), 1), nfirst);
        bid = __builtin_amdgcn_readfirstlane(bid);
        if (sched_body<KIND, ROLE>(P, D, actions, reward, terminated, truncated, info, bid)) return;
        __syncthreads();
    }
}
#ifndef BP_SCHED_WAVES
#define BP_SCHED_WAVES 2   // wavefronts per SIMD the scheduled ship-ice kernels are compiled for (3: the occupancy experiment, tools/build_variant.sh w3)
#endif
__global__ __launch_bounds__(64, BP_SCHED_WAVES) void k_physics_step_schedl(const DevParams *Pg, const DevPtrs *Dg, const double *__restrict__ actions,
                                                             double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                             unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_resident<0, 0>(Pg, Dg, actions, reward, terminated, truncated, info);
}
__global__ __launch_bounds__(64, 2) void k_physics_step_schedl_maze(const DevParams *Pg, const DevPtrs *Dg, const double *__restrict__ actions,
                                                                 double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                                 unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_resident<BP_ENV_MAZE, 0>(Pg, Dg, actions, reward, terminated, truncated, info);
}
// ---- a pairing launch on resident wavefronts (k_physics_step_schedr): the two step bodies as separate FUNCTIONS ---------------------------------------------------
// One kernel that holds both bodies inline pays for it in its register allocation (the solo body 6 % slower, 203 spilled VGPRs in the paired one, 266 with the
// resident loop around them).  Here each body is a function of its own -- its own allocation -- called from a small resident loop that picks the tasks; the launch
// constants reach them as two pointers (a by-value DevParams / DevPtrs argument of a non-inlined function is a 1.7 KB copy on the stack of every lane).  Arguments
// of a device function arrive in VGPRs: the functions make them wave-uniform again first.
#define BP_UNIFORM_PTR(T, p) ((T)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(p) >> 32)) << 32) | \
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(p))))
template <int KIND>
__device__ __attribute__((noinline)) int sched_fn_pair(const DevParams *Pg_, const DevPtrs *Dg_, const double *actions_, double *reward_, unsigned char *terminated_,
                                                       unsigned char *truncated_, double *info_, int pe0, int pe1, int x, int first)
{
    const DevParams &P = *(const DevParams *)(const __attribute__((address_space(4))) DevParams *)BP_UNIFORM_PTR(unsigned long long, Pg_);
    const DevPtrs &D = *(const DevPtrs *)(const __attribute__((address_space(4))) DevPtrs *)BP_UNIFORM_PTR(unsigned long long, Dg_);
    return sched_paired_block<KIND>(P, D, BP_UNIFORM_PTR(const double *, actions_), BP_UNIFORM_PTR(double *, reward_), BP_UNIFORM_PTR(unsigned char *, terminated_),
                                    BP_UNIFORM_PTR(unsigned char *, truncated_), BP_UNIFORM_PTR(double *, info_), __builtin_amdgcn_readfirstlane(pe0),
                                    __builtin_amdgcn_readfirstlane(pe1), __builtin_amdgcn_readfirstlane(x), __builtin_amdgcn_readfirstlane(first) != 0);
}
template <int KIND>
__device__ __attribute__((noinline)) void sched_fn_solo(const DevParams *Pg_, const DevPtrs *Dg_, const double *actions_, double *reward_, unsigned char *terminated_,
                                                        unsigned char *truncated_, double *info_, int item, int lev, int x)
{
    const DevParams &P = *(const DevParams *)(const __attribute__((address_space(4))) DevParams *)BP_UNIFORM_PTR(unsigned long long, Pg_);
    const DevPtrs &D = *(const DevPtrs *)(const __attribute__((address_space(4))) DevPtrs *)BP_UNIFORM_PTR(unsigned long long, Dg_);
    SchedTraceCtx T;
    T.trw = 0ull; T.idle = 0; T.first = 0; T.home = 0;   // (the trace build records the tasks of the kernels that inline sched_body)
    sched_solo_block<KIND>(P, D, BP_UNIFORM_PTR(const double *, actions_), BP_UNIFORM_PTR(double *, reward_), BP_UNIFORM_PTR(unsigned char *, terminated_),
                           BP_UNIFORM_PTR(unsigned char *, truncated_), BP_UNIFORM_PTR(double *, info_), __builtin_amdgcn_readfirstlane(item),
                           __builtin_amdgcn_readfirstlane(lev), __builtin_amdgcn_readfirstlane(x), false, T);
}
// the resident loop of a pairing launch: what sched_body<KIND, 1> does before its task (first tasks by position, then the queues of either kind), then the functions
template <int KIND>
__device__ __forceinline__ void sched_resident_pairing(const DevParams *Pg, const DevPtrs *Dg, const double *__restrict__ actions,
                                                       double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                       unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    const int lane = lane_id();
    const int home = sq_xcc_id();
    for (;;) {
        unsigned long long pa = (unsigned long long)Pg, da = (unsigned long long)Dg;
        asm volatile("" : "+s"(pa), "+s"(da));
        const DevParams &P = *(const DevParams *)(const __attribute__((address_space(4))) DevParams *)pa;
        const DevPtrs &D = *(const DevPtrs *)(const __attribute__((address_space(4))) DevPtrs *)da;
        __builtin_amdgcn_s_setprio(0);
        const int npairs = (P.num_envs - P.pair_solo + 1) / 2, nfirst = P.pair_solo + npairs;
        int bid = nfirst;
        if (lane == 0 && sq_ld(sq_nextpos(D)) < nfirst) bid = min(atomicAdd(sq_nextpos(D), 1), nfirst);
        bid = __builtin_amdgcn_readfirstlane(bid);
        int item = -1, lev = 0, x = home, pe0 = -1, pe1 = -1;
        if (bid >= P.pair_solo && bid < nfirst) {
            const int p0 = P.pp_snake ? bid : P.pair_solo + 2 * (bid - P.pair_solo), p1 = P.pp_snake ? P.num_envs - 1 - (bid - P.pair_solo) : p0 + 1;
            pe0 = D.order != nullptr ? D.order[p0] : p0;
            pe1 = (p1 < P.num_envs && p1 != p0) ? (D.order != nullptr ? D.order[p1] : p1) : -1;
            if (lane == 0) atomicAdd(sq_started(D), pe1 >= 0 ? 2 : 1);
        } else if (bid < nfirst) {
            const int cls = (bid < P.num_envs / 4) ? 3 : (bid < P.num_envs / 2) ? 1 : 0;
            item = (D.order != nullptr ? D.order[bid] : bid) | (cls << 24);
            if (lane == 0) atomicAdd(sq_started(D), 1);
        } else {
            int kind = 0, mate = -1;
            if (lane == 0) {
                const int limit = P.sq_debug ? 64 : (1 << 20);
                for (int idle = 0;; idle++) {
                    item = sq_pop(P, D, home, lev, kind);
                    if (item < 0 && (idle & 3) == 3)
                        for (int o = 1; o < 8 && item < 0; o++) { const int y = (home + o) & 7; item = sq_pop(P, D, y, lev, kind); if (item >= 0) x = y; }
                    if (item >= 0 || sq_ld(sq_finished(D)) >= P.num_envs || sq_ld(sq_abort(D)) != 0) break;
                    if (idle > limit) { if (atomicExch(sq_abort(D), 1) == 0) atomicAdd(&D.sq_warn[0], 1); break; }
                    for (int q = 0; q < 4; q++) __builtin_amdgcn_s_sleep(127);
                }
                if (item >= 0 && kind == 1 && sq_ld(sq_waiting(D, x + 8)) > 0)
                    for (int l = lev; l < P.sq_levels && mate < 0; l++) mate = sq_pop_level(P, D, x + 8, l);
            }
            item = __builtin_amdgcn_readfirstlane(item); lev = __builtin_amdgcn_readfirstlane(lev); x = __builtin_amdgcn_readfirstlane(x);
            mate = __builtin_amdgcn_readfirstlane(mate);
            if (item < 0) return;
            if (mate >= 0) { pe0 = item & 0xFFFFFF; pe1 = mate & 0xFFFFFF; item = -1; }
        }
        if (pe0 >= 0) {
            const int r = sched_fn_pair<KIND>(Pg, Dg, actions, reward, terminated, truncated, info, pe0, pe1, x, bid < nfirst ? 1 : 0);
            if (r < 0) { __syncthreads(); continue; }
            item = r & 0x3FFFFFF; lev = (int)((unsigned)r >> 26);
        }
        sched_fn_solo<KIND>(Pg, Dg, actions, reward, terminated, truncated, info, item, lev, x);
        __syncthreads();
    }
}
__global__ __launch_bounds__(64, 2) void k_physics_step_schedr(const DevParams *Pg, const DevPtrs *Dg, const double *__restrict__ actions,
                                                            double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                            unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_resident_pairing<0>(Pg, Dg, actions, reward, terminated, truncated, info);
}
__global__ void k_store_params(const DevParams P, const DevPtrs D, DevParams *Pg, DevPtrs *Dg)
{
    static_assert(sizeof(DevParams) % 4 == 0 && sizeof(DevPtrs) % 4 == 0, "copied as 32-bit words");
    const unsigned *sp = (const unsigned *)&P, *sd = (const unsigned *)&D;
    for (unsigned i = threadIdx.x; i < sizeof(DevParams) / 4; i += blockDim.x) ((unsigned *)Pg)[i] = sp[i];
    for (unsigned i = threadIdx.x; i < sizeof(DevPtrs) / 4; i += blockDim.x) ((unsigned *)Dg)[i] = sd[i];
}
__global__ __launch_bounds__(64, BP_SCHED_WAVES) void k_physics_step_sched(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                           double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                           unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_body<0, 0>(P, D, actions, reward, terminated, truncated, info);
}
// the paired half of a pairing launch (two environments per wavefront, bp_physics_pair.hpp): runs beside k_physics_step_sched, which starts the P.pair_solo
// envs that begin alone
__global__ __launch_bounds__(64, 2) void k_physics_step_schedp(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                            double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                            unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_body<0, 1>(P, D, actions, reward, terminated, truncated, info);
}
__global__ __launch_bounds__(64, 2) void k_physics_step_sched_maze(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                                double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                                unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    sched_body<BP_ENV_MAZE>(P, D, actions, reward, terminated, truncated, info);
}
// reset() of the masked envs: new space from the next trial + 1000 settle sub-steps
__global__ __launch_bounds__(64, 2) void k_physics_reset(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                      double *__restrict__ info, const int tmpl)
{
    physics_body<MODE_RESET, 0>(P, D, nullptr, mask, nullptr, nullptr, nullptr, tmpl ? nullptr : info, tmpl);
}
// space.damping != 0 (bp_config.damping_pow != 0; no shipped config): the generic instantiation (vertex loops of BP_MAXV, so it serves ship-ice and maze
// handles alike) with substep<0, DAMP = true>; one wavefront per env for the whole step, no scheduler
__global__ __launch_bounds__(64, 2) void k_physics_step_damp(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                          double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                          unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    physics_body<MODE_STEP, 0, false, true>(P, D, actions, nullptr, reward, terminated, truncated, info, 0);
}
__global__ __launch_bounds__(64, 2) void k_physics_reset_damp(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                           double *__restrict__ info, const int tmpl)
{
    physics_body<MODE_RESET, 0, false, true>(P, D, nullptr, mask, nullptr, nullptr, nullptr, tmpl ? nullptr : info, tmpl);
}
// maze-NAMO-v0 instantiations (vertex loops of 8)
__global__ __launch_bounds__(64, 2) void k_physics_step_maze(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                          double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                          unsigned char *__restrict__ truncated, double *__restrict__ info)
{
    physics_body<MODE_STEP, BP_ENV_MAZE>(P, D, actions, nullptr, reward, terminated, truncated, info, 0);
}
__global__ __launch_bounds__(64, 2) void k_physics_reset_maze(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                           double *__restrict__ info, const int tmpl)
{
    physics_body<MODE_RESET, BP_ENV_MAZE>(P, D, nullptr, mask, nullptr, nullptr, nullptr, tmpl ? nullptr : info, tmpl);
}
// box-delivery: new space + 1000 settle sub-steps with the boundary handlers (box_delivery_env.py:239-285)
__global__ __launch_bounds__(64, 2) void k_bd_settle(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                  double *__restrict__ info, const int tmpl)
{
    physics_body<MODE_RESET, BP_ENV_BOX>(P, D, nullptr, mask, nullptr, nullptr, nullptr, tmpl ? nullptr : info, tmpl);
}
__global__ __launch_bounds__(64, 2) void k_bd_settle_damp(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                       double *__restrict__ info, const int tmpl)
{
    physics_body<MODE_RESET, BP_ENV_BOX, false, true>(P, D, nullptr, mask, nullptr, nullptr, nullptr, tmpl ? nullptr : info, tmpl);
}

// Dispatch order for the next step: envs sorted by the cycles their last step took, heaviest first (bucket sort).
// Workgroups start in index order, so the long-running environments start first and the launch tail shrinks.
// The order only permutes independent environments: results do not depend on it.
// thr (may be null): [0] = the cost of the env at rank n / 256 from the top (the reference of the scheduler's pace-based issue priorities), to the resolution of
// the 256 buckets
__global__ __launch_bounds__(1024) void k_make_order(const unsigned *__restrict__ cost, int *__restrict__ order, int n, unsigned *__restrict__ thr)
{
    __shared__ unsigned hist[257];
    __shared__ unsigned smax;
    const int tid = threadIdx.x;
    if (tid < 257) hist[tid] = 0;
    if (tid == 0) smax = 1;
    __syncthreads();
    unsigned mx = 0;
    for (int i = tid; i < n; i += blockDim.x) mx = max(mx, cost[i]);
    atomicMax(&smax, mx);
    __syncthreads();
    const unsigned long long m = smax;
    for (int i = tid; i < n; i += blockDim.x) {
        const unsigned b = 255u - (unsigned)(((unsigned long long)cost[i] * 255ull) / m);
        atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0;
        const unsigned rank = (unsigned)max(1, n / 256);
        int tb = -1;
        for (int b = 0; b < 256; b++) { const unsigned c = hist[b]; hist[b] = acc; acc += c; if (tb < 0 && acc >= rank) tb = b; }
        if (thr != nullptr) thr[0] = (unsigned)((m * (unsigned long long)(255 - max(tb, 0))) / 255ull);
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const unsigned b = 255u - (unsigned)(((unsigned long long)cost[i] * 255ull) / m);
        const unsigned pos = atomicAdd(&hist[b], 1u);
        order[pos] = i;
    }
}


// Per-launch cost statistics for bench.py's two ceilings (bp_get_cost_stats; launched only while kernel timing is on): ring[slot] = (sum over the
// envs of the wave cycles >> 8 their step took, the largest of them).  The sum over the device's wave slots bounds the launch from below when
// the slots are perfectly packed, the maximum is the busy time of the heaviest env -- both in this run's own clock.
__global__ __launch_bounds__(1024) void k_cost_stats(const unsigned *__restrict__ cost, int n, unsigned long long *__restrict__ ring, int slot)
{
    __shared__ unsigned long long ssum;
    __shared__ unsigned smax;
    if (threadIdx.x == 0) { ssum = 0ull; smax = 0u; }
    __syncthreads();
    unsigned long long s = 0ull;
    unsigned mx = 0u;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const unsigned c = cost[i]; s += c; mx = max(mx, c); }
    atomicAdd(&ssum, s);
    atomicMax(&smax, mx);
    __syncthreads();
    if (threadIdx.x == 0) { ring[2 * slot] = ssum; ring[2 * slot + 1] = smax; }
}

// reset() from the settled per-trial template (ship_ice_env.py:223-249 is a pure function of the trial when
// random_start is off): copy the template state of trial (global_env_id + episode) % T into the env.
template <typename T>
__device__ __forceinline__ void copy_span(T *dst, const T *src, size_t n, int tid, int nt)
{
    for (size_t i = tid; i < n; i += nt) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_reset_copy(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                    double *__restrict__ info)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int episode = D.e_episode[env] + 1;
    const int trial = (int)(((long long)P.env_offset + env + episode) % P.num_trials);
    const size_t nb = P.nbcap, d = (size_t)env * nb, s = (size_t)(P.num_envs + trial) * nb;
    copy_span(D.pxy + d, D.pxy + s, nb, tid, nt);
    copy_span(D.ang + d, D.ang + s, nb, tid, nt);
    copy_span(D.rot + d, D.rot + s, nb, tid, nt);
    copy_span(D.velv + d, D.velv + s, nb, tid, nt);
    copy_span(D.velw + d, D.velw + s, nb, tid, nt);
    copy_span(D.velb + d, D.velb + s, nb, tid, nt);
    copy_span(D.wv + d * BP_MAXV, D.wv + s * BP_MAXV, nb * BP_MAXV, tid, nt);
    copy_span(D.wn + d * BP_MAXV, D.wn + s * BP_MAXV, nb * BP_MAXV, tid, nt);
    copy_span(D.pv + d * BP_MAXV, D.pv + s * BP_MAXV, nb * BP_MAXV, tid, nt);
    copy_span(D.bb + d, D.bb + s, nb, tid, nt);
    copy_span(D.fat + d, D.fat + s, nb, tid, nt);
    copy_span((unsigned long long *)(D.adj + d * BP_KADJ), (const unsigned long long *)(D.adj + s * BP_KADJ), nb * BP_KADJ / 4, tid, nt);
    copy_span(D.hint + d * BP_KADJ, D.hint + s * BP_KADJ, nb * BP_KADJ, tid, nt);
    copy_span((unsigned long long *)(D.adjn + d), (const unsigned long long *)(D.adjn + s), nb / 8, tid, nt);
    const size_t ad = (size_t)env * BP_ACAP, as = (size_t)(P.num_envs + trial) * BP_ACAP;
    copy_span(D.a_key + ad, D.a_key + as, BP_ACAP, tid, nt);
    copy_span(D.a_stamp + ad, D.a_stamp + as, BP_ACAP, tid, nt);
    copy_span(D.a_sc + ad, D.a_sc + as, BP_ACAP, tid, nt);
    copy_span(D.a_h0 + ad, D.a_h0 + as, BP_ACAP, tid, nt);
    copy_span(D.a_h1 + ad, D.a_h1 + as, BP_ACAP, tid, nt);
    copy_span(D.a_d + ad * 14, D.a_d + as * 14, (size_t)BP_ACAP * 14, tid, nt);
    if (tid == 0) {
        const int se = P.num_envs + trial;
        D.e_trial[env] = trial; D.e_episode[env] = episode; D.e_nb[env] = D.e_nb[se];
        D.e_stamp[env] = D.e_stamp[se]; D.e_currdt[env] = D.e_currdt[se];
        D.e_total_work[env] = 0.0; D.e_ke[env] = D.e_ke[se]; D.e_imp[env] = D.e_imp[se];
        D.e_flags[env] = 0; D.e_prevdist[env] = 0.0;
        for (int q = 0; q < 4; q++) D.e_cnt[env * 4 + q] = D.e_cnt[se * 4 + q];
        if (D.e_err[se]) atomicOr(&D.e_err[env], D.e_err[se]);
        if (info) {
            double *o = info + (size_t)env * BP_INFO_COUNT;
            for (int q = 0; q < BP_INFO_COUNT; q++) o[q] = 0.0;
            o[BP_I_X] = D.pxy[s].x; o[BP_I_Y] = D.pxy[s].y; o[BP_I_THETA] = D.ang[s];
            o[BP_I_KE] = D.e_ke[se]; o[BP_I_IMPULSE] = D.e_imp[se];
            o[BP_I_NPOST] = (double)D.e_cnt[se * 4 + 0]; o[BP_I_NCONTACT] = (double)D.e_cnt[se * 4 + 1]; o[BP_I_NFIRST] = (double)D.e_cnt[se * 4 + 2];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_observe: egocentric observation, uint8 [4][H][W] per env (ship_ice_env.py:378-409 + occupancy_map.py).
// One 256-thread workgroup per env.  Every output pixel is computed directly ("gather"): the 1000x300 global maps
// of the reference are never materialised.
//   ch0 footprint   free 0.5 / ship 1.0 / out-of-map 0     occupancy_map.py:300-337,379-410
//   ch1 goal line   max(0, goal - i*0.04)/goal, OOB 1      occupancy_map.py:413-433,492-521
//   ch2 heading     cv2.line head->tail 0.5, head pixel 1  occupancy_map.py:524-587
//   ch3 occupancy   skimage.draw.polygon of floes within 12 m, 25 px/m   occupancy_map.py:37-65,112-140
// ------------------------------------------------------------------------------------------------------------
#define OBS_THREADS 256
#ifndef OBS_THREADS_SHIP
#define OBS_THREADS_SHIP 512   // k_observe (ship-ice): 8 waves per workgroup, 4 workgroups per CU = the 32 wave slots of a CU
#endif
#define OBS_MAXCAND 96      // floes whose pixel AABB meets the 150x150 window (6 m x 6 m; typically 15-40)
#define OBS_CHUNK 32        // candidates rasterised per pass: 10 KB of vertex staging instead of 30 KB -> 4 workgroups per CU instead of 2
#ifndef OBS_SNAP_Y
#define OBS_SNAP_Y 1e-6     // a vertex this close to a raster row sends the row to the per-pixel test
#define OBS_SNAP_X 1e-6     // a crossing this close to a pixel centre is decided by the exact test on that pixel
#endif

// skimage._shared.geometry.point_in_polygon restated (third-party, unpinned): non-zero = inside / edge / vertex.
// The crossing tests `(x0*y1 - x1*y0) / (y1 - y0) > 0` (`< 0`) are evaluated as sign tests: for the finite,
// well-scaled raster coordinates here the quotient can neither overflow nor underflow, so it has the sign of
// numerator * denominator (and is 0 exactly when the numerator is 0).  Bit-identical to the oracle's division.
__device__ __forceinline__ bool pip_arrays(const double *xp, const double *yp, int n, double x, double y)
{
    const double eps = 1e-12;
    unsigned l_cross = 0, r_cross = 0;
    double x1 = xp[n - 1] - x, y1 = yp[n - 1] - y;
    for (int i = 0; i < n; i++) {
        const double x0 = xp[i] - x, y0 = yp[i] - y;
        if ((-eps < x0 && x0 < eps) && (-eps < y0 && y0 < eps)) return true;
        const bool up = (y0 > 0) != (y1 > 0), dn = (y0 < 0) != (y1 < 0);
        if (up || dn) {
            const double num = x0 * y1 - x1 * y0, den = y1 - y0;
            const bool pos = (num > 0 && den > 0) || (num < 0 && den < 0);
            const bool neg = (num > 0 && den < 0) || (num < 0 && den > 0);
            if (up && pos) r_cross++;
            if (dn && neg) l_cross++;
        }
        x1 = x0; y1 = y0;
    }
    if ((r_cross & 1) != (l_cross & 1)) return true;
    return (r_cross & 1) != 0;
}

__device__ __forceinline__ long long to_u16(double v) { return ((long long)v) & 0xFFFF; }

// cv2.clipLine restated (third-party, unpinned)
__device__ __forceinline__ bool clip_line(long long W, long long H, long long &x1, long long &y1, long long &x2, long long &y2)
{
    const long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

struct LineSpec { int valid; long long x0, y0, dx, dy, sx, sy; int vert; };

// 8-connected Bresenham of cv2.line in closed form: after t steps along the driving axis the minor coordinate has
// advanced m_t = floor((2*dmin*t + dmaj - 1) / (2*dmaj)) (dmaj > 0); left-to-right start like cv::LineIterator.
__device__ __forceinline__ LineSpec make_line(long long W, long long H, long long x1, long long y1, long long x2, long long y2)
{
    LineSpec s;
    s.valid = 1;
    if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
        (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
        if (!clip_line(W, H, x1, y1, x2, y2)) { s.valid = 0; }
    }
    long long dx = x2 - x1, dy = y2 - y1;
    long long px = x1, py = y1;
    if (dx < 0) { dx = -dx; dy = -dy; px = x2; py = y2; }
    long long sy = 1;
    if (dy < 0) { dy = -dy; sy = -1; }
    s.x0 = px; s.y0 = py; s.dx = dx; s.dy = dy; s.sx = 1; s.sy = sy; s.vert = dy > dx;
    return s;
}
__device__ __forceinline__ bool on_line(const LineSpec &s, long long x, long long y)
{
    if (!s.valid) return false;
    const long long ax = (x - s.x0) * s.sx, ay = (y - s.y0) * s.sy; // steps along +x / signed y from the start
    long long t, m, dmaj, dmin;
    if (!s.vert) { t = ax; m = ay; dmaj = s.dx; dmin = s.dy; }
    else { t = ay; m = ax; dmaj = s.dy; dmin = s.dx; }
    if (t < 0 || t > dmaj || m < 0 || m > dmin) return false; // m_t lies in [0, dmin]
    if (dmaj == 0) return m == 0;
    const long long mt = (2 * dmin * t + dmaj - 1) / (2 * dmaj);
    return m == mt;
}

// Sequential-sum restatements of poly_area / poly_centroid on split coordinate arrays (same operation order as the *_seq forms)
__device__ __forceinline__ d2 poly_centroid_xy(const double *x, const double *y, int n)
{
    double d1 = 0.0, d2_ = 0.0;
    for (int i = 0; i < n; i++) {
        const int p = (i == 0) ? n - 1 : i - 1;
        d1 += x[i] * y[p];
        d2_ += y[i] * x[p];
    }
    const double A = 0.5 * __builtin_fabs(d1 - d2_);
    double sx = 0.0, sy = 0.0;
    for (int i = 0; i < n; i++) {
        const int p = (i == 0) ? n - 1 : i - 1;
        const double u = x[i] * y[p] - x[p] * y[i];
        sx += (x[i] + x[p]) * u;
        sy += (y[i] + y[p]) * u;
    }
    const double f = 1.0 / (6.0 * A);
    return mk2(__builtin_fabs(f * sx), __builtin_fabs(f * sy));
}

// byte mask of the first k bytes of a little-endian word (k <= 0: none, k >= 4: all)
__device__ __forceinline__ unsigned obs_upto(int k) { return k <= 0 ? 0u : (k >= 4 ? 0xFFFFFFFFu : ((1u << (8 * k)) - 1u)); }
// flag |= on the bytes [b0, b1] of an LDS byte image through word atomics (several wavefronts may touch one word)
__device__ __forceinline__ void obs_or_bytes(unsigned *img32, int b0, int b1, unsigned char flag)
{
    for (int w = b0 >> 2; w <= (b1 >> 2); w++) {
        const int lo = max(b0, 4 * w) - 4 * w, hi = min(b1, 4 * w + 3) - 4 * w;
        atomicOr(&img32[w], (obs_upto(hi + 1) & ~obs_upto(lo)) & ((unsigned)flag * 0x01010101u));
    }
}

// One raster row of skimage.draw.polygon for a CONVEX polygon (xp = columns, yp = rows, raster coordinates), columns [c0, c1]:
// the inside pixels of the row form one span between the two edges that straddle it.  With every vertex at least OBS_SNAP_Y away
// from the row, point_in_polygon's crossing tests have the sign of (X - x) * dy for the crossing abscissa X of an edge (error
// ~1e-13 px), so pixels strictly between the two crossings are inside and the others outside; a crossing within OBS_SNAP_X of a
// pixel centre is decided by the exact test on that pixel.  Rows with a vertex within OBS_SNAP_Y (vertex / edge rules of
// point_in_polygon) or not exactly two crossings fall back to the exact test on every pixel.  Identical to testing every pixel.
// OR_BITS: the bytes keep their other flag bits; `row` is then the image base and pixel gj is byte off + gj of it (word atomics).
template <bool OR_BITS>
__device__ __forceinline__ void raster_row_convex(const double *xp, const double *yp, int n, int gi, int c0, int c1, unsigned char *row,
                                                  unsigned char flag, const int off = 0)
{
    const double y = (double)gi;
    bool slow = false;
    int ncross = 0, ia = 0, ib = 0;   // the (at most two) edges i-1 -> i that straddle the row; their abscissae are computed after the
    double y1 = yp[n - 1] - y;        // loop, so that the lanes of a wave do not serialise one division per vertex
    for (int i = 0; i < n; i++) {
        const double y0 = yp[i] - y;
        if (__builtin_fabs(y0) < OBS_SNAP_Y) slow = true;
        if ((y0 > 0) != (y1 > 0)) {
            if (ncross == 0) ia = i; else ib = i;
            ncross++;
        }
        y1 = y0;
    }
    double xa = 0.0, xb = 0.0;
    if (ncross == 2) {
        const int pa = (ia == 0) ? n - 1 : ia - 1, pb = (ib == 0) ? n - 1 : ib - 1;
        { const double x0 = xp[ia], y0 = yp[ia] - y, x1 = xp[pa], y1_ = yp[pa] - y; xa = x0 - y0 * ((x1 - x0) / (y1_ - y0)); }
        { const double x0 = xp[ib], y0 = yp[ib] - y, x1 = xp[pb], y1_ = yp[pb] - y; xb = x0 - y0 * ((x1 - x0) / (y1_ - y0)); }
    }
    if (slow || (ncross != 0 && ncross != 2)) {
        for (int gj = c0; gj <= c1; gj++)
            if (pip_arrays(xp, yp, n, (double)gj, y)) { if (OR_BITS) obs_or_bytes((unsigned *)row, off + gj, off + gj, flag); else row[gj] = flag; }
        return;
    }
    if (ncross == 0) return;
    const double xl = fmin(xa, xb), xr = fmax(xa, xb);
    int first, last;
    {
        const double rl = __builtin_rint(xl);
        if (__builtin_fabs(xl - rl) < OBS_SNAP_X) first = pip_arrays(xp, yp, n, rl, y) ? (int)rl : (int)rl + 1;
        else first = (int)__builtin_ceil(xl);
        const double rr = __builtin_rint(xr);
        if (__builtin_fabs(xr - rr) < OBS_SNAP_X) last = pip_arrays(xp, yp, n, rr, y) ? (int)rr : (int)rr - 1;
        else last = (int)__builtin_floor(xr);
    }
    first = max(first, c0); last = min(last, c1);
    if (OR_BITS) { if (last >= first) obs_or_bytes((unsigned *)row, off + first, off + last, flag); return; }
    // plain stores: whole 32-bit words where the span covers them (every byte of such a word belongs to this span, and any other
    // writer of this phase stores the same flag), single bytes at the two ends
    unsigned char *p = row + first, *const pe = row + last + 1;
    while (p < pe && ((uintptr_t)p & 3u)) *p++ = flag;
    const unsigned w4 = (unsigned)flag * 0x01010101u;
    for (; p + 4 <= pe; p += 4) *(unsigned *)p = w4;
    while (p < pe) *p++ = flag;
}


// Flag bits of the LDS window image
#ifdef BP_PROF
#define OPROF(k) { if (tid == 0 && D.prof != nullptr) D.prof[(size_t)env * BP_PROFN + 56 + (k)] = __builtin_amdgcn_s_memtime(); }   // slots 56..63: k_observe's stamps
#else
#define OPROF(k)
#endif
#define OBS_F_OCC 1u
#define OBS_F_SHIP 2u
#define OBS_F_LINE 4u
#define OBS_F_HEAD 8u

__global__ __launch_bounds__(OBS_THREADS_SHIP, 8) void k_observe(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                         unsigned char *__restrict__ obs)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    const int tid = threadIdx.x;
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap;
    const int nb = D.e_nb[env];
    const d2 *wv = D.wv + eb * BP_MAXV;
    const int *nv = D.sc_nv + (size_t)D.e_trial[env] * nbcap;
    const int Hg = P.grid_h, Wg = P.grid_w, LH = P.obs_h, LW = P.obs_w;
    const int npix = LH * LW;

    extern __shared__ double2 obs_smem[];
    // LDS: window image of flag bits [npix] u8 | goal-distance row table [LH] u8 | candidate polygons (world, then raster coordinates)
    unsigned char *s_img = (unsigned char *)obs_smem;
    unsigned char *s_edt = s_img + ((npix + 15) & ~15);
    double *s_px = (double *)(s_edt + ((LH + 15) & ~15));   // [OBS_CHUNK][BP_MAXV]: the candidates are processed OBS_CHUNK at a time
    double *s_py = s_px + OBS_CHUNK * BP_MAXV;
    __shared__ int s_npass;
    __shared__ unsigned short s_list[OBS_MAXCAND];
    __shared__ int s_bbx[OBS_CHUNK][4]; // window-clipped pixel box: r0, r1, c0, c1 (global raster coordinates); r1 < r0: not a candidate
    __shared__ unsigned char s_cn[OBS_CHUNK];
    __shared__ double s_fr[BP_MAX_SHIP_VERTS], s_fc[BP_MAX_SHIP_VERTS];
    __shared__ int s_fcnt, s_fbb[4];
    OPROF(0)
    // the first AABB of every thread is requested together with the env scalars and the ship pose (its address needs neither): one cold
    // round trip at the start of the kernel instead of three dependent ones
    const double4 bb_first = D.bb[eb + min(1 + tid, nbcap - 1)];
    if (tid == 0) s_npass = 0;
    for (int w = tid; w < (npix + 3) / 4; w += OBS_THREADS_SHIP) ((unsigned *)s_img)[w] = 0u;
    __syncthreads();
    OPROF(1)

    const d2 sp = D.pxy[eb];
    const d2 srot = D.rot[eb]; // (cos, sin) == bp_sincos(angle)
    const double m2gx = (double)Wg / P.map_w, m2gy = (double)Hg / P.map_h;
    const int wx = (int)(sp.x * m2gx);
    const int wy = (int)((sp.y + P.vshift) * m2gy);
    // window rows/cols in global raster coordinates (occupancy_map.py:112-140): g = int(l + w - L/2)
    const int gi0 = (int)((double)(0 + wy) - ((double)LH / 2)), gj0 = (int)((double)(0 + wx) - ((double)LW / 2));
    const int gi1 = gi0 + LH - 1, gj1 = gj0 + LW - 1;
    const int bh = (int)(P.map_h * P.m_to_pix), bw = (int)(P.map_w * P.m_to_pix);
    // ---- 1. conservative pre-test on the body AABBs (min/max of the same world vertices, grown by the shape radius): a floe whose
    //         AABB, grown by a pixel, misses the window cannot pass the exact pixel-box test of step 3 -> its vertices are not read
    for (int s = 1 + tid; s < nb; s += OBS_THREADS_SHIP) {
        const double4 b = (s == 1 + tid) ? bb_first : D.bb[eb + s];
        if (__builtin_floor(b.y * P.m_to_pix) - 1.0 > (double)gi1 || __builtin_ceil(b.w * P.m_to_pix) + 1.0 < (double)gi0 ||
            __builtin_floor(b.x * P.m_to_pix) - 1.0 > (double)gj1 || __builtin_ceil(b.z * P.m_to_pix) + 1.0 < (double)gj0) continue;
        const int slot = atomicAdd(&s_npass, 1);
        if (slot < OBS_MAXCAND) s_list[slot] = (unsigned short)s;
    }
    // ---- ship footprint polygon in grid coordinates (vertices outside the grid are dropped, occupancy_map.py:318-325)
    LineSpec line;
    long long hpx, hpy;
    {
        const double ch = srot.x, sh = srot.y;
        if (tid == 0) {
            int cnt = 0;
            double rmin = BP_INF, rmax = -BP_INF, cmin = BP_INF, cmax = -BP_INF;
            for (int i = 0; i < P.num_ship_verts; i++) {
                const double vx = P.ship_verts[i][0] * ch + P.ship_verts[i][1] * -sh + sp.x;
                const double vy = P.ship_verts[i][0] * sh + P.ship_verts[i][1] * ch + sp.y;
                const double gx = vx * m2gx, gy = vy * m2gy;
                if (gy < 0 || gy >= Hg || gx < 0 || gx >= Wg) continue;
                s_fr[cnt] = gy; s_fc[cnt] = gx; cnt++;
                rmin = fmin(rmin, gy); rmax = fmax(rmax, gy); cmin = fmin(cmin, gx); cmax = fmax(cmax, gx);
            }
            s_fcnt = cnt;
            if (cnt > 0) {
                s_fbb[0] = (int)(long long)fmax(0.0, rmin); s_fbb[1] = (int)(long long)__builtin_ceil(rmax);
                s_fbb[2] = (int)(long long)fmax(0.0, cmin); s_fbb[3] = (int)(long long)__builtin_ceil(cmax);
            }
        }
        line.valid = 0; hpx = hpy = 0;
        if (tid >= 64 && tid < 128) { // only the second wavefront draws the heading line and the head pixel (step 5)
            const double hx = P.ship_head[0] * ch + P.ship_head[1] * -sh + sp.x, hy = P.ship_head[0] * sh + P.ship_head[1] * ch + sp.y;
            const double tx = P.ship_tail[0] * ch + P.ship_tail[1] * -sh + sp.x, ty = P.ship_tail[0] * sh + P.ship_tail[1] * ch + sp.y;
            hpx = to_u16(hx * m2gx); hpy = to_u16(hy * m2gy);
            const long long tpx = to_u16(tx * m2gx), tpy = to_u16(ty * m2gy);
            line = make_line(Wg, Hg, hpx, hpy, tpx, tpy);
            hpx = hpx < 0 ? 0 : (hpx > Wg - 1 ? Wg - 1 : hpx);
            hpy = hpy < 0 ? 0 : (hpy > Hg - 1 ? Hg - 1 : hpy);
        }
    }
    // goal-distance value per window row (occupancy_map.py:413-433): max(0, goal - i*g2m)/goal, out of map -> 1
    const double g2m = P.map_h / (double)Hg;
    for (int li = tid; li < LH; li += OBS_THREADS_SHIP) {
        const int gi = gi0 + li;
        unsigned char e = 255;
        if (gi >= 0 && gi < Hg) {
            double dd = P.goal_y - gi * g2m;
            if (dd < 0) dd = 0;
            e = (unsigned char)((dd / P.goal_y) * 255);
        }
        s_edt[li] = e;
    }
    __syncthreads();
    OPROF(2)
    const int ncand_all = min(s_npass, OBS_MAXCAND);
    if (tid == 0 && s_npass > OBS_MAXCAND) atomicOr(&D.e_err[env], BP_ERR_LEVEL_OVERFLOW);
    for (int cbase = 0; cbase < ncand_all; cbase += OBS_CHUNK) {
    const int ncand = min(OBS_CHUNK, ncand_all - cbase);
    // ---- 2. world vertices of the pre-test survivors -> LDS, one (floe, vertex) item per thread ----
    for (int idx = tid; idx < ncand * BP_MAXV; idx += OBS_THREADS_SHIP) {
        const int k = idx / BP_MAXV, q = idx - k * BP_MAXV;
        const int s = s_list[cbase + k];
        const d2 v = wv[(size_t)s * BP_MAXV + q]; // unconditional (slots >= n hold stale vertices of the same array): travels with nv[s]
        const int n = nv[s];
        if (q == 0) s_cn[k] = (unsigned char)n;
        if (q < n) { s_px[idx] = v.x; s_py[idx] = v.y; }
    }
    __syncthreads();
    OPROF(3)
    // ---- 3. exact candidate test, one floe per thread: |centroid| range culling (occupancy_map.py:44-49) + pixel box vs window;
    //         the vertices are converted to raster coordinates in place ----
    if (tid < ncand) {
        const int k = tid;
        const int n = s_cn[k];
        double *xp = s_px + k * BP_MAXV, *yp = s_py + k * BP_MAXV;
        bool keep = true;
        // poly_centroid_xy's three sums (area, x and y moments) and the pixel box in ONE pass over the vertices: five independent accumulators,
        // each in the order of the sequential forms; the previous vertex stays in registers, so the in-place conversion to raster
        // coordinates can happen in the same pass
        double d1 = 0.0, d2_ = 0.0, sx = 0.0, sy = 0.0;
        double rmin = BP_INF, rmax = -BP_INF, cmin = BP_INF, cmax = -BP_INF;
        double xq = xp[n - 1], yq = yp[n - 1];
        for (int i = 0; i < n; i++) {
            const double xi = xp[i], yi = yp[i];
            const double a = xi * yq, b = yi * xq;
            d1 += a; d2_ += b;
            const double u = a - xq * yi;
            sx += (xi + xq) * u;
            sy += (yi + yq) * u;
            const double r = yi * P.m_to_pix, cc = xi * P.m_to_pix;
            xp[i] = cc; yp[i] = r;
            rmin = fmin(rmin, r); rmax = fmax(rmax, r); cmin = fmin(cmin, cc); cmax = fmax(cmax, cc);
            xq = xi; yq = yi;
        }
        {
            const double A = 0.5 * __builtin_fabs(d1 - d2_);
            const double f = 1.0 / (6.0 * A);
            const double cx = __builtin_fabs(__builtin_fabs(f * sx)), cy = __builtin_fabs(__builtin_fabs(f * sy));
            if (__builtin_fabs(sp.x - cx) > P.obs_range || __builtin_fabs(sp.y - cy) > P.obs_range) keep = false;
        }
        long long minr = (long long)fmax(0.0, rmin), maxr = (long long)__builtin_ceil(rmax);
        long long minc = (long long)fmax(0.0, cmin), maxc = (long long)__builtin_ceil(cmax);
        if (maxr > bh - 1) maxr = bh - 1;
        if (maxc > bw - 1) maxc = bw - 1;
        if (maxr < gi0 || minr > gi1 || maxc < gj0 || minc > gj1 || maxr < minr || maxc < minc) keep = false;
        if (keep) {
            s_bbx[k][0] = (int)max(minr, (long long)gi0); s_bbx[k][1] = (int)min(maxr, (long long)gi1);
            s_bbx[k][2] = (int)max(minc, (long long)gj0); s_bbx[k][3] = (int)min(maxc, (long long)gj1);
        } else { s_bbx[k][0] = 0; s_bbx[k][1] = -1; s_bbx[k][2] = 0; s_bbx[k][3] = -1; }
    }
    __syncthreads();
    OPROF(4)
    // ---- 4. occupancy: skimage.draw.polygon of each candidate as exact scanline spans (raster_row_convex): OBS_THREADS_SHIP / OBS_CHUNK
    //         threads per candidate walk its raster rows ----
    {
        int TPC = OBS_THREADS_SHIP / OBS_CHUNK;                 // threads per candidate: all 256 threads share the chunk's candidates
        while (TPC * ncand * 2 <= OBS_THREADS_SHIP) TPC *= 2;
        const int k = tid / TPC, sub = tid - k * TPC;
        if (k < ncand && s_bbx[k][1] >= s_bbx[k][0]) {
            const int c0 = max(s_bbx[k][2], 0), c1 = min(s_bbx[k][3], Wg - 1);
            const int n = s_cn[k];
            const double *xp = s_px + k * BP_MAXV, *yp = s_py + k * BP_MAXV;
            for (int gi = s_bbx[k][0] + sub; gi <= s_bbx[k][1]; gi += TPC) {
                if (gi < 0 || gi >= Hg) continue;
                raster_row_convex<false>(xp, yp, n, gi, c0, c1, s_img + (gi - gi0) * LW - gj0, (unsigned char)OBS_F_OCC);
            }
        }
    }
    __syncthreads();
    }
    OPROF(5)
    // ---- 5. footprint (skimage.draw.polygon of the hull outline, exact test over its pixel box) and heading line -> flag bits.
    //         The first wavefront sets the ship bits, the second the heading line and the head pixel, both with word atomics (the two overlap,
    //         and a word may straddle two rows), so they need no order between them; the occupancy stores are behind the barrier above. ----
    if (tid < 64) {
        if (s_fcnt > 0) {   // the hull outline (vertices outside the grid dropped) stays convex: same exact scanline spans as the floes
            const int r0 = max(s_fbb[0], max(gi0, 0)), r1 = min(s_fbb[1], min(gi1, Hg - 1));
            const int c0 = max(s_fbb[2], max(gj0, 0)), c1 = min(s_fbb[3], min(gj1, Wg - 1));
            if (c1 >= c0)
                for (int gi = r0 + tid; gi <= r1; gi += 64)
                    raster_row_convex<true>(s_fc, s_fr, s_fcnt, gi, c0, c1, s_img, (unsigned char)OBS_F_SHIP, (gi - gi0) * LW - gj0);
        }
    } else if (tid < 128) {
        const int l2 = tid - 64;
        if (line.valid) {   // cv2.line: pixel t of the 8-connected walk, t = 0 .. dmaj (closed form of on_line)
            const long long dmaj = line.vert ? line.dy : line.dx, dmin = line.vert ? line.dx : line.dy;
            for (long long t = l2; t <= dmaj; t += 64) {
                // coordinates are u16 (cast by the reference); inside the 1000 x 300 map 2 * dmin * t + dmaj fits u32
                const long long mt = (dmaj == 0) ? 0 : ((dmaj < 32768 && dmin < 32768)
                                         ? (long long)(((unsigned)(2 * dmin * t) + (unsigned)dmaj - 1u) / (unsigned)(2 * dmaj))
                                         : (2 * dmin * t + dmaj - 1) / (2 * dmaj));
                const long long ax = line.vert ? mt : t, ay = line.vert ? t : mt;
                const long long gj = line.x0 + ax * line.sx, gi = line.y0 + ay * line.sy;
                if (gi >= gi0 && gi <= gi1 && gj >= gj0 && gj <= gj1 && gi >= 0 && gi < Hg && gj >= 0 && gj < Wg) {
                    const int b = ((int)gi - gi0) * LW + ((int)gj - gj0);
                    obs_or_bytes((unsigned *)s_img, b, b, (unsigned char)OBS_F_LINE);
                }
            }
        }
        if (l2 == 0 && hpy >= gi0 && hpy <= gi1 && hpx >= gj0 && hpx <= gj1) {
            const int b = ((int)hpy - gi0) * LW + ((int)hpx - gj0);
            obs_or_bytes((unsigned *)s_img, b, b, (unsigned char)OBS_F_HEAD);
        }
    }
    __syncthreads();
    OPROF(6)
    // ---- 6. compose the four channels from the flag image, 4 pixels per 32-bit store ----
    //   ch0: in map 127, ship 255, out of map 0; ch1: row value, out of map 255; ch2: line 127, head 255; ch3: occupied 255
    const size_t plane = (size_t)npix;
    unsigned *o32 = (unsigned *)(obs + (size_t)env * BP_OBS_C * plane);
    const int nwords = npix / 4;
    // LW and LH are even (checked on the host), so two window rows are WP = LW / 2 whole words.  Thread (g, j) composes word j of the row pairs
    // g, g + G, ...: which of its bytes lie in the first row of the pair and which lie inside the map's columns depends on j alone, so the byte
    // masks are computed once per thread; per word only the two rows' in-map tests and goal-distance bytes change.
    const int WP = LW / 2, G = OBS_THREADS_SHIP / WP;
    const int g = tid / WP, j = tid - g * WP;
    const int ljlo = max(0, -gj0), ljhi = min(LW - 1, Wg - 1 - gj0);
    if (g < G) {
        const int bs = LW - 4 * j;             // bytes [0, bs) of the word lie in the first row of the pair
        const unsigned mlo = obs_upto(bs);
        const unsigned mA = mlo & obs_upto(ljhi - 4 * j + 1) & ~obs_upto(ljlo - 4 * j);
        const unsigned mB = ~mlo & obs_upto(bs + ljhi + 1) & ~obs_upto(bs + ljlo);
        for (int rp = g; 2 * rp < LH; rp += G) {
            const int w = rp * WP + j;
            const unsigned f = ((const unsigned *)s_img)[w];
            const int giA = gi0 + 2 * rp, giB = giA + 1;
            const unsigned rowA = (giA >= 0 && giA < Hg) ? 0xFFFFFFFFu : 0u, rowB = (giB >= 0 && giB < Hg) ? 0xFFFFFFFFu : 0u;
            const unsigned inb = (rowA & mA) | (rowB & mB);
            const unsigned e2 = ((const unsigned short *)s_edt)[rp];
            const unsigned eA = (e2 & 0xFFu) * 0x01010101u, eB = (e2 >> 8) * 0x01010101u;
            const unsigned e = ((eA & mlo) | (eB & ~mlo)) & inb;
            const unsigned ship = (f >> 1) & 0x01010101u, ln = (f >> 2) & 0x01010101u, hd = (f >> 3) & 0x01010101u;
            o32[w] = inb & (0x7F7F7F7Fu + ship * 0x80u);
            o32[nwords + w] = e | ~inb;
            o32[2 * nwords + w] = inb & ((ln & ~hd) * 127u + hd * 255u);
            o32[3 * nwords + w] = (f & 0x01010101u) * 255u;
        }
    }
    OPROF(7)
}

// ------------------------------------------------------------------------------------------------------------
// k_observe_maze: maze-NAMO-v0 observation, uint8 [4][192][192] per env (maze_NAMO_env.py:514-525 ->
// OccupancyGrid.ego_view_map_maze, occupancy_map.py:142-202): channels [robot footprint, boxes, walls, goal map];
// a 288x288 axis-aligned window around the robot is rotated by (heading - pi/2) with scipy.ndimage.rotate
// (order 1, reshape=False, cval 0/0/0/1) and its centre 192x192 kept.
// The boxes and the footprint are rasterised (skimage.draw.polygon rule) into two LDS bit-images of the window; the
// static wall / goal maps are read from HBM; every output pixel then evaluates the order-1 spline of
// NI_GeometricTransform on its 2x2 source cells.
// ------------------------------------------------------------------------------------------------------------
#define MZ_MAXBOX 64
__global__ __launch_bounds__(OBS_THREADS) void k_observe_maze(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                              unsigned char *__restrict__ obs)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    const int tid = threadIdx.x;
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap;
    const int nb = D.e_nb[env];
    const size_t tb = (size_t)D.e_trial[env] * nbcap;
    const d2 *wv = D.wv + eb * BP_MAXV;
    const int *nv = D.sc_nv + tb;
    const int *kind = D.sc_kind + tb;
    const int Hg = P.grid_h, Wg = P.grid_w, LH = P.obs_h, LW = P.obs_w;
    const int infl = (LW > LH ? LW : LH) / 2;
    const int IH = LH + infl, IW = LW + infl;
    const int nwords = (IH * IW + 31) / 32;

    extern __shared__ double2 obs_smem[];
    unsigned *s_box = (unsigned *)obs_smem;          // bit-image of the window: boxes
    unsigned *s_foot = s_box + nwords;               // bit-image of the window: robot footprint
    double *s_px = (double *)(s_foot + ((nwords + 1) & ~1));  // [MZ_MAXBOX][4] box vertices in raster coordinates
    double *s_py = s_px + MZ_MAXBOX * 4;
    __shared__ int s_nbox;
    __shared__ int s_bbx[MZ_MAXBOX][4];
    __shared__ double s_fr[BP_MAX_SHIP_VERTS], s_fc[BP_MAX_SHIP_VERTS];
    __shared__ int s_fcnt, s_fbb[4];
    if (tid == 0) s_nbox = 0;
    for (int w = tid; w < nwords; w += OBS_THREADS) { s_box[w] = 0u; s_foot[w] = 0u; }
    __syncthreads();

    const d2 sp = D.pxy[eb];
    const double sa = D.ang[eb];
    const d2 srot = D.rot[eb];
    const int wx = (int)(sp.x * P.m_to_pix), wy = (int)(sp.y * P.m_to_pix);
    const int gi0 = (int)((double)(0 + wy) - ((double)IH / 2)), gj0 = (int)((double)(0 + wx) - ((double)IW / 2));
    const int gi1 = gi0 + IH - 1, gj1 = gj0 + IW - 1;
    // boxes (collision type 2): compute_occ_img without range culling (occupancy_map.py:37-65)
    for (int s = tid; s < nb; s += OBS_THREADS) {
        if (kind_ctype(kind[s]) != 2) continue;
        const int n = nv[s];
        const d2 *v = wv + (size_t)s * BP_MAXV;
        double rmin = v[0].y * P.m_to_pix, rmax = rmin, cmin = v[0].x * P.m_to_pix, cmax = cmin;
        for (int i = 1; i < n; i++) {
            const double r = v[i].y * P.m_to_pix, cc = v[i].x * P.m_to_pix;
            rmin = fmin(rmin, r); rmax = fmax(rmax, r); cmin = fmin(cmin, cc); cmax = fmax(cmax, cc);
        }
        long long minr = (long long)fmax(0.0, rmin), maxr = (long long)__builtin_ceil(rmax);
        long long minc = (long long)fmax(0.0, cmin), maxc = (long long)__builtin_ceil(cmax);
        if (maxr > Hg - 1) maxr = Hg - 1;
        if (maxc > Wg - 1) maxc = Wg - 1;
        if (maxr < gi0 || minr > gi1 || maxc < gj0 || minc > gj1 || maxr < minr || maxc < minc) continue;
        const int slot = atomicAdd(&s_nbox, 1);
        if (slot < MZ_MAXBOX && n <= 4) {
            s_bbx[slot][0] = (int)max(minr, (long long)gi0); s_bbx[slot][1] = (int)min(maxr, (long long)gi1);
            s_bbx[slot][2] = (int)max(minc, (long long)gj0); s_bbx[slot][3] = (int)min(maxc, (long long)gj1);
            for (int i = 0; i < 4; i++) { s_px[slot * 4 + i] = v[i].x * P.m_to_pix; s_py[slot * 4 + i] = v[i].y * P.m_to_pix; }
        }
    }
    // robot footprint polygon (_compute_global_footprint_maze, occupancy_map.py:340-376)
    if (tid == 0) {
        const double ch = srot.x, sh = srot.y;
        const double m2gx = (double)Wg / P.map_w, m2gy = (double)Hg / P.map_h;
        int cnt = 0;
        double rmin = BP_INF, rmax = -BP_INF, cmin = BP_INF, cmax = -BP_INF;
        for (int i = 0; i < P.num_ship_verts; i++) {
            const double vx = P.ship_verts[i][0] * ch + P.ship_verts[i][1] * -sh + sp.x;
            const double vy = P.ship_verts[i][0] * sh + P.ship_verts[i][1] * ch + sp.y;
            const double gx = vx * m2gx, gy = vy * m2gy;
            if (gy < 0 || gy >= Hg || gx < 0 || gx >= Wg) continue;
            s_fr[cnt] = gy; s_fc[cnt] = gx; cnt++;
            rmin = fmin(rmin, gy); rmax = fmax(rmax, gy); cmin = fmin(cmin, gx); cmax = fmax(cmax, gx);
        }
        s_fcnt = cnt;
        if (cnt > 0) {
            s_fbb[0] = max((int)(long long)fmax(0.0, rmin), gi0); s_fbb[1] = min((int)(long long)__builtin_ceil(rmax), gi1);
            s_fbb[2] = max((int)(long long)fmax(0.0, cmin), gj0); s_fbb[3] = min((int)(long long)__builtin_ceil(cmax), gj1);
        }
    }
    __syncthreads();
    const int nbox = min(s_nbox, MZ_MAXBOX);
    if (tid == 0 && s_nbox > MZ_MAXBOX) atomicOr(&D.e_err[env], BP_ERR_LEVEL_OVERFLOW);

    for (int kx = 0; kx < nbox; kx++) {
        const int r0 = s_bbx[kx][0], r1 = s_bbx[kx][1], c0 = s_bbx[kx][2], c1 = s_bbx[kx][3];
        const int wbox = c1 - c0 + 1, npx = (r1 - r0 + 1) * wbox;
        for (int q = tid; q < npx; q += OBS_THREADS) {
            const int rr = q / wbox, cc = q - rr * wbox;
            const int gi = r0 + rr, gj = c0 + cc;
            if (gi < 0 || gi >= Hg || gj < 0 || gj >= Wg) continue;
            if (pip_arrays(s_px + kx * 4, s_py + kx * 4, 4, (double)gj, (double)gi)) {
                const int bit = (gi - gi0) * IW + (gj - gj0);
                atomicOr(&s_box[bit >> 5], 1u << (bit & 31));
            }
        }
    }
    if (s_fcnt > 0) {
        const int r0 = s_fbb[0], r1 = s_fbb[1], c0 = s_fbb[2], c1 = s_fbb[3];
        const int wbox = c1 - c0 + 1, npx = (r1 >= r0 && c1 >= c0) ? (r1 - r0 + 1) * wbox : 0;
        for (int q = tid; q < npx; q += OBS_THREADS) {
            const int rr = q / wbox, cc = q - rr * wbox;
            const int gi = r0 + rr, gj = c0 + cc;
            if (gi < 0 || gi >= Hg || gj < 0 || gj >= Wg) continue;
            if (pip_arrays(s_fc, s_fr, s_fcnt, (double)gj, (double)gi)) {
                const int bit = (gi - gi0) * IW + (gj - gj0);
                atomicOr(&s_foot[bit >> 5], 1u << (bit & 31));
            }
        }
    }
    __syncthreads();
    // rotation (scipy.ndimage.rotate, reshape=False, order=1): input = rot @ output + offset, rot = [[c, s], [-s, c]]
    double rs, rc;
    bp_sincos(sa - BP_PI / 2, rs, rc);
    const double ctr = ((double)IH - 1) / 2;
    const double oc0 = rc * ctr + rs * ctr, oc1 = -rs * ctr + rc * ctr;
    const double off0 = ctr - oc0, off1 = ctr - oc1;
    const int half = infl / 2;
    const size_t plane = (size_t)LH * LW;
    unsigned char *o = obs + (size_t)env * BP_OBS_C * plane;
    // one output pixel = scipy's order-1 sample of the four channels (2 x 2 cells, per-cell ((v * w_row) * w_col) accumulated row-major); a thread takes four
    // neighbouring pixels of a row and stores one 32-bit word per channel (byte stores were a quarter of this kernel's instructions)
    // Branch-free: the sixteen cell reads of a thread's four pixels (one global word + two LDS words each) are issued together instead of one dependent
    // round trip per cell behind a branch; cells outside the window or the map are read at a clamped address and replaced by the channels' cval.
    // One output pixel = scipy's order-1 sample of the four channels (2 x 2 cells, per-cell ((v * w_row) * w_col) accumulated row-major).  Footprint, boxes
    // and walls are 0 / 1: 1 * w_row is w_row and 0 * w is +0, which a non-negative sum absorbs bit for bit, so those three channels add the cell's weight
    // product or nothing; the distance channel keeps its two products.  Walls and distances come from ONE word per cell (sign = wall, magnitude = normalised
    // distance, bp_load_maze).  Branch-free: cells outside the window or the map are read at a clamped address and replaced by the channels' cval.  A thread
    // takes four neighbouring pixels of a row and stores one 32-bit word per channel.  The kernel is bound by its arithmetic (about 700 cycles per 64 pixels,
    // a good part of it quarter-rate int <-> double conversions), not by memory: without any load or store it still takes 0.94 of 1.17 ms (r05 variants:
    // two-phase fetch / blend, 8 x 8 output tiles per wavefront and a wave-level skip of the empty binary channels were all slower or equal).
    const int nbits = IH * IW;
    auto sample = [&](const double di, const double dj, unsigned &b0, unsigned &b1, unsigned &b2, unsigned &b3) {
        double c0 = 0.0, c1 = 0.0;
        c0 += di * rc; c0 += dj * rs; c0 += off0;
        c1 += di * -rs; c1 += dj * rc; c1 += off1;
        const bool outside = (c0 < 0 || c0 > IH - 1 || c1 < 0 || c1 > IW - 1);
        const double c0c = outside ? 0.0 : c0, c1c = outside ? 0.0 : c1;
        const double fl0 = __builtin_floor(c0c), fl1 = __builtin_floor(c1c);   // the double of the integer cell index: x = c - floor(c) needs no conversion back
        const int s0 = (int)fl0, s1 = (int)fl1;
        const double x0 = c0c - fl0, x1 = c1c - fl1;
        const double w0[2] = {1.0 - x0, x0}, w1[2] = {1.0 - x1, x1};
        double pm[4];
        unsigned wf[4], wb[4];
        bool in[4];
        int sh[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ii = s0 + (k >> 1), jj = s1 + (k & 1);
            const int gi = gi0 + ii, gj = gj0 + jj;
            in[k] = !(ii > IH - 1 || jj > IW - 1) && !(gi < 0 || gi >= Hg || gj < 0 || gj >= Wg);   // outside the map the window keeps its initial values
            const int gic = min(max(gi, 0), Hg - 1), gjc = min(max(gj, 0), Wg - 1);
            const int bit = min(max(ii * IW + jj, 0), nbits - 1);
            sh[k] = bit & 31;
            pm[k] = *(const double *)((const char *)D.maze_obs_map + (size_t)((unsigned)(gic * Wg + gjc) * 8u));
            wf[k] = s_foot[bit >> 5]; wb[k] = s_box[bit >> 5];
        }
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int a = k >> 1, b = k & 1;
            const double ww = w0[a] * w1[b];
            const bool f0 = in[k] && ((wf[k] >> sh[k]) & 1u) != 0u, f1 = in[k] && ((wb[k] >> sh[k]) & 1u) != 0u, f2 = in[k] && pm[k] < 0.0;
            const double v3 = in[k] ? __builtin_fabs(pm[k]) : 1.0;   // cval of the distance channel
            t0 += f0 ? ww : 0.0;
            t1 += f1 ? ww : 0.0;
            t2 += f2 ? ww : 0.0;
            double q3 = v3;
            q3 *= w0[a]; q3 *= w1[b]; t3 += q3;
        }
        if (outside) { t0 = 0.0; t1 = 0.0; t2 = 0.0; t3 = 1.0; }
        b0 = (unsigned)(unsigned char)(t0 * 255); b1 = (unsigned)(unsigned char)(t1 * 255);
        b2 = (unsigned)(unsigned char)(t2 * 255); b3 = (unsigned)(unsigned char)(t3 * 255);
    };
    if ((LW & 3) == 0) {
        unsigned *o32 = (unsigned *)o;
        const int wpr = LW / 4;
        for (int w = tid; w < LH * wpr; w += OBS_THREADS) {
            const int oi = w / wpr, oj = (w - oi * wpr) * 4;
            const double di = (double)(half + oi), dj = (double)(half + oj);   // small integers: dj + k is exactly (double)(j + k)
            unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned b0, b1, b2, b3;
                sample(di, dj + (double)k, b0, b1, b2, b3);
                p0 |= b0 << (8 * k); p1 |= b1 << (8 * k); p2 |= b2 << (8 * k); p3 |= b3 << (8 * k);
            }
            const size_t wi = (size_t)oi * wpr + (oj >> 2);
            o32[0 * (plane / 4) + wi] = p0; o32[1 * (plane / 4) + wi] = p1; o32[2 * (plane / 4) + wi] = p2; o32[3 * (plane / 4) + wi] = p3;
        }
    } else {
        for (int px = tid; px < LH * LW; px += OBS_THREADS) {
            const int oi = px / LW, oj = px - oi * LW;
            unsigned b0, b1, b2, b3;
            sample((double)(half + oi), (double)(half + oj), b0, b1, b2, b3);
            o[0 * plane + px] = (unsigned char)b0; o[1 * plane + px] = (unsigned char)b1; o[2 * plane + px] = (unsigned char)b2; o[3 * plane + px] = (unsigned char)b3;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// k_observe_global: planner observation of ship-ice (cfg.egocentric_obs: false, ship_ice_env.py:96-99,394-406):
// uint8 [2][map_h/0.2][map_w/0.2]; ch0 = 5x5 block mean of the full 25 px/m occupancy raster (every floe, no range
// culling; occupancy_map.py:37-65,97-109), ch1 = compute_ship_footprint_planner on the 0.2 m grid (:253-296).
// The full 1000x300 raster lives as a bit-image in LDS (37.5 KB): floes are scattered into it polygon by polygon
// with the skimage.draw.polygon rule, then each coarse cell counts its 25 bits.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OBS_THREADS) void k_observe_global(const DevParams P, const DevPtrs D, const unsigned char *__restrict__ mask,
                                                                unsigned char *__restrict__ obs, const int cell_px)
{
    const int env = blockIdx.x;
    if (mask != nullptr && mask[env] == 0) return;
    const int tid = threadIdx.x;
    const int nbcap = P.nbcap;
    const size_t eb = (size_t)env * nbcap;
    const int nb = D.e_nb[env];
    const d2 *wv = D.wv + eb * BP_MAXV;
    const int *nv = D.sc_nv + (size_t)D.e_trial[env] * nbcap;
    const int bh = P.grid_h, bw = P.grid_w;              // fine raster (25 px/m)
    const int Hc = bh / cell_px, Wc = bw / cell_px;      // coarse grid
    const int nwords = (bh * bw + 31) / 32;
    extern __shared__ double2 obs_smem[];
    unsigned *s_bits = (unsigned *)obs_smem;
    __shared__ double s_x[BP_MAXV], s_y[BP_MAXV];
    __shared__ int s_box[4];
    __shared__ double s_fr[BP_MAX_SHIP_VERTS], s_fc[BP_MAX_SHIP_VERTS];
    __shared__ int s_fcnt;
    for (int w = tid; w < nwords; w += OBS_THREADS) s_bits[w] = 0u;
    __syncthreads();
    for (int s = 1; s < nb; s++) {
        const int n = nv[s];
        if (tid < n) { const d2 v = wv[(size_t)s * BP_MAXV + tid]; s_x[tid] = v.x * P.m_to_pix; s_y[tid] = v.y * P.m_to_pix; }
        __syncthreads();
        if (tid == 0) {
            double rmin = s_y[0], rmax = rmin, cmin = s_x[0], cmax = cmin;
            for (int i = 1; i < n; i++) { rmin = fmin(rmin, s_y[i]); rmax = fmax(rmax, s_y[i]); cmin = fmin(cmin, s_x[i]); cmax = fmax(cmax, s_x[i]); }
            long long minr = (long long)fmax(0.0, rmin), maxr = (long long)__builtin_ceil(rmax);
            long long minc = (long long)fmax(0.0, cmin), maxc = (long long)__builtin_ceil(cmax);
            if (maxr > bh - 1) maxr = bh - 1;
            if (maxc > bw - 1) maxc = bw - 1;
            s_box[0] = (int)minr; s_box[1] = (int)maxr; s_box[2] = (int)minc; s_box[3] = (int)maxc;
        }
        __syncthreads();
        const int r0 = s_box[0], r1 = s_box[1], c0 = s_box[2], c1 = s_box[3];
        if (r1 >= r0 && c1 >= c0) {
            const int wbox = c1 - c0 + 1, npx = (r1 - r0 + 1) * wbox;
            for (int q = tid; q < npx; q += OBS_THREADS) {
                const int rr = q / wbox, cc = q - rr * wbox;
                const int gi = r0 + rr, gj = c0 + cc;
                if (pip_arrays(s_x, s_y, n, (double)gj, (double)gi)) {
                    const int bit = gi * bw + gj;
                    atomicOr(&s_bits[bit >> 5], 1u << (bit & 31));
                }
            }
        }
        __syncthreads();
    }
    // ship footprint on the coarse grid
    const d2 sp = D.pxy[eb];
    const d2 srot = D.rot[eb];
    if (tid == 0) {
        const double ch = srot.x, sh = srot.y;
        const double m2gx = (double)Wc / P.map_w, m2gy = (double)Hc / P.map_h;
        int cnt = 0;
        for (int i = 0; i < P.num_ship_verts; i++) {
            const double vx = P.ship_verts[i][0] * ch + P.ship_verts[i][1] * -sh + sp.x;
            const double vy = P.ship_verts[i][0] * sh + P.ship_verts[i][1] * ch + sp.y;
            const double gx = vx * m2gx, gy = vy * m2gy;
            if (gy < 0 || gy >= Hc || gx < 0 || gx >= Wc) continue;
            s_fr[cnt] = gy; s_fc[cnt] = gx; cnt++;
        }
        s_fcnt = cnt;
    }
    __syncthreads();
    const size_t plane = (size_t)Hc * Wc;
    unsigned char *o = obs + (size_t)env * 2 * plane;
    const double denom = (double)(cell_px * cell_px);
    for (int cidx = tid; cidx < Hc * Wc; cidx += OBS_THREADS) {
        const int gi = cidx / Wc, gj = cidx - gi * Wc;
        int cnt = 0;
        for (int a = 0; a < cell_px; a++)
            for (int b = 0; b < cell_px; b++) {
                const int bit = (gi * cell_px + a) * bw + (gj * cell_px + b);
                cnt += (s_bits[bit >> 5] >> (bit & 31)) & 1u;
            }
        const double mean = (double)cnt / denom; // block sum is an exact integer; np.mean divides it by 25
        o[cidx] = (unsigned char)(mean * 255);
        unsigned char f = 0;
        if (s_fcnt > 0 && pip_arrays(s_fc, s_fr, s_fcnt, (double)gj, (double)gi)) f = 255;
        o[plane + cidx] = f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Planner cost map (common/cost_map.py:27-126,284-287): kinetic-energy-loss cost of every floe, max-combined per cell.
// k_costmap_init clears the maps and writes the boundary columns; k_costmap handles one floe per wavefront:
// scale, horizon cull, resample_vertices(decimals=0), skimage.draw.polygon over the floe's pixel box (pass 1: pixel
// count and coordinate sums -> pixel centroid; pass 2: cost), poly_radius, poly_area, and an atomic max per cell on the
// (non-negative) float64 bit pattern -- max is exact, so the order of the floes does not matter.
// ------------------------------------------------------------------------------------------------------------
#define BP_MAX_COST 1e10
__global__ __launch_bounds__(256) void k_costmap_init(double *__restrict__ out, int E, int H, int W, int margin)
{
    const size_t total = (size_t)E * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % W);
        out[i] = (margin && (j < margin || j >= W - margin)) ? BP_MAX_COST : 0.0;
    }
}
__global__ __launch_bounds__(256) void k_costmap(const DevParams P, const DevPtrs D, const double scale, const int H, const int W,
                                                 const double alpha, const double ship_mass, const double hz,
                                                 const double *__restrict__ ship_pos_y, const double vs, double *__restrict__ out)
{
    const int env = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = P.nkin + blockIdx.y * 4 + wave;
    if (s >= D.e_nb[env]) return;
    __shared__ double s_c[4][BP_MAXV], s_r[4][BP_MAXV];
    const size_t eb = (size_t)env * P.nbcap;
    const int n0 = D.sc_nv[(size_t)D.e_trial[env] * P.nbcap + s];
    const double spy = ship_pos_y ? ship_pos_y[env] : 0.0;
    double ox = 0.0, oy = 0.0;
    if (lane < n0) { const d2 v = D.wv[(eb + s) * BP_MAXV + lane]; ox = v.x * scale; oy = v.y * scale; }
    if (hz != 0.0) {   // discard obstacles that lie entirely outside [ship_pos_y, ship_pos_y + horizon]
        const bool inside = (lane < n0) && !(oy > (spy + hz) || oy < spy);
        if (ballot(inside) == 0ull) return;
    }
    // resample_vertices(decimals=0): a vertex whose rounded coordinates repeat an earlier vertex's is dropped
    const double rx = __builtin_rint(ox), ry = __builtin_rint(oy);
    bool dup = false;
    for (int j = 0; j < n0; j++) {
        const double jx = __shfl(rx, j), jy = __shfl(ry, j);
        if (j < lane && jx == rx && jy == ry) dup = true;
    }
    const bool keep = (lane < n0) && !dup;
    const unsigned long long km = ballot(keep);
    const int k = __popcll(km);
    if (keep) { const int pos = popc_below(km, lane); s_c[wave][pos] = ox; s_r[wave][pos] = oy; }
    lds_sync();
    const double *c = s_c[wave], *r = s_r[wave];
    double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
    for (int i = 1; i < k; i++) { rmin = fmin(rmin, r[i]); rmax = fmax(rmax, r[i]); cmin = fmin(cmin, c[i]); cmax = fmax(cmax, c[i]); }
    long long minr = (long long)fmax(0.0, rmin), maxr = (long long)__builtin_ceil(rmax);
    long long minc = (long long)fmax(0.0, cmin), maxc = (long long)__builtin_ceil(cmax);
    if (maxr > H - 1) maxr = H - 1;
    if (maxc > W - 1) maxc = W - 1;
    if (maxr < minr || maxc < minc) return;
    const int wbox = (int)(maxc - minc + 1), npx = (int)(maxr - minr + 1) * wbox;
    // pass 1: pixels inside (skimage.draw.polygon) -> count and coordinate sums (integers: exact in any order)
    long long cnt = 0, sr = 0, sc = 0;
    for (int q = lane; q < npx; q += 64) {
        const int rr = q / wbox, cc = q - rr * wbox;
        const long long ri = minr + rr, ci = minc + cc;
        if (pip_arrays(c, r, k, (double)ci, (double)ri)) { cnt++; sr += ri; sc += ci; }
    }
    for (int off = 32; off >= 1; off >>= 1) { cnt += __shfl_xor(cnt, off); sr += __shfl_xor(sr, off); sc += __shfl_xor(sc, off); }
    if (cnt == 0) return;
    const double cx = (double)sc / (double)cnt, cy = (double)sr / (double)cnt;
    double rad = 0.0;   // poly_radius: largest vertex distance from the pixel centroid
    for (int i = 0; i < k; i++) {
        const double d = __builtin_sqrt((c[i] - cx) * (c[i] - cx) + (r[i] - cy) * (r[i] - cy));
        if (i == 0 || d > rad) rad = d;
    }
    double d1 = 0.0, d2_ = 0.0;   // poly_area(vertices / scale), sequential sums
    for (int i = 0; i < k; i++) {
        const int p = (i - 1 + k) % k;
        d1 += (c[i] / scale) * (r[p] / scale);
        d2_ += (r[i] / scale) * (c[p] / scale);
    }
    const double mi = 0.5 * __builtin_fabs(d1 - d2_);
    const double norm = alpha * ((vs * vs) * (mi * mi)) / (2 * (ship_mass + mi));
    unsigned long long *o = (unsigned long long *)(out + (size_t)env * H * W);
    for (int q = lane; q < npx; q += 64) {
        const int rr = q / wbox, cc = q - rr * wbox;
        const long long ri = minr + rr, ci = minc + cc;
        if (pip_arrays(c, r, k, (double)ci, (double)ri)) {
            const double dist = __builtin_sqrt(((double)ri - cy) * ((double)ri - cy) + ((double)ci - cx) * ((double)ci - cx));
            const double nc = fmax(0.0, (rad * rad - dist * dist) / (rad * rad));
            const double v = fmin(BP_MAX_COST, nc * norm);
            atomicMax(&o[(size_t)ri * W + ci], (unsigned long long)__double_as_longlong(v + 0.0));
        }
    }
}

