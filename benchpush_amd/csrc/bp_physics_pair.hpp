// Two environments per wavefront (ship-ice, space.damping == 0): lanes 0..31 serve one environment, lanes 32..63 another, under ONE instruction
// stream.  The mean env keeps 4 moving bodies, 3 manifolds and 3 warm arbiters busy -- 28 % of the lanes of a wavefront of its own -- and its sub-step
// is bound by the number of wave-instructions issued, so serving two light environments per instruction halves what each
// of them costs wherever a phase is overhead rather than items.  substep_pair() is substep<BP_ENV_SHIP_ICE, false>() of bp_physics.hpp restated for
// half-waves:
//   * every quantity that is wave-uniform there (trip counts, ballots, the sub-step state) is uniform per HALF here and lives in a VGPR; a ballot is
//     the half's 32 bits of the wave ballot (lanes whose half does not take a branch are masked off by EXEC and contribute zeros);
//   * loops run to the larger of the two trip counts, branches are taken when either half takes them;
//   * a value that travelled by v_readlane with a uniform lane index (pair keys to the arbiter lanes, colouring, ordered sums) goes through a few words
//     of the half's LDS instead;
//   * each half has its own 10 240-byte LDS image (40 velocity slots, 32 arbiter lanes, 48 support queries, 32-entry pair tables), two halves = the
//     20 480 bytes of a solo wavefront, so paired and solo workgroups share the launch and the occupancy (two waves per SIMD).
// The arithmetic of every item is the solo kernel's, instruction for instruction, and min / max / atomics are order-free, so the results are those of
// k_physics_step bit for bit (tests/test_gpu_pair.py).  An environment that outgrows the half-wave capacities -- or simply turns heavy -- is parked at
// a sub-step boundary exactly like at a chunk boundary of the preemptive scheduler and carries on in a wavefront of its own.
#pragma once
#include "bp_physics.hpp"

// pair_task is inlined into the scheduler kernel beside the solo step body (as a function of its own -- __attribute__((noinline)) -- the launch was 20 %
// slower: 1.7 KB of stack per lane for the by-value structs)
#ifndef BP_PAIR_TASK_INLINE
#define BP_PAIR_TASK_INLINE __forceinline__
#endif
#define PP_NSLOT 40        // velocity slots per half (+ 1 scratch slot)
#define PP_QCAP 48         // support queries per batch
#define PP_MBOX 8          // manifold mailbox entries per hand-over batch
#define PP_MVCAP 68        // moving list: the ship + two bodies per arbiter lane at most (65)
#define PP_NBCAP 448       // body slots per env the half-wave image is laid out for (30 %: 272, 50 %: 360; bp_load_scenarios enables pairing only up to this)
#define PP_CC 64           // candidate-cache entries: the first TWO candidate rounds of 32 (the mean env has 34 candidate slots)

// LDS image of one half (byte offsets from the half's base).  Compile-time constants: the accesses fold them into the instruction's offset field.
enum : unsigned {
    PL_SV = 0, PL_SW = PL_SV + 16u * (PP_NSLOT + 1), PL_SB = PL_SW + 16u * (PP_NSLOT + 1), PL_SP = PL_SB + 16u * (PP_NSLOT + 1),
    PL_AG = PL_SP + 16u * (PP_NSLOT + 1),
    PL_QDIR = PL_AG + 32u,                       // also: transforms of the integrate phase ([32][2] d2), manifold mailbox, ordered-sum scratch
    PL_QC = PL_QDIR + 16u * PP_QCAP, PL_RVAL = PL_QC + 8u * PP_QCAP, PL_QMETA = PL_RVAL + 8u * PP_QCAP, PL_QAUX = PL_QMETA + 4u * PP_QCAP,
    PL_RIDX = PL_QAUX + 4u * PP_QCAP,
    PL_PTA = PL_RIDX + 4u * PP_QCAP, PL_PTTHR = PL_PTA + 16u * 32,
    PL_CC = PL_PTTHR + 16u * 32, PL_CCHW = PL_CC + 8u * PP_CC,
    PL_RSMA = PL_CCHW + 8u * PP_CC,              // res_smA .. res_jB contiguous: the AABB keys of the integrate phase ([32][4] u64) alias them
    PL_RSMB = PL_RSMA + 8u * 32, PL_RIA = PL_RSMB + 8u * 32, PL_RIB = PL_RIA + 4u * 32, PL_RJA = PL_RIB + 4u * 32, PL_RJB = PL_RJA + 4u * 32,
    PL_MVS = PL_RJB + 4u * 32,
    PL_OWNER = PL_MVS + 2u * PP_NBCAP,           // mvs: 16-bit stamps relative to the task's first sub-step (at most 400 per step)
    PL_COLMASK = PL_OWNER + 2u * (PP_NSLOT + 4), PL_MVO = PL_COLMASK + 2u * (PP_NSLOT + 4),
    PL_MV = PL_MVO + 4u * (PP_NSLOT + 2),
    PL_SLOTOF = PL_MV + 2u * PP_MVCAP,
    PL_RF = PL_SLOTOF + PP_NBCAP,
    PL_HS = (PL_RF + 32u + 7u) & ~7u,            // half scalars: [0] curr_dt, [1] ecoef_e, [2] ecoef, [3] scratch double, then 8 u32 words
    PL_KQ = PL_HS + 8u * 8,                      // u32 [32] keys / colouring records
    PL_CQ = PL_KQ + 4u * 32,
    PL_END = PL_CQ + 4u * 32,
    PL_HALF = 10240u
};
static_assert(PL_END <= PL_HALF, "half-wave LDS image exceeds 10 240 bytes");
static_assert(16u * 32 * 2 <= 16u * PP_QCAP + 8u * PP_QCAP, "the integrate transforms alias q_dir + q_c");
static_assert(96u * PP_MBOX <= 16u * PP_QCAP, "the manifold mailbox aliases q_dir");
static_assert(PP_NSLOT + 1 <= 64, "pair_compact_slots moves at most two slots per lane");
static_assert(PP_NSLOT + 1 <= 255 && PP_MVCAP >= 66, "slot_of sentinel / moving-list bound");

// What a lane knows about the half it serves.
struct PW {
    int h, hl;           // half (0 / 1), lane within the half
    unsigned lb;         // LDS byte offset of the half's image
    unsigned eo, to;     // element offsets env * nbcap / trial * nbcap into the per-env / per-trial arrays
    int nb;              // bodies of the env
    int env;
};
// per-half sub-step state kept in registers (the cold part lives in the half's LDS scalars)
struct PState {
    unsigned stamp, stamp0;   // stamp0: the env's stamp when this task took it over ("moved in sub-step" marks in LDS are relative to it)
    int nmv, nslots, nlevels;
    unsigned prev_amask;
    int cc_ok, cc_kmax;
    int quiescent, err;
    unsigned costp;
    double total_ke, total_imp;
    unsigned n_post, n_contact, n_first, ship_post, ship_contacts;
    int yaw_violated, boundary_violated;
    int nkeys, nact, nwarm;   // arbiter lanes in use / active / warm in the last sub-step (pairing thresholds)
};

extern __shared__ double2 bp_smem[];
// Diagnostic build (-DBP_PAIR_PROF, tools/prof_pair.py): cycle stamps of the paired sub-step's phases, taken by lane 0 for the wave, kept in the spare
// bytes of half 0's LDS image and added to D.prof[env of half 0] at the end of the task.
#ifdef BP_PAIR_PROF
#define PPROF_SLOT(k) (((unsigned long long *)((char *)bp_smem + PL_END))[k])
#define PPROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime();
#define PPROF(k) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) PPROF_SLOT(k) += _n - _pt; _pt = _n; }
#define PCNT(k, v) { if ((threadIdx.x & 63) == 0) PPROF_SLOT(k) += (unsigned long long)(v); }
static_assert(PL_END + 8u * 19 <= PL_HALF, "phase counters fit the spare bytes of the half-wave image");
#else
#define PPROF_DECL
#define PPROF(k)
#define PCNT(k, v)
#endif
#define PLDS(T, off) ((T *)((char *)bp_smem + W.lb + (off)))

// Element `idx` of a per-env / per-trial array: the byte offset is formed in 32 bits (bp_load_scenarios enables pairing only while every such array stays
// below 4 GB), so the access is "uniform base + 32-bit lane offset" -- one address instruction per access instead of a 64-bit multiply-add chain.
template <typename T> __device__ __forceinline__ T &gA(T *const base, const unsigned idx) { return *(T *)((char *)base + (size_t)(idx * (unsigned)sizeof(T))); }
__device__ __forceinline__ unsigned hballot(const bool p, const int h)
{
    const unsigned long long b = __ballot(p);
    return h ? (unsigned)(b >> 32) : (unsigned)b;
}
__device__ __forceinline__ int popc_below32(const unsigned m, const int hl) { return __popc(m & ((1u << hl) - 1u)); }
// hand-over through global memory between lanes of the wave (vertices, AABBs, poses written by one lane and read by another): drain the stores.  The
// workgroup is one wave, so no barrier is involved -- which also makes the point legal inside half-divergent control flow.
__device__ __forceinline__ void pair_gsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int pair_slot_get(const DevPtrs &D, const PW &W, PState &S, const int body)
{
    int s = PLDS(unsigned char, PL_SLOTOF)[body];
    if (BP_UNLIKELY(s == 255)) {
        s = S.nslots;
        if (s >= PP_NSLOT) { S.err |= BP_ERR_ARB_OVERFLOW; s = PP_NSLOT - 1; }
        else S.nslots = s + 1;
        if (W.hl == 0) {
            PLDS(unsigned char, PL_SLOTOF)[body] = (unsigned char)s;
            PLDS(d2, PL_SV)[s] = mk2(0.0, 0.0); PLDS(d2, PL_SW)[s] = mk2(0.0, 0.0); PLDS(d2, PL_SB)[s] = mk2(0.0, 0.0);
            PLDS(d2, PL_SP)[s] = gA(D.pxy, W.eo + body);
        }
        lds_sync();
    }
    return s;
}

// refresh_body() of bp_physics.hpp for one half (i is uniform within the half)
__device__ __forceinline__ void pair_refresh_body(const DevParams &P, const DevPtrs &D, const PW &W, const int i, int &err)
{
    const int hl = W.hl;
    const unsigned eo = W.eo;
    const double4 b = gA(D.bb, eo + i);
    double4 nf;
    nf.x = b.x - P.skin; nf.y = b.y - P.skin; nf.z = b.z + P.skin; nf.w = b.w + P.skin;
    if (hl == 0) gA(D.fat, eo + i) = nf;
    pair_gsync();
    int cnt = 0;
    for (int base = 0; base < W.nb; base += 32) {
        const int j = base + hl;
        const bool valid = (j < W.nb) && (j != i);
        const double4 fj = valid ? gA(D.fat, eo + j) : nf;
        const bool ov = valid && bb_overlap(nf, fj);
        const unsigned m = hballot(ov, W.h);
        const int pos = cnt + popc_below32(m, hl);
        if (ov && pos < BP_KADJ) { gA(D.adj, (eo + i) * BP_KADJ + pos) = (unsigned short)j; gA(D.hint, (eo + i) * BP_KADJ + pos) = 0; }
        cnt += __popc(m);
        if (ov && kind_btype(gA(D.sc_kind, W.to + j)) != BODY_STATIC) {
            int nj = gA(D.adjn, eo + j);
            bool found = false;
            for (int s2 = 0; s2 < nj; s2++) found = found || (gA(D.adj, (eo + j) * BP_KADJ + s2) == (unsigned short)i);
            if (!found) {
                if (nj >= BP_KADJ) {
                    int w = 0;
                    for (int s2 = 0; s2 < nj; s2++) {
                        const int k = gA(D.adj, (eo + j) * BP_KADJ + s2);
                        const double4 fk = (k == i) ? nf : gA(D.fat, eo + k);
                        if (bb_overlap(fj, fk)) {
                            gA(D.adj, (eo + j) * BP_KADJ + w) = (unsigned short)k;
                            gA(D.hint, (eo + j) * BP_KADJ + w) = gA(D.hint, (eo + j) * BP_KADJ + s2);
                            w++;
                        }
                    }
                    nj = w;
                }
                if (nj < BP_KADJ) {
                    gA(D.adj, (eo + j) * BP_KADJ + nj) = (unsigned short)i;
                    gA(D.hint, (eo + j) * BP_KADJ + nj) = 0;
                    gA(D.adjn, eo + j) = (unsigned char)(nj + 1);
                } else {
                    gA(D.adjn, eo + j) = (unsigned char)nj;
                    err |= BP_ERR_ADJ_OVERFLOW;
                }
            }
        }
    }
    if (cnt > BP_KADJ) { err |= BP_ERR_ADJ_OVERFLOW; cnt = BP_KADJ; }
    if (hl == 0) gA(D.adjn, eo + i) = (unsigned char)cnt;
    pair_gsync();
}

// support_queries() of bp_physics.hpp for one half: four queries per trip (eight lanes each)
template <int VL>
__device__ __forceinline__ void pair_support_queries(const DevPtrs &D, const PW &W, const int nq)
{
    const int l8 = W.hl & 7, g = W.hl >> 3;
    for (int base = 0; base < nq; base += 4) {
        const int k = base + g;
        const bool act = k < nq;
        const d2 dir = PLDS(d2, PL_QDIR)[act ? k : 0];
        const unsigned meta = PLDS(unsigned, PL_QMETA)[act ? k : 0];
        const int body = (int)(meta & 0xFFFFu), nv = (int)(meta >> 16);
        double best = BP_INF;
        int bi = 255;
#pragma unroll
        for (int t = 0; t < (VL + 7) / 8; t++) {
            const int q = l8 + 8 * t;
            const bool ok = act && (q < nv);
            const d2 v = gA(D.wv, (W.eo + body) * BP_MAXV + (ok ? q : 0));
            const double d = vdot(dir, v);
            if (ok && d < best) { best = d; bi = q; }
        }
        const double m = oct_min_f64(best);
        const int mi = oct_min_i32((best == m) ? bi : 255);
        if (act && best == m && bi == mi) { PLDS(double, PL_RVAL)[k] = best; PLDS(unsigned, PL_RIDX)[k] = (unsigned)bi; }
    }
}

// One sub-step of the two environments of the wave.  `run`: this lane's half takes part (the other half may have finished or been parked).
__device__ __forceinline__ void substep_pair(const DevParams &P, const DevPtrs &D, const PW &W, ArbReg &A, PState &S, const double dt)
{
    constexpr int VL = BP_MAXV;
    const int hl = W.hl, h = W.h;
    const unsigned eo = W.eo, to = W.to;
    d2 *const Lsv = PLDS(d2, PL_SV), *const Lsw = PLDS(d2, PL_SW), *const Lsb = PLDS(d2, PL_SB), *const Lsp = PLDS(d2, PL_SP), *const Lag = PLDS(d2, PL_AG);
    unsigned char *const Lslot_of = PLDS(unsigned char, PL_SLOTOF), *const Lrf = PLDS(unsigned char, PL_RF);
    unsigned *const Lmvo = PLDS(unsigned, PL_MVO);
    unsigned short *const Lmvs = PLDS(unsigned short, PL_MVS);
    unsigned short *const Lmv = PLDS(unsigned short, PL_MV), *const Lowner = PLDS(unsigned short, PL_OWNER), *const Lcolmask = PLDS(unsigned short, PL_COLMASK);
    d2 *const Ltf = PLDS(d2, PL_QDIR), *const Lq_dir = PLDS(d2, PL_QDIR), *const Lmbox = PLDS(d2, PL_QDIR), *const Lpt_thr = PLDS(d2, PL_PTTHR);
    double *const Lq_c = PLDS(double, PL_QC), *const Lr_val = PLDS(double, PL_RVAL), *const Lhs = PLDS(double, PL_HS);
    unsigned *const Lq_meta = PLDS(unsigned, PL_QMETA), *const Lq_aux = PLDS(unsigned, PL_QAUX), *const Lr_idx = PLDS(unsigned, PL_RIDX);
    uint4 *const Lpt_a = PLDS(uint4, PL_PTA);
    unsigned long long *const Lcc = PLDS(unsigned long long, PL_CC), *const Lcc_hw = PLDS(unsigned long long, PL_CCHW);
    unsigned long long *const Lres_smA = PLDS(unsigned long long, PL_RSMA), *const Lres_smB = PLDS(unsigned long long, PL_RSMB);
    unsigned *const Lres_iA = PLDS(unsigned, PL_RIA), *const Lres_iB = PLDS(unsigned, PL_RIB), *const Lres_jA = PLDS(unsigned, PL_RJA), *const Lres_jB = PLDS(unsigned, PL_RJB);
    unsigned *const Lkq = PLDS(unsigned, PL_KQ), *const Lcq = PLDS(unsigned, PL_CQ);

    S.stamp += 1u;
    const unsigned now = S.stamp;
    const unsigned short nowr = (unsigned short)(now - S.stamp0);   // 1 .. 400: what the LDS marks of "moved in this sub-step" hold
    const double prev_dt = Lhs[0];
    lds_sync();
    if (hl == 0) Lhs[0] = dt;
    if (A.key != ARB_FREE_KEY && A.stamp == now - 1u) A.state = ARB_NORMAL;
    PPROF_DECL
    PCNT(12, 1)

    // ---- 1. integrate positions of the moving bodies; world geometry; AABBs ----------------------------------
    for (int k0 = 0; k0 < S.nmv; k0 += 32) {
        const int k = k0 + hl;
        double rad = 0.0;
        double4 fatb; fatb.x = fatb.y = -BP_INF; fatb.z = fatb.w = BP_INF;
        if (k < S.nmv) {
            const int i = Lmv[k];
            const int sl = Lslot_of[i];
            d2 v = mk2(0.0, 0.0), w2 = mk2(0.0, 0.0), vb = mk2(0.0, 0.0);
            if (sl != 255) { v = Lsv[sl]; w2 = Lsw[sl]; vb = Lsb[sl]; }
            d2 p = gA(D.pxy, eo + i);
            if (sl != 255) p = Lsp[sl];
            const double a = gA(D.ang, eo + i);
            d2 r = gA(D.rot, eo + i);
            const double4 ms = gA(D.sc_mass, to + i);
            rad = gA(D.sc_prop, to + i).x;
            fatb = gA(D.fat, eo + i);
            Lrf[hl] = (unsigned char)gA(D.sc_nv, to + i);
            p.x = p.x + (v.x + vb.x) * dt;
            p.y = p.y + (v.y + vb.y) * dt;
            const double a2 = a + (w2.x + w2.y) * dt;
            if (a2 != a) { double sn, cs; bp_sincos(a2, sn, cs); r = mk2(cs, sn); }
            gA(D.pxy, eo + i) = p; gA(D.ang, eo + i) = a2; gA(D.rot, eo + i) = r;
            if (i == 0) { Lag[0] = mk2(a2, 0.0); Lag[1] = r; }
            if (sl != 255) { Lsb[sl] = mk2(0.0, 0.0); Lsw[sl].y = 0.0; Lsp[sl] = p; }
            double4 t;
            t.x = r.x; t.y = r.y;
            t.z = p.x - (ms.z * r.x - ms.w * r.y);
            t.w = p.y - (ms.z * r.y + ms.w * r.x);
            Ltf[2 * hl] = mk2(t.x, t.y);
            Ltf[2 * hl + 1] = mk2(t.z, t.w);
            Lmvs[i] = nowr;
        }
        const int cnt = min(32, S.nmv - k0);
        unsigned long long *bbk = Lres_smA; // [32][4] = min x, max x, min y, max y
        if (hl < cnt) { bbk[hl * 4 + 0] = ~0ull; bbk[hl * 4 + 1] = 0ull; bbk[hl * 4 + 2] = ~0ull; bbk[hl * 4 + 3] = 0ull; }
        lds_sync();
        for (int t0 = 0; t0 < cnt * VL; t0 += 32) {
            const int t = t0 + hl;
            const int kk = t / VL, q = t - kk * VL;
            if (kk < cnt) {
                const int i = Lmv[k0 + kk];
                if (q < (int)Lrf[kk]) {
                    const d2 t0_ = Ltf[2 * kk], t1_ = Ltf[2 * kk + 1];
                    const double c = t0_.x, s = t0_.y;
                    const d2 lv = gA(D.sc_lv, (to + i) * BP_MAXV + q), ln = gA(D.sc_ln, (to + i) * BP_MAXV + q);
                    const double vx = (c * lv.x + (-s) * lv.y) + t1_.x;
                    const double vy = (s * lv.x + c * lv.y) + t1_.y;
                    const double nx = c * ln.x + (-s) * ln.y;
                    const double ny = s * ln.x + c * ln.y;
                    gA(D.wv, (eo + i) * BP_MAXV + q) = mk2(vx, vy);
                    gA(D.wn, (eo + i) * BP_MAXV + q) = mk2(nx, ny);
                    const unsigned long long kx = f64_key(vx), ky = f64_key(vy);
                    atomicMin(&bbk[kk * 4 + 0], kx); atomicMax(&bbk[kk * 4 + 1], kx);
                    atomicMin(&bbk[kk * 4 + 2], ky); atomicMax(&bbk[kk * 4 + 3], ky);
                }
            }
        }
        lds_sync();
        bool leftfat = false;
        if (hl < cnt) {
            const int i = Lmv[k0 + hl];
            double4 nbb;
            nbb.x = key_f64(bbk[hl * 4 + 0]) - rad; nbb.y = key_f64(bbk[hl * 4 + 2]) - rad;
            nbb.z = key_f64(bbk[hl * 4 + 1]) + rad; nbb.w = key_f64(bbk[hl * 4 + 3]) + rad;
            gA(D.bb, eo + i) = nbb;
            leftfat = !(nbb.x >= fatb.x && nbb.y >= fatb.y && nbb.z <= fatb.z && nbb.w <= fatb.w);
        }
        // ---- 2. Verlet refresh --------------------------------------------------------------------------------
        unsigned rm = hballot(leftfat, h);
        if (BP_UNLIKELY(rm != 0)) { S.cc_ok = 0; pair_gsync(); }
        while (BP_UNLIKELY(rm != 0)) {
            const int kk = __ffs((int)rm) - 1;
            rm &= rm - 1u;
            pair_refresh_body(P, D, W, Lmv[k0 + kk], S.err);
        }
        lds_sync();
    }
    pair_gsync();
    PPROF(0)

    // ---- 3./4. candidate pairs of moving bodies ----------------------------------------------------------------
    int kmax = S.cc_kmax;
    if (!S.cc_ok) {
        kmax = 0;
        for (int k0 = 0; k0 < S.nmv; k0 += 32) {
            const int k = k0 + hl;
            const int cnt = (k < S.nmv) ? (int)gA(D.adjn, eo + Lmv[k]) : 0;
            int m = 0;
            for (int bit = 16; bit >= 1; bit >>= 1) { if (hballot(cnt >= (m | bit), h)) m |= bit; }
            kmax = max(kmax, m);
        }
        S.cc_kmax = kmax;
    }
    const int ncand_slots = S.nmv * kmax;
    for (int base = 0; base < ncand_slots; base += 32) {
        int i, j, s, nA_h, nB_h;
        bool valid;
        unsigned long long hw;
        const bool incache = base < PP_CC;
        const bool cached = S.cc_ok && incache;
        if (cached) {
            const unsigned long long c = Lcc[base + hl];
            hw = Lcc_hw[base + hl];
            i = (int)(c & ((1u << BP_CC_IDX_BITS) - 1u)); j = (int)((c >> BP_CC_IDX_BITS) & ((1u << BP_CC_IDX_BITS) - 1u)); s = (int)((c >> 28) & 31u);
            nA_h = (int)((c >> 33) & 31u); nB_h = (int)((c >> 38) & 31u);
            valid = ((c >> 43) & 1u) != 0;
        } else {
            const int idx = base + hl;
            const int k = idx / kmax;
            s = idx - k * kmax;
            const bool inlist = k < S.nmv;
            i = inlist ? (int)Lmv[k] : 0;
            const int sc = min(s, BP_KADJ - 1);
            const int adjn_i = gA(D.adjn, eo + i);
            const int jr = gA(D.adj, (eo + i) * BP_KADJ + sc);
            j = jr < W.nb ? jr : 0;
            hw = gA(D.hint, (eo + i) * BP_KADJ + sc);
            valid = inlist && (s < adjn_i);
            if (valid && Lmvs[j] == nowr && j < i) valid = false; // pair is evaluated from j's list
            const int ki = gA(D.sc_kind, to + i), kj = gA(D.sc_kind, to + j);
            const double mi = gA(D.sc_mass, to + i).x, mj = gA(D.sc_mass, to + j).x;
            nA_h = gA(D.sc_nv, to + min(i, j)); nB_h = gA(D.sc_nv, to + max(i, j));
            if (valid) {
                if (kind_group(ki) != 0 && kind_group(ki) == kind_group(kj)) valid = false; // shapes of one body
                else if (mi == 0.0 && mj == 0.0) valid = false;                            // two infinite masses are never solved (no wall handler in ship-ice)
            }
            s = min(s, 31);
            if (incache) {
                Lcc[base + hl] = (unsigned long long)(unsigned)i | ((unsigned long long)(unsigned)j << BP_CC_IDX_BITS) | ((unsigned long long)(unsigned)s << 28) |
                                 ((unsigned long long)(unsigned)nA_h << 33) | ((unsigned long long)(unsigned)nB_h << 38) |
                                 ((unsigned long long)(valid ? 1u : 0u) << 43);
                Lcc_hw[base + hl] = hw;
            }
        }
        const int sa = min(i, j), sb = max(i, j);
        const double4 bbi = gA(D.bb, eo + i);
        const double4 bbj = gA(D.bb, eo + j);
        const double radA = gA(D.sc_prop, to + sa).x, radB = gA(D.sc_prop, to + sb).x;
        const double rsum = radA + radB;
        const int hA = HW_PLANE_A(hw) < BP_MAXV ? HW_PLANE_A(hw) : 0, hB = HW_PLANE_B(hw) < BP_MAXV ? HW_PLANE_B(hw) : 0;
        const int cnA = HW_NV_A(hw), cnB = HW_NV_B(hw);
        const int jA0 = HW_VERT_A(hw) < BP_MAXV ? HW_VERT_A(hw) : 0, jB0 = HW_VERT_B(hw) < BP_MAXV ? HW_VERT_B(hw) : 0;
        const int jAm = (jA0 == 0) ? max(min(cnB, BP_MAXV) - 1, 0) : jA0 - 1, jAp = (jA0 + 1 >= cnB) ? 0 : jA0 + 1;   // on B
        const int jBm = (jB0 == 0) ? max(min(cnA, BP_MAXV) - 1, 0) : jB0 - 1, jBp = (jB0 + 1 >= cnA) ? 0 : jB0 + 1;   // on A
        const unsigned oA = (eo + (unsigned)sa) * BP_MAXV, oB = (eo + (unsigned)sb) * BP_MAXV;   // world vertices / normals of the pair's two shapes
#define Av(k) gA(D.wv, oA + (unsigned)(k))
#define An(k) gA(D.wn, oA + (unsigned)(k))
#define Bv(k) gA(D.wv, oB + (unsigned)(k))
#define Bn(k) gA(D.wn, oB + (unsigned)(k))
        const d2 fnA = An(hA), fpA = Av(hA);
        const d2 fnB = Bn(hB), fpB = Bv(hB);
        const d2 vAm = Bv(jAm), vA0 = Bv(jA0), vAp = Bv(jAp);
        const d2 vBm = Av(jBm), vB0 = Av(jB0), vBp = Av(jBp);
        if (valid) valid = bb_overlap(bbi, bbj);
        if (hballot(valid, h) == 0) continue;
        // ---- 4a. cached planes -----------------------------------------------------------------------------------------
        const bool evA = valid && (hw & HW_HAS_A) && ((hw & HW_BOTH) || !(hw & HW_PRIM_B));
        const bool evB = valid && (hw & HW_HAS_B) && ((hw & HW_BOTH) || (hw & HW_PRIM_B));
        double sepAc = -BP_INF, sepBc = -BP_INF;
        int jAc = jA0, jBc = jB0;
        const double cA = vdot(fnA, fpA), cB = vdot(fnB, fpB);
        bool qryA = evA, qryB = evB;
        {
            const double dm = vdot(fnA, vAm), d0 = vdot(fnA, vA0), dp = vdot(fnA, vAp);
            if (evA && cnB == nB_h && cnB >= 2 && d0 + BP_SUPPORT_MARGIN <= dm && d0 + BP_SUPPORT_MARGIN <= dp) { sepAc = (d0 - cA) + 0.0; qryA = false; }
        }
        {
            const double dm = vdot(fnB, vBm), d0 = vdot(fnB, vB0), dp = vdot(fnB, vBp);
            if (evB && cnA == nA_h && cnA >= 2 && d0 + BP_SUPPORT_MARGIN <= dm && d0 + BP_SUPPORT_MARGIN <= dp) { sepBc = (d0 - cB) + 0.0; qryB = false; }
        }
        {
            const unsigned mqA = hballot(qryA, h), mqB = hballot(qryB, h);
            const int nqA = __popc(mqA), nq1 = nqA + __popc(mqB);
            if (BP_UNLIKELY(nq1)) {
                const int slA = popc_below32(mqA, hl), slB = nqA + popc_below32(mqB, hl);
                for (int q0 = 0; q0 < nq1; q0 += PP_QCAP) {
                    const bool inA = qryA && slA >= q0 && slA < q0 + PP_QCAP, inB = qryB && slB >= q0 && slB < q0 + PP_QCAP;
                    if (inA) { Lq_dir[slA - q0] = fnA; Lq_meta[slA - q0] = (unsigned)sb | ((unsigned)nB_h << 16); }
                    if (inB) { Lq_dir[slB - q0] = fnB; Lq_meta[slB - q0] = (unsigned)sa | ((unsigned)nA_h << 16); }
                    lds_sync();
                    pair_support_queries<VL>(D, W, min(nq1 - q0, PP_QCAP));
                    lds_sync();
                    if (inA) { sepAc = (Lr_val[slA - q0] - cA) + 0.0; jAc = (int)Lr_idx[slA - q0]; }
                    if (inB) { sepBc = (Lr_val[slB - q0] - cB) + 0.0; jBc = (int)Lr_idx[slB - q0]; }
                    lds_sync();
                }
            }
        }
        if (valid && (sepAc > rsum || sepBc > rsum)) {
            valid = false;
            if (hw & HW_BOTH) {
                const unsigned long long nh = (hw & ~(HW_BOTH | HW_PRIM_B)) | ((sepAc > rsum) ? 0ull : HW_PRIM_B);
                gA(D.hint, (eo + i) * BP_KADJ + s) = nh;
                if (incache) Lcc_hw[base + hl] = nh;
            }
        }
        const unsigned cm = hballot(valid, h);
        PPROF(1)
        PCNT(13, 1)
        if (cm == 0) continue;
        // ---- 4a'. every other plane of the surviving pairs: one (pair, side) per round -- the half's 32 lanes cover the up to BP_MAXV planes of a side;
        //      the two sides of a pair are taken together so that their loads travel together ------------------------------------------------------
        const int nc = __popc(cm);
        const int myr = popc_below32(cm, hl);
        const int nA_l = valid ? nA_h : 0, nB_l = valid ? nB_h : 0;
        if (valid) {
            uint4 pa;
            pa.x = (unsigned)sa | ((unsigned)sb << 16);
            pa.y = (unsigned)nA_l | ((unsigned)nB_l << 8) | ((unsigned)hA << 16) | ((unsigned)hB << 24);
            pa.z = (evA ? 1u : 0u) | (evB ? 2u : 0u) | ((unsigned)jAc << 8) | ((unsigned)jBc << 16);
            pa.w = 0u;
            Lpt_a[myr] = pa;
            Lpt_thr[myr] = mk2(sepAc, sepBc);
            Lres_smA[myr] = evA ? f64_key(sepAc) : 0ull; Lres_iA[myr] = evA ? (unsigned)hA : 0xFFFFFFFFu; Lres_jA[myr] = (unsigned)jAc;
            Lres_smB[myr] = evB ? f64_key(sepBc) : 0ull; Lres_iB[myr] = evB ? (unsigned)hB : 0xFFFFFFFFu; Lres_jB[myr] = (unsigned)jBc;
        }
        lds_sync();
        {
            int nq = 0, g0 = 0;
            const int f = hl;
            auto round_addr = [&](const uint4 pa, const int side, int &pbody, int &qbody, int &np, int &nqv, int &hX, bool &evX, int &jc, int &jm, int &jp) {
                const int psa = (int)(pa.x & 0xFFFFu), psb = (int)(pa.x >> 16);
                const int pna = (int)(pa.y & 0xFFu), pnb = (int)((pa.y >> 8) & 0xFFu);
                pbody = side ? psb : psa; qbody = side ? psa : psb;
                np = side ? pnb : pna; nqv = side ? pna : pnb;
                hX = (int)((pa.y >> (side ? 24 : 16)) & 0xFFu);
                evX = ((pa.z >> side) & 1u) != 0;
                jc = (int)((pa.z >> (side ? 16 : 8)) & 0xFFu);
                jm = (jc == 0) ? max(nqv, 1) - 1 : jc - 1; jp = (jc + 1 >= nqv) ? 0 : jc + 1;
            };
            // the search of the collected queries and the resolution of their maxima (lowest plane index on ties, its support vertex)
            auto flush = [&](const int rr_end) {
                pair_support_queries<VL>(D, W, nq);
                lds_sync();
                for (int s0 = 0; s0 < nq; s0 += 32) {
                    const int sl = s0 + hl;
                    if (sl < nq) {
                        const unsigned aux = Lq_aux[sl];
                        const int r = (int)(aux & 0xFFu);
                        const double sp = (Lr_val[sl] - Lq_c[sl]) + 0.0;
                        const unsigned long long key = f64_key(sp);
                        Lq_c[sl] = __builtin_bit_cast(double, key);
                        atomicMax((aux & 0x100u) ? &Lres_smB[r] : &Lres_smA[r], key);
                    }
                }
                lds_sync();
                if (valid && myr >= g0 && myr < rr_end) {
                    if (evA && f64_key(sepAc) != Lres_smA[myr]) Lres_iA[myr] = 0xFFFFFFFFu;
                    if (evB && f64_key(sepBc) != Lres_smB[myr]) Lres_iB[myr] = 0xFFFFFFFFu;
                }
                lds_sync();
                for (int s0 = 0; s0 < nq; s0 += 32) {
                    const int sl = s0 + hl;
                    if (sl < nq) {
                        const unsigned aux = Lq_aux[sl];
                        const int r = (int)(aux & 0xFFu);
                        const unsigned long long key = __builtin_bit_cast(unsigned long long, Lq_c[sl]);
                        if (key == ((aux & 0x100u) ? Lres_smB[r] : Lres_smA[r])) atomicMin((aux & 0x100u) ? &Lres_iB[r] : &Lres_iA[r], aux >> 16);
                    }
                }
                lds_sync();
                for (int s0 = 0; s0 < nq; s0 += 32) {
                    const int sl = s0 + hl;
                    if (sl < nq) {
                        const unsigned aux = Lq_aux[sl];
                        const int r = (int)(aux & 0xFFu);
                        const unsigned long long key = __builtin_bit_cast(unsigned long long, Lq_c[sl]);
                        const bool onB = (aux & 0x100u) != 0;
                        if (key == (onB ? Lres_smB[r] : Lres_smA[r]) && (aux >> 16) == (onB ? Lres_iB[r] : Lres_iA[r])) {
                            if (onB) Lres_jB[r] = Lr_idx[sl]; else Lres_jA[r] = Lr_idx[sl];
                        }
                    }
                }
                lds_sync();
                nq = 0; g0 = rr_end;
            };
            for (int r0 = 0; r0 < nc; r0++) {
                // a pair's two sides hold at most 2 * BP_MAXV survivors: search what has been collected when they might not fit
                if (nq + 2 * BP_MAXV > PP_QCAP && nq > 0) flush(r0);
                const uint4 pa = Lpt_a[r0];
                const d2 thr = Lpt_thr[r0];
                int pb0, qb0, np0, nqv0, hX0, jc0, jm0, jp0, pb1, qb1, np1, nqv1, hX1, jc1, jm1, jp1; bool ev0, ev1;
                round_addr(pa, 0, pb0, qb0, np0, nqv0, hX0, ev0, jc0, jm0, jp0);
                round_addr(pa, 1, pb1, qb1, np1, nqv1, hX1, ev1, jc1, jm1, jp1);
                const int fc0 = (f < np0) ? f : 0, fc1 = (f < np1) ? f : 0;
                const unsigned o0 = (eo + (unsigned)pb0) * BP_MAXV, o1 = (eo + (unsigned)pb1) * BP_MAXV;
                const d2 fn0 = gA(D.wn, o0 + (unsigned)(fc0)), fp0 = gA(D.wv, o0 + (unsigned)(fc0)), vb0 = gA(D.wv, o1 + (unsigned)(jc0)), vm0 = gA(D.wv, o1 + (unsigned)(jm0)), vp0 = gA(D.wv, o1 + (unsigned)(jp0));   // side 0: planes of A (= pb0) against vertices of B (= qb0 = pb1)
                const d2 fn1 = gA(D.wn, o1 + (unsigned)(fc1)), fp1 = gA(D.wv, o1 + (unsigned)(fc1)), vb1 = gA(D.wv, o0 + (unsigned)(jc1)), vm1 = gA(D.wv, o0 + (unsigned)(jm1)), vp1 = gA(D.wv, o0 + (unsigned)(jp1));   // side 1: planes of B against vertices of A
                {
                    const double th = thr.x;
                    const bool pv = (f < np0) && !(ev0 && f == hX0);
                    const double c = vdot(fn0, fp0);
                    const double bound = (fmin(vdot(fn0, vb0), fmin(vdot(fn0, vm0), vdot(fn0, vp0))) - c) + 0.0;
                    const bool surv = pv && (!ev0 || bound >= th);
                    const unsigned sm = hballot(surv, h);
                    if (surv) {
                        const int sl = nq + popc_below32(sm, hl);
                        Lq_dir[sl] = fn0; Lq_meta[sl] = (unsigned)qb0 | ((unsigned)nqv0 << 16);
                        Lq_aux[sl] = (unsigned)r0 | (0u << 8) | ((unsigned)f << 16);
                        Lq_c[sl] = c;
                    }
                    nq += __popc(sm);
                }
                {
                    const double th = thr.y;
                    const bool pv = (f < np1) && !(ev1 && f == hX1);
                    const double c = vdot(fn1, fp1);
                    const double bound = (fmin(vdot(fn1, vb1), fmin(vdot(fn1, vm1), vdot(fn1, vp1))) - c) + 0.0;
                    const bool surv = pv && (!ev1 || bound >= th);
                    const unsigned sm = hballot(surv, h);
                    if (surv) {
                        const int sl = nq + popc_below32(sm, hl);
                        Lq_dir[sl] = fn1; Lq_meta[sl] = (unsigned)qb1 | ((unsigned)nqv1 << 16);
                        Lq_aux[sl] = (unsigned)r0 | (1u << 8) | ((unsigned)f << 16);
                        Lq_c[sl] = c;
                    }
                    nq += __popc(sm);
                }
                lds_sync();
            }
            if (nq > 0) flush(nc);
        }
        PPROF(2)
        // ---- 4b. closest features -> normal -> Chipmunk ContactPoints, one pair per lane ------------------------------
        Manifold M;
        M.count = 0; M.h0 = M.h1 = 0; M.n = mk2(0, 0);
        M.p1_0 = M.p2_0 = M.p1_1 = M.p2_1 = mk2(0, 0);
        bool touching = false;
        int src = 2;
        d2 n = mk2(0, 0);
        int iA = 0, iB = 0, jA = 0, jB = 0;
        int i1A = 0, i1B = 0;
        bool needA = false, needB = false;
        const int nA = nA_l, nB = nB_l;
        if (valid) {
            const double sA = key_f64(Lres_smA[myr]), sB = key_f64(Lres_smB[myr]);
            iA = (int)Lres_iA[myr]; iB = (int)Lres_iB[myr]; jA = (int)Lres_jA[myr]; jB = (int)Lres_jB[myr];
            const bool useA = (sA >= sB);
            const double smax = useA ? sA : sB;
            touching = true;
            const int iA0 = (iA == 0) ? nA - 1 : iA - 1, iB0 = (iB == 0) ? nB - 1 : iB - 1;
            const d2 nAi = An(iA), nBi = Bn(iB);
            const d2 aA = Av(iA0), bA = Av(iA), qA = Bv(jA);
            const d2 aB = Bv(iB0), bB = Bv(iB), qB = Av(jB);
            const int iA0m = (iA0 == 0) ? nA - 1 : iA0 - 1, iAp = (iA + 1 >= nA) ? 0 : iA + 1;
            const int iB0m = (iB0 == 0) ? nB - 1 : iB0 - 1, iBp = (iB + 1 >= nB) ? 0 : iB + 1;
            const d2 oA0 = Av(iA0m), oA1 = Av(iAp), oB0 = Bv(iB0m), oB1 = Bv(iBp);
            if (smax > rsum) touching = false;
            else if (smax <= 0.0) { n = useA ? nAi : vneg(nBi); src = useA ? 0 : 1; }
            else {
                const d2 eA = vsub(bA, aA);
                const double uA = vdot(vsub(qA, aA), eA), eeA = vdot(eA, eA);
                const bool spanA = !(uA < 0.0) && !(uA > eeA);
                const d2 eB = vsub(bB, aB);
                const double uB = vdot(vsub(qB, aB), eB), eeB = vdot(eB, eB);
                const bool spanB = !(uB < 0.0) && !(uB > eeB);
                auto tie_partner = [&](const d2 aP, const d2 eP, const double eeP, const d2 nP, const unsigned oQ, const int nQ, const int j_, const d2 q0, const double u) -> bool {
                    const int jn = (u < 0.0) ? ((j_ == 0) ? nQ - 1 : j_ - 1) : ((j_ + 1 >= nQ) ? 0 : j_ + 1);
                    const d2 q1 = gA(D.wv, oQ + (unsigned)jn);
                    const double u1 = vdot(vsub(q1, aP), eP);
                    return !(u1 < 0.0) && !(u1 > eeP) && (vdot(nP, q1) - vdot(nP, q0) <= BP_TIE_TOL);
                };
                if (useA) {
                    if (spanA) { n = nAi; src = 0; }
                    else if (sB > 0.0 && spanB) { n = vneg(nBi); src = 1; }
                    else if (tie_partner(aA, eA, eeA, nAi, oB, nB, jA, qA, uA)) { n = nAi; src = 0; }
                    else if (sB > 0.0 && tie_partner(aB, eB, eeB, nBi, oA, nA, jB, qB, uB)) { n = vneg(nBi); src = 1; }
                    else {
                        const d2 pp = vsub(qA, (uA < 0.0) ? aA : bA);
                        const double dl = vlen(pp);
                        if (dl > rsum) touching = false;
                        n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                    }
                } else {
                    if (spanB) { n = vneg(nBi); src = 1; }
                    else if (sA > 0.0 && spanA) { n = nAi; src = 0; }
                    else if (tie_partner(aB, eB, eeB, nBi, oA, nA, jB, qB, uB)) { n = vneg(nBi); src = 1; }
                    else if (sA > 0.0 && tie_partner(aA, eA, eeA, nAi, oB, nB, jA, qA, uA)) { n = nAi; src = 0; }
                    else {
                        const d2 pp = vsub((uB < 0.0) ? aB : bB, qB);
                        const double dl = vlen(pp);
                        if (dl > rsum) touching = false;
                        n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                    }
                }
            }
            i1A = jB; i1B = jA;
            needA = touching && src != 1; needB = touching && src != 0;
            if (touching && src == 0) {
                const double c0 = vdot(aA, n), c1 = vdot(bA, n), o0 = vdot(oA0, n), o1 = vdot(oA1, n);
                const double cmx = (c0 > c1) ? c0 : c1, om = (o0 > o1) ? o0 : o1;
                if (nA == 2 || cmx >= om + BP_SUPPORT_MARGIN) { i1A = (c0 > c1) ? iA0 : (c1 > c0) ? iA : min(iA0, iA); needA = false; }
            }
            if (touching && src == 1) {
                const d2 nn = vneg(n);
                const double c0 = vdot(aB, nn), c1 = vdot(bB, nn), o0 = vdot(oB0, nn), o1 = vdot(oB1, nn);
                const double cmx = (c0 > c1) ? c0 : c1, om = (o0 > o1) ? o0 : o1;
                if (nB == 2 || cmx >= om + BP_SUPPORT_MARGIN) { i1B = (c0 > c1) ? iB0 : (c1 > c0) ? iB : min(iB0, iB); needB = false; }
            }
            const unsigned long long nh = (unsigned long long)((unsigned)iA | ((unsigned)iB << 5) | ((unsigned)jA << 10) | ((unsigned)jB << 15) |
                                                               ((unsigned)nA << 20) | ((unsigned)nB << 25)) |
                                          HW_HAS_A | HW_HAS_B | (useA ? 0ull : HW_PRIM_B) | ((smax > rsum) ? 0ull : HW_BOTH);
            gA(D.hint, (eo + i) * BP_KADJ + s) = nh;
            if (incache) Lcc_hw[base + hl] = nh;
        }
        {
            const unsigned mA = hballot(needA, h), mB = hballot(needB, h);
            const int nqa = __popc(mA), nq2 = nqa + __popc(mB);
            const int slA = popc_below32(mA, hl), slB = nqa + popc_below32(mB, hl);
            for (int q0 = 0; q0 < nq2; q0 += PP_QCAP) {
                const bool inA = needA && slA >= q0 && slA < q0 + PP_QCAP, inB = needB && slB >= q0 && slB < q0 + PP_QCAP;
                if (inA) { Lq_dir[slA - q0] = vneg(n); Lq_meta[slA - q0] = (unsigned)sa | ((unsigned)nA << 16); }
                if (inB) { Lq_dir[slB - q0] = n; Lq_meta[slB - q0] = (unsigned)sb | ((unsigned)nB << 16); }
                lds_sync();
                pair_support_queries<VL>(D, W, min(nq2 - q0, PP_QCAP));
                lds_sync();
                if (inA) i1A = (int)Lr_idx[slA - q0];
                if (inB) i1B = (int)Lr_idx[slB - q0];
                lds_sync();
            }
        }
        if (touching) {
            const d2 nn = vneg(n);
            d2 e1a, e1b, e2a, e2b;
            int e1ia, e1ib, e2ia, e2ib;
            {
                const int a0 = (i1A == 0) ? nA - 1 : i1A - 1, a2 = (i1A + 1 == nA) ? 0 : i1A + 1;
                const int b0 = (i1B == 0) ? nB - 1 : i1B - 1, b2 = (i1B + 1 == nB) ? 0 : i1B + 1;
                const d2 nA1 = An(i1A), nA2 = An(a2), vA0_ = Av(a0), vA1 = Av(i1A), vA2 = Av(a2);
                const d2 nB1 = Bn(i1B), nB2 = Bn(b2), vB0_ = Bv(b0), vB1 = Bv(i1B), vB2 = Bv(b2);
                const bool fa = vdot(n, nA1) > vdot(n, nA2);
                e1a = fa ? vA0_ : vA1; e1ia = fa ? a0 : i1A; e1b = fa ? vA1 : vA2; e1ib = fa ? i1A : a2;
                const bool fb = vdot(nn, nB1) > vdot(nn, nB2);
                e2a = fb ? vB0_ : vB1; e2ia = fb ? b0 : i1B; e2b = fb ? vB1 : vB2; e2ib = fb ? i1B : b2;
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
                    const unsigned hh = ((unsigned)e1ib << 8) | (unsigned)e2ia;
                    if (M.count == 0) { M.p1_0 = p1; M.p2_0 = p2; M.h0 = hh; M.count = 1; }
                    else { M.p1_1 = p1; M.p2_1 = p2; M.h1 = hh; M.count = 2; }
                }
            }
        }
        PPROF(3)
        // ---- 4c. cpArbiterUpdate: hand each manifold to the lane of the half that owns the pair's arbiter ---------------------
        const unsigned dmk = hballot(valid && M.count > 0, h);
        const int drank = popc_below32(dmk, hl);
        const int ndel = __popc(dmk);
        lds_sync(); // the mailbox aliases the plane-search scratch: all reads of it are done
        const unsigned keyv = ((unsigned)sa << 16) | (unsigned)sb;
        for (int dbase = 0; dbase < ndel; dbase += PP_MBOX) {
            const bool mine = valid && M.count > 0 && drank >= dbase && drank < dbase + PP_MBOX;
            if (mine) {
                d2 *mb = Lmbox + (drank - dbase) * 6;
                mb[0] = M.n; mb[1] = M.p1_0; mb[2] = M.p2_0; mb[3] = M.p1_1; mb[4] = M.p2_1;
                mb[5] = mk2(__hiloint2double((int)M.h0, M.count), __hiloint2double((int)M.h1, (int)keyv));
            }
            lds_sync();
            int my_mb = -1;
            bool fresh = false;
            const int nbat = min(PP_MBOX, ndel - dbase);
            for (int e = 0; e < nbat; e++) { // every arbiter lane looks for its pair among the delivered ones; a pair nobody owns gets a free lane (rare)
                const unsigned key = (unsigned)__double2loint(Lmbox[e * 6 + 5].y);
                const bool own = (A.key == key);
                if (BP_UNLIKELY2(!hballot(own, h))) {
                    const unsigned om = hballot(A.key == ARB_FREE_KEY, h);
                    if (!om) S.err |= BP_ERR_ARB_OVERFLOW;
                    else {
                        const int owner = __ffs((int)om) - 1;
                        const int s1 = pair_slot_get(D, W, S, (int)(key >> 16)), s2 = pair_slot_get(D, W, S, (int)(key & 0xFFFFu));
                        if (hl == owner) { my_mb = e; fresh = true; A.key = key; A.slotA = s1; A.slotB = s2; }
                    }
                } else if (own) my_mb = e;
            }
            if (my_mb >= 0) {
                const d2 *mb = Lmbox + my_mb * 6;
                const d2 mn_ = mb[0], mp10 = mb[1], mp20 = mb[2], mp11 = mb[3], mp21 = mb[4], mh = mb[5];
                const d2 pa = Lsp[A.slotA], pbp = Lsp[A.slotB];
                const unsigned mh0 = (unsigned)__double2hiint(mh.x), mh1 = (unsigned)__double2hiint(mh.y);
                const int mcount = __double2loint(mh.x);
                if (BP_UNLIKELY2(fresh)) {
                    A.state = ARB_FIRST; A.count = 0; A.h0 = A.h1 = 0; A.jn0 = A.jt0 = A.jn1 = A.jt1 = 0.0;
                    const int usa = (int)(A.key >> 16), usb = (int)(A.key & 0xFFFFu);
                    const double4 m1 = gA(D.sc_mass, to + usa), m2 = gA(D.sc_mass, to + usb);
                    A.ma = m1.x; A.ia = m1.y; A.mb = m2.x; A.ib = m2.y;
                    const double4 q1 = gA(D.sc_prop, to + usa), q2 = gA(D.sc_prop, to + usb);
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
    S.cc_ok = 1;
    PPROF(4)
    // arbiters whose bodies did not move keep last sub-step's contacts
    if (A.key != ARB_FREE_KEY && A.stamp == now - 1u) {
        const int a = (int)(A.key >> 16), b = (int)(A.key & 0xFFFFu);
        if (Lmvs[a] != nowr && Lmvs[b] != nowr) A.stamp = now;
    }
    // ---- 5. cpSpaceArbiterSetFilter ---------------------------------------------------------------------------
    if (A.key != ARB_FREE_KEY) {
        const unsigned ticks = now - A.stamp;
        if (ticks >= 1u && A.state != ARB_CACHED) A.state = ARB_CACHED;
        if (ticks >= (unsigned)P.persistence) A.key = ARB_FREE_KEY;
    }
    const bool active = (A.key != ARB_FREE_KEY) && (A.stamp == now);
    const unsigned amask = hballot(active, h);
    S.nkeys = __popc(hballot(A.key != ARB_FREE_KEY, h));
    S.nact = __popc(amask);
    const int ba = (int)(A.key >> 16), bbi = (int)(A.key & 0xFFFFu);

    // ---- 6a. prestep (cpArbiterPreStep) -----------------------------------------------------------------------
    double nMass0 = 0.0, tMass0 = 0.0, bias0 = 0.0, bounce0 = 0.0, jBias0 = 0.0;
    double nMass1 = 0.0, tMass1 = 0.0, bias1 = 0.0, bounce1 = 0.0, jBias1 = 0.0;
    if (active) {
        const d2 pa = Lsp[A.slotA], pb = Lsp[A.slotB];
        const d2 va = Lsv[A.slotA], vb = Lsv[A.slotB];
        const double wa = Lsw[A.slotA].x, wb = Lsw[A.slotB].x;
        const d2 n = A.n;
        const d2 body_delta = vsub(pb, pa);
        const d2 t = vperp(n);
        {
            const double rcn1 = vcross(A.r1_0, n), rcn2 = vcross(A.r2_0, n);
            nMass0 = 1.0 / ((A.ma + A.ia * rcn1 * rcn1) + (A.mb + A.ib * rcn2 * rcn2));
            const double rct1 = vcross(A.r1_0, t), rct2 = vcross(A.r2_0, t);
            tMass0 = 1.0 / ((A.ma + A.ia * rct1 * rct1) + (A.mb + A.ib * rct2 * rct2));
            const double dist = vdot(vadd(vsub(A.r2_0, A.r1_0), body_delta), n);
            bias0 = -P.bias_coef * fmin(0.0, dist + P.slop);
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
    // the division by dt changes nothing for a signed zero, so it runs for the whole wave as soon as one lane of either half has a bias term
    if (__ballot(active && (bias0 != 0.0 || bias1 != 0.0))) { bias0 = bias0 / dt; bias1 = bias1 / dt; }
    // ---- warm set ------------------------------------------------------------------------------------------------------
    bool warm = false;
    if (active) {
        warm = (A.jn0 != 0.0) || (A.jt0 != 0.0) || (bias0 != 0.0) || (bounce0 != 0.0);
        if (A.count > 1) warm = warm || (A.jn1 != 0.0) || (A.jt1 != 0.0) || (bias1 != 0.0) || (bounce1 != 0.0);
        if (A.ma == 0.0) { const d2 v = Lsv[A.slotA]; warm = warm || (v.x != 0.0) || (v.y != 0.0) || (Lsw[A.slotA].x != 0.0); }
        if (A.mb == 0.0) { const d2 v = Lsv[A.slotB]; warm = warm || (v.x != 0.0) || (v.y != 0.0) || (Lsw[A.slotB].x != 0.0); }
        if (A.ma != 0.0) Lowner[A.slotA] = 0;
        if (A.mb != 0.0) Lowner[A.slotB] = 0;
    }
    unsigned wmask = hballot(warm, h);
    if (wmask != 0 && wmask != amask) {
        lds_sync();
        for (;;) {
            if (warm) { if (A.ma != 0.0) Lowner[A.slotA] = 1; if (A.mb != 0.0) Lowner[A.slotB] = 1; }
            lds_sync();
            if (active && !warm) warm = (A.ma != 0.0 && Lowner[A.slotA] != 0) || (A.mb != 0.0 && Lowner[A.slotB] != 0);
            const unsigned nm = hballot(warm, h);
            if (nm == wmask) break;
            wmask = nm;
        }
    }
    S.nwarm = __popc(wmask);
    // bias terms: one copy of the loop for the wave -- the copy with bias arithmetic leaves an env without bias terms bit-identical (its bias velocities and
    // impulses are and stay +0), so it is taken as soon as either half needs it
    const bool any_bias = __ballot(warm && ((bias0 != 0.0) || (A.count > 1 && bias1 != 0.0))) != 0ull;
    PPROF(5)
    // ---- solve order: greedy colouring of the active set in ascending key order (cached while the set is unchanged) ----------
    if (BP_UNLIKELY(amask != S.prev_amask)) {
        lds_sync();
        Lkq[hl] = A.key;
        lds_sync();
        int rank = 0;
        for (unsigned m = amask; m; m &= m - 1u) {
            const int l = __ffs((int)m) - 1;
            rank += (Lkq[l] < A.key) ? 1 : 0;
        }
        A.rank = rank;
        if (active) {
            Lcolmask[A.slotA] = 0; Lcolmask[A.slotB] = 0;
            Lcq[rank] = (unsigned)A.slotA | ((unsigned)A.slotB << 8) | ((A.ma != 0.0) ? 0x10000u : 0u) | ((A.mb != 0.0) ? 0x20000u : 0u);
        }
        lds_sync();
        const int nact = __popc(amask);
        int nlev = 0;
        for (int r = 0; r < nact; r++) {
            const unsigned rec = Lcq[r];
            const int a = (int)(rec & 0xFFu), b = (int)((rec >> 8) & 0xFFu);
            const bool adyn = (rec & 0x10000u) != 0, bdyn = (rec & 0x20000u) != 0;
            const unsigned ua = adyn ? (unsigned)Lcolmask[a] : 0u, ub = bdyn ? (unsigned)Lcolmask[b] : 0u;
            const unsigned used = ua | ub;
            int c = __ffs((int)~used) - 1;
            if (c > 15) { c = 15; S.err |= BP_ERR_LEVEL_OVERFLOW; }
            lds_sync();
            if (hl == 0) {   // one writer per half (the two bodies of an arbiter are different bodies, hence different slots)
                if (adyn) Lcolmask[a] = (unsigned short)(ua | (1u << c));
                if (bdyn) Lcolmask[b] = (unsigned short)(ub | (1u << c));
            }
            lds_sync();
            if (active && A.rank == r) A.level = c + 1;
            nlev = max(nlev, c + 1);
        }
        S.nlevels = nlev;
        S.prev_amask = amask;
    }
    S.costp += 16u + 2u * (unsigned)__popc(amask) + 4u * (unsigned)(__popc(wmask) * S.nlevels);
    lds_sync();
    PPROF(6)
    // ---- 6b. velocity integrate: damping^dt == 0 -> dynamic bodies' v, w := +0 -----------------------------------------
    for (int k0 = 0; k0 < S.nmv; k0 += 32) {
        const int k = k0 + hl;
        if (k < S.nmv) {
            const int i = Lmv[k];
            const int sl = Lslot_of[i];
            if (sl != 255 && i >= P.nkin) { Lsv[sl] = mk2(0.0, 0.0); Lsw[sl].x = 0.0; }
        }
    }
    lds_sync();
    // ---- 6c. warm start (cpArbiterApplyCachedImpulse) -----------------------------------------------------------
    const double dt_coef = (prev_dt == 0.0) ? 0.0 : (prev_dt == dt) ? 1.0 : dt / prev_dt;
    unsigned lvlmask = 0; // colours that hold at least one warm arbiter
    if (wmask) {
        unsigned *const scr = (unsigned *)(Lhs + 4);
        if (hl == 0) scr[0] = 0u;
        lds_sync();
        if (warm) atomicOr(&scr[0], 1u << A.level);
        lds_sync();
        lvlmask = scr[0];
        lds_sync();
    }
    for (unsigned lm = lvlmask; lm; lm &= lm - 1u) {
        const int lvl = __ffs((int)lm) - 1;
        if (warm && A.level == lvl && A.state != ARB_FIRST) {
            d2 va = Lsv[A.slotA], vb = Lsv[A.slotB];
            double wa = Lsw[A.slotA].x, wb = Lsw[A.slotB].x;
            {
                const d2 j = vmul(vrotate(A.n, mk2(A.jn0, A.jt0)), dt_coef);
                apply_contact_impulses(A, 0, va, wa, vb, wb, j);
            }
            if (A.count > 1) {
                const d2 j = vmul(vrotate(A.n, mk2(A.jn1, A.jt1)), dt_coef);
                apply_contact_impulses(A, 1, va, wa, vb, wb, j);
            }
            if (A.ma != 0.0) { Lsv[A.slotA] = va; Lsw[A.slotA].x = wa; }
            if (A.mb != 0.0) { Lsv[A.slotB] = vb; Lsw[A.slotB].x = wb; }
        }
        lds_sync();
    }
    PPROF(7)
    // ---- 6d. sequential impulses (cpArbiterApplyImpulse) ------------------------------------------------------
    const int wA = (A.ma != 0.0) ? A.slotA : PP_NSLOT, wB = (A.mb != 0.0) ? A.slotB : PP_NSLOT;
    auto iterate = [&](auto bias_tag) {
        constexpr bool AB = decltype(bias_tag)::value;
        auto gather = [&](d2 &va, d2 &vb, d2 &wa2, d2 &wb2, d2 &vba, d2 &vbb) {
            va = Lsv[A.slotA]; vb = Lsv[A.slotB];
            wa2 = mk2(0.0, 0.0); wb2 = mk2(0.0, 0.0); vba = mk2(0.0, 0.0); vbb = mk2(0.0, 0.0);
            if (AB) { wa2 = Lsw[A.slotA]; wb2 = Lsw[A.slotB]; vba = Lsb[A.slotA]; vbb = Lsb[A.slotB]; }
            else { wa2.x = Lsw[A.slotA].x; wb2.x = Lsw[A.slotB].x; }
        };
        auto scatter = [&](const d2 va, const d2 vb, const d2 wa2, const d2 wb2, const d2 vba, const d2 vbb) {
            Lsv[wA] = va; if (AB) { Lsw[wA] = wa2; Lsb[wA] = vba; } else Lsw[wA].x = wa2.x;
            Lsv[wB] = vb; if (AB) { Lsw[wB] = wb2; Lsb[wB] = vbb; } else Lsw[wB].x = wb2.x;
        };
        auto contacts = [&](d2 &va, d2 &vb, d2 &wa2, d2 &wb2, d2 &vba, d2 &vbb, double &chg) {
            const d2 n = A.n;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (c == 0 || A.count > 1) {
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
                    if (AB) {
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
                    chg += __builtin_fabs(djn); chg += __builtin_fabs(djt);
                    const d2 j = vrotate(n, mk2(djn, djt));
                    apply_contact_impulses(A, c, va, wa2.x, vb, wb2.x, j);
                }
            }
        };
        // ---- at most one warm colour in EITHER half (what light envs, the ones that are paired, mostly are): within a half the warm arbiters share no
        // dynamic body, so nobody else reads or writes the velocities a lane works on between its passes -- they stay in registers for the ten iterations
        // (one gather, one scatter), exactly as in the solo kernel: an infinite-mass body's velocity is re-read unchanged by every pass of the slot loop,
        // while here the (zero) impulse is added to the lane's copy, and x + (+-0) == x bit for bit unless a component of x is a negative zero -- lanes check
        // their infinite-mass sides once and the wave takes the slot loop if any such component exists.
        if (!AB && __ballot((lvlmask & (lvlmask - 1u)) != 0u) == 0ull) {
            d2 va = mk2(0.0, 0.0), vb = va, wa2 = va, wb2 = va, vba = va, vbb = va;
            if (warm) gather(va, vb, wa2, wb2, vba, vbb);
            auto negzero = [](double x) { return (((unsigned)__double2hiint(x) ^ 0x80000000u) | (unsigned)__double2loint(x)) == 0u; };
            bool nz = false;
            if (warm && A.ma == 0.0) nz = negzero(va.x) || negzero(va.y) || negzero(wa2.x);
            if (warm && A.mb == 0.0) nz = nz || negzero(vb.x) || negzero(vb.y) || negzero(wb2.x);
            if (!__ballot(nz)) {
                bool going1 = lvlmask != 0u;
                for (int it = 0; it < P.iterations; it++) {
                    if (!__ballot(going1)) break;
                    double chg = 0.0;
                    PCNT(15, 1)
                    if (going1 && warm) contacts(va, vb, wa2, wb2, vba, vbb, chg);
                    if (going1 && !hballot(warm && chg != 0.0, h)) going1 = false;
                }
                if (warm) scatter(va, vb, wa2, wb2, vba, vbb);
                lds_sync();
                return;
            }
        }
        // each half walks its own colours: pass p of the loop is the half's p-th warm colour, and a half that has reached its fixed point drops out
        bool going = lvlmask != 0u;
        for (int it = 0; it < P.iterations; it++) {
            if (!__ballot(going)) break;
            double chg = 0.0;
            if (going) {
                for (unsigned lm = lvlmask; lm; lm &= lm - 1u) {
                    const int lvl = __ffs((int)lm) - 1;
                    PCNT(14, 1)
                    if (warm && A.level == lvl) {
                        d2 va, vb, wa2, wb2, vba, vbb;
                        gather(va, vb, wa2, wb2, vba, vbb);
                        contacts(va, vb, wa2, wb2, vba, vbb, chg);
                        scatter(va, vb, wa2, wb2, vba, vbb);
                    }
                    lds_sync();
                }
                // an iteration that changed no accumulated impulse applied only zero impulses: a fixed point, the remaining iterations would repeat it
                if (!hballot(warm && chg != 0.0, h)) going = false;
            }
        }
    };
    if (any_bias) iterate(std::true_type{}); else iterate(std::false_type{});
    PPROF(8)
    // ---- 7. post-solve bookkeeping for ship(0) x floe arbiters, ascending key order ------------------------------
    {
        const bool shiparb = active && ba == 0;
        const unsigned sm = hballot(shiparb, h);
        const unsigned sm2 = hballot(shiparb && A.count > 1, h);
        S.ship_post = (unsigned)__popc(sm);
        S.ship_contacts = (unsigned)__popc(sm) + (unsigned)__popc(sm2);
        if (sm) {
            S.n_post += (unsigned)__popc(sm);
            S.n_contact += (unsigned)__popc(sm) + (unsigned)__popc(sm2);
            S.n_first += (unsigned)__popc(hballot(shiparb && A.state == ARB_FIRST, h));
            const bool ws = shiparb && warm;
            const unsigned wsm = hballot(ws, h);
            if (wsm) {
                // (1 - e) / (1 + e): one division per distinct elasticity product of the half instead of one per sub-step
                lds_sync();
                if (ws && hl == __ffs((int)wsm) - 1) Lhs[3] = A.e;
                lds_sync();
                const double e_first = Lhs[3];
                double eCoef;
                if (!hballot(ws && A.e != e_first, h)) {
                    if (e_first != Lhs[1]) { const double ec = (1 - e_first) / (1 + e_first); lds_sync(); if (hl == 0) { Lhs[1] = e_first; Lhs[2] = ec; } lds_sync(); }
                    eCoef = Lhs[2];
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
                // ship arbiters have the smallest keys of the active set: ranks 0 .. ns-1; the sums run in that order.  A cold ship arbiter is skipped
                // by the solo kernel and adds exactly +0 here: both sums are sums of non-negative terms (never -0), so x + 0 == x bit for bit
                d2 *const tmp = Lq_dir;
                if (shiparb) tmp[A.rank] = ws ? mk2(ke, imp) : mk2(0.0, 0.0);
                lds_sync();
                const int ns = __popc(sm);
                for (int r = 0; r < ns; r++) { const d2 v = tmp[r]; S.total_ke += v.x; S.total_imp += v.y; }
                lds_sync();
            }
        }
    }
    // ---- agent rules applied after every sub-step: yaw limits + channel boundary (ship_ice_env.py:284-290) ------------------------
    {
        const double a0 = Lag[0].x;
        const double x0 = Lsp[0].x;
        if (a0 <= 0.0 || a0 >= BP_PI) {
            if (hl < P.nkin) Lsw[hl] = mk2(0.0, Lsw[hl].y);
            S.yaw_violated = 1;
        }
        if (x0 < 0.0 || x0 > P.map_w) S.boundary_violated = 1;
    }
    lds_sync();
    PPROF(9)
    // ---- next sub-step's moving list: bodies of active arbiters with a non-zero velocity, plus the ship ----------
    {
        bool wantA = false, wantB = false;
        if (active) {
            if (A.ma != 0.0) {
                const d2 v = Lsv[A.slotA], w2 = Lsw[A.slotA], vb = Lsb[A.slotA];
                wantA = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
            }
            if (A.mb != 0.0) {
                const d2 v = Lsv[A.slotB], w2 = Lsw[A.slotB], vb = Lsb[A.slotB];
                wantB = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
            }
        }
        const d2 v0 = Lsv[0], w0 = Lsw[0];
        const int shipmv = (v0.x != 0.0 || v0.y != 0.0 || w0.x != 0.0) ? P.nkin : 0;
        bool differs = false;
        if (hl < shipmv) { differs = Lmv[hl] != (unsigned short)hl; Lmv[hl] = (unsigned short)hl; }
        const bool gotA = wantA && atomicMax(&Lmvo[A.slotA], now) < now;
        const bool gotB = wantB && atomicMax(&Lmvo[A.slotB], now) < now;
        const unsigned mA = hballot(gotA, h), mB = hballot(gotB, h);
        const int nA_ = __popc(mA);
        if (gotA) { const int pos = shipmv + popc_below32(mA, hl); differs = differs || Lmv[pos] != (unsigned short)ba; Lmv[pos] = (unsigned short)ba; }
        if (gotB) { const int pos = shipmv + nA_ + popc_below32(mB, hl); differs = differs || Lmv[pos] != (unsigned short)bbi; Lmv[pos] = (unsigned short)bbi; }
        const int newn = shipmv + nA_ + __popc(mB);
        if (newn != S.nmv || hballot(differs, h)) S.cc_ok = 0;
        S.nmv = newn;
    }
    S.quiescent = (S.nmv == 0) && (wmask == 0);
    lds_sync();
    PPROF(10)
#undef Av
#undef An
#undef Bv
#undef Bn
}

// ---- state of one half: persistent env state <-> LDS image / registers (load_state_a / load_state_b / store_state of bp_kernels.hpp for 32 lanes) ----
// `resumed`: the env was parked inside this step (by the scheduler or by a paired wave): the agent's control and the step-local flags come from the parked state.
__device__ __forceinline__ void pair_load_state(const DevParams &P, const DevPtrs &D, const PW &W, ArbReg &A, PState &S, const double *__restrict__ actions,
                                                const bool resumed)
{
    const int hl = W.hl, h = W.h, env = W.env;
    const unsigned eo = W.eo, to = W.to;
    d2 *const Lsv = PLDS(d2, PL_SV), *const Lsw = PLDS(d2, PL_SW), *const Lsb = PLDS(d2, PL_SB), *const Lsp = PLDS(d2, PL_SP), *const Lag = PLDS(d2, PL_AG);
    unsigned char *const Lslot_of = PLDS(unsigned char, PL_SLOTOF);
    unsigned *const Lmvo = PLDS(unsigned, PL_MVO), *const Lkq = PLDS(unsigned, PL_KQ);
    unsigned short *const Lmv = PLDS(unsigned short, PL_MV), *const Lmvs = PLDS(unsigned short, PL_MVS);
    double *const Lhs = PLDS(double, PL_HS);
    for (int i = hl; i < PP_NBCAP; i += 32) { Lmvs[i] = 0; Lslot_of[i] = (i < P.nkin) ? (unsigned char)i : 255; }
    for (int i = hl; i < PP_NSLOT + 2; i += 32) Lmvo[i] = 0u;
    if (hl < P.nkin) { Lsv[hl] = gA(D.velv, eo + hl); Lsw[hl] = gA(D.velw, eo + hl); Lsb[hl] = gA(D.velb, eo + hl); Lsp[hl] = gA(D.pxy, eo + hl); }
    if (hl == 0) {
        Lag[0] = mk2(gA(D.ang, eo), 0.0); Lag[1] = gA(D.rot, eo);
        Lhs[0] = gA(D.e_currdt, env); Lhs[1] = -1.0; Lhs[2] = 0.0; Lhs[3] = 0.0;
    }
    S.nslots = P.nkin;
    S.costp = 0u; S.cc_ok = 0; S.cc_kmax = 0; S.quiescent = 0; S.ship_post = 0; S.ship_contacts = 0; S.err = 0;
    S.yaw_violated = 0; S.boundary_violated = 0; S.prev_amask = 0u; S.nlevels = 0; S.nact = 0; S.nwarm = 0;
    A.level = 0; A.rank = 0; A.ma = A.ia = A.mb = A.ib = 0.0; A.e = 0.0; A.u = 0.0;
    // persisted arbiters: BP_ACAP = 64 entries per env, of which a half-wave holds 32 -- the live ones are compacted onto the lanes (which lane holds an
    // arbiter is free: pairs are matched by key and solved in (colour, key) order)
    const size_t ab0 = (size_t)env * BP_ACAP;
    int nlive = 0;
    lds_sync();
    for (int part = 0; part < 2; part++) {
        const int idx = part * 32 + hl;
        const bool live = gA(D.a_key, ab0 + idx) != ARB_FREE_KEY;
        const unsigned m = hballot(live, h);
        const int pos = nlive + popc_below32(m, hl);
        if (live && pos < 32) Lkq[pos] = (unsigned)idx;
        nlive += __popc(m);
    }
    S.nkeys = nlive;
    if (nlive > 32) { S.err |= BP_ERR_ARB_OVERFLOW; nlive = 32; }
    lds_sync();
    A.key = ARB_FREE_KEY; A.stamp = 0; A.state = ARB_FIRST; A.count = 0; A.h0 = A.h1 = 0;
    A.jn0 = A.jt0 = A.jn1 = A.jt1 = 0.0;
    A.n = mk2(0, 0); A.r1_0 = A.r2_0 = A.r1_1 = A.r2_1 = mk2(0, 0);
    A.slotA = A.slotB = 0;
    if (hl < nlive) {
        const size_t ab = ab0 + Lkq[hl];
        A.key = gA(D.a_key, ab); A.stamp = gA(D.a_stamp, ab);
        { const unsigned sc = gA(D.a_sc, ab); A.state = (int)(sc & 0xFF); A.count = (int)(sc >> 8); }
        A.h0 = gA(D.a_h0, ab); A.h1 = gA(D.a_h1, ab);
        const double *ad = D.a_d + ab * 14;
        A.jn0 = ad[0]; A.jt0 = ad[1]; A.jn1 = ad[2]; A.jt1 = ad[3];
        A.n = mk2(ad[4], ad[5]);
        A.r1_0 = mk2(ad[6], ad[7]); A.r2_0 = mk2(ad[8], ad[9]); A.r1_1 = mk2(ad[10], ad[11]); A.r2_1 = mk2(ad[12], ad[13]);
        const double4 m1 = gA(D.sc_mass, to + (A.key >> 16)), m2 = gA(D.sc_mass, to + (A.key & 0xFFFFu));
        A.ma = m1.x; A.ia = m1.y; A.mb = m2.x; A.ib = m2.y;
        const double4 q1 = gA(D.sc_prop, to + (A.key >> 16)), q2 = gA(D.sc_prop, to + (A.key & 0xFFFFu));
        A.e = q1.y * q2.y; A.u = q1.z * q2.z;
    }
    S.stamp = gA(D.e_stamp, env); S.stamp0 = S.stamp;
    S.total_ke = gA(D.e_ke, env); S.total_imp = gA(D.e_imp, env);
    S.n_post = gA(D.e_cnt, env * 4 + 0); S.n_contact = gA(D.e_cnt, env * 4 + 1); S.n_first = gA(D.e_cnt, env * 4 + 2);
    if (resumed) {
        const unsigned *cy = D.sq_carry + (size_t)env * 4;
        S.yaw_violated = (int)(cy[0] & 1u); S.boundary_violated = (int)(cy[1] & 1u); S.costp = cy[2];
    }
    lds_sync();
    // ship control (ship_ice_env.py:265-274): set once per env step
    if (!resumed && hl < P.nkin) {
        const double act = actions[env] * P.max_yaw_rate;
        const d2 r = gA(D.rot, eo);
        Lsv[hl] = mk2(r.x * P.target_speed + -r.y * 0.0, r.y * P.target_speed + r.x * 0.0);
        Lsw[hl] = mk2(act, Lsw[hl].y);
    }
    lds_sync();
    // moving list: every body with a non-zero velocity gets a velocity slot (the ship owns slot 0)
    int n = 0;
    for (int base = 0; base < W.nb; base += 32) {
        const int i = base + hl;
        bool mvg = false;
        d2 v = mk2(0.0, 0.0), w2 = v, vb = v;
        if (i < W.nb) {
            if (i < P.nkin) { v = Lsv[i]; w2 = Lsw[i]; vb = Lsb[i]; }
            else { v = gA(D.velv, eo + i); w2 = gA(D.velw, eo + i); vb = gA(D.velb, eo + i); }
            mvg = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
        }
        const unsigned m = hballot(mvg, h);
        const unsigned ms = hballot(mvg && i >= P.nkin, h);
        if (mvg) { const int pos = n + popc_below32(m, hl); if (pos < PP_MVCAP) Lmv[pos] = (unsigned short)i; }
        if (mvg && i >= P.nkin) {
            int sl = S.nslots + popc_below32(ms, hl);
            if (sl >= PP_NSLOT) { S.err |= BP_ERR_ARB_OVERFLOW; sl = PP_NSLOT - 1; }
            Lslot_of[i] = (unsigned char)sl;
            Lsv[sl] = v; Lsw[sl] = w2; Lsb[sl] = vb; Lsp[sl] = gA(D.pxy, eo + i);
        }
        n += __popc(m);
        S.nslots = min(S.nslots + __popc(ms), PP_NSLOT);
    }
    if (n > PP_MVCAP) { S.err |= BP_ERR_ARB_OVERFLOW; n = PP_MVCAP; }
    S.nmv = n;
    lds_sync();
    // velocity slots for the bodies of the persisted arbiters, in lane order
    for (int l = 0; l < nlive; l++) {
        lds_sync();
        if (hl == l) Lkq[0] = A.key;
        lds_sync();
        const unsigned key = Lkq[0];
        const int s1 = pair_slot_get(D, W, S, (int)(key >> 16)), s2 = pair_slot_get(D, W, S, (int)(key & 0xFFFFu));
        if (hl == l) { A.slotA = s1; A.slotB = s2; }
    }
    lds_sync();
}

__device__ __forceinline__ void pair_store_state(const DevParams &P, const DevPtrs &D, const PW &W, const ArbReg &A, const PState &S)
{
    const int hl = W.hl, env = W.env;
    const unsigned eo = W.eo;
    const d2 *const Lsv = PLDS(d2, PL_SV), *const Lsw = PLDS(d2, PL_SW), *const Lsb = PLDS(d2, PL_SB);
    const unsigned char *const Lslot_of = PLDS(unsigned char, PL_SLOTOF);
    for (int i = hl; i < P.nbcap; i += 32) {
        const int sl = Lslot_of[i];
        const d2 z = mk2(0.0, 0.0);
        gA(D.velv, eo + i) = (sl != 255) ? Lsv[sl] : z; gA(D.velw, eo + i) = (sl != 255) ? Lsw[sl] : z; gA(D.velb, eo + i) = (sl != 255) ? Lsb[sl] : z;
    }
    {
        const size_t ab = (size_t)env * BP_ACAP + hl;
        gA(D.a_key, ab) = A.key; gA(D.a_stamp, ab) = A.stamp; gA(D.a_sc, ab) = (unsigned)A.state | ((unsigned)A.count << 8);
        gA(D.a_h0, ab) = A.h0; gA(D.a_h1, ab) = A.h1;
        double *ad = D.a_d + ab * 14;
        ad[0] = A.jn0; ad[1] = A.jt0; ad[2] = A.jn1; ad[3] = A.jt1; ad[4] = A.n.x; ad[5] = A.n.y;
        ad[6] = A.r1_0.x; ad[7] = A.r1_0.y; ad[8] = A.r2_0.x; ad[9] = A.r2_0.y;
        ad[10] = A.r1_1.x; ad[11] = A.r1_1.y; ad[12] = A.r2_1.x; ad[13] = A.r2_1.y;
        gA(D.a_key, ab + 32) = ARB_FREE_KEY;   // the upper half of the env's persisted arbiter entries is unused while it runs paired
    }
    if (hl == 0) {
        gA(D.e_stamp, env) = S.stamp; gA(D.e_currdt, env) = PLDS(double, PL_HS)[0];
        gA(D.e_ke, env) = S.total_ke; gA(D.e_imp, env) = S.total_imp;
        gA(D.e_cnt, env * 4 + 0) = S.n_post; gA(D.e_cnt, env * 4 + 1) = S.n_contact; gA(D.e_cnt, env * 4 + 2) = S.n_first;
        if (S.err) atomicOr(&gA(D.e_err, env), S.err & (BP_ERR_ADJ_OVERFLOW | BP_ERR_ARB_OVERFLOW | BP_ERR_LEVEL_OVERFLOW));
    }
}

// Velocity slots are handed out on first use and never returned within a task, and a half has 40 of them instead of 96: over the 400 sub-steps of a
// step the ship touches more bodies than that.  A slot whose body has no arbiter lane (its pair has aged out) and an all-zero velocity is exactly "no
// slot" -- that is how a parked env comes back (load_state_b hands slots to bodies with a velocity or an arbiter) -- so such slots are dropped and the
// rest moved down in order.  Called between sub-steps, by the halves that need it.
__device__ __forceinline__ void pair_compact_slots(const DevParams &P, const PW &W, ArbReg &A, PState &S)
{
    const int hl = W.hl, h = W.h;
    d2 *const Lsv = PLDS(d2, PL_SV), *const Lsw = PLDS(d2, PL_SW), *const Lsb = PLDS(d2, PL_SB), *const Lsp = PLDS(d2, PL_SP);
    unsigned char *const Lslot_of = PLDS(unsigned char, PL_SLOTOF);
    unsigned *const Lmvo = PLDS(unsigned, PL_MVO);
    unsigned short *const keep = PLDS(unsigned short, PL_OWNER), *const remap = PLDS(unsigned short, PL_COLMASK);   // both are scratch between sub-steps
    const int n = S.nslots;
    lds_sync();
    for (int s = hl; s < n; s += 32) {
        const d2 v = Lsv[s], w2 = Lsw[s], vb = Lsb[s];
        keep[s] = (s < P.nkin || v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0) ? 1 : 0;
    }
    lds_sync();
    if (A.key != ARB_FREE_KEY) { keep[A.slotA] = 1; keep[A.slotB] = 1; }
    lds_sync();
    int base = 0;
    for (int s0 = 0; s0 < n; s0 += 32) {
        const int s = s0 + hl;
        const bool k = s < n && keep[s] != 0;
        const unsigned m = hballot(k, h);
        if (s < n) remap[s] = k ? (unsigned short)(base + popc_below32(m, hl)) : (unsigned short)0xFFFF;
        base += __popc(m);
    }
    lds_sync();
    {   // every kept slot is read before any is written: a slot moves down onto an index that may still hold another kept slot's data
        const int s0 = hl, s1 = hl + 32;
        const bool k0 = s0 < n && remap[s0] != 0xFFFF, k1 = s1 < n && remap[s1] != 0xFFFF;
        d2 a0 = mk2(0, 0), b0 = a0, c0 = a0, d0 = a0, a1 = a0, b1 = a0, c1 = a0, d1 = a0;
        unsigned m0 = 0, m1 = 0;
        if (k0) { a0 = Lsv[s0]; b0 = Lsw[s0]; c0 = Lsb[s0]; d0 = Lsp[s0]; m0 = Lmvo[s0]; }
        if (k1) { a1 = Lsv[s1]; b1 = Lsw[s1]; c1 = Lsb[s1]; d1 = Lsp[s1]; m1 = Lmvo[s1]; }
        lds_sync();
        if (k0) { const int r = remap[s0]; Lsv[r] = a0; Lsw[r] = b0; Lsb[r] = c0; Lsp[r] = d0; Lmvo[r] = m0; }
        if (k1) { const int r = remap[s1]; Lsv[r] = a1; Lsw[r] = b1; Lsb[r] = c1; Lsp[r] = d1; Lmvo[r] = m1; }
    }
    lds_sync();
    for (int i = hl; i < W.nb; i += 32) {
        const int sl = Lslot_of[i];
        if (sl != 255) { const unsigned r = remap[sl]; Lslot_of[i] = (r == 0xFFFFu) ? (unsigned char)255 : (unsigned char)r; }
    }
    if (A.key != ARB_FREE_KEY) { A.slotA = remap[A.slotA]; A.slotB = remap[A.slotB]; }
    S.nslots = base;
    lds_sync();
}

// When does a half leave the pair?  Capacity first: the half-wave image holds 32 arbiter lanes, PP_NSLOT velocity slots and a moving list of PP_MVCAP; an env
// that approaches them is parked at the sub-step boundary (the margins are what one sub-step can add at most in practice, and a capacity that is hit
// anyway raises the same per-env error bits as in the solo kernel).  Heaviness second: an env with many active arbiters and colours sets the pace of the
// wave for its mate and would itself run faster alone.
struct PairLimits { int max_keys, max_slots, max_mv, max_act, max_work, gc_slots, max_rate; };   // gc_slots: pair_compact_slots runs above this many slots in use
__device__ __forceinline__ bool pair_should_leave(const PState &S, const PairLimits &Q)
{
    return S.nkeys > Q.max_keys || S.nslots > Q.max_slots || S.nmv > Q.max_mv || S.nact > Q.max_act || S.nwarm * S.nlevels > Q.max_work;
}

// One paired task: up to two envs (env1 may be -1), each from its own sub-step `it` on, to the end of the env step or until it leaves the pair.
// Returns per half (in the lane's registers): status 1 = step complete (outputs written), 2 = parked at sub-step *it_out, 0 = no env.
template <bool CAN_LEAVE, typename BehindFn>
__device__ BP_PAIR_TASK_INLINE int pair_task(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions, double *__restrict__ reward,
                                         unsigned char *__restrict__ terminated, unsigned char *__restrict__ truncated, double *__restrict__ info,
                                         const int env0, const int env1, const PairLimits Q, int &it_out, int &score_out, int &heavy_out,
                                         BehindFn behind)
{
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const int lane = (int)(threadIdx.x & 63);
#ifdef BP_PAIR_PROF
    if (lane < 19) PPROF_SLOT(lane) = 0ull;
    lds_sync();
#endif
    PW W;
    W.h = lane >> 5; W.hl = lane & 31;
    W.lb = (unsigned)W.h * PL_HALF;
    W.env = W.h ? env1 : env0;
    const bool have = W.env >= 0;
    const int env = have ? W.env : env0;          // a half without an env addresses its mate's arrays but never runs
    W.env = env;
    const int nbcap = P.nbcap;
    W.eo = (unsigned)env * (unsigned)nbcap;
    const int trial = gA(D.e_trial, env);
    W.to = (unsigned)trial * (unsigned)nbcap;
    W.nb = gA(D.e_nb, env);
    const int it_first = CAN_LEAVE ? gA(D.sq_sub, env) : 0;
    const bool resumed = it_first > 0;
    ArbReg A;
    PState S;
    int status = 0;
    int it = it_first;
    if (have) pair_load_state(P, D, W, A, S, actions, resumed);
    // An env that does not fit the half-wave as it stands (arbiter lanes, velocity slots or moving list above the leave limits: a heavy env that the
    // dispatch order took for a light one) is declined before anything has been written: its state is untouched, and so is its mate's -- the wave hands
    // both back (status 2, `it` where they were) and sched_body carries on with the heavier one alone.
    if (CAN_LEAVE) {
        const bool declined = have && (S.err != 0 || S.nkeys > Q.max_keys || S.nslots > Q.max_slots || S.nmv > Q.max_mv);
        if (__ballot(declined) != 0ull) {
            it_out = it;
            score_out = have ? S.nkeys * 64 + S.nmv : 0;
            heavy_out = declined ? 1 : 0;
            return have ? 2 : 0;
        }
        if (have && !resumed) { unsigned char *mvd_ = D.sq_moved + (size_t)env * nbcap; for (int i = W.hl; i < nbcap; i += 32) mvd_[i] = 0; }
    }
    const int nsub = P.steps;
    const unsigned costp0 = S.costp;
    bool running = have;
    bool heavy = false;
    int to_boundary = P.sq_chunk;     // sub-steps of this task until the next look at the queues
    while (__ballot(running)) {
        if (running) {
            substep_pair(P, D, W, A, S, P.dt_sub);
            it++;
            if (BP_UNLIKELY2(S.quiescent)) {
                // nothing moves and no arbiter is warm: the remaining sub-steps in closed form (physics_body of bp_kernels.hpp)
                const unsigned k = (unsigned)(nsub - it);
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
                it = nsub;
            }
            if (it >= nsub) { running = false; status = 1; }
            else if (S.nslots > Q.gc_slots) pair_compact_slots(P, W, A, S);   // slots leak over a step: drop the ones nobody needs any more
        }
        // a half that must leave takes its mate along: the wave carries on with one of them alone (sched_body), the other waits in a queue -- an env that is
        // heavy must not wait for its mate to finish, nor run on in half a wavefront
        if (CAN_LEAVE) {
            const bool lv = running && pair_should_leave(S, Q);
            if (__ballot(lv) != 0ull) { if (running) { running = false; status = 2; heavy = lv; } }
            else if (--to_boundary == 0) {
                // chunk boundary of the task: like a solo wave, a paired wave yields to envs that are further behind than its least advanced env
                to_boundary = P.sq_chunk;
                const int itv = running ? it : 0x7FFFFFFF;
                const int itmin = min(__builtin_amdgcn_readlane(itv, 0), __builtin_amdgcn_readlane(itv, 32));
                if (itmin != 0x7FFFFFFF && behind(itmin / P.sq_chunk) && running) { running = false; status = 2; }
            }
        }
    }
#ifdef BP_PAIR_PROF
    if (D.prof != nullptr) {
        lds_sync();
        if (lane == 0) PPROF_SLOT(11) = __builtin_amdgcn_s_memtime() - t_begin;
        lds_sync();
        if (lane < 19) D.prof[(size_t)env0 * BP_PROFN + lane] += PPROF_SLOT(lane);
    }
#endif
    it_out = it;
    {
        const unsigned rate = have ? (S.costp - costp0) / (unsigned)max(it - it_first, 1) : 0u;   // mean work proxy per sub-step of this task
        score_out = (int)min(rate, 0x7FFFFFFFu);
        // parked without having asked for it (its mate left, or the wave yielded): it still counts as heavy if its own work rate is above the pairing limit
        heavy_out = (heavy || (have && rate > (unsigned)Q.max_rate)) ? 1 : 0;
    }   // mean work proxy per sub-step of this task: which of two parked envs is the heavier
    if (!have) return 0;
    const unsigned short *const Lmvs = PLDS(unsigned short, PL_MVS);
    if (status == 2) {
        // ---- parked at a sub-step boundary: exactly the park of the preemptive scheduler (step-local flags and the shapes moved so far go along)
        unsigned char *mvd_ = D.sq_moved + (size_t)env * nbcap;
        for (int i = W.hl; i < W.nb; i += 32) if (Lmvs[i] != 0) mvd_[i] = 1;
        pair_gsync();
        pair_store_state(P, D, W, A, S);
        if (W.hl == 0) {
            unsigned *cy = D.sq_carry + (size_t)env * 4;
            cy[0] = (unsigned)S.yaw_violated; cy[1] = (unsigned)S.boundary_violated; cy[2] = S.costp;
            cy[3] = (resumed ? cy[3] : 0u) + (unsigned)((__builtin_amdgcn_s_memtime() - t_begin) >> 8);
            gA(D.sq_sub, env) = it;
        }
        return 2;
    }
    // ---- end of step: work / reward / termination (ship_ice_env.py:291-345), as the tail of physics_body ----
    double work = 0.0;
    {
        double *const scr = PLDS(double, PL_QDIR);   // [32] contributions of a chunk of bodies
        for (int base = 0; base < W.nb; base += 32) {
            const int i = base + W.hl;
            const bool mvd = (i < W.nb) && ((Lmvs[i] != 0) || (resumed && gA(D.sq_moved, (size_t)env * nbcap + i) != 0)) &&
                             (kind_ctype(gA(D.sc_kind, W.to + i)) == 2);
            double contrib = 0.0;
            if (mvd) {
                const int n = gA(D.sc_nv, W.to + i);
                d2 *prev = D.pv + (size_t)(W.eo + i) * BP_MAXV;
                const d2 *nowv = D.wv + (size_t)(W.eo + i) * BP_MAXV;
                const double area = poly_area_seq(prev, n);
                const d2 ca = poly_centroid_seq(prev, n);
                const d2 cb = poly_centroid_seq(nowv, n);
                const double d = __builtin_sqrt((ca.x - cb.x) * (ca.x - cb.x) + (ca.y - cb.y) * (ca.y - cb.y));
                contrib = d * area;
                for (int q = 0; q < n; q++) prev[q] = nowv[q];
            }
            lds_sync();
            scr[W.hl] = contrib;
            lds_sync();
            for (unsigned m = hballot(mvd, W.h); m; m &= m - 1u) work += scr[__ffs((int)m) - 1];   // ascending floe order, like the python loop
            lds_sync();
        }
    }
    pair_gsync();
    pair_store_state(P, D, W, A, S);
    if (W.hl == 0) {
        const d2 sp = gA(D.pxy, W.eo);
        const double sa = gA(D.ang, W.eo);
        // cost of the env for the next step's dispatch order: the wave's cycles of this task up to here -- written after the pair loop, so a half that finished
        // first is charged its mate's remaining sub-steps too (an over-estimate that only moves it forward in the heaviest-first order) -- plus its earlier tasks'
        gA(D.e_cost, env) = (unsigned)((__builtin_amdgcn_s_memtime() - t_begin) >> 8) + (resumed ? gA(D.sq_carry, (size_t)env * 4 + 3) : 0u);
        const double total_work = gA(D.e_total_work, env) + work;
        gA(D.e_total_work, env) = total_work;
        int boundary_terminal = 0;
        if (sp.x < 0.0 && __builtin_fabs(sp.x - 0.0) >= 0.0) boundary_terminal = 1;
        if (sp.x > P.map_w && __builtin_fabs(sp.x - P.map_w) >= 0.0) boundary_terminal = 1;
        int term = 0;
        if (sp.y >= P.goal_y) term = 1;
        else if (boundary_terminal) term = 1;
        double dist_reward = 0.0;
        if (sp.y < P.goal_y) {
            const d2 r = gA(D.rot, W.eo);
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
        gA(D.e_lastrew, env) = rwd; gA(D.e_lastflag, env) = term | (success << 1);
        if (info) {
            double *o = info + (size_t)env * BP_INFO_COUNT;
            o[BP_I_X] = sp.x; o[BP_I_Y] = sp.y; o[BP_I_THETA] = sa; o[BP_I_TOTAL_WORK] = total_work; o[BP_I_WORK] = work;
            o[BP_I_COLL_REWARD] = coll; o[BP_I_SCALED_COLL] = coll * P.beta; o[BP_I_DIST_REWARD] = dist_reward;
            o[BP_I_SUCCESS] = success; o[BP_I_BOUNDARY] = S.boundary_violated; o[BP_I_YAW] = S.yaw_violated;
            o[BP_I_KE] = S.total_ke; o[BP_I_IMPULSE] = S.total_imp;
            o[BP_I_NPOST] = (double)S.n_post; o[BP_I_NCONTACT] = (double)S.n_contact; o[BP_I_NFIRST] = (double)S.n_first;
        }
    }
    return 1;
}
