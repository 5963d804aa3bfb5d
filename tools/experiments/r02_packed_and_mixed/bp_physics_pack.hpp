// k_physics_step_pack<K, R>: env.step() physics of ship-ice-v0 with K environments per 64-lane wavefront.
//
// Why: one wavefront per environment (k_physics_step) leaves ~80 % of the lanes idle -- a sub-step of a typical env has 4
// moving bodies, 3 narrow-phase pairs and 3 warm arbiters in 2 colours -- and every one of its ~6 000 instructions per
// sub-step serves a single env.  Here the 64 lanes are a flat pool shared by K envs: every item (moving body, vertex,
// candidate, plane, pair, arbiter) carries a 2-bit env tag and addresses its env's arrays through it, so the instruction
// stream of the low-occupancy phases (solver colours, manifolds, candidates, bookkeeping) is shared K ways, and the
// item-parallel phases (vertex transform, plane search) pack the items of all K envs into dense 64-lane rounds.
// Arbiters stay in registers, R sets per lane (64 * R slots per wave, shared by the K envs on demand); velocity slots
// are one LDS pool per wave.  With K = 4 a 4096-env launch is 1024 single-wave workgroups = one per SIMD, all resident.
//
// Semantics are those of substep() in bp_physics.hpp (same arithmetic, same operation order per item -> bit-identical
// results; the order in which independent items are visited differs, which nothing observable depends on):
// Chipmunk2D 7.0.3 cpSpaceStep as called by ShipIceEnv.step (ship_ice_env.py:261-355).
// Restrictions (checked by the host before this kernel is chosen): ship-ice handles only (one kinematic shape, no static
// shapes, no collision groups), nb_cap < 16384.
#pragma once
#include "bp_physics.hpp"

#define PK_NS 176     // velocity slots per wave (shared by the K envs); slots [0, K) are the ships
#define PK_MVC 128    // moving-list entries per wave
#define PK_PC 128     // narrow-phase pair list per wave (processed 64 pairs at a time)

struct PkLds {
    d2 *sv, *sw, *sb;              // [PK_NS] (vx,vy) (w,w_bias) (vbx,vby)
    d2 *ship;                      // [K] (x, angle) of the ship: read by the per-sub-step yaw / boundary rules
    uint4 *ctx;                    // [4] per env: env * nbcap, trial * nbcap, bodies (0 = no env in this seat), env id
    double *dtc;                   // [4] per env: dt / prev_dt of the current sub-step
    unsigned *stamp0;              // [4] per env: space stamp at the start of this env step
    d2 *tf;                        // [64][2]
    unsigned long long *bbk;       // [64][4]
    unsigned long long *res_smA, *res_smB;          // [64]
    unsigned *res_iA, *res_iB, *res_jA, *res_jB;    // [64]
    unsigned *pq_a, *pq_b;         // [64] global body index (env base + body) of the pair's two shapes
    unsigned short *pl_off;        // [128 + 2]
    unsigned char *pl_na, *pl_nb;  // [64]
    unsigned *pr_ab, *pr_is;       // [PK_PC] pair list: sa | sb << 14 | env << 28 ; i | s << 14
    unsigned short *pr_n;          // [PK_PC] nA | nB << 8
    unsigned *s_mvs;               // [PK_NS] sub-step (1-based within this env step) in which the slot's body last moved, 0 = never
    unsigned short *s_col, *s_own, *s_body;   // [PK_NS] colour mask / owner scratch / body id
    unsigned char *s_env;          // [PK_NS]
    unsigned char *slot_of;        // [K][nbcap] velocity slot of a body, 255 = none (velocity exactly zero)
    unsigned *mv;                  // [PK_MVC] moving list: body | env << 14 | slot << 16
    unsigned char *rf;             // [64]
};

static __host__ __device__ inline size_t pk_lds_bytes(int K, int nbcap)
{
    size_t b = 0;
    b += sizeof(d2) * PK_NS * 3 + sizeof(d2) * 4 + 64 + 32 + 16 + sizeof(d2) * 128 + 8 * 256;
    b += 8 * 64 * 2 + 4 * 64 * 4 + 4 * 64 * 2 + 2 * 136 + 64 * 2;
    b += 4 * PK_PC * 2 + 2 * PK_PC;
    b += 4 * PK_NS + 2 * PK_NS * 3 + PK_NS;
    b += (size_t)K * (size_t)((nbcap + 15) & ~15);
    b += 4 * PK_MVC + 64 + 64;
    return (b + 15) & ~(size_t)15;
}

template <int K>
__device__ __forceinline__ void pk_carve(const DevParams &P, PkLds &L)
{
    char *p = (char *)bp_smem;
    L.sv = (d2 *)p; p += sizeof(d2) * PK_NS;
    L.sw = (d2 *)p; p += sizeof(d2) * PK_NS;
    L.sb = (d2 *)p; p += sizeof(d2) * PK_NS;
    L.ship = (d2 *)p; p += sizeof(d2) * 4;
    L.ctx = (uint4 *)p; p += 64;
    L.dtc = (double *)p; p += 32;
    L.stamp0 = (unsigned *)p; p += 16;
    L.tf = (d2 *)p; p += sizeof(d2) * 128;
    L.bbk = (unsigned long long *)p; p += 8 * 256;
    L.res_smA = (unsigned long long *)p; p += 8 * 64;
    L.res_smB = (unsigned long long *)p; p += 8 * 64;
    L.res_iA = (unsigned *)p; p += 4 * 64;
    L.res_iB = (unsigned *)p; p += 4 * 64;
    L.res_jA = (unsigned *)p; p += 4 * 64;
    L.res_jB = (unsigned *)p; p += 4 * 64;
    L.pq_a = (unsigned *)p; p += 4 * 64;
    L.pq_b = (unsigned *)p; p += 4 * 64;
    L.pl_off = (unsigned short *)p; p += 2 * 136;
    L.pl_na = (unsigned char *)p; p += 64;
    L.pl_nb = (unsigned char *)p; p += 64;
    L.pr_ab = (unsigned *)p; p += 4 * PK_PC;
    L.pr_is = (unsigned *)p; p += 4 * PK_PC;
    L.pr_n = (unsigned short *)p; p += 2 * PK_PC;
    L.s_mvs = (unsigned *)p; p += 4 * PK_NS;
    L.s_col = (unsigned short *)p; p += 2 * PK_NS;
    L.s_own = (unsigned short *)p; p += 2 * PK_NS;
    L.s_body = (unsigned short *)p; p += 2 * PK_NS;
    L.s_env = (unsigned char *)p; p += PK_NS;
    L.slot_of = (unsigned char *)p; p += (size_t)K * (size_t)((P.nbcap + 15) & ~15);
    L.mv = (unsigned *)p; p += 4 * PK_MVC;
    L.rf = (unsigned char *)p; p += 64;
}

#define PK_KEY(e, sa, sb) (((unsigned)(e) << 28) | ((unsigned)(sa) << 14) | (unsigned)(sb))
#define PK_KEY_E(k) ((int)((k) >> 28))
#define PK_KEY_A(k) ((int)(((k) >> 14) & 0x3FFFu))
#define PK_KEY_B(k) ((int)((k) & 0x3FFFu))

#define R_MAX 3
template <int K>
struct PkCtx {
    int env[K], nb[K];
    unsigned eb[K], tb[K];     // env * nbcap, trial * nbcap
    bool valid[K];
    int stride;                // (nbcap + 15) & ~15: slot_of row length
    int nslots, nmv;
    int err;
    unsigned n_post[K], n_contact[K], n_first[K];
    double total_ke[K], total_imp[K], prev_dt[K];
    int yaw[K], boundary[K];
    unsigned cost[K];
    unsigned long long prev_mask[K][R_MAX];   // active set (lane masks per register set) the env's colouring was made for
    int nlevels[K];
#ifdef BP_PROF
    unsigned long long prof[24];
#endif
};

#ifdef BP_PROF
#define PK_PROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime();
#define PK_PROF_ACC(slot) { unsigned long long _n = __builtin_amdgcn_s_memtime(); C.prof[slot] += _n - _pt; _pt = _n; }
#define PK_PROF_CNT(slot, v) { C.prof[slot] += (unsigned long long)(v); }
#else
#define PK_PROF_DECL
#define PK_PROF_ACC(slot)
#define PK_PROF_CNT(slot, v)
#endif

// velocity slot of (env e, body): allocate a zeroed one on first use (wave-uniform call)
template <int K>
__device__ __forceinline__ int pk_slot_get(const PkLds &L, PkCtx<K> &C, int e, int body)
{
    int s = L.slot_of[e * C.stride + body];
    if (s == 255) {
        s = C.nslots;
        if (s >= PK_NS || s >= 255) { C.err |= BP_ERR_ARB_OVERFLOW; s = PK_NS - 1; }
        else C.nslots = s + 1;
        if (lane_id() == 0) {
            L.slot_of[e * C.stride + body] = (unsigned char)s;
            L.sv[s] = mk2(0.0, 0.0); L.sw[s] = mk2(0.0, 0.0); L.sb[s] = mk2(0.0, 0.0);
            L.s_mvs[s] = 0u; L.s_body[s] = (unsigned short)body; L.s_env[s] = (unsigned char)e;
        }
        lds_sync();
    }
    return s;
}

// ---- narrow phase of the listed pairs [p0, p0 + 64): plane search, manifolds, hand-over to the arbiter slots ---------
template <int K, int R>
__device__ __forceinline__ void pk_pairs(const DevParams &P, const DevPtrs &D, const PkLds &L, PkCtx<K> &C, ArbReg (&A)[R],
                                         const int p0, const int np, const int now)
{
    const int lane = lane_id();
    constexpr int VL = BP_MAXV;
    const bool valid = (p0 + lane) < np;
    const int pidx = valid ? p0 + lane : p0;
    const unsigned pab = L.pr_ab[pidx], pis = L.pr_is[pidx];
    const unsigned pn = L.pr_n[pidx];
    const int sa = (int)(pab & 0x3FFFu), sb = (int)((pab >> 14) & 0x3FFFu), e = (int)(pab >> 28);
    const int ci = (int)(pis & 0x3FFFu), cs = (int)(pis >> 14);
    const uint4 cx_eb = L.ctx[e]; const unsigned eb = cx_eb.x, tb = cx_eb.y;
    const int nA_l = valid ? (int)(pn & 0xFFu) : 0, nB_l = valid ? (int)(pn >> 8) : 0;
    const unsigned long long cm = ballot(valid);
    const int nc = __popcll(cm);
    const int myr = lane; // pairs of this batch sit in lanes [0, nc)
    PK_PROF_DECL
    // block layout: the planes of one (pair, side) never straddle a 64-item round
    int total = 0;
    {
        for (int r = 0; r < nc; r++) {
            const int cA = __builtin_amdgcn_readlane(nA_l, r), cB = __builtin_amdgcn_readlane(nB_l, r);
            if ((total & 63) + cA > 64) total = (total + 63) & ~63;
            const int offA = total;
            total += cA;
            if ((total & 63) + cB > 64) total = (total + 63) & ~63;
            const int offB = total;
            total += cB;
            if (lane == r) {
                L.pl_off[2 * r] = (unsigned short)offA; L.pl_off[2 * r + 1] = (unsigned short)offB;
                L.pq_a[r] = eb + (unsigned)sa; L.pq_b[r] = eb + (unsigned)sb;
                L.pl_na[r] = (unsigned char)nA_l; L.pl_nb[r] = (unsigned char)nB_l;
                L.res_smA[r] = 0ull; L.res_smB[r] = 0ull; L.res_iA[r] = 0xFFFFFFFFu; L.res_iB[r] = 0xFFFFFFFFu;
            }
        }
    }
    lds_sync();
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        int eb_ = 0;
        {   // largest block (2 * rank + side) with pl_off[block] <= t
            int lo = 0, hi = 2 * nc - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)L.pl_off[mid] <= t) lo = mid; else hi = mid - 1; }
            eb_ = lo;
        }
        const int r = eb_ >> 1;
        const bool onA = (eb_ & 1) == 0;
        const int pna = L.pl_na[r], pnb = L.pl_nb[r];
        int fidx = t - (int)L.pl_off[eb_];
        const bool tv = (fidx >= 0) && (fidx < (onA ? pna : pnb));
        unsigned long long skey = 0ull;
        int jm = 0;
        if (tv) {
            const unsigned ga = L.pq_a[r], gb = L.pq_b[r];
            const size_t pbody = onA ? ga : gb, qbody = onA ? gb : ga;
            const int f = fidx;
            const int nq = onA ? pnb : pna;
            const d2 fn = D.wn[pbody * BP_MAXV + f], fp = D.wv[pbody * BP_MAXV + f];
            double mn = BP_INF;
#pragma unroll 10
            for (int q = 0; q < VL; q++) { // slots >= nq repeat vertex 0, which cannot win the strict '<'
                const double d = vdot(fn, D.wv[qbody * BP_MAXV + (q < nq ? q : 0)]);
                if (d < mn) { mn = d; jm = q; }
            }
            const double sp = (mn - vdot(fn, fp)) + 0.0; // "+ 0.0": -0 and +0 share one key
            skey = f64_key(sp);
            atomicMax(onA ? &L.res_smA[r] : &L.res_smB[r], skey);
        }
        lds_sync();
        bool isbest = false;
        if (tv) {
            isbest = (skey == (onA ? L.res_smA[r] : L.res_smB[r]));
            if (isbest) atomicMin(onA ? &L.res_iA[r] : &L.res_iB[r], (unsigned)fidx); // ties -> lowest plane index
        }
        lds_sync();
        if (isbest && (unsigned)fidx == (onA ? L.res_iA[r] : L.res_iB[r])) { if (onA) L.res_jA[r] = (unsigned)jm; else L.res_jB[r] = (unsigned)jm; }
    }
    lds_sync();
    PK_PROF_ACC(3)
    // ---- closest features -> normal -> Chipmunk ContactPoints, one pair per lane ----------------------------------
    Manifold M;
    M.count = 0; M.h0 = M.h1 = 0; M.n = mk2(0, 0); M.newhint = 255;
    M.p1_0 = M.p2_0 = M.p1_1 = M.p2_1 = mk2(0, 0);
    if (valid) {
        const int nA = nA_l, nB = nB_l;
        const d2 *Av = D.wv + (size_t)(eb + sa) * BP_MAXV, *An = D.wn + (size_t)(eb + sa) * BP_MAXV;
        const d2 *Bv = D.wv + (size_t)(eb + sb) * BP_MAXV, *Bn = D.wn + (size_t)(eb + sb) * BP_MAXV;
        const double r1 = D.sc_prop[tb + sa].x, r2 = D.sc_prop[tb + sb].x;
        const double rsum = r1 + r2;
        const double sA = key_f64(L.res_smA[myr]), sB = key_f64(L.res_smB[myr]);
        const int iA = (int)L.res_iA[myr], iB = (int)L.res_iB[myr], jA = (int)L.res_jA[myr], jB = (int)L.res_jB[myr];
        const bool useA = (sA >= sB);
        const double smax = useA ? sA : sB;
        bool touching = true;
        d2 n = mk2(0, 0);
        const int iA0 = (iA == 0) ? nA - 1 : iA - 1, iB0 = (iB == 0) ? nB - 1 : iB - 1;
        const d2 nAi = An[iA], nBi = Bn[iB];
        const d2 aA = Av[iA0], bA = Av[iA], qA = Bv[jA];
        const d2 aB = Bv[iB0], bB = Bv[iB], qB = Av[jB];
        if (smax > rsum) { M.newhint = useA ? iA : (nA + iB); touching = false; }
        else if (smax <= 0.0) { n = useA ? nAi : vneg(nBi); }
        else {
            const d2 eA = vsub(bA, aA);
            const double uA = vdot(vsub(qA, aA), eA), eeA = vdot(eA, eA);
            const bool spanA = !(uA < 0.0) && !(uA > eeA);
            const d2 eB = vsub(bB, aB);
            const double uB = vdot(vsub(qB, aB), eB), eeB = vdot(eB, eB);
            const bool spanB = !(uB < 0.0) && !(uB > eeB);
            if (useA) {
                if (spanA) n = nAi;
                else if (sB > 0.0 && spanB) n = vneg(nBi);
                else {
                    const d2 pp = vsub(qA, (uA < 0.0) ? aA : bA);
                    const double dl = vlen(pp);
                    if (dl > rsum) touching = false;
                    n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                }
            } else {
                if (spanB) n = vneg(nBi);
                else if (sA > 0.0 && spanA) n = nAi;
                else {
                    const d2 pp = vsub((uB < 0.0) ? aB : bB, qB);
                    const double dl = vlen(pp);
                    if (dl > rsum) touching = false;
                    n = vmul(pp, 1.0 / (dl + BP_DBL_MIN));
                }
            }
        }
        if (touching) {
            const d2 nn = vneg(n);
            int i1A = 0, i1B = 0;
            {
                double mx = -BP_INF;
#pragma unroll 10
                for (int q = 0; q < VL; q++) {
                    const double d = vdot(Av[q < nA ? q : 0], n);
                    if (d > mx) { mx = d; i1A = q; }
                }
                mx = -BP_INF;
#pragma unroll 10
                for (int q = 0; q < VL; q++) {
                    const double d = vdot(Bv[q < nB ? q : 0], nn);
                    if (d > mx) { mx = d; i1B = q; }
                }
            }
            d2 e1a, e1b, e2a, e2b;
            int e1ia, e1ib, e2ia, e2ib;
            {
                const int i0 = (i1A == 0) ? nA - 1 : i1A - 1, i2 = (i1A + 1 == nA) ? 0 : i1A + 1;
                if (vdot(n, An[i1A]) > vdot(n, An[i2])) { e1a = Av[i0]; e1ia = i0; e1b = Av[i1A]; e1ib = i1A; }
                else { e1a = Av[i1A]; e1ia = i1A; e1b = Av[i2]; e1ib = i2; }
            }
            {
                const int i0 = (i1B == 0) ? nB - 1 : i1B - 1, i2 = (i1B + 1 == nB) ? 0 : i1B + 1;
                if (vdot(nn, Bn[i1B]) > vdot(nn, Bn[i2])) { e2a = Bv[i0]; e2ia = i0; e2b = Bv[i1B]; e2ib = i1B; }
                else { e2a = Bv[i1B]; e2ia = i1B; e2b = Bv[i2]; e2ib = i2; }
            }
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
        D.hint[(size_t)(eb + ci) * BP_KADJ + cs] = (unsigned char)M.newhint;
    }
    PK_PROF_ACC(10)
    // ---- cpArbiterUpdate: the manifold goes to the lane / register set that owns the pair's arbiter -----------------
    // Hand-over through LDS (the plane-search scratch is free now): the pair lane publishes its manifold, the owner reads it.
    const unsigned long long dm = ballot(valid && M.count > 0);
    lds_sync();
    d2 *mbox = (d2 *)L.bbk; // [16][6] (aliases the AABB scratch: 2 KB >= 16 * 96 B)
    const int drank = popc_below(dm, lane);
    const int ndel = __popcll(dm);
    for (int dbase = 0; dbase < ndel; dbase += 16) {
        const bool mine = valid && M.count > 0 && drank >= dbase && drank < dbase + 16;
        if (mine) {
            d2 *mb = mbox + (drank - dbase) * 6;
            mb[0] = M.n; mb[1] = M.p1_0; mb[2] = M.p2_0; mb[3] = M.p1_1; mb[4] = M.p2_1;
            mb[5] = mk2(__hiloint2double((int)M.h0, M.count), __hiloint2double((int)M.h1, 0));
        }
        lds_sync();
        int my_mb[R];
        bool fresh[R];
#pragma unroll
        for (int s = 0; s < R; s++) { my_mb[s] = -1; fresh[s] = false; }
        unsigned long long m = dm;
        int dr = 0;
        while (m) { // slot search / allocation in pair order (wave-uniform)
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (dr >= dbase && dr < dbase + 16) {
                const unsigned key = PK_KEY(__builtin_amdgcn_readlane(e, l), __builtin_amdgcn_readlane(sa, l), __builtin_amdgcn_readlane(sb, l));
                int oset = -1, owner = 0;
                bool fr = false;
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const unsigned long long om = ballot(A[s].key == key);
                    if (oset < 0 && om) { oset = s; owner = __ffsll((long long)om) - 1; }
                }
                if (oset < 0) {
                    fr = true;
#pragma unroll
                    for (int s = 0; s < R; s++) {
                        const unsigned long long om = ballot(A[s].key == ARB_FREE_KEY);
                        if (oset < 0 && om) { oset = s; owner = __ffsll((long long)om) - 1; }
                    }
                }
                if (oset < 0) { C.err |= BP_ERR_ARB_OVERFLOW; }
                else {
                    int s1 = 0, s2 = 0;
                    if (fr) { s1 = pk_slot_get<K>(L, C, PK_KEY_E(key), PK_KEY_A(key)); s2 = pk_slot_get<K>(L, C, PK_KEY_E(key), PK_KEY_B(key)); }
#pragma unroll
                    for (int s = 0; s < R; s++) {
                        if (s == oset && lane == owner) {
                            my_mb[s] = dr - dbase; fresh[s] = fr; A[s].key = key;
                            if (fr) { A[s].slotA = s1; A[s].slotB = s2; }
                        }
                    }
                }
            }
            dr++;
        }
#pragma unroll
        for (int s = 0; s < R; s++) {
            if (my_mb[s] >= 0) {
                ArbReg &a = A[s];
                const d2 *mb = mbox + my_mb[s] * 6;
                const d2 mn_ = mb[0], mp10 = mb[1], mp20 = mb[2], mp11 = mb[3], mp21 = mb[4], mh = mb[5];
                const unsigned mh0 = (unsigned)__double2hiint(mh.x), mh1 = (unsigned)__double2hiint(mh.y);
                const int mcount = __double2loint(mh.x);
                if (fresh[s]) { a.state = ARB_FIRST; a.count = 0; a.h0 = a.h1 = 0; a.jn0 = a.jt0 = a.jn1 = a.jt1 = 0.0; }
                const int ke = PK_KEY_E(a.key), usa = PK_KEY_A(a.key), usb = PK_KEY_B(a.key);
                const uint4 cx_keb = L.ctx[ke]; const unsigned keb = cx_keb.x, ktb = cx_keb.y;
                const d2 pa = D.pxy[keb + usa], pbp = D.pxy[keb + usb];
                double njn0 = 0.0, njt0 = 0.0, njn1 = 0.0, njt1 = 0.0;
                if (a.count > 0 && a.h0 == mh0) { njn0 = a.jn0; njt0 = a.jt0; }
                if (a.count > 1 && a.h1 == mh0) { njn0 = a.jn1; njt0 = a.jt1; }
                if (mcount > 1) {
                    if (a.count > 0 && a.h0 == mh1) { njn1 = a.jn0; njt1 = a.jt0; }
                    if (a.count > 1 && a.h1 == mh1) { njn1 = a.jn1; njt1 = a.jt1; }
                }
                a.jn0 = njn0; a.jt0 = njt0; a.jn1 = njn1; a.jt1 = njt1;
                a.h0 = mh0; a.h1 = mh1;
                a.r1_0 = vsub(mp10, pa); a.r2_0 = vsub(mp20, pbp);
                a.r1_1 = vsub(mp11, pa); a.r2_1 = vsub(mp21, pbp);
                a.count = mcount;
                a.n = mn_;
                if (a.state == ARB_CACHED) a.state = ARB_FIRST;
                a.stamp = (unsigned)now;
                const double4 m1 = D.sc_mass[ktb + usa], m2 = D.sc_mass[ktb + usb];
                a.ma = m1.x; a.ia = m1.y; a.mb = m2.x; a.ib = m2.y;
                const double4 q1 = D.sc_prop[ktb + usa], q2 = D.sc_prop[ktb + usb];
                a.e = q1.y * q2.y; a.u = q1.z * q2.z;
            }
        }
        lds_sync();
    }
    PK_PROF_ACC(4)
}

// ---- one sub-step of the K environments of the wave -------------------------------------------------------------------
template <int K, int R>
__device__ __forceinline__ void pk_substep(const DevParams &P, const DevPtrs &D, const PkLds &L, PkCtx<K> &C, ArbReg (&A)[R],
                                           const int now, const double dt)
{
    const int lane = lane_id();
    constexpr int VL = BP_MAXV;
#pragma unroll
    for (int s = 0; s < R; s++)
        if (A[s].key != ARB_FREE_KEY && A[s].stamp == (unsigned)(now - 1)) A[s].state = ARB_NORMAL;
    PK_PROF_DECL
    PK_PROF_CNT(16, C.nmv)
    const int M = C.nmv;

    // ---- 1. integrate positions of the moving bodies; world geometry; AABBs ----------------------------------------
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        if (k < M) {
            const unsigned mvk = L.mv[k];
            const int i = (int)(mvk & 0x3FFFu), e = (int)((mvk >> 14) & 3u), sl = (int)(mvk >> 16);
            const uint4 cx_eb = L.ctx[e]; const unsigned eb = cx_eb.x, tb = cx_eb.y;
            const d2 v = L.sv[sl], w2 = L.sw[sl], vb = L.sb[sl];
            d2 p = D.pxy[eb + i];
            p.x = p.x + (v.x + vb.x) * dt;
            p.y = p.y + (v.y + vb.y) * dt;
            const double a = D.ang[eb + i];
            const double a2 = a + (w2.x + w2.y) * dt;
            d2 r = D.rot[eb + i];
            if (a2 != a) { double sn, cs; bp_sincos(a2, sn, cs); r = mk2(cs, sn); }
            D.pxy[eb + i] = p; D.ang[eb + i] = a2; D.rot[eb + i] = r;
            L.sb[sl] = mk2(0.0, 0.0); L.sw[sl] = mk2(w2.x, 0.0);
            const double4 ms = D.sc_mass[tb + i];
            double4 t;
            t.x = r.x; t.y = r.y;
            t.z = p.x - (ms.z * r.x - ms.w * r.y);
            t.w = p.y - (ms.z * r.y + ms.w * r.x);
            L.tf[2 * lane] = mk2(t.x, t.y);
            L.tf[2 * lane + 1] = mk2(t.z, t.w);
            L.s_mvs[sl] = (unsigned)now;
            if (i == 0) L.ship[e] = mk2(p.x, a2);
        }
        lds_sync();
        const int cnt = min(64, M - k0);
        unsigned long long *bbk = L.bbk; // [64][4] = min x, max x, min y, max y
        if (lane < cnt) { bbk[lane * 4 + 0] = ~0ull; bbk[lane * 4 + 1] = 0ull; bbk[lane * 4 + 2] = ~0ull; bbk[lane * 4 + 3] = 0ull; }
        lds_sync();
        for (int t0 = 0; t0 < cnt * VL; t0 += 64) {
            const int t = t0 + lane;
            const int kk = t / VL, q = t - kk * VL;
            if (kk < cnt) {
                const unsigned mvk = L.mv[k0 + kk];
                const int i = (int)(mvk & 0x3FFFu), e = (int)((mvk >> 14) & 3u);
                const uint4 cx_eb = L.ctx[e]; const unsigned eb = cx_eb.x, tb = cx_eb.y;
                if (q < D.sc_nv[tb + i]) {
                    const d2 t0_ = L.tf[2 * kk], t1_ = L.tf[2 * kk + 1];
                    const double c = t0_.x, s = t0_.y;
                    const d2 lv = D.sc_lv[(size_t)(tb + i) * BP_MAXV + q], ln = D.sc_ln[(size_t)(tb + i) * BP_MAXV + q];
                    const double vx = (c * lv.x + (-s) * lv.y) + t1_.x;
                    const double vy = (s * lv.x + c * lv.y) + t1_.y;
                    const double nx = c * ln.x + (-s) * ln.y;
                    const double ny = s * ln.x + c * ln.y;
                    D.wv[(size_t)(eb + i) * BP_MAXV + q] = mk2(vx, vy);
                    D.wn[(size_t)(eb + i) * BP_MAXV + q] = mk2(nx, ny);
                    const unsigned long long kx = f64_key(vx), ky = f64_key(vy);
                    atomicMin(&bbk[kk * 4 + 0], kx); atomicMax(&bbk[kk * 4 + 1], kx);
                    atomicMin(&bbk[kk * 4 + 2], ky); atomicMax(&bbk[kk * 4 + 3], ky);
                }
            }
        }
        lds_sync();
        if (lane < cnt) {
            const unsigned mvk = L.mv[k0 + lane];
            const int i = (int)(mvk & 0x3FFFu), e = (int)((mvk >> 14) & 3u);
            const uint4 cx_eb = L.ctx[e]; const unsigned eb = cx_eb.x, tb = cx_eb.y;
            const double rad = D.sc_prop[tb + i].x;
            double4 nbb;
            nbb.x = key_f64(bbk[lane * 4 + 0]) - rad; nbb.y = key_f64(bbk[lane * 4 + 2]) - rad;
            nbb.z = key_f64(bbk[lane * 4 + 1]) + rad; nbb.w = key_f64(bbk[lane * 4 + 3]) + rad;
            D.bb[eb + i] = nbb;
            const double4 f = D.fat[eb + i];
            L.rf[lane] = !(nbb.x >= f.x && nbb.y >= f.y && nbb.z <= f.z && nbb.w <= f.w);
        }
        lds_sync();
        PK_PROF_ACC(0)
        // ---- 2. Verlet refresh (rare): the whole wave rebuilds one body's neighbour list at a time ------------------
        unsigned long long rm = ballot((lane < cnt) && L.rf[lane < cnt ? lane : 0]);
        if (rm) __syncthreads(); // refresh_body reads the AABBs other lanes have just stored
        while (rm) {
            const int kk = __ffsll((long long)rm) - 1;
            rm &= rm - 1;
            const unsigned mvk = (unsigned)__builtin_amdgcn_readfirstlane((int)L.mv[k0 + kk]);
            const int i = (int)(mvk & 0x3FFFu), e = (int)((mvk >> 14) & 3u);
            EnvCtx E;
            const uint4 cxr = L.ctx[e];
            const int env_u = __builtin_amdgcn_readfirstlane((int)cxr.w);
            const unsigned tb_u = (unsigned)__builtin_amdgcn_readfirstlane((int)cxr.y);
            E.nb = __builtin_amdgcn_readfirstlane((int)cxr.z);
            env_ctx(P, D, env_u, (int)(tb_u / (unsigned)P.nbcap), E);
            int rerr = 0;
            refresh_body(P, E, i, rerr);
            C.err |= rerr;
            PK_PROF_CNT(17, 1)
        }
        PK_PROF_ACC(1)
    }
    __syncthreads();

    // ---- 3./4. candidate pairs of the moving bodies -> pair list -> narrow phase ----------------------------------
    int kmax = 0; // largest neighbour count among the moving bodies of the wave
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        int cnt = 0;
        if (k < M) {
            const unsigned mvk = L.mv[k];
            cnt = (int)D.adjn[L.ctx[(mvk >> 14) & 3u].x + (mvk & 0x3FFFu)];
        }
        int m = 0;
        for (int bit = 16; bit >= 1; bit >>= 1) { if (ballot(cnt >= (m | bit))) m |= bit; }
        kmax = max(kmax, m);
    }
    const int ncand_slots = M * kmax;
    int np = 0;
    for (int base = 0; base < ncand_slots; base += 64) {
        const int idx = base + lane;
        const int k = idx / kmax, s = idx - k * kmax;
        const bool inlist = k < M;
        const unsigned mvk = L.mv[inlist ? k : 0];
        const int i = (int)(mvk & 0x3FFFu), e = (int)((mvk >> 14) & 3u);
        const uint4 cx_eb = L.ctx[e]; const unsigned eb = cx_eb.x, tb = cx_eb.y;
        const int nb_e = (int)cx_eb.z;
        const int sc = min(s, BP_KADJ - 1);
        const int adjn_i = D.adjn[eb + i];
        const int jraw = D.adj[(size_t)(eb + i) * BP_KADJ + sc];
        const int j = jraw < nb_e ? jraw : 0;
        const int h = D.hint[(size_t)(eb + i) * BP_KADJ + sc];
        const double4 bbi = D.bb[eb + i];
        bool valid = inlist && (s < adjn_i);
        {
            const int slj = L.slot_of[e * C.stride + j];
            const bool movedj = (slj != 255) && (L.s_mvs[slj != 255 ? slj : 0] == (unsigned)now);
            if (valid && movedj && j < i) valid = false; // pair is evaluated from j's list
        }
        const double4 bbj = D.bb[eb + j];
        const int sa = min(i, j), sb = max(i, j);
        const double rsum = D.sc_prop[tb + sa].x + D.sc_prop[tb + sb].x;
        const int nA_h = D.sc_nv[tb + sa], nB_h = D.sc_nv[tb + sb];
        if (valid) valid = bb_overlap(bbi, bbj);
        if (valid && h != 255) {
            int pb, qb, fi;
            if (h < nA_h) { pb = sa; qb = sb; fi = h; } else { pb = sb; qb = sa; fi = min(h - nA_h, nB_h - 1); }
            const d2 fn = D.wn[(size_t)(eb + pb) * BP_MAXV + fi], fp = D.wv[(size_t)(eb + pb) * BP_MAXV + fi];
            const int nq = (qb == sa) ? nA_h : nB_h;
            const d2 *qv = D.wv + (size_t)(eb + qb) * BP_MAXV;
            double mn = BP_INF;
#pragma unroll 10
            for (int q = 0; q < VL; q++) {
                const double d = vdot(fn, qv[q < nq ? q : 0]);
                if (d < mn) mn = d;
            }
            const double sep = mn - vdot(fn, fp);
            if (sep > rsum) valid = false;
        }
        const unsigned long long cm = ballot(valid);
        PK_PROF_ACC(2)
        PK_PROF_CNT(18, __popcll(cm))
        if (cm) {
            if (valid) {
                const int pos = np + popc_below(cm, lane);
                L.pr_ab[pos] = (unsigned)sa | ((unsigned)sb << 14) | ((unsigned)e << 28);
                L.pr_is[pos] = (unsigned)i | ((unsigned)s << 14);
                L.pr_n[pos] = (unsigned short)(nA_h | (nB_h << 8));
            }
            np += __popcll(cm);
            lds_sync();
        }
        if (np > PK_PC - 64 || (base + 64 >= ncand_slots && np > 0)) { // list full, or last round: run the narrow phase
            for (int p0 = 0; p0 < np; p0 += 64) pk_pairs<K, R>(P, D, L, C, A, p0, np, now);
            np = 0;
        }
    }
    PK_PROF_ACC(4)
    // ---- arbiters whose bodies did not move keep last sub-step's contacts; 5. cpSpaceArbiterSetFilter -------------
    bool active[R];
    unsigned long long amask[R];
#pragma unroll
    for (int s = 0; s < R; s++) {
        ArbReg &a = A[s];
        if (a.key != ARB_FREE_KEY && a.stamp == (unsigned)(now - 1)) {
            if (L.s_mvs[a.slotA] != (unsigned)now && L.s_mvs[a.slotB] != (unsigned)now) a.stamp = (unsigned)now;
        }
        if (a.key != ARB_FREE_KEY) {
            const int ticks = now - (int)a.stamp;
            if (ticks >= 1 && a.state != ARB_CACHED) a.state = ARB_CACHED;
            if (ticks >= P.persistence) a.key = ARB_FREE_KEY;
        }
        active[s] = (a.key != ARB_FREE_KEY) && (a.stamp == (unsigned)now);
        amask[s] = ballot(active[s]);
    }
    PK_PROF_ACC(5)
    // ---- 6a. prestep (cpArbiterPreStep) -----------------------------------------------------------------------------
    bool warm[R];
#pragma unroll
    for (int s = 0; s < R; s++) {
        ArbReg &a = A[s];
        warm[s] = false;
        if (amask[s] == 0) continue;
        if (active[s]) {
            const int ke = PK_KEY_E(a.key), ba = PK_KEY_A(a.key), bbi = PK_KEY_B(a.key);
            const unsigned eb = L.ctx[ke].x;
            const d2 pa = D.pxy[eb + ba], pb = D.pxy[eb + bbi];
            const d2 va = L.sv[a.slotA], vb = L.sv[a.slotB];
            const double wa = L.sw[a.slotA].x, wb = L.sw[a.slotB].x;
            const d2 n = a.n;
            const d2 body_delta = vsub(pb, pa);
            const d2 t = vperp(n);
            {
                const double rcn1 = vcross(a.r1_0, n), rcn2 = vcross(a.r2_0, n);
                a.nMass0 = 1.0 / ((a.ma + a.ia * rcn1 * rcn1) + (a.mb + a.ib * rcn2 * rcn2));
                const double rct1 = vcross(a.r1_0, t), rct2 = vcross(a.r2_0, t);
                a.tMass0 = 1.0 / ((a.ma + a.ia * rct1 * rct1) + (a.mb + a.ib * rct2 * rct2));
                const double dist = vdot(vadd(vsub(a.r2_0, a.r1_0), body_delta), n);
                a.bias0 = -P.bias_coef * fmin(0.0, dist + P.slop) / dt;
                a.jBias0 = 0.0;
                const d2 v1 = vadd(va, vmul(vperp(a.r1_0), wa));
                const d2 v2 = vadd(vb, vmul(vperp(a.r2_0), wb));
                a.bounce0 = vdot(vsub(v2, v1), n) * a.e;
            }
            if (a.count > 1) {
                const double rcn1 = vcross(a.r1_1, n), rcn2 = vcross(a.r2_1, n);
                a.nMass1 = 1.0 / ((a.ma + a.ia * rcn1 * rcn1) + (a.mb + a.ib * rcn2 * rcn2));
                const double rct1 = vcross(a.r1_1, t), rct2 = vcross(a.r2_1, t);
                a.tMass1 = 1.0 / ((a.ma + a.ia * rct1 * rct1) + (a.mb + a.ib * rct2 * rct2));
                const double dist = vdot(vadd(vsub(a.r2_1, a.r1_1), body_delta), n);
                a.bias1 = -P.bias_coef * fmin(0.0, dist + P.slop) / dt;
                a.jBias1 = 0.0;
                const d2 v1 = vadd(va, vmul(vperp(a.r1_1), wa));
                const d2 v2 = vadd(vb, vmul(vperp(a.r2_1), wb));
                a.bounce1 = vdot(vsub(v2, v1), n) * a.e;
            }
            // warm-set seeds: a kinematic body that moves, a cached impulse, a bias or a bounce term
            bool w = (a.jn0 != 0.0) || (a.jt0 != 0.0) || (a.bias0 != 0.0) || (a.bounce0 != 0.0);
            if (a.count > 1) w = w || (a.jn1 != 0.0) || (a.jt1 != 0.0) || (a.bias1 != 0.0) || (a.bounce1 != 0.0);
            if (a.ma == 0.0) w = w || (va.x != 0.0) || (va.y != 0.0) || (wa != 0.0);
            if (a.mb == 0.0) w = w || (vb.x != 0.0) || (vb.y != 0.0) || (wb != 0.0);
            warm[s] = w;
            if (a.ma != 0.0) L.s_own[a.slotA] = 0;
            if (a.mb != 0.0) L.s_own[a.slotB] = 0;
        }
    }
    // warm set: closed under "shares a dynamic body"; every other arbiter provably keeps all its impulses at exactly 0
    {
        unsigned long long wm[R], am_all = 0, wm_all = 0;
        bool differs = false;
#pragma unroll
        for (int s = 0; s < R; s++) { wm[s] = ballot(warm[s]); am_all |= amask[s]; wm_all |= wm[s]; differs = differs || (wm[s] != amask[s]); }
        if (wm_all != 0 && differs) {
            lds_sync();
            for (;;) {
#pragma unroll
                for (int s = 0; s < R; s++)
                    if (warm[s]) { if (A[s].ma != 0.0) L.s_own[A[s].slotA] = 1; if (A[s].mb != 0.0) L.s_own[A[s].slotB] = 1; }
                lds_sync();
                bool grew = false;
#pragma unroll
                for (int s = 0; s < R; s++) {
                    if (active[s] && !warm[s])
                        warm[s] = (A[s].ma != 0.0 && L.s_own[A[s].slotA] != 0) || (A[s].mb != 0.0 && L.s_own[A[s].slotB] != 0);
                    const unsigned long long nm = ballot(warm[s]);
                    grew = grew || (nm != wm[s]);
                    wm[s] = nm;
                }
                if (!grew) break;
            }
        }
    }
    unsigned long long wmask[R], wm_any = 0;
    bool any_bias = false;
#pragma unroll
    for (int s = 0; s < R; s++) {
        wmask[s] = ballot(warm[s]);
        wm_any |= wmask[s];
        any_bias = any_bias || (ballot(warm[s] && ((A[s].bias0 != 0.0) || (A[s].count > 1 && A[s].bias1 != 0.0))) != 0);
    }
    // ---- solve order per env: greedy colouring of the env's active set in ascending key order (cached while the set is
    //      unchanged); arbiters of one colour share no dynamic body, so a colour runs in parallel; order = (colour, key) ----
    int nact_e[K];
#pragma unroll
    for (int e = 0; e < K; e++) {
        unsigned long long em[R];
        bool changed = false;
        int nact = 0;
#pragma unroll
        for (int s = 0; s < R; s++) {
            em[s] = ballot(active[s] && PK_KEY_E(A[s].key) == e);
            changed = changed || (em[s] != C.prev_mask[e][s]);
            nact += __popcll(em[s]);
        }
        nact_e[e] = nact;
        if (changed) {
#pragma unroll
            for (int s = 0; s < R; s++) A[s].rank = (active[s] && PK_KEY_E(A[s].key) == e) ? 0 : A[s].rank;
#pragma unroll
            for (int s2 = 0; s2 < R; s2++) {
                unsigned long long m = em[s2];
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)A[s2].key, l);
#pragma unroll
                    for (int s = 0; s < R; s++) A[s].rank += (active[s] && PK_KEY_E(A[s].key) == e && k < A[s].key) ? 1 : 0;
                }
            }
            lds_sync();
#pragma unroll
            for (int s = 0; s < R; s++)
                if (active[s] && PK_KEY_E(A[s].key) == e) { L.s_col[A[s].slotA] = 0; L.s_col[A[s].slotB] = 0; }
            lds_sync();
            int nlev = 0;
            for (int r = 0; r < nact; r++) {
                int hs = -1, l = 0;
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const unsigned long long rm = ballot(active[s] && PK_KEY_E(A[s].key) == e && A[s].rank == r);
                    if (hs < 0 && rm) { hs = s; l = __ffsll((long long)rm) - 1; }
                }
                int sla = 0, slb = 0;
                bool adyn = false, bdyn = false;
#pragma unroll
                for (int s = 0; s < R; s++) {
                    if (s == hs) {
                        sla = __builtin_amdgcn_readlane(A[s].slotA, l); slb = __builtin_amdgcn_readlane(A[s].slotB, l);
                        adyn = __builtin_amdgcn_readlane(__double2hiint(A[s].ma), l) != 0 || __builtin_amdgcn_readlane(__double2loint(A[s].ma), l) != 0;
                        bdyn = __builtin_amdgcn_readlane(__double2hiint(A[s].mb), l) != 0 || __builtin_amdgcn_readlane(__double2loint(A[s].mb), l) != 0;
                    }
                }
                const unsigned ua = adyn ? (unsigned)L.s_col[sla] : 0u, ub = bdyn ? (unsigned)L.s_col[slb] : 0u;
                const unsigned used = ua | ub;
                int c = __ffs(~used) - 1;
                if (c > 15) { c = 15; C.err |= BP_ERR_LEVEL_OVERFLOW; }
                if (adyn) L.s_col[sla] = (unsigned short)(ua | (1u << c));
                if (bdyn) L.s_col[slb] = (unsigned short)(ub | (1u << c));
#pragma unroll
                for (int s = 0; s < R; s++) if (s == hs && lane == l) A[s].level = c + 1;
                nlev = max(nlev, c + 1);
                lds_sync();
            }
            C.nlevels[e] = nlev;
#pragma unroll
            for (int s = 0; s < R; s++) C.prev_mask[e][s] = em[s];
        }
    }
    lds_sync();
    // ---- 6b. velocity integrate: damping^dt == 0, no gravity/forces -> dynamic bodies' v, w := +0 ---------------------
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        if (k < M) {
            const unsigned mvk = L.mv[k];
            const int i = (int)(mvk & 0x3FFFu), sl = (int)(mvk >> 16);
            if (i >= P.nkin) { L.sv[sl] = mk2(0.0, 0.0); L.sw[sl] = mk2(0.0, L.sw[sl].y); }
        }
    }
    lds_sync();
    PK_PROF_ACC(6)
    // ---- 6c. warm start (cpArbiterApplyCachedImpulse) -----------------------------------------------------------------
    int maxlev = 0;
#pragma unroll
    for (int e = 0; e < K; e++) maxlev = max(maxlev, C.nlevels[e]);
    if (now <= 2) { // dt / prev_dt: the previous env step's dt for the first sub-step, 1 afterwards
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < K; e++) L.dtc[e] = (C.prev_dt[e] == 0.0) ? 0.0 : dt / C.prev_dt[e];
        }
        lds_sync();
    }
    if (!wm_any) maxlev = 0;
    unsigned lvlmask = 0; // colours that hold at least one warm arbiter
    for (int lvl = 1; lvl <= maxlev; lvl++) {
        bool any = false;
#pragma unroll
        for (int s = 0; s < R; s++) any = any || (ballot(warm[s] && A[s].level == lvl) != 0);
        if (any) lvlmask |= 1u << lvl;
    }
    for (int lvl = 1; lvl <= maxlev; lvl++) {
        if (!(lvlmask & (1u << lvl))) continue;
#pragma unroll
        for (int s = 0; s < R; s++) {
            ArbReg &a = A[s];
            if (warm[s] && a.level == lvl && a.state != ARB_FIRST) {
                const double dt_coef = L.dtc[PK_KEY_E(a.key)];
                d2 va = L.sv[a.slotA], vb = L.sv[a.slotB];
                d2 wa2 = L.sw[a.slotA], wb2 = L.sw[a.slotB];
                {
                    const d2 j = vmul(vrotate(a.n, mk2(a.jn0, a.jt0)), dt_coef);
                    apply_contact_impulses(a, 0, va, wa2.x, vb, wb2.x, j);
                }
                if (a.count > 1) {
                    const d2 j = vmul(vrotate(a.n, mk2(a.jn1, a.jt1)), dt_coef);
                    apply_contact_impulses(a, 1, va, wa2.x, vb, wb2.x, j);
                }
                if (a.ma != 0.0) { L.sv[a.slotA] = va; L.sw[a.slotA] = wa2; }
                if (a.mb != 0.0) { L.sv[a.slotB] = vb; L.sw[a.slotB] = wb2; }
            }
        }
        lds_sync();
    }
    PK_PROF_ACC(7)
    // ---- 6d. sequential impulses (cpArbiterApplyImpulse) --------------------------------------------------------------
    unsigned long long wenv[K][R]; // warm arbiters of env e in set s
    int nwarm_e[K];
#pragma unroll
    for (int e = 0; e < K; e++) {
        nwarm_e[e] = 0;
#pragma unroll
        for (int s = 0; s < R; s++) { wenv[e][s] = ballot(warm[s] && PK_KEY_E(A[s].key) == e); nwarm_e[e] += __popcll(wenv[e][s]); }
    }
    PK_PROF_CNT(21, __popcll(wm_any))
    auto iterate = [&](auto bias_tag) {
    constexpr bool AB = decltype(bias_tag)::value;
    bool alive[R];
#pragma unroll
    for (int s = 0; s < R; s++) alive[s] = warm[s];
    for (int it = 0; it < P.iterations; it++) {
        bool changed[R];
#pragma unroll
        for (int s = 0; s < R; s++) changed[s] = false;
        for (int lvl = 1; lvl <= maxlev; lvl++) {
            if (!(lvlmask & (1u << lvl))) continue;
#pragma unroll
            for (int s = 0; s < R; s++) {
                ArbReg &a = A[s];
                if (alive[s] && a.level == lvl) {
                    d2 va = L.sv[a.slotA], vb = L.sv[a.slotB];
                    d2 wa2 = L.sw[a.slotA], wb2 = L.sw[a.slotB];
                    d2 vba = mk2(0.0, 0.0), vbb = mk2(0.0, 0.0);
                    if (AB) { vba = L.sb[a.slotA]; vbb = L.sb[a.slotB]; }
                    const d2 n = a.n;
                    bool ch = false;
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        if (c == 0 || a.count > 1) { // an arbiter always has its first contact
                            const d2 r1 = c ? a.r1_1 : a.r1_0, r2 = c ? a.r2_1 : a.r2_0;
                            const double nMass = c ? a.nMass1 : a.nMass0, tMass = c ? a.tMass1 : a.tMass0;
                            const double bias = c ? a.bias1 : a.bias0, bounce = c ? a.bounce1 : a.bounce0;
                            const d2 v1 = vadd(va, vmul(vperp(r1), wa2.x));
                            const d2 v2 = vadd(vb, vmul(vperp(r2), wb2.x));
                            const d2 vr = vsub(v2, v1);
                            const double vrn = vdot(vr, n);
                            const double vrt = vdot(vr, vperp(n));
                            const double jbnOld = c ? a.jBias1 : a.jBias0;
                            double jBias = jbnOld;
                            if (AB) { // with no bias term anywhere every bias impulse stays exactly 0
                                const d2 vb1 = vadd(vba, vmul(vperp(r1), wa2.y));
                                const d2 vb2 = vadd(vbb, vmul(vperp(r2), wb2.y));
                                const double vbn = vdot(vsub(vb2, vb1), n);
                                const double jbn = (bias - vbn) * nMass;
                                jBias = fmax(jbnOld + jbn, 0.0);
                            }
                            const double jn = -(bounce + vrn) * nMass;
                            const double jnOld = c ? a.jn1 : a.jn0;
                            const double jnAcc = fmax(jnOld + jn, 0.0);
                            const double jtMax = a.u * jnAcc;
                            const double jt = -vrt * tMass;
                            const double jtOld = c ? a.jt1 : a.jt0;
                            const double jtAcc = fclampd(jtOld + jt, -jtMax, jtMax);
                            ch = ch || (jnAcc != jnOld) || (jtAcc != jtOld) || (jBias != jbnOld);
                            if (c) { a.jBias1 = jBias; a.jn1 = jnAcc; a.jt1 = jtAcc; }
                            else   { a.jBias0 = jBias; a.jn0 = jnAcc; a.jt0 = jtAcc; }
                            if (AB) {
                                const d2 jb = vmul(n, jBias - jbnOld);
                                const d2 jbneg = vneg(jb);
                                vba = vadd(vba, vmul(jbneg, a.ma));
                                wa2.y += a.ia * vcross(r1, jbneg);
                                vbb = vadd(vbb, vmul(jb, a.mb));
                                wb2.y += a.ib * vcross(r2, jb);
                            }
                            const d2 j = vrotate(n, mk2(jnAcc - jnOld, jtAcc - jtOld));
                            apply_contact_impulses(a, c, va, wa2.x, vb, wb2.x, j);
                        }
                    }
                    changed[s] = changed[s] || ch;
                    if (a.ma != 0.0) { L.sv[a.slotA] = va; L.sw[a.slotA] = wa2; if (AB) L.sb[a.slotA] = vba; }
                    if (a.mb != 0.0) { L.sv[a.slotB] = vb; L.sw[a.slotB] = wb2; if (AB) L.sb[a.slotB] = vbb; }
                }
            }
            lds_sync();
        }
        // an env whose iteration changed no accumulated impulse applied only zero impulses: its state is a fixed point and
        // its remaining iterations would repeat it exactly -> its lanes stop; the wave stops when every env has
        unsigned long long cm[R], keep[R];
#pragma unroll
        for (int s = 0; s < R; s++) { cm[s] = ballot(changed[s]); keep[s] = 0ull; }
        bool any = false;
#pragma unroll
        for (int e = 0; e < K; e++) {
            bool ch = false;
#pragma unroll
            for (int s = 0; s < R; s++) ch = ch || ((cm[s] & wenv[e][s]) != 0ull);
            if (ch) {
#pragma unroll
                for (int s = 0; s < R; s++) keep[s] |= wenv[e][s];
            }
            any = any || ch;
        }
        if (!any) break;
#pragma unroll
        for (int s = 0; s < R; s++) alive[s] = alive[s] && (((keep[s] >> lane) & 1ull) != 0ull);
    }
    };
    if (any_bias) iterate(std::true_type{}); else iterate(std::false_type{});
    PK_PROF_ACC(8)
    // ---- 7. post-solve bookkeeping for ship(0) x floe arbiters of every env, ascending key order ------------------------
#pragma unroll
    for (int e = 0; e < K; e++) {
        bool shiparb[R];
        int ns = 0, n2 = 0, nf = 0;
        unsigned long long wsm_any = 0ull;
#pragma unroll
        for (int s = 0; s < R; s++) {
            shiparb[s] = active[s] && PK_KEY_E(A[s].key) == e && PK_KEY_A(A[s].key) == 0;
            const unsigned long long sm = ballot(shiparb[s]);
            ns += __popcll(sm);
            if (sm) {
                n2 += __popcll(ballot(shiparb[s] && A[s].count > 1));
                nf += __popcll(ballot(shiparb[s] && A[s].state == ARB_FIRST));
                wsm_any |= ballot(shiparb[s] && warm[s]);
            }
        }
        // integer bookkeeping is order-free; cold arbiters add exactly +0 to the float sums
        C.n_post[e] += (unsigned)ns; C.n_contact[e] += (unsigned)(ns + n2); C.n_first[e] += (unsigned)nf;
        if (wsm_any) {
            double ke[R], imp[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const ArbReg &a = A[s];
                ke[s] = 0.0; imp[s] = 0.0;
                if (shiparb[s] && warm[s]) {
                    const double eCoef = (1 - a.e) / (1 + a.e);
                    double k2 = 0.0;
                    d2 js = mk2(0.0, 0.0);
                    k2 += eCoef * a.jn0 * a.jn0 / a.nMass0 + a.jt0 * a.jt0 / a.tMass0;
                    js = vadd(js, vrotate(a.n, mk2(a.jn0, a.jt0)));
                    if (a.count > 1) {
                        k2 += eCoef * a.jn1 * a.jn1 / a.nMass1 + a.jt1 * a.jt1 / a.tMass1;
                        js = vadd(js, vrotate(a.n, mk2(a.jn1, a.jt1)));
                    }
                    ke[s] = k2; imp[s] = vlen(js);
                }
            }
            for (int r = 0; r < ns; r++) { // ship arbiters have the smallest keys of the env's active set: ranks 0..ns-1
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const unsigned long long rm = ballot(shiparb[s] && warm[s] && A[s].rank == r);
                    if (rm) {
                        const int l = __ffsll((long long)rm) - 1;
                        C.total_ke[e] += __shfl(ke[s], l);
                        C.total_imp[e] += __shfl(imp[s], l);
                    }
                }
            }
        }
        C.cost[e] += 16u + 2u * (unsigned)nact_e[e] + 4u * (unsigned)(nwarm_e[e] * C.nlevels[e]);
    }
    // ---- agent rules applied after every sub-step: yaw limits + channel boundary (ship_ice_env.py:284-290) ---------------
    {
        const d2 sp = L.ship[lane < K ? lane : 0];
        const bool yv = (lane < K) && (sp.y <= 0.0 || sp.y >= BP_PI);
        const bool bv = (lane < K) && (sp.x < 0.0 || sp.x > P.map_w);
        if (yv && L.ctx[lane].z != 0u) L.sw[lane] = mk2(0.0, L.sw[lane].y);
        const unsigned long long ym = ballot(yv), bm = ballot(bv);
#pragma unroll
        for (int e = 0; e < K; e++) {
            if (C.valid[e] && ((ym >> e) & 1ull)) C.yaw[e] = 1;
            if (C.valid[e] && ((bm >> e) & 1ull)) C.boundary[e] = 1;
        }
    }
    lds_sync();
    // ---- next sub-step's moving list: bodies of active arbiters with a non-zero velocity, plus the ships ------------------
    {
        bool wantA[R], wantB[R];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const ArbReg &a = A[s];
            wantA[s] = wantB[s] = false;
            if (active[s]) {
                if (a.ma != 0.0) {
                    const d2 v = L.sv[a.slotA], w2 = L.sw[a.slotA], vb = L.sb[a.slotA];
                    wantA[s] = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
                }
                if (a.mb != 0.0) {
                    const d2 v = L.sv[a.slotB], w2 = L.sw[a.slotB], vb = L.sb[a.slotB];
                    wantB[s] = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < R; s++) if (wantA[s]) L.s_own[A[s].slotA] = (unsigned short)((s * 64 + lane) * 2 + 2);
        lds_sync();
#pragma unroll
        for (int s = 0; s < R; s++) if (wantB[s]) L.s_own[A[s].slotB] = (unsigned short)((s * 64 + lane) * 2 + 3);
        lds_sync();
        // the ships: every env whose ship has a non-zero velocity
        bool shipmv = false;
        if (lane < K) {
            const d2 v0 = L.sv[lane], w0 = L.sw[lane];
            shipmv = (L.ctx[lane].z != 0u) && (v0.x != 0.0 || v0.y != 0.0 || w0.x != 0.0);
        }
        const unsigned long long smk = ballot(shipmv);
        int n = __popcll(smk);
        if (shipmv) L.mv[popc_below(smk, lane)] = (unsigned)(0u | ((unsigned)lane << 14) | ((unsigned)lane << 16));
#pragma unroll
        for (int s = 0; s < R; s++) {
            const ArbReg &a = A[s];
            const bool gotA = wantA[s] && L.s_own[a.slotA] == (unsigned short)((s * 64 + lane) * 2 + 2);
            const bool gotB = wantB[s] && L.s_own[a.slotB] == (unsigned short)((s * 64 + lane) * 2 + 3);
            const unsigned long long mA = ballot(gotA), mB = ballot(gotB);
            const int nA_ = __popcll(mA);
            const unsigned ke = (unsigned)PK_KEY_E(a.key);
            if (gotA) { const int pos = n + popc_below(mA, lane); if (pos < PK_MVC) L.mv[pos] = (unsigned)PK_KEY_A(a.key) | (ke << 14) | ((unsigned)a.slotA << 16); }
            if (gotB) { const int pos = n + nA_ + popc_below(mB, lane); if (pos < PK_MVC) L.mv[pos] = (unsigned)PK_KEY_B(a.key) | (ke << 14) | ((unsigned)a.slotB << 16); }
            n += nA_ + __popcll(mB);
        }
        if (n > PK_MVC) { C.err |= BP_ERR_ARB_OVERFLOW; n = PK_MVC; }
        C.nmv = n;
    }
#pragma unroll
    for (int e = 0; e < K; e++) C.prev_dt[e] = dt;
    lds_sync();
    PK_PROF_ACC(9)
}

// ---- env.step() of K environments per wavefront --------------------------------------------------------------------------
template <int K, int R>
__device__ __forceinline__ void pk_body(const DevParams &P, const DevPtrs &D, const double *__restrict__ actions,
                                        double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                        unsigned char *__restrict__ truncated, double *__restrict__ info, const int pos0, const int npos)
{
    const int lane = lane_id();
    const int W = (int)gridDim.x;
    const int nbcap = P.nbcap;
    PkLds L;
    pk_carve<K>(P, L);
    PkCtx<K> C;
    C.stride = (nbcap + 15) & ~15;
    C.err = 0; C.nslots = K; C.nmv = 0;
    unsigned stamp0[K];
#ifdef BP_PROF
    for (int q = 0; q < 24; q++) C.prof[q] = 0;
    const unsigned long long _t_kernel0 = __builtin_amdgcn_s_memtime();
#endif
    // wave b takes the envs at positions b, 2W-1-b, 2W+b, 4W-1-b of the cost-sorted order (heaviest first): a snake, so
    // that the per-wave sums of the previous step's costs are about equal
#pragma unroll
    for (int e = 0; e < K; e++) {
        const int rel = P.pack_adjacent ? (K * (int)blockIdx.x + e) : ((e & 1) ? ((e + 1) * W - 1 - (int)blockIdx.x) : (e * W + (int)blockIdx.x));
        const bool valid = rel < npos;   // the launch covers positions [pos0, pos0 + npos) of the dispatch order
        const int pos = pos0 + rel;
        const int env = valid ? (D.order != nullptr ? D.order[pos] : pos) : 0;
        C.valid[e] = valid; C.env[e] = env;
        const int trial = valid ? D.e_trial[env] : 0;
        C.nb[e] = valid ? D.e_nb[env] : 0;
        C.eb[e] = (unsigned)env * (unsigned)nbcap; C.tb[e] = (unsigned)trial * (unsigned)nbcap;
        stamp0[e] = D.e_stamp[env];
        C.prev_dt[e] = D.e_currdt[env];
        C.total_ke[e] = D.e_ke[env]; C.total_imp[e] = D.e_imp[env];
        C.n_post[e] = D.e_cnt[env * 4 + 0]; C.n_contact[e] = D.e_cnt[env * 4 + 1]; C.n_first[e] = D.e_cnt[env * 4 + 2];
        C.yaw[e] = 0; C.boundary[e] = 0; C.cost[e] = 0; C.nlevels[e] = 0;
#pragma unroll
        for (int s = 0; s < R_MAX; s++) C.prev_mask[e][s] = 0ull;
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < K; e++) {
            uint4 cx; cx.x = C.eb[e]; cx.y = C.tb[e]; cx.z = (unsigned)C.nb[e]; cx.w = (unsigned)C.env[e];
            L.ctx[e] = cx; L.stamp0[e] = stamp0[e]; L.dtc[e] = 0.0;
        }
        for (int e = K; e < 4; e++) { uint4 cx; cx.x = cx.y = cx.z = cx.w = 0u; L.ctx[e] = cx; }
    }
    lds_sync();
    // ---- persistent state -> LDS / registers ------------------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < K; e++)
        for (int base = 0; base < C.stride; base += 64) {
            const int i = base + lane;
            if (i < C.stride) L.slot_of[e * C.stride + i] = (i == 0) ? (unsigned char)e : (unsigned char)255;
        }
    if (lane < K) {
        const unsigned eb = L.ctx[lane].x;
        const bool v = L.ctx[lane].z != 0u;
        d2 sv = mk2(0.0, 0.0), sw = sv, sb = sv, shp = sv;
        if (v) {
            // ship control (ship_ice_env.py:265-274): set once per env step
            const double act = actions[L.ctx[lane].w] * P.max_yaw_rate;
            const d2 r = D.rot[eb];
            sv = mk2(r.x * P.target_speed + -r.y * 0.0, r.y * P.target_speed + r.x * 0.0);
            sw = mk2(act, D.velw[eb].y);
            sb = D.velb[eb];
            shp = mk2(D.pxy[eb].x, D.ang[eb]);
        }
        L.sv[lane] = sv; L.sw[lane] = sw; L.sb[lane] = sb; L.ship[lane] = shp;
        L.s_mvs[lane] = 0u; L.s_body[lane] = 0; L.s_env[lane] = (unsigned char)lane;
    }
    lds_sync();
    // moving list: every body with a non-zero velocity gets a velocity slot (the ship of env e owns slot e)
    {
        int n = 0;
#pragma unroll
        for (int e = 0; e < K; e++) {
            const unsigned eb = C.eb[e];
            for (int base = 0; base < C.nb[e]; base += 64) {
                const int i = base + lane;
                bool mvg = false;
                d2 v = mk2(0.0, 0.0), w2 = v, vb = v;
                if (i < C.nb[e]) {
                    if (i == 0) { v = L.sv[e]; w2 = L.sw[e]; vb = L.sb[e]; }
                    else { v = D.velv[eb + i]; w2 = D.velw[eb + i]; vb = D.velb[eb + i]; }
                    mvg = (v.x != 0.0 || v.y != 0.0 || w2.x != 0.0 || w2.y != 0.0 || vb.x != 0.0 || vb.y != 0.0);
                }
                const unsigned long long m = ballot(mvg);
                const unsigned long long ms = ballot(mvg && i != 0);
                int sl = e;
                if (mvg && i != 0) {
                    sl = C.nslots + popc_below(ms, lane);
                    if (sl >= PK_NS) sl = PK_NS - 1;
                    L.slot_of[e * C.stride + i] = (unsigned char)sl;
                    L.sv[sl] = v; L.sw[sl] = w2; L.sb[sl] = vb;
                    L.s_mvs[sl] = 0u; L.s_body[sl] = (unsigned short)i; L.s_env[sl] = (unsigned char)e;
                }
                if (mvg) { const int pos = n + popc_below(m, lane); if (pos < PK_MVC) L.mv[pos] = (unsigned)i | ((unsigned)e << 14) | ((unsigned)sl << 16); }
                n += __popcll(m);
                if (C.nslots + __popcll(ms) > PK_NS) { C.err |= BP_ERR_ARB_OVERFLOW; C.nslots = PK_NS; }
                else C.nslots += __popcll(ms);
            }
        }
        if (n > PK_MVC) { C.err |= BP_ERR_ARB_OVERFLOW; n = PK_MVC; }
        C.nmv = n;
    }
    lds_sync();
    // persisted arbiters of the K envs -> the 64 * R register slots of the wave (compacted)
    ArbReg A[R];
    {
        unsigned *scr = (unsigned *)L.bbk; // [512]
        int narb = 0;
#pragma unroll
        for (int e = 0; e < K; e++) {
            const unsigned kp = C.valid[e] ? D.a_key[(size_t)C.env[e] * BP_ACAP + lane] : ARB_FREE_KEY;
            const bool have = kp != ARB_FREE_KEY;
            const unsigned long long m = ballot(have);
            const int idx = narb + popc_below(m, lane);
            if (have && idx < 64 * R) scr[idx] = (unsigned)(e * 64 + lane);
            narb += __popcll(m);
        }
        if (narb > 64 * R) { C.err |= BP_ERR_ARB_OVERFLOW; narb = 64 * R; }
        lds_sync();
#pragma unroll
        for (int s = 0; s < R; s++) {
            ArbReg &a = A[s];
            a.key = ARB_FREE_KEY; a.stamp = 0; a.state = ARB_FIRST; a.count = 0; a.h0 = a.h1 = 0; a.level = 0; a.rank = 0;
            a.jn0 = a.jt0 = a.jn1 = a.jt1 = 0.0;
            a.n = mk2(0, 0); a.r1_0 = a.r2_0 = a.r1_1 = a.r2_1 = mk2(0, 0);
            a.nMass0 = a.tMass0 = a.bias0 = a.bounce0 = a.jBias0 = 0.0;
            a.nMass1 = a.tMass1 = a.bias1 = a.bounce1 = a.jBias1 = 0.0;
            a.ma = a.ia = a.mb = a.ib = 0.0; a.e = 0.0; a.u = 0.0;
            a.slotA = a.slotB = 0;
            const int t = s * 64 + lane;
            if (t < narb) {
                const unsigned src = scr[t];
                const int e = (int)(src >> 6);
                const uint4 cxa = L.ctx[e];
                const size_t ab = (size_t)cxa.w * BP_ACAP + (src & 63u);
                const unsigned kp = D.a_key[ab];
                const int usa = (int)(kp >> 16), usb = (int)(kp & 0xFFFFu);
                a.key = PK_KEY(e, usa, usb);
                a.stamp = D.a_stamp[ab] - L.stamp0[e]; // relative to the env's stamp at the start of this step (<= 0)
                { const unsigned sc = D.a_sc[ab]; a.state = (int)(sc & 0xFF); a.count = (int)(sc >> 8); }
                a.h0 = D.a_h0[ab]; a.h1 = D.a_h1[ab];
                const double *ad = D.a_d + ab * 14;
                a.jn0 = ad[0]; a.jt0 = ad[1]; a.jn1 = ad[2]; a.jt1 = ad[3];
                a.n = mk2(ad[4], ad[5]);
                a.r1_0 = mk2(ad[6], ad[7]); a.r2_0 = mk2(ad[8], ad[9]); a.r1_1 = mk2(ad[10], ad[11]); a.r2_1 = mk2(ad[12], ad[13]);
                const unsigned tb = cxa.y;
                const double4 m1 = D.sc_mass[tb + usa], m2 = D.sc_mass[tb + usb];
                a.ma = m1.x; a.ia = m1.y; a.mb = m2.x; a.ib = m2.y;
                const double4 q1 = D.sc_prop[tb + usa], q2 = D.sc_prop[tb + usb];
                a.e = q1.y * q2.y; a.u = q1.z * q2.z;
            }
        }
        lds_sync();
        // velocity slots for the bodies of the persisted arbiters
#pragma unroll
        for (int s = 0; s < R; s++) {
            unsigned long long m = ballot(A[s].key != ARB_FREE_KEY);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const unsigned key = (unsigned)__builtin_amdgcn_readlane((int)A[s].key, l);
                const int s1 = pk_slot_get<K>(L, C, PK_KEY_E(key), PK_KEY_A(key)), s2 = pk_slot_get<K>(L, C, PK_KEY_E(key), PK_KEY_B(key));
                if (lane == l) { A[s].slotA = s1; A[s].slotB = s2; }
            }
        }
    }
    __syncthreads();

    for (int it = 0; it < P.steps; it++) pk_substep<K, R>(P, D, L, C, A, it + 1, P.dt_sub);

    __syncthreads();
    // ---- end of step: work (evaluation/metrics.py:96-113), velocities and arbiters back to HBM, reward / termination --------
    double work[K];
#pragma unroll
    for (int e = 0; e < K; e++) {
        work[e] = 0.0;
        const unsigned eb = C.eb[e], tb = C.tb[e];
        for (int base = 0; base < C.nb[e]; base += 64) {
            const int i = base + lane;
            bool mvd = false;
            if (i < C.nb[e]) {
                const int sl = L.slot_of[e * C.stride + i];
                mvd = (sl != 255) && (L.s_mvs[sl != 255 ? sl : 0] != 0u) && (kind_ctype(D.sc_kind[tb + i]) == 2); // floes only
            }
            double contrib = 0.0;
            if (mvd) {
                const int n = D.sc_nv[tb + i];
                d2 *prev = D.pv + (size_t)(eb + i) * BP_MAXV;
                const d2 *nowv = D.wv + (size_t)(eb + i) * BP_MAXV;
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
                work[e] += __shfl(contrib, l);
            }
        }
    }
    // velocities of every body that holds a slot (all others are exactly zero, in HBM as well)
    for (int base = 0; base < C.nslots; base += 64) {
        const int sl = base + lane;
        if (sl < C.nslots) {
            const int e = L.s_env[sl];
            const uint4 cxs = L.ctx[e];
            if (cxs.z != 0u) {
                const size_t o = (size_t)cxs.x + L.s_body[sl];
                D.velv[o] = L.sv[sl]; D.velw[o] = L.sw[sl]; D.velb[o] = L.sb[sl];
            }
        }
    }
    // arbiters: env e's go to its persistent slots [0, n), the rest are freed
#pragma unroll
    for (int e = 0; e < K; e++) {
        if (!C.valid[e]) continue;
        int n = 0;
        const size_t ab0 = (size_t)C.env[e] * BP_ACAP;
#pragma unroll
        for (int s = 0; s < R; s++) {
            const ArbReg &a = A[s];
            const bool mine = (a.key != ARB_FREE_KEY) && PK_KEY_E(a.key) == e;
            const unsigned long long m = ballot(mine);
            const int pos = n + popc_below(m, lane);
            if (mine && pos < BP_ACAP) {
                const size_t ab = ab0 + pos;
                D.a_key[ab] = ((unsigned)PK_KEY_A(a.key) << 16) | (unsigned)PK_KEY_B(a.key);
                D.a_stamp[ab] = a.stamp + stamp0[e];
                D.a_sc[ab] = (unsigned)a.state | ((unsigned)a.count << 8);
                D.a_h0[ab] = a.h0; D.a_h1[ab] = a.h1;
                double *ad = D.a_d + ab * 14;
                ad[0] = a.jn0; ad[1] = a.jt0; ad[2] = a.jn1; ad[3] = a.jt1; ad[4] = a.n.x; ad[5] = a.n.y;
                ad[6] = a.r1_0.x; ad[7] = a.r1_0.y; ad[8] = a.r2_0.x; ad[9] = a.r2_0.y;
                ad[10] = a.r1_1.x; ad[11] = a.r1_1.y; ad[12] = a.r2_1.x; ad[13] = a.r2_1.y;
            }
            n += __popcll(m);
        }
        if (n > BP_ACAP) { C.err |= BP_ERR_ARB_OVERFLOW; n = BP_ACAP; }
        if (lane >= n) D.a_key[ab0 + lane] = ARB_FREE_KEY;
    }
    const int err_any = (ballot((C.err & BP_ERR_ADJ_OVERFLOW) != 0) ? BP_ERR_ADJ_OVERFLOW : 0) |
                        (ballot((C.err & BP_ERR_ARB_OVERFLOW) != 0) ? BP_ERR_ARB_OVERFLOW : 0) |
                        (ballot((C.err & BP_ERR_LEVEL_OVERFLOW) != 0) ? BP_ERR_LEVEL_OVERFLOW : 0);
#ifdef BP_PROF
    if (D.prof != nullptr && lane == 0) {
        C.prof[23] = __builtin_amdgcn_s_memtime() - _t_kernel0;
        for (int e = 0; e < K; e++)
            if (C.valid[e]) for (int q = 0; q < 24; q++) D.prof[(size_t)C.env[e] * 24 + q] = C.prof[q];
    }
#endif
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < K; e++) {
            if (!C.valid[e]) continue;
            const int env = C.env[e];
            const d2 sp = D.pxy[C.eb[e]];
            const double sa = D.ang[C.eb[e]];
            D.e_stamp[env] = stamp0[e] + (unsigned)P.steps; D.e_currdt[env] = P.dt_sub;
            D.e_cost[env] = C.cost[e];
            D.e_ke[env] = C.total_ke[e]; D.e_imp[env] = C.total_imp[e];
            D.e_cnt[env * 4 + 0] = C.n_post[e]; D.e_cnt[env * 4 + 1] = C.n_contact[e]; D.e_cnt[env * 4 + 2] = C.n_first[e];
            if (err_any) atomicOr(&D.e_err[env], err_any);
            const double total_work = D.e_total_work[env] + work[e];
            D.e_total_work[env] = total_work;
            int boundary_terminal = 0;
            if (sp.x < 0.0 && __builtin_fabs(sp.x - 0.0) >= 0.0) boundary_terminal = 1;
            if (sp.x > P.map_w && __builtin_fabs(sp.x - P.map_w) >= 0.0) boundary_terminal = 1;
            int term = 0;
            if (sp.y >= P.goal_y) term = 1;
            else if (boundary_terminal) term = 1;
            double dist_reward = 0.0;
            if (sp.y < P.goal_y) {
                const d2 r = D.rot[C.eb[e]];
                dist_reward = 1.0 * (r.x * 0.0 + r.y * 1.0);
            }
            const double coll = -work[e];
            double rwd = P.beta * coll + dist_reward;
            if (C.yaw[e]) rwd += 0.0;
            if (C.boundary[e]) rwd += P.boundary_penalty;
            int success = 0;
            if (term && !boundary_terminal) { rwd += P.terminal_reward; success = 1; }
            if (reward) reward[env] = rwd;
            if (terminated) terminated[env] = (unsigned char)term;
            if (truncated) truncated[env] = 0;
            D.e_lastrew[env] = rwd; D.e_lastflag[env] = term | (success << 1);
            if (info) {
                double *o = info + (size_t)env * BP_INFO_COUNT;
                o[BP_I_X] = sp.x; o[BP_I_Y] = sp.y; o[BP_I_THETA] = sa; o[BP_I_TOTAL_WORK] = total_work; o[BP_I_WORK] = work[e];
                o[BP_I_COLL_REWARD] = coll; o[BP_I_SCALED_COLL] = coll * P.beta; o[BP_I_DIST_REWARD] = dist_reward;
                o[BP_I_SUCCESS] = success; o[BP_I_BOUNDARY] = C.boundary[e]; o[BP_I_YAW] = C.yaw[e];
                o[BP_I_KE] = C.total_ke[e]; o[BP_I_IMPULSE] = C.total_imp[e];
                o[BP_I_NPOST] = (double)C.n_post[e]; o[BP_I_NCONTACT] = (double)C.n_contact[e]; o[BP_I_NFIRST] = (double)C.n_first[e];
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_physics_step_pack4(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                           double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                           unsigned char *__restrict__ truncated, double *__restrict__ info,
                                                           const int pos0, const int npos)
{
    pk_body<4, 2>(P, D, actions, reward, terminated, truncated, info, pos0, npos);
}
__global__ __launch_bounds__(64, 2) void k_physics_step_pack2(const DevParams P, const DevPtrs D, const double *__restrict__ actions,
                                                           double *__restrict__ reward, unsigned char *__restrict__ terminated,
                                                           unsigned char *__restrict__ truncated, double *__restrict__ info,
                                                           const int pos0, const int npos)
{
    pk_body<2, 1>(P, D, actions, reward, terminated, truncated, info, pos0, npos);
}
