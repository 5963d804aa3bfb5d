/*
 * bp_oracle.c -- CPU restatement ("oracle") of the BenchPush ship-ice env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under benchpush_amd/ (the product) may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" for the physics (pymunk==6.5.1 / Chipmunk2D 7.0.3 is an
 * un-vendored third-party dependency of the reference, requirements.txt:10, absent from
 * /root/reference and from this image) and for the skimage.draw.polygon / cv2.line raster
 * rules (unpinned third-party, absent).  The numpy-only pieces (poly_area, poly_centroid,
 * total_work_done, goal-distance crop, window arithmetic) ARE pinned by golden vectors
 * generated from the reference itself (tests/golden/), and the step loop's control / yaw / boundary / reward / termination logic is
 * pinned against the reference's ShipIceEnv / MazeNAMO classes running on a stand-in space (tests/golden/make_golden_step_logic.py).
 *
 * What is restated, with the reference call sites it follows (paths relative to /root/reference):
 *   - ShipIceEnv.step / reset / reward / termination   benchpush/environments/ship_ice_nav/ship_ice_env.py:223-355
 *   - body/shape construction                          benchpush/common/utils/sim_utils.py:136-163, benchpush/common/ship.py:77-98
 *   - pymunk.Space.step(dt)  (Chipmunk2D 7.0.3 cpSpaceStep, published algorithm restated below)
 *                                                      call sites ship_ice_env.py:219,281
 *   - CostMap.get_obs_from_poly                        benchpush/common/cost_map.py:275-281
 *   - total_work_done / poly_area / poly_centroid      benchpush/common/evaluation/metrics.py:96-113,197-198; geometry/polygon.py:25-41
 *   - generate_observation + OccupancyGrid             ship_ice_env.py:378-409; occupancy_grid/occupancy_map.py:37-65,97-140,300-337,379-433,492-587
 *
 * Arithmetic: IEEE binary64 throughout, no FMA contraction (build with -ffp-contract=off), every
 * sum in the index order written here.  sin/cos use the deterministic bp_sincos below (fdlibm-style
 * kernels, <=2 ulp from libm) so that a GPU implementation can be bit-identical.
 *
 * Deliberate, documented choices where Chipmunk's behaviour is not a function of its inputs alone:
 *   - Arbiter (contact pair) solve order: ascending (colour, shapeA, shapeB) with shapeA < shapeB and colour from a
 *     greedy colouring in ascending (shapeA, shapeB) order (see space_step).  Chipmunk's order falls out of its
 *     BB-tree / hash-set internals; any fixed order is an equally valid Gauss-Seidel sweep.
 *   - The closest-feature query (Chipmunk: GJK+EPA, cpCollision.c) is restated as an exact
 *     separating-axis / closest-feature search that yields the same (normal, touching?) answer up
 *     to rounding; only the normal and the boolean reach cpCollision.c:ContactPoints.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXV 24
#define ORC_OBS_C 4

typedef struct { double x, y; } vec;

static inline vec V(double x, double y) { vec r = {x, y}; return r; }
static inline vec vadd(vec a, vec b) { return V(a.x + b.x, a.y + b.y); }
static inline vec vsub(vec a, vec b) { return V(a.x - b.x, a.y - b.y); }
static inline vec vneg(vec a) { return V(-a.x, -a.y); }
static inline vec vmult(vec a, double s) { return V(a.x * s, a.y * s); }
static inline double vdot(vec a, vec b) { return a.x * b.x + a.y * b.y; }
static inline double vcross(vec a, vec b) { return a.x * b.y - a.y * b.x; }
static inline vec vperp(vec a) { return V(-a.y, a.x); }
static inline vec vrperp(vec a) { return V(a.y, -a.x); }
static inline vec vrotate(vec a, vec b) { return V(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
static inline double vlength(vec a) { return sqrt(vdot(a, a)); }
static inline vec vnormalize(vec a) { return vmult(a, 1.0 / (vlength(a) + DBL_MIN)); }
static inline vec vlerp(vec a, vec b, double t) { return vadd(vmult(a, 1.0 - t), vmult(b, t)); }
static inline double fclamp(double f, double lo, double hi) { return fmin(fmax(f, lo), hi); }
static inline double fclamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }

/* ---------------------------------------------------------------------------------------------
 * Deterministic sin/cos (replaces libm's in cpvforangle, cpBody.c:SetTransform).
 * Cody-Waite reduction by pi/2 (fdlibm e_rem_pio2.c medium path) + fdlibm k_sin/k_cos polynomials.
 * ------------------------------------------------------------------------------------------- */
static void bp_sincos(double x, double *sn, double *cs)
{
    static const double invpio2 = 6.36619772367581382433e-01;
    static const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    static const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = rint(x * invpio2);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y0 = r - w;
    if (fabs(y0) < fabs(x) * 7.62939453125e-06 /* 2^-17 */) {
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
    }
    double y1 = (r - y0) - w;
    int q = (int)((long long)fn & 3);
    /* kernels on (y0, y1) */
    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double kc = 1.0 - (0.5 * z - (z * rc - y0 * y1));
    switch (q) {
    case 0: *sn = ks;  *cs = kc;  break;
    case 1: *sn = kc;  *cs = -ks; break;
    case 2: *sn = -ks; *cs = -kc; break;
    default: *sn = -kc; *cs = ks; break;
    }
}

/* --------------------------------------------------------------------------------------------- */
enum { BODY_DYNAMIC = 0, BODY_KINEMATIC = 1, BODY_STATIC = 2 };
enum { ARB_FIRST = 0, ARB_NORMAL = 1, ARB_IGNORE = 2, ARB_CACHED = 3 };

typedef struct {
    int type;
    vec p; double a;
    vec v; double w;
    vec vb; double wb;
    double m, i, m_inv, i_inv;
    vec cog;
    double ta, tb, tc, td, tx, ty; /* cpTransform {a,b,c,d,tx,ty} */
} body_t;

typedef struct {
    int body, n, ctype;
    double r, e, u;
    vec lv[ORC_MAXV], ln[ORC_MAXV];
    vec wv[ORC_MAXV], wn[ORC_MAXV];
    double bl, bb, br, bt;
} shape_t;

typedef struct {
    vec r1, r2;
    double nMass, tMass, bounce, bias, jnAcc, jtAcc, jBias;
    uint32_t hash;
} contact_t;

typedef struct {
    int sa, sb, state, count;
    long stamp;
    contact_t con[2];
    vec n;
    double e, u;
} arb_t;

typedef struct orc_params {
    double dt;               /* env step, config.yaml:4 */
    int    steps;            /* sub-steps per env step, config.yaml:39 */
    int    iterations;       /* config.yaml:43 */
    int    persistence;      /* cpSpace collisionPersistence default 3 */
    int    settle_steps;     /* ship_ice_env.py:218 */
    int    brute_force;      /* 1: all-pairs broadphase (cross-check), 0: sweep */
    double damping_pow;      /* pow(space.damping, dt_sub)  (cpSpaceStep) */
    double bias_coef;        /* 1 - pow(collisionBias, dt_sub), collisionBias = pow(0.9, 60) */
    double slop;             /* cpSpace collisionSlop default 0.1 */
    double target_speed;     /* config.yaml:5 */
    double max_yaw_rate;     /* ship_ice_env.py:71 */
    double map_w, map_h;     /* config.yaml:52-53 */
    double goal_y;           /* config.yaml:62 */
    double m_to_pix;         /* config.yaml:51 */
    double density;          /* config.yaml:45 */
    double poly_radius;      /* sim_utils.py:144, ship.py:90 */
    double elasticity;       /* 0.01 */
    double friction;         /* 1.0 */
    double beta;             /* ship_ice_env.py:60 */
    double boundary_penalty; /* -50 */
    double terminal_reward;  /* 200 */
    double local_w, local_h; /* 6, 6 (ship_ice_env.py:91) */
    double vshift;           /* local_window_v_shift 2 */
    double obs_range;        /* local_range 12 (ship_ice_env.py:383) */
    /* maze-NAMO-v0 only (maze_NAMO_env.py; appended so that the ship-ice prefix keeps its layout) */
    double goal_x;           /* cfg.env.goal_x (goal_y above) */
    double goal_reach;       /* cfg.goal_radius + cfg.robot.min_r, maze_NAMO_env.py:528-536 */
    double k_increment;      /* 150, maze_NAMO_env.py:82 */
    double wall_radius;      /* 0.5, sim_utils.py:177 */
} orc_params;

typedef struct orc_env {
    orc_params P;
    int nb, ns;
    body_t *bodies;
    shape_t *shapes;
    /* arbiter table sorted by key */
    arb_t *arbs; int narb, caparb;
    int *active; int nactive, capactive;
    int *solve, *color; int capsolve; uint32_t *used; long stat_ncol_max;
    /* order-sensitivity study (orc_set_solve_order): 0 = the documented (colour, key) order, 1 = ascending key, 2 = the order in
     * which the y-sweep met the pairs this sub-step, 3 = a seeded random permutation per sub-step, 4 = descending key */
    int solve_order; uint64_t order_seed; int *disc; int capdisc;
    long stamp;
    double curr_dt;
    /* sweep order */
    int *order;
    /* ship config */
    int ship_nv; vec ship_verts[ORC_MAXV + 8];
    vec head, tail;
    /* episode state */
    double *prev_wv; /* previous world verts [ns-1][ORC_MAXV][2] */
    double total_work;
    /* bookkeeping counters (ship_ice_env.py:150-180) */
    double total_ke, total_impulse;
    long n_post_solve, n_contact_pts, n_first_contact;
    /* stats for design studies */
    long stat_pairs_bb, stat_narrow, stat_arb_sum, stat_moving_sum, stat_arb_max, stat_substeps;
    long stat_hot_sum;
    /* maze-NAMO-v0 */
    int wall_collision;        /* sticky flag set by the (1,3) pre_solve handler, maze_NAMO_env.py:203-205 */
    int nwalls; double walls[16][4];
    int robot_nv; vec robot_verts[ORC_MAXV];
    double *dist_map, *wall_map; int map_h, map_w; /* normalised BFS goal map, wall raster */
    double *dist_raw;
    int have_prev_dist; double prev_dist;
    double *ras_occ, *ras_foot, *ras_orient; size_t ras_n; /* raster scratch */
    /* box-delivery-v0 (bp_oracle_bd.c) */
    int kind;                  /* 0 = ship-ice / maze, 2 = box-delivery (collision handlers of box_delivery_env.py:208-229) */
    unsigned char *removed;    /* per shape: space.remove(body, shape) was called (box_delivery_env.py:765-767) */
    int robot_hit;             /* robot_hit_obstacle, written by the (1,3) pre_solve (box_delivery_env.py:208-210) */
    struct { uint32_t key; vec n, r1, r2; } *events; int nevents, capevents; /* (1,3)/(2,3) pre_solve calls of this sub-step */
} orc_env;

/* ---- Chipmunk geometry helpers (cpPolyline.c cpConvexHull / cpChipmunk.c) restated ---- */
static void loop_indexes(const vec *v, int n, int *start, int *end)
{
    *start = *end = 0;
    vec mn = v[0], mx = v[0];
    for (int i = 1; i < n; i++) {
        vec q = v[i];
        if (q.x < mn.x || (q.x == mn.x && q.y < mn.y)) { mn = q; *start = i; }
        else if (q.x > mx.x || (q.x == mx.x && q.y > mx.y)) { mx = q; *end = i; }
    }
}
#define SWAPV(a, b) do { vec _t = (a); (a) = (b); (b) = _t; } while (0)
static int qhull_partition(vec *v, int n, vec a, vec b, double tol)
{
    if (n == 0) return 0;
    double mx = 0; int pivot = 0;
    vec delta = vsub(b, a);
    double valueTol = tol * vlength(delta);
    int head = 0;
    for (int tail = n - 1; head <= tail;) {
        double value = vcross(vsub(v[head], a), delta);
        if (value > valueTol) {
            if (value > mx) { mx = value; pivot = head; }
            head++;
        } else {
            SWAPV(v[head], v[tail]);
            tail--;
        }
    }
    if (pivot != 0) SWAPV(v[0], v[pivot]);
    return head;
}
static int qhull_reduce(double tol, vec *v, int n, vec a, vec pivot, vec b, vec *result)
{
    if (n < 0) return 0;
    if (n == 0) { result[0] = pivot; return 1; }
    int left = qhull_partition(v, n, a, pivot, tol);
    int index = qhull_reduce(tol, v + 1, left - 1, a, v[0], pivot, result);
    result[index++] = pivot;
    int right = qhull_partition(v + left, n - left, pivot, b, tol);
    return index + qhull_reduce(tol, v + left + 1, right - 1, pivot, v[left], b, result + index);
}
/* cpConvexHull(count, verts, result, NULL, 0.0); result may not alias verts here. */
static int convex_hull(int n, const vec *verts, vec *result)
{
    memcpy(result, verts, (size_t)n * sizeof(vec));
    int start, end;
    loop_indexes(verts, n, &start, &end);
    if (start == end) return 1;
    SWAPV(result[0], result[start]);
    SWAPV(result[1], result[end == 0 ? start : end]);
    vec a = result[0], b = result[1];
    return qhull_reduce(0.0, result + 2, n - 2, a, b, a, result + 1) + 1;
}
static vec centroid_for_poly(int n, const vec *v)
{
    double sum = 0.0; vec vsum = V(0, 0);
    for (int i = 0; i < n; i++) {
        vec v1 = v[i], v2 = v[(i + 1) % n];
        double cr = vcross(v1, v2);
        sum += cr;
        vsum = vadd(vsum, vmult(vadd(v1, v2), cr));
    }
    return vmult(vsum, 1.0 / (3.0 * sum));
}
static double area_for_poly(int n, const vec *v, double r)
{
    double area = 0.0, perim = 0.0;
    for (int i = 0; i < n; i++) {
        vec v1 = v[i], v2 = v[(i + 1) % n];
        area += vcross(v1, v2);
        perim += vlength(vsub(v1, v2));
    }
    return r * (M_PI * fabs(r) + perim) + area / 2.0;
}
static double moment_for_poly(double m, int n, const vec *v, vec off)
{
    double sum1 = 0.0, sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        vec v1 = vadd(v[i], off), v2 = vadd(v[(i + 1) % n], off);
        double a = vcross(v2, v1);
        double b = vdot(v1, v1) + vdot(v1, v2) + vdot(v2, v2);
        sum1 += a * b;
        sum2 += a;
    }
    return (m * sum1) / (6.0 * sum2);
}
/* cpPolyShape.c SetVerts: plane i = (v0 = verts[i], n = normalize(rperp(verts[i] - verts[i-1]))) */
static void shape_set_verts(shape_t *s, int n, const vec *v)
{
    s->n = n;
    for (int i = 0; i < n; i++) {
        vec a = v[(i - 1 + n) % n], b = v[i];
        s->lv[i] = b;
        s->ln[i] = vnormalize(vrperp(vsub(b, a)));
    }
}

/* ---- body helpers (cpBody.c) ---- */
static void body_set_transform(body_t *b)
{
    double sn, cs;
    bp_sincos(b->a, &sn, &cs);
    vec c = b->cog;
    b->ta = cs; b->tb = sn; b->tc = -sn; b->td = cs;
    b->tx = b->p.x - (c.x * cs - c.y * sn);
    b->ty = b->p.y - (c.x * sn + c.y * cs);
}
static void shape_cache_bb(shape_t *s, const body_t *b)
{
    double l = INFINITY, r = -INFINITY, bo = INFINITY, t = -INFINITY;
    for (int i = 0; i < s->n; i++) {
        vec lv = s->lv[i], ln = s->ln[i];
        vec v = V(b->ta * lv.x + b->tc * lv.y + b->tx, b->tb * lv.x + b->td * lv.y + b->ty);
        vec n = V(b->ta * ln.x + b->tc * ln.y, b->tb * ln.x + b->td * ln.y);
        s->wv[i] = v; s->wn[i] = n;
        l = fmin(l, v.x); r = fmax(r, v.x); bo = fmin(bo, v.y); t = fmax(t, v.y);
    }
    s->bl = l - s->r; s->bb = bo - s->r; s->br = r + s->r; s->bt = t + s->r;
}

/* ---- narrow phase: closest features of two convex polygons -> Chipmunk ContactPoints ---- */
typedef struct { vec a, b; int ia, ib; double r; vec n; } edge_t;

static int support_index(const shape_t *s, vec n)
{
    double mx = -INFINITY; int idx = 0;
    for (int i = 0; i < s->n; i++) {
        double d = vdot(s->wv[i], n);
        if (d > mx) { mx = d; idx = i; }
    }
    return idx;
}
/* cpCollision.c SupportEdgeForPoly */
static edge_t support_edge(const shape_t *s, vec n)
{
    int cnt = s->n;
    int i1 = support_index(s, n);
    int i0 = (i1 - 1 + cnt) % cnt;
    int i2 = (i1 + 1) % cnt;
    edge_t e;
    if (vdot(n, s->wn[i1]) > vdot(n, s->wn[i2])) {
        e.a = s->wv[i0]; e.ia = i0; e.b = s->wv[i1]; e.ib = i1; e.r = s->r; e.n = s->wn[i1];
    } else {
        e.a = s->wv[i1]; e.ia = i1; e.b = s->wv[i2]; e.ib = i2; e.r = s->r; e.n = s->wn[i2];
    }
    return e;
}
/* separation of plane i of P from polygon Q: min_j n_i.q_j - n_i.p_i ; also returns argmin j */
static double face_sep(const shape_t *P, int i, const shape_t *Q, int *jmin)
{
    vec n = P->wn[i];
    double mn = INFINITY; int jm = 0;
    for (int j = 0; j < Q->n; j++) {
        double d = vdot(n, Q->wv[j]);
        if (d < mn) { mn = d; jm = j; }
    }
    *jmin = jm;
    return mn - vdot(n, P->wv[i]);
}
/* Is vertex q inside the span of edge (i-1 -> i) of P?  If not, *k = nearer end vertex index. */
static int in_span(const shape_t *P, int i, vec q, int *k)
{
    int i0 = (i - 1 + P->n) % P->n;
    vec a = P->wv[i0], b = P->wv[i];
    vec e = vsub(b, a);
    double u = vdot(vsub(q, a), e);
    double ee = vdot(e, e);
    if (u < 0.0) { *k = i0; return 0; }
    if (u > ee) { *k = i; return 0; }
    *k = -1;
    return 1;
}

/* Facing edges that are parallel (to rounding) and overlap only partly: the plane search's support vertex of Q -- the FIRST minimum of n . q, a tie or
 * one ulp apart between the two ends of Q's facing edge -- can be the end that lies beyond edge i of P while the other end lies over it, and the same
 * on the other side; the closest features are then the two edges (distance = the plane separation), not the vertex pair.  k = the end of P's edge
 * that the support vertex j lies beyond (in_span); the neighbour of j towards the span (both hulls are counter-clockwise, so along Q's facing side the
 * tangential coordinate falls with the index) takes its place if it lies over the edge and no higher above the plane than BP_TIE_TOL.  Chipmunk's GJK
 * returns a point pair of the overlap with the edge normal in this configuration (cpCollision.c ClosestPoints). */
#define BP_TIE_TOL 1e-9
static int tie_partner_in_span(const shape_t *P, int i, const shape_t *Q, int j, int k)
{
    if (k < 0 || Q->n < 2) return 0;
    int i0 = (i - 1 + P->n) % P->n;
    int jn = (k == i0) ? (j - 1 + Q->n) % Q->n : (j + 1) % Q->n;
    vec n = P->wn[i];
    int kk;
    if (!in_span(P, i, Q->wv[jn], &kk)) return 0;
    return vdot(n, Q->wv[jn]) - vdot(n, Q->wv[j]) <= BP_TIE_TOL;
}

typedef struct { int count; vec n; vec p1[2], p2[2]; uint32_t hash[2]; } manifold_t;

/* Returns 1 and fills *n when the core polygons are within rsum of each other (Chipmunk:
 * GJK/EPA closest points with d - r1 - r2 <= 0), n pointing from A to B. */
static int closest_normal(const shape_t *A, const shape_t *B, vec *nout)
{
    double rsum = A->r + B->r;
    double sA = -INFINITY, sB = -INFINITY; int iA = 0, jA = 0, iB = 0, jB = 0;
    for (int i = 0; i < A->n; i++) {
        int j; double s = face_sep(A, i, B, &j);
        if (s > sA) { sA = s; iA = i; jA = j; }
    }
    for (int i = 0; i < B->n; i++) {
        int j; double s = face_sep(B, i, A, &j);
        if (s > sB) { sB = s; iB = i; jB = j; }
    }
    int useA = (sA >= sB);
    double smax = useA ? sA : sB;
    if (smax > rsum) return 0;
    if (smax <= 0.0) { /* overlapping cores: minimum-penetration axis (EPA result) */
        *nout = useA ? A->wn[iA] : vneg(B->wn[iB]);
        return 1;
    }
    /* separated cores, 0 < smax <= rsum: vertex/edge or vertex/vertex */
    int k;
    if (useA) {
        if (in_span(A, iA, B->wv[jA], &k)) { *nout = A->wn[iA]; return 1; }
        int k2 = -1;
        if (sB > 0.0 && in_span(B, iB, A->wv[jB], &k2)) { *nout = vneg(B->wn[iB]); return 1; }
        if (tie_partner_in_span(A, iA, B, jA, k)) { *nout = A->wn[iA]; return 1; }
        if (sB > 0.0 && tie_partner_in_span(B, iB, A, jB, k2)) { *nout = vneg(B->wn[iB]); return 1; }
        vec p = vsub(B->wv[jA], A->wv[k]);
        double d2 = vlength(p);
        if (d2 > rsum) return 0;
        *nout = vmult(p, 1.0 / (d2 + DBL_MIN));
        return 1;
    } else {
        if (in_span(B, iB, A->wv[jB], &k)) { *nout = vneg(B->wn[iB]); return 1; }
        int k2 = -1;
        if (sA > 0.0 && in_span(A, iA, B->wv[jA], &k2)) { *nout = A->wn[iA]; return 1; }
        if (tie_partner_in_span(B, iB, A, jB, k)) { *nout = vneg(B->wn[iB]); return 1; }
        if (sA > 0.0 && tie_partner_in_span(A, iA, B, jA, k2)) { *nout = A->wn[iA]; return 1; }
        vec p = vsub(B->wv[k], A->wv[jB]);
        double d2 = vlength(p);
        if (d2 > rsum) return 0;
        *nout = vmult(p, 1.0 / (d2 + DBL_MIN));
        return 1;
    }
}

/* cpCollision.c ContactPoints */
static void contact_points(edge_t e1, edge_t e2, vec n, manifold_t *m)
{
    m->count = 0; m->n = n;
    double d_e1_a = vcross(e1.a, n), d_e1_b = vcross(e1.b, n);
    double d_e2_a = vcross(e2.a, n), d_e2_b = vcross(e2.b, n);
    double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
    double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
    {
        vec p1 = vadd(vmult(n, e1.r), vlerp(e1.a, e1.b, fclamp01((d_e2_b - d_e1_a) * e1_denom)));
        vec p2 = vadd(vmult(n, -e2.r), vlerp(e2.a, e2.b, fclamp01((d_e1_a - d_e2_a) * e2_denom)));
        double dist = vdot(vsub(p2, p1), n);
        if (dist <= 0.0) {
            int c = m->count++;
            m->p1[c] = p1; m->p2[c] = p2; m->hash[c] = ((uint32_t)e1.ia << 8) | (uint32_t)e2.ib;
        }
    }
    {
        vec p1 = vadd(vmult(n, e1.r), vlerp(e1.a, e1.b, fclamp01((d_e2_a - d_e1_a) * e1_denom)));
        vec p2 = vadd(vmult(n, -e2.r), vlerp(e2.a, e2.b, fclamp01((d_e1_b - d_e2_a) * e2_denom)));
        double dist = vdot(vsub(p2, p1), n);
        if (dist <= 0.0) {
            int c = m->count++;
            m->p1[c] = p1; m->p2[c] = p2; m->hash[c] = ((uint32_t)e1.ib << 8) | (uint32_t)e2.ia;
        }
    }
}

static void collide_poly_poly(const shape_t *A, const shape_t *B, manifold_t *m)
{
    vec n;
    m->count = 0;
    if (!closest_normal(A, B, &n)) return;
    contact_points(support_edge(A, n), support_edge(B, vneg(n)), n, m);
}

/* ---- arbiter table ---- */
static inline uint32_t arb_key(int sa, int sb) { return ((uint32_t)sa << 16) | (uint32_t)sb; }
static int arb_find(const orc_env *E, uint32_t key, int *pos)
{
    int lo = 0, hi = E->narb;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        uint32_t k = arb_key(E->arbs[mid].sa, E->arbs[mid].sb);
        if (k < key) lo = mid + 1; else hi = mid;
    }
    *pos = lo;
    return lo < E->narb && arb_key(E->arbs[lo].sa, E->arbs[lo].sb) == key;
}
static arb_t *arb_get_or_insert(orc_env *E, int sa, int sb)
{
    int pos;
    if (arb_find(E, arb_key(sa, sb), &pos)) return &E->arbs[pos];
    if (E->narb == E->caparb) {
        E->caparb = E->caparb ? E->caparb * 2 : 64;
        E->arbs = (arb_t *)realloc(E->arbs, (size_t)E->caparb * sizeof(arb_t));
    }
    memmove(&E->arbs[pos + 1], &E->arbs[pos], (size_t)(E->narb - pos) * sizeof(arb_t));
    E->narb++;
    arb_t *a = &E->arbs[pos];
    memset(a, 0, sizeof(*a));
    a->sa = sa; a->sb = sb; a->state = ARB_FIRST; a->count = 0; a->stamp = E->stamp; /* cpArbiterInit */
    return a;
}

/* cpSpaceStep.c cpSpaceCollideShapes for one candidate pair (sa < sb) */
static void collide_pair(orc_env *E, int sa, int sb)
{
    shape_t *A = &E->shapes[sa], *B = &E->shapes[sb];
    /* QueryReject */
    if (!(A->bl <= B->br && B->bl <= A->br && A->bb <= B->bt && B->bb <= A->bt)) return;
    if (A->body == B->body) return;
    if (E->bodies[A->body].type == BODY_STATIC && E->bodies[B->body].type == BODY_STATIC) return; /* never queried by Chipmunk */
    if (E->removed && (E->removed[sa] || E->removed[sb])) return;
    E->stat_pairs_bb++;
    manifold_t m;
    collide_poly_poly(A, B, &m);
    E->stat_narrow++;
    if (m.count == 0) return;
    arb_t *arb = arb_get_or_insert(E, sa, sb);
    body_t *a = &E->bodies[A->body], *b = &E->bodies[B->body];
    /* cpArbiterUpdate */
    contact_t nc[2];
    for (int i = 0; i < m.count; i++) {
        contact_t *c = &nc[i];
        memset(c, 0, sizeof(*c));
        c->r1 = vsub(m.p1[i], a->p);
        c->r2 = vsub(m.p2[i], b->p);
        c->hash = m.hash[i];
        c->jnAcc = c->jtAcc = 0.0;
        for (int j = 0; j < arb->count; j++) {
            if (arb->con[j].hash == c->hash) { c->jnAcc = arb->con[j].jnAcc; c->jtAcc = arb->con[j].jtAcc; }
        }
    }
    for (int i = 0; i < m.count; i++) arb->con[i] = nc[i];
    arb->count = m.count;
    arb->n = m.n;
    arb->e = A->e * B->e;
    arb->u = A->u * B->u;
    if (arb->state == ARB_CACHED) arb->state = ARB_FIRST;
    /* begin/pre_solve handlers of the reference always return True (ship_ice_env.py:150-153);
     * the maze's (1,3) robot x wall pre_solve additionally raises a flag (maze_NAMO_env.py:203-205) */
    if ((A->ctype == 1 && B->ctype == 3) || (A->ctype == 3 && B->ctype == 1)) E->wall_collision = 1;
    int presolve_ok = 1;
    if (E->kind == 2) {
        /* box_delivery_env.py:208-229: (1,3) and (2,3) pre_solve -> prevent_boundary_intersection (deferred to the end of
         * the collision phase and run in ascending key order, see space_step); (2,4) pre_solve returns False;
         * (1,4) begin returns False (both bodies have infinite mass anyway).  Shape order: A is the robot/box, B the static. */
        if ((A->ctype == 1 || A->ctype == 2) && B->ctype == 3) {
            if (E->nevents == E->capevents) {
                E->capevents = E->capevents ? E->capevents * 2 : 32;
                E->events = realloc(E->events, (size_t)E->capevents * sizeof(*E->events));
            }
            E->events[E->nevents].key = arb_key(sa, sb);
            E->events[E->nevents].n = arb->n; E->events[E->nevents].r1 = arb->con[0].r1; E->events[E->nevents].r2 = arb->con[0].r2;
            E->nevents++;
        }
        if (A->ctype == 2 && B->ctype == 4) presolve_ok = 0;
    }
    if (presolve_ok && !(a->m_inv == 0.0 && b->m_inv == 0.0)) {
        if (E->nactive == E->capactive) {
            E->capactive = E->capactive ? E->capactive * 2 : 64;
            E->active = (int *)realloc(E->active, (size_t)E->capactive * sizeof(int));
        }
        E->active[E->nactive++] = (int)arb_key(sa, sb);
    } else {
        arb->count = 0;
        if (arb->state != ARB_IGNORE) arb->state = ARB_NORMAL;
    }
    arb->stamp = E->stamp;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

/* cpArbiter.c */
static void arb_prestep(orc_env *E, arb_t *arb, double dt)
{
    body_t *a = &E->bodies[E->shapes[arb->sa].body], *b = &E->bodies[E->shapes[arb->sb].body];
    vec n = arb->n;
    vec body_delta = vsub(b->p, a->p);
    for (int i = 0; i < arb->count; i++) {
        contact_t *c = &arb->con[i];
        double rcn1 = vcross(c->r1, n), rcn2 = vcross(c->r2, n);
        c->nMass = 1.0 / ((a->m_inv + a->i_inv * rcn1 * rcn1) + (b->m_inv + b->i_inv * rcn2 * rcn2));
        vec t = vperp(n);
        double rct1 = vcross(c->r1, t), rct2 = vcross(c->r2, t);
        c->tMass = 1.0 / ((a->m_inv + a->i_inv * rct1 * rct1) + (b->m_inv + b->i_inv * rct2 * rct2));
        double dist = vdot(vadd(vsub(c->r2, c->r1), body_delta), n);
        c->bias = -E->P.bias_coef * fmin(0.0, dist + E->P.slop) / dt;
        c->jBias = 0.0;
        vec v1 = vadd(a->v, vmult(vperp(c->r1), a->w));
        vec v2 = vadd(b->v, vmult(vperp(c->r2), b->w));
        c->bounce = vdot(vsub(v2, v1), n) * arb->e;
    }
}
static inline void apply_impulse(body_t *b, vec j, vec r)
{
    b->v = vadd(b->v, vmult(j, b->m_inv));
    b->w += b->i_inv * vcross(r, j);
}
static inline void apply_bias_impulse(body_t *b, vec j, vec r)
{
    b->vb = vadd(b->vb, vmult(j, b->m_inv));
    b->wb += b->i_inv * vcross(r, j);
}
static void arb_apply_cached(orc_env *E, arb_t *arb, double dt_coef)
{
    if (arb->state == ARB_FIRST) return;
    body_t *a = &E->bodies[E->shapes[arb->sa].body], *b = &E->bodies[E->shapes[arb->sb].body];
    for (int i = 0; i < arb->count; i++) {
        contact_t *c = &arb->con[i];
        vec j = vmult(vrotate(arb->n, V(c->jnAcc, c->jtAcc)), dt_coef);
        apply_impulse(a, vneg(j), c->r1);
        apply_impulse(b, j, c->r2);
    }
}
static void arb_apply_impulse(orc_env *E, arb_t *arb)
{
    body_t *a = &E->bodies[E->shapes[arb->sa].body], *b = &E->bodies[E->shapes[arb->sb].body];
    vec n = arb->n;
    double friction = arb->u;
    for (int i = 0; i < arb->count; i++) {
        contact_t *c = &arb->con[i];
        vec r1 = c->r1, r2 = c->r2;
        vec vb1 = vadd(a->vb, vmult(vperp(r1), a->wb));
        vec vb2 = vadd(b->vb, vmult(vperp(r2), b->wb));
        vec v1 = vadd(a->v, vmult(vperp(r1), a->w));
        vec v2 = vadd(b->v, vmult(vperp(r2), b->w));
        vec vr = vsub(v2, v1); /* surface_vr == 0 */
        double vbn = vdot(vsub(vb2, vb1), n);
        double vrn = vdot(vr, n);
        double vrt = vdot(vr, vperp(n));
        double jbn = (c->bias - vbn) * c->nMass;
        double jbnOld = c->jBias;
        c->jBias = fmax(jbnOld + jbn, 0.0);
        double jn = -(c->bounce + vrn) * c->nMass;
        double jnOld = c->jnAcc;
        c->jnAcc = fmax(jnOld + jn, 0.0);
        double jtMax = friction * c->jnAcc;
        double jt = -vrt * c->tMass;
        double jtOld = c->jtAcc;
        c->jtAcc = fclamp(jtOld + jt, -jtMax, jtMax);
        vec jb = vmult(n, c->jBias - jbnOld);
        apply_bias_impulse(a, vneg(jb), r1);
        apply_bias_impulse(b, jb, r2);
        vec j = vrotate(n, V(c->jnAcc - jnOld, c->jtAcc - jtOld));
        apply_impulse(a, vneg(j), r1);
        apply_impulse(b, j, r2);
    }
}

/* BoxDeliveryEnv.prevent_boundary_intersection (box_delivery_env.py:294-311) for every (1,3)/(2,3) pair that touched in
 * this sub-step.  pymunk calls it from inside the collision phase in BB-tree pair order; here the calls run after the
 * collision phase in ascending (shapeA, shapeB) order (same documented choice as the solve order).  contact_point_set:
 * cpArbiterGetContactPointSet (normal = n, distance = dot((b.p + r2) - (a.p + r1), n) with the bodies' current positions). */
static void bd_run_presolve(orc_env *E)
{
    qsort(E->events, (size_t)E->nevents, sizeof(*E->events), cmp_u32);
    for (int k = 0; k < E->nevents; k++) {
        int sa = (int)(E->events[k].key >> 16), sb = (int)(E->events[k].key & 0xFFFFu);
        body_t *a = &E->bodies[E->shapes[sa].body], *b = &E->bodies[E->shapes[sb].body];
        vec n = E->events[k].n;
        vec v = a->v;
        double f = 2 * (v.x * n.x + v.y * n.y);
        vec refl = V(v.x - n.x * f, v.y - n.y * f);
        vec nv = V(refl.x * 0.5, refl.y * 0.5);
        vec p1 = vadd(a->p, E->events[k].r1), p2 = vadd(b->p, E->events[k].r2);
        double depth = vdot(vsub(p2, p1), n);
        if (E->shapes[sa].ctype == 1) E->robot_hit = depth < 0;
        a->p = V(a->p.x + n.x * depth, a->p.y + n.y * depth);
        body_set_transform(a);
        a->v = nv;
    }
    E->nevents = 0;
}

/* pymunk.Space.step(dt) == Chipmunk2D 7.0.3 cpSpaceStep (cpSpaceStep.c), restated */
static void space_step(orc_env *E, double dt)
{
    if (dt == 0.0) return;
    E->stamp++;
    double prev_dt = E->curr_dt;
    E->curr_dt = dt;
    /* reset arbiter states of last step's active list */
    for (int k = 0; k < E->nactive; k++) {
        int pos;
        if (arb_find(E, (uint32_t)E->active[k], &pos)) E->arbs[pos].state = ARB_NORMAL;
    }
    E->nactive = 0;
    /* integrate positions (cpBodyUpdatePosition) of dynamic + kinematic bodies */
    long moving = 0;
    for (int i = 0; i < E->nb; i++) {
        body_t *b = &E->bodies[i];
        if (b->type == BODY_STATIC) continue;
        if (b->v.x != 0.0 || b->v.y != 0.0 || b->w != 0.0 || b->vb.x != 0.0 || b->vb.y != 0.0 || b->wb != 0.0) moving++;
        b->p = vadd(b->p, vmult(vadd(b->v, b->vb), dt));
        b->a = b->a + (b->w + b->wb) * dt;
        body_set_transform(b);
        b->vb = V(0, 0); b->wb = 0.0;
    }
    E->stat_moving_sum += moving;
    /* cache shapes */
    for (int s = 0; s < E->ns; s++) {
        shape_t *sh = &E->shapes[s];
        if (E->bodies[sh->body].type == BODY_STATIC) continue;
        shape_cache_bb(sh, &E->bodies[sh->body]);
    }
    /* broadphase + narrowphase: every pair whose BBs intersect */
    if (E->P.brute_force) {
        for (int sa = 0; sa < E->ns; sa++)
            for (int sb = sa + 1; sb < E->ns; sb++) collide_pair(E, sa, sb);
    } else {
        /* sweep along y: insertion-sort shapes by bb bottom, scan forward while bottoms <= top */
        int *o = E->order;
        for (int i = 1; i < E->ns; i++) {
            int s = o[i]; double key = E->shapes[s].bb; int j = i - 1;
            while (j >= 0 && E->shapes[o[j]].bb > key) { o[j + 1] = o[j]; j--; }
            o[j + 1] = s;
        }
        for (int i = 0; i < E->ns; i++) {
            const shape_t *si = &E->shapes[o[i]];
            for (int j = i + 1; j < E->ns && E->shapes[o[j]].bb <= si->bt; j++) {
                int sa = o[i], sb = o[j];
                if (sa > sb) { int t = sa; sa = sb; sb = t; }
                collide_pair(E, sa, sb);
            }
        }
        if (E->solve_order == 2) {
            if (E->nactive > E->capdisc) { E->capdisc = E->nactive * 2; E->disc = (int *)realloc(E->disc, (size_t)E->capdisc * sizeof(int)); }
            memcpy(E->disc, E->active, (size_t)E->nactive * sizeof(int));
        }
        qsort(E->active, (size_t)E->nactive, sizeof(int), cmp_u32);
    }
    if (E->kind == 2 && E->nevents) bd_run_presolve(E);
    /* cpSpaceArbiterSetFilter over the cached arbiter set */
    {
        int w = 0;
        for (int k = 0; k < E->narb; k++) {
            arb_t *arb = &E->arbs[k];
            long ticks = E->stamp - arb->stamp;
            if (ticks >= 1 && arb->state != ARB_CACHED) arb->state = ARB_CACHED;
            if (ticks >= E->P.persistence) continue; /* dropped */
            if (w != k) E->arbs[w] = *arb;
            w++;
        }
        E->narb = w;
    }
    /* Solve order.  Chipmunk sweeps its arbiter array in BB-tree/hash order; any fixed order is a valid Gauss-Seidel
     * sweep.  Here: greedy colouring in ascending key order (an arbiter takes the smallest colour not yet used by
     * either of its finite-mass bodies), then ascending (colour, key).  Arbiters of one colour share no dynamic body,
     * which is what lets a GPU run a colour in parallel with a bit-identical result. */
    if (E->nactive > E->capsolve) {
        E->capsolve = E->nactive * 2;
        E->solve = (int *)realloc(E->solve, (size_t)E->capsolve * sizeof(int));
        E->color = (int *)realloc(E->color, (size_t)E->capsolve * sizeof(int));
    }
    {
        if (!E->used) E->used = (uint32_t *)calloc((size_t)E->nb + 1, sizeof(uint32_t));
        for (int k = 0; k < E->nactive; k++) {
            int pos; arb_find(E, (uint32_t)E->active[k], &pos);
            E->used[E->shapes[E->arbs[pos].sa].body] = 0; E->used[E->shapes[E->arbs[pos].sb].body] = 0;
        }
        int ncol = 0;
        for (int k = 0; k < E->nactive; k++) {
            int pos; arb_find(E, (uint32_t)E->active[k], &pos);
            int ba = E->shapes[E->arbs[pos].sa].body, bb = E->shapes[E->arbs[pos].sb].body;
            uint32_t m = 0;
            if (E->bodies[ba].m_inv != 0.0) m |= E->used[ba];
            if (E->bodies[bb].m_inv != 0.0) m |= E->used[bb];
            int c = 0;
            while (m & (1u << c)) c++;
            E->color[k] = c;
            if (E->bodies[ba].m_inv != 0.0) E->used[ba] |= 1u << c;
            if (E->bodies[bb].m_inv != 0.0) E->used[bb] |= 1u << c;
            if (c + 1 > ncol) ncol = c + 1;
        }
        int w = 0;
        for (int c = 0; c < ncol; c++)
            for (int k = 0; k < E->nactive; k++) if (E->color[k] == c) E->solve[w++] = E->active[k];
        if (ncol > E->stat_ncol_max) E->stat_ncol_max = ncol;
        /* Alternative Gauss-Seidel sweeps for the order-sensitivity envelope (tests/test_order_envelope.py): what
         * Chipmunk's own order -- a by-product of its BB-tree and hash set, not recoverable here -- could do to the results. */
        if (E->solve_order == 1) memcpy(E->solve, E->active, (size_t)E->nactive * sizeof(int));
        else if (E->solve_order == 2 && !E->P.brute_force) memcpy(E->solve, E->disc, (size_t)E->nactive * sizeof(int));
        else if (E->solve_order == 4) { for (int k = 0; k < E->nactive; k++) E->solve[k] = E->active[E->nactive - 1 - k]; }
        else if (E->solve_order == 3) {
            memcpy(E->solve, E->active, (size_t)E->nactive * sizeof(int));
            uint64_t st = E->order_seed ^ ((uint64_t)E->stamp * 0x9E3779B97F4A7C15ull);
            for (int k = E->nactive - 1; k > 0; k--) {
                st += 0x9E3779B97F4A7C15ull;
                uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                int j = (int)(z % (uint64_t)(k + 1));
                int t = E->solve[k]; E->solve[k] = E->solve[j]; E->solve[j] = t;
            }
        }
    }
    /* prestep */
    for (int k = 0; k < E->nactive; k++) {
        int pos; arb_find(E, (uint32_t)E->active[k], &pos);
        arb_prestep(E, &E->arbs[pos], dt);
    }
    /* integrate velocities (cpBodyUpdateVelocity; gravity 0, no forces) */
    for (int i = 0; i < E->nb; i++) {
        body_t *b = &E->bodies[i];
        if (b->type != BODY_DYNAMIC) continue;
        b->v = vadd(vmult(b->v, E->P.damping_pow), vmult(vadd(V(0, 0), vmult(V(0, 0), b->m_inv)), dt));
        b->w = b->w * E->P.damping_pow + 0.0 * b->i_inv * dt;
    }
    /* warm start */
    double dt_coef = (prev_dt == 0.0 ? 0.0 : dt / prev_dt);
    for (int k = 0; k < E->nactive; k++) {
        int pos; arb_find(E, (uint32_t)E->solve[k], &pos);
        arb_apply_cached(E, &E->arbs[pos], dt_coef);
    }
    /* solver */
    for (int it = 0; it < E->P.iterations; it++)
        for (int k = 0; k < E->nactive; k++) {
            int pos; arb_find(E, (uint32_t)E->solve[k], &pos);
            arb_apply_impulse(E, &E->arbs[pos]);
        }
    /* post-solve callbacks: ship(type 1) x floe(type 2) bookkeeping, ship_ice_env.py:155-173 */
    long hot = 0;
    for (int k = 0; k < E->nactive; k++) {
        int pos; arb_find(E, (uint32_t)E->active[k], &pos);
        arb_t *arb = &E->arbs[pos];
        int any = 0;
        for (int i = 0; i < arb->count; i++) if (arb->con[i].jnAcc != 0.0 || arb->con[i].jtAcc != 0.0 || arb->con[i].jBias != 0.0) any = 1;
        hot += any;
        if (E->shapes[arb->sa].ctype == 1 && E->shapes[arb->sb].ctype == 2) {
            double eCoef = (1 - arb->e) / (1 + arb->e);
            double sum = 0.0; vec jsum = V(0, 0);
            for (int i = 0; i < arb->count; i++) {
                contact_t *c = &arb->con[i];
                sum += eCoef * c->jnAcc * c->jnAcc / c->nMass + c->jtAcc * c->jtAcc / c->tMass;
                jsum = vadd(jsum, vrotate(arb->n, V(c->jnAcc, c->jtAcc)));
            }
            E->total_ke += sum;
            E->total_impulse += vlength(jsum);
            E->n_post_solve++;
            E->n_contact_pts += arb->count;
            if (arb->state == ARB_FIRST) E->n_first_contact++;
        }
    }
    E->stat_hot_sum += hot;
    E->stat_arb_sum += E->nactive;
    if (E->nactive > E->stat_arb_max) E->stat_arb_max = E->nactive;
    E->stat_substeps++;
}

/* ------------------------------------------------------------------------------------------- */
/* numpy-only reference pieces, restated with sequential sums */
static double poly_area_np(int n, const double *xy) /* geometry/polygon.py:25-29 */
{
    double d1 = 0.0, d2 = 0.0;
    for (int i = 0; i < n; i++) {
        int p = (i - 1 + n) % n;
        d1 += xy[2 * i] * xy[2 * p + 1];
        d2 += xy[2 * i + 1] * xy[2 * p];
    }
    return 0.5 * fabs(d1 - d2);
}
static void poly_centroid_np(int n, const double *xy, double *cx, double *cy) /* polygon.py:32-41 */
{
    double A = poly_area_np(n, xy);
    double sx = 0.0, sy = 0.0;
    for (int i = 0; i < n; i++) {
        int p = (i - 1 + n) % n;
        double u = xy[2 * i] * xy[2 * p + 1] - xy[2 * p] * xy[2 * i + 1];
        sx += (xy[2 * i] + xy[2 * p]) * u;
        sy += (xy[2 * i + 1] + xy[2 * p + 1]) * u;
    }
    double f = 1.0 / (6.0 * A);
    *cx = fabs(f * sx);
    *cy = fabs(f * sy);
}

/* ------------------------------------------------------------------------------------------- */
orc_env *orc_create(const orc_params *P)
{
    orc_env *E = (orc_env *)calloc(1, sizeof(orc_env));
    E->P = *P;
    return E;
}
void orc_destroy(orc_env *E)
{
    if (!E) return;
    free(E->bodies); free(E->shapes); free(E->arbs); free(E->active); free(E->order); free(E->prev_wv);
    free(E->solve); free(E->color); free(E->used); free(E->disc); free(E->dist_map); free(E->wall_map); free(E->dist_raw);
    free(E->ras_occ); free(E->ras_foot); free(E->ras_orient); free(E->removed); free(E->events);
    free(E);
}

static void snapshot_world(orc_env *E)
{
    for (int s = 1; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double *dst = E->prev_wv + (size_t)(s - 1) * ORC_MAXV * 2;
        for (int i = 0; i < sh->n; i++) { dst[2 * i] = sh->wv[i].x; dst[2 * i + 1] = sh->wv[i].y; }
    }
}

/* reset(): new space + bodies from the trial, 1000 settle sub-steps (ship_ice_env.py:186-249).
 * floe_verts: raw polygon vertices as stored in the trial ('vertices'), concatenated; counts[nf];
 * centres[nf][2] ('centre'); ship_verts[nsv][2] (cfg.ship.vertices); start = (x, y, theta).
 * Returns the number of floes kept (zero-area floes are dropped, ship_ice_env.py:206). */
int orc_reset(orc_env *E, int nf, const double *floe_verts, const int *counts, const double *centres,
              int nsv, const double *ship_verts, const double *head, const double *tail, const double *start)
{
    free(E->bodies); free(E->shapes); free(E->order); free(E->prev_wv); free(E->used); E->used = NULL;
    E->bodies = (body_t *)calloc((size_t)nf + 1, sizeof(body_t));
    E->shapes = (shape_t *)calloc((size_t)nf + 1, sizeof(shape_t));
    E->order = (int *)calloc((size_t)nf + 1, sizeof(int));
    E->prev_wv = (double *)calloc((size_t)(nf > 0 ? nf : 1) * ORC_MAXV * 2, sizeof(double));
    E->narb = 0; E->nactive = 0; E->stamp = 0; E->curr_dt = 0.0;
    E->total_work = 0.0; E->total_ke = 0.0; E->total_impulse = 0.0;
    E->n_post_solve = E->n_contact_pts = E->n_first_contact = 0;
    E->ship_nv = nsv;
    for (int i = 0; i < nsv; i++) E->ship_verts[i] = V(ship_verts[2 * i], ship_verts[2 * i + 1]);
    E->head = V(head[0], head[1]); E->tail = V(tail[0], tail[1]);

    /* ship: Ship.sim (ship.py:77-98): KINEMATIC body, Poly(radius) = convex hull of the listed verts */
    {
        body_t *b = &E->bodies[0];
        b->type = BODY_KINEMATIC; b->m = b->i = INFINITY; b->m_inv = b->i_inv = 0.0;
        b->p = V(start[0], start[1]); b->a = start[2]; b->cog = V(0, 0);
        vec tmp[ORC_MAXV + 8], hull[ORC_MAXV + 8];
        for (int i = 0; i < nsv; i++) tmp[i] = E->ship_verts[i];
        int hn = convex_hull(nsv, tmp, hull);
        shape_t *s = &E->shapes[0];
        s->body = 0; s->r = E->P.poly_radius; s->e = E->P.elasticity; s->u = E->P.friction; s->ctype = 1;
        shape_set_verts(s, hn, hull);
        body_set_transform(b);
    }
    /* floes: generate_sim_obs -> create_polygon (sim_utils.py:136-163) */
    int kept = 0, off = 0;
    for (int f = 0; f < nf; f++) {
        int n = counts[f];
        const double *raw = floe_verts + 2 * (size_t)off;
        off += n;
        if (poly_area_np(n, raw) == 0.0) continue; /* ship_ice_env.py:206 */
        vec tmp[ORC_MAXV + 8], hull[ORC_MAXV + 8];
        vec centre = V(centres[2 * f], centres[2 * f + 1]);
        for (int i = 0; i < n; i++) tmp[i] = V(raw[2 * i] - centre.x, raw[2 * i + 1] - centre.y);
        int hn = convex_hull(n, tmp, hull);           /* dummy_shape = pymunk.Poly(None, vertices) */
        vec cog = centroid_for_poly(hn, hull);         /* dummy_shape.center_of_gravity */
        for (int i = 0; i < n; i++) tmp[i] = V(tmp[i].x - cog.x, tmp[i].y - cog.y);
        hn = convex_hull(n, tmp, hull);                /* pymunk.Poly(body, vs, radius=0.02) */
        int bi = kept + 1;
        shape_t *s = &E->shapes[bi];
        body_t *b = &E->bodies[bi];
        s->body = bi; s->r = E->P.poly_radius; s->e = E->P.elasticity; s->u = E->P.friction; s->ctype = 2;
        shape_set_verts(s, hn, hull);
        /* shape.density -> cpShapeSetMass(density*area) -> cpBodyAccumulateMassFromShapes */
        vec scog = centroid_for_poly(hn, hull);
        double area = area_for_poly(hn, hull, s->r);
        double m = E->P.density * area;
        double i_per_m = moment_for_poly(1.0, hn, hull, vneg(scog));
        b->type = BODY_DYNAMIC;
        {
            double bm = 0.0, bI = 0.0; vec bc = V(0, 0);
            double msum = bm + m;
            bI += m * i_per_m + vdot(vsub(bc, scog), vsub(bc, scog)) * (m * bm) / msum;
            bc = vlerp(bc, scog, m / msum);
            bm = msum;
            b->m = bm; b->i = bI; b->cog = bc;
        }
        b->m_inv = 1.0 / b->m; b->i_inv = 1.0 / b->i;
        /* body.position = (x, y) was set before the shape was attached; cpBodySetPosition keeps
         * the user position, so p (centre of gravity in world) = rot(cog) + position */
        b->a = 0.0;
        b->p = vadd(V(b->cog.x * 1.0 - b->cog.y * 0.0, b->cog.x * 0.0 + b->cog.y * 1.0), centre);
        body_set_transform(b);
        kept++;
    }
    E->nb = E->ns = kept + 1;
    for (int i = 0; i < E->ns; i++) E->order[i] = i;
    for (int s = 0; s < E->ns; s++) shape_cache_bb(&E->shapes[s], &E->bodies[E->shapes[s].body]);
    double dts = E->P.dt / E->P.steps;
    for (int k = 0; k < E->P.settle_steps; k++) space_step(E, dts);
    snapshot_world(E);
    return kept;
}

/* info scalars written by orc_step / orc_get_state */
enum { ORC_I_X = 0, ORC_I_Y, ORC_I_THETA, ORC_I_TOTAL_WORK, ORC_I_WORK, ORC_I_COLL_REWARD, ORC_I_SCALED_COLL, ORC_I_DIST_REWARD,
       ORC_I_SUCCESS, ORC_I_BOUNDARY, ORC_I_YAW, ORC_I_KE, ORC_I_IMPULSE, ORC_I_NPOST, ORC_I_NCONTACT, ORC_I_NFIRST, ORC_I_COUNT };

void orc_observe(orc_env *E, uint8_t *obs);

/* ShipIceEnv.step (ship_ice_env.py:261-355). action is the raw policy action in [-1, 1]. */
void orc_step(orc_env *E, double action, uint8_t *obs, double *reward, int *terminated, double *info)
{
    const orc_params *P = &E->P;
    body_t *ship = &E->bodies[0];
    double act = action * P->max_yaw_rate;
    /* global_velocity = R(angle) @ [speed, 0] */
    double sn, cs;
    bp_sincos(ship->a, &sn, &cs);
    ship->w = act;
    ship->v = V(cs * P->target_speed + -sn * 0.0, sn * P->target_speed + cs * 0.0);
    int yaw_violated = 0, boundary_violated = 0, boundary_terminal = 0;
    double dts = P->dt / P->steps;
    for (int k = 0; k < P->steps; k++) {
        space_step(E, dts);
        if (ship->a <= 0.0 || ship->a >= M_PI) { ship->w = 0.0; yaw_violated = 1; }
        if (ship->p.x < 0.0 || ship->p.x > P->map_w) boundary_violated = 1;
    }
    if (ship->p.x < 0.0 && fabs(ship->p.x - 0.0) >= 0.0) boundary_terminal = 1;
    if (ship->p.x > P->map_w && fabs(ship->p.x - P->map_w) >= 0.0) boundary_terminal = 1;
    /* work (metrics.py:96-113) on world vertices before/after */
    double work = 0.0;
    for (int s = 1; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double now[ORC_MAXV * 2];
        for (int i = 0; i < sh->n; i++) { now[2 * i] = sh->wv[i].x; now[2 * i + 1] = sh->wv[i].y; }
        const double *prev = E->prev_wv + (size_t)(s - 1) * ORC_MAXV * 2;
        double area = poly_area_np(sh->n, prev);
        double ax, ay, bx, by;
        poly_centroid_np(sh->n, prev, &ax, &ay);
        poly_centroid_np(sh->n, now, &bx, &by);
        double d = sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
        work += d * area;
    }
    E->total_work += work;
    snapshot_world(E);
    int term = 0;
    if (ship->p.y >= P->goal_y) term = 1;
    else if (boundary_terminal) term = 1;
    double dist_reward = 0.0;
    if (ship->p.y < P->goal_y) {
        bp_sincos(ship->a, &sn, &cs);
        dist_reward = 1.0 * (cs * 0.0 + sn * 1.0);
    }
    double coll = -work;
    double r = P->beta * coll + dist_reward;
    if (yaw_violated) r += 0.0;
    if (boundary_violated) r += P->boundary_penalty;
    int success = 0;
    if (term && !boundary_terminal) { r += P->terminal_reward; success = 1; }
    *reward = r; *terminated = term;
    if (info) {
        info[ORC_I_X] = ship->p.x; info[ORC_I_Y] = ship->p.y; info[ORC_I_THETA] = ship->a;
        info[ORC_I_TOTAL_WORK] = E->total_work; info[ORC_I_WORK] = work; info[ORC_I_COLL_REWARD] = coll;
        info[ORC_I_SCALED_COLL] = coll * P->beta; info[ORC_I_DIST_REWARD] = dist_reward;
        info[ORC_I_SUCCESS] = success; info[ORC_I_BOUNDARY] = boundary_violated; info[ORC_I_YAW] = yaw_violated;
        info[ORC_I_KE] = E->total_ke; info[ORC_I_IMPULSE] = E->total_impulse;
        info[ORC_I_NPOST] = (double)E->n_post_solve; info[ORC_I_NCONTACT] = (double)E->n_contact_pts; info[ORC_I_NFIRST] = (double)E->n_first_contact;
    }
    if (obs) orc_observe(E, obs);
}

int orc_num_shapes(const orc_env *E) { return E->ns; }
void orc_set_solve_order(orc_env *E, int mode, uint64_t seed) { E->solve_order = mode; E->order_seed = seed; }
/* body state: [nb][9] = x, y, a, vx, vy, w, vbx, vby, wb */
void orc_get_bodies(const orc_env *E, double *out)
{
    for (int i = 0; i < E->nb; i++) {
        const body_t *b = &E->bodies[i];
        double *o = out + 9 * (size_t)i;
        o[0] = b->p.x; o[1] = b->p.y; o[2] = b->a; o[3] = b->v.x; o[4] = b->v.y; o[5] = b->w; o[6] = b->vb.x; o[7] = b->vb.y; o[8] = b->wb;
    }
}
/* mass data: [nb][5] = m_inv, i_inv, cog.x, cog.y, nverts */
void orc_get_mass(const orc_env *E, double *out)
{
    for (int i = 0; i < E->nb; i++) {
        const body_t *b = &E->bodies[i];
        double *o = out + 5 * (size_t)i;
        o[0] = b->m_inv; o[1] = b->i_inv; o[2] = b->cog.x; o[3] = b->cog.y; o[4] = E->shapes[i].n;
    }
}
/* world polygons (info['obs'], cost_map.py:275-281) incl. ship at index 0: [ns][ORC_MAXV][2], counts[ns] */
void orc_get_world_polys(const orc_env *E, double *out, int *counts)
{
    for (int s = 0; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        counts[s] = sh->n;
        for (int i = 0; i < sh->n; i++) {
            out[((size_t)s * ORC_MAXV + i) * 2] = sh->wv[i].x;
            out[((size_t)s * ORC_MAXV + i) * 2 + 1] = sh->wv[i].y;
        }
    }
}
void orc_get_local_polys(const orc_env *E, double *verts, double *normals)
{
    for (int s = 0; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        for (int i = 0; i < sh->n; i++) {
            verts[((size_t)s * ORC_MAXV + i) * 2] = sh->lv[i].x; verts[((size_t)s * ORC_MAXV + i) * 2 + 1] = sh->lv[i].y;
            normals[((size_t)s * ORC_MAXV + i) * 2] = sh->ln[i].x; normals[((size_t)s * ORC_MAXV + i) * 2 + 1] = sh->ln[i].y;
        }
    }
}
void orc_get_stats(const orc_env *E, long *out)
{
    out[0] = E->stat_substeps; out[1] = E->stat_pairs_bb; out[2] = E->stat_narrow; out[3] = E->stat_arb_sum;
    out[4] = E->stat_arb_max; out[5] = E->stat_moving_sum; out[6] = E->stat_hot_sum; out[7] = E->stat_ncol_max;
}
void orc_get_info(const orc_env *E, double *info)
{
    const body_t *ship = &E->bodies[0];
    memset(info, 0, sizeof(double) * ORC_I_COUNT);
    info[ORC_I_X] = ship->p.x; info[ORC_I_Y] = ship->p.y; info[ORC_I_THETA] = ship->a; info[ORC_I_TOTAL_WORK] = E->total_work;
    info[ORC_I_KE] = E->total_ke; info[ORC_I_IMPULSE] = E->total_impulse;
    info[ORC_I_NPOST] = (double)E->n_post_solve; info[ORC_I_NCONTACT] = (double)E->n_contact_pts; info[ORC_I_NFIRST] = (double)E->n_first_contact;
}
/* exposed for unit tests */
void orc_sincos(double x, double *s, double *c) { bp_sincos(x, s, c); }
double orc_poly_area(int n, const double *xy) { return poly_area_np(n, xy); }
void orc_poly_centroid(int n, const double *xy, double *c) { poly_centroid_np(n, xy, &c[0], &c[1]); }
int orc_convex_hull(int n, const double *xy, double *out)
{
    vec tmp[64], hull[64];
    for (int i = 0; i < n; i++) tmp[i] = V(xy[2 * i], xy[2 * i + 1]);
    int hn = convex_hull(n, tmp, hull);
    for (int i = 0; i < hn; i++) { out[2 * i] = hull[i].x; out[2 * i + 1] = hull[i].y; }
    return hn;
}
/* narrow phase on two explicit convex polygons (CCW world verts): returns count, fills n[2], p1[4], p2[4], hash[2] */
int orc_collide(int na, const double *a, double ra, int nb, const double *b, double rb, double *n, double *p1, double *p2, unsigned *hash)
{
    shape_t A, B;
    memset(&A, 0, sizeof(A)); memset(&B, 0, sizeof(B));
    vec va[ORC_MAXV], vb[ORC_MAXV];
    for (int i = 0; i < na; i++) va[i] = V(a[2 * i], a[2 * i + 1]);
    for (int i = 0; i < nb; i++) vb[i] = V(b[2 * i], b[2 * i + 1]);
    shape_set_verts(&A, na, va); shape_set_verts(&B, nb, vb);
    A.r = ra; B.r = rb;
    for (int i = 0; i < na; i++) { A.wv[i] = A.lv[i]; A.wn[i] = A.ln[i]; }
    for (int i = 0; i < nb; i++) { B.wv[i] = B.lv[i]; B.wn[i] = B.ln[i]; }
    manifold_t m;
    collide_poly_poly(&A, &B, &m);
    n[0] = m.n.x; n[1] = m.n.y;
    for (int i = 0; i < m.count; i++) {
        p1[2 * i] = m.p1[i].x; p1[2 * i + 1] = m.p1[i].y; p2[2 * i] = m.p2[i].x; p2[2 * i + 1] = m.p2[i].y; hash[i] = m.hash[i];
    }
    return m.count;
}

/* ------------------------------------------------------------------------------------------- */
/* Observation raster (ship_ice_env.py:378-409)                                                 */

/* skimage._shared.geometry.point_in_polygon (unpinned third-party; restated): 0 outside, else in/edge/vertex */
static int point_in_polygon(int n, const double *xp, const double *yp, double x, double y)
{
    const double eps = 1e-12;
    unsigned l_cross = 0, r_cross = 0;
    double x1 = xp[n - 1] - x, y1 = yp[n - 1] - y;
    for (int i = 0; i < n; i++) {
        double x0 = xp[i] - x, y0 = yp[i] - y;
        if ((-eps < x0 && x0 < eps) && (-eps < y0 && y0 < eps)) return 2;
        if ((y0 > 0) != (y1 > 0)) {
            if (((x0 * y1 - x1 * y0) / (y1 - y0)) > 0) r_cross++;
        }
        if ((y0 < 0) != (y1 < 0)) {
            if (((x0 * y1 - x1 * y0) / (y1 - y0)) < 0) l_cross++;
        }
        x1 = x0; y1 = y0;
    }
    if ((r_cross & 1) != (l_cross & 1)) return 3;
    if (r_cross & 1) return 1;
    return 0;
}
/* skimage.draw.polygon(r, c, shape): fill 'val' into img[H][W] */
static void draw_polygon(int n, const double *r, const double *c, int H, int W, int clip, double *img, double val)
{
    double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
    for (int i = 1; i < n; i++) {
        rmin = fmin(rmin, r[i]); rmax = fmax(rmax, r[i]); cmin = fmin(cmin, c[i]); cmax = fmax(cmax, c[i]);
    }
    long minr = (long)fmax(0.0, rmin), maxr = (long)ceil(rmax);
    long minc = (long)fmax(0.0, cmin), maxc = (long)ceil(cmax);
    if (clip) { if (maxr > H - 1) maxr = H - 1; if (maxc > W - 1) maxc = W - 1; }
    for (long ri = minr; ri <= maxr; ri++)
        for (long ci = minc; ci <= maxc; ci++)
            if (point_in_polygon(n, c, r, (double)ci, (double)ri)) {
                if (ri >= 0 && ri < H && ci >= 0 && ci < W) img[ri * W + ci] = val;
            }
}
/* cv2.clipLine (imgproc drawing.cpp; unpinned third-party, restated) */
static int clip_line(long W, long H, long *px1, long *py1, long *px2, long *py2)
{
    long x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2;
    long right = W - 1, bottom = H - 1;
    if (W <= 0 || H <= 0) return 0;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
    return (c1 | c2) == 0;
}
/* cv2.line(img, pt1, pt2, color, thickness=1) -> 8-connected LineIterator, leftToRight (restated) */
static void cv_line(double *img, int H, int W, long x1, long y1, long x2, long y2, double val)
{
    if ((unsigned long)x1 >= (unsigned long)W || (unsigned long)x2 >= (unsigned long)W ||
        (unsigned long)y1 >= (unsigned long)H || (unsigned long)y2 >= (unsigned long)H) {
        if (!clip_line(W, H, &x1, &y1, &x2, &y2)) return;
    }
    long dx = x2 - x1, dy = y2 - y1;
    long delta_x = 1, delta_y = 1;
    long px = x1, py = y1;
    if (dx < 0) { dx = -dx; dy = -dy; px = x2; py = y2; }
    if (dy < 0) { dy = -dy; delta_y = -1; }
    int vert = dy > dx;
    if (vert) { long t = dx; dx = dy; dy = t; t = delta_x; delta_x = delta_y; delta_y = t; }
    long err = dx - (dy + dy);
    long plusDelta = dx + dx, minusDelta = -(dy + dy);
    long minusShift = delta_x, plusShift = 0, minusStep = 0, plusStep = delta_y;
    long count = dx + 1;
    if (vert) { long t = plusStep; plusStep = plusShift; plusShift = t; t = minusStep; minusStep = minusShift; minusShift = t; }
    for (long i = 0; i < count; i++) {
        if (px >= 0 && px < W && py >= 0 && py < H) img[py * W + px] = val;
        long mask = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & mask);
        py += minusStep + (plusStep & mask);
        px += minusShift + (plusShift & mask);
    }
}
static long to_u16(double v) /* numpy .astype(np.uint16) on x86-64: truncate, wrap mod 2^16 */
{
    long long t = (long long)v;
    return (long)(t & 0xFFFF);
}

void orc_observe(orc_env *E, uint8_t *obs)
{
    const orc_params *P = &E->P;
    const body_t *ship = &E->bodies[0];
    /* OccupancyGrid.__init__ (occupancy_map.py:11-35) with grid = 1/m_to_pix */
    double grid = 1.0 / P->m_to_pix;
    int Wg = (int)(P->map_w / grid), Hg = (int)(P->map_h / grid);
    int LH = (int)(P->local_h * P->m_to_pix), LW = (int)(P->local_w * P->m_to_pix);
    int bw = (int)(P->map_w * P->m_to_pix), bh = (int)(P->map_h * P->m_to_pix); /* ice_binary_w/h */
    size_t npx = (size_t)Hg * Wg;
    /* persistent per-env raster buffers (the reference allocates fresh numpy arrays; values are identical) */
    if (E->ras_n < npx || (size_t)bh * bw > E->ras_n) {
        size_t need = npx > (size_t)bh * bw ? npx : (size_t)bh * bw;
        free(E->ras_occ); free(E->ras_foot); free(E->ras_orient);
        E->ras_occ = (double *)malloc(need * sizeof(double)); E->ras_foot = (double *)malloc(need * sizeof(double));
        E->ras_orient = (double *)malloc(need * sizeof(double)); E->ras_n = need;
    }
    double *occ = E->ras_occ, *foot = E->ras_foot, *orient = E->ras_orient;
    memset(occ, 0, (size_t)bh * bw * sizeof(double));
    memset(orient, 0, npx * sizeof(double));
    double sx = ship->p.x, sy = ship->p.y, sa = ship->a;
    /* compute_occ_img (occupancy_map.py:37-65) */
    for (int s = 1; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double xy[ORC_MAXV * 2], r[ORC_MAXV], c[ORC_MAXV];
        for (int i = 0; i < sh->n; i++) { xy[2 * i] = sh->wv[i].x; xy[2 * i + 1] = sh->wv[i].y; }
        double cx, cy;
        poly_centroid_np(sh->n, xy, &cx, &cy);
        cx = fabs(cx); cy = fabs(cy);
        if (fabs(sx - cx) > P->obs_range || fabs(sy - cy) > P->obs_range) continue;
        for (int i = 0; i < sh->n; i++) { c[i] = xy[2 * i] * P->m_to_pix; r[i] = xy[2 * i + 1] * P->m_to_pix; }
        draw_polygon(sh->n, r, c, bh, bw, 1, occ, 1.0);
    }
    /* block_reduce with block (int(grid*m_to_pix), ...) = (1,1) mean: identity (occupancy_map.py:97-109) */
    /* _compute_global_footprint (occupancy_map.py:300-337) */
    double m2gx = (double)Wg / P->map_w, m2gy = (double)Hg / P->map_h;
    for (size_t i = 0; i < npx; i++) foot[i] = 0.5;
    {
        double ch, shh;
        bp_sincos(sa, &shh, &ch);
        double r[ORC_MAXV + 8], c[ORC_MAXV + 8]; int cnt = 0;
        for (int i = 0; i < E->ship_nv; i++) {
            double vx = E->ship_verts[i].x * ch + E->ship_verts[i].y * -shh + sx;
            double vy = E->ship_verts[i].x * shh + E->ship_verts[i].y * ch + sy;
            double gx = vx * m2gx, gy = vy * m2gy;
            if (gy < 0 || gy >= Hg || gx < 0 || gx >= Wg) continue;
            r[cnt] = gy; c[cnt] = gx; cnt++;
        }
        if (cnt > 0) draw_polygon(cnt, r, c, Hg, Wg, 0, foot, 1.0);
        /* global_orientation_map (occupancy_map.py:524-554) */
        double hx = E->head.x * ch + E->head.y * -shh + sx, hy = E->head.x * shh + E->head.y * ch + sy;
        double tx = E->tail.x * ch + E->tail.y * -shh + sx, ty = E->tail.x * shh + E->tail.y * ch + sy;
        long hpx = to_u16(hx * m2gx), hpy = to_u16(hy * m2gy);
        long tpx = to_u16(tx * m2gx), tpy = to_u16(ty * m2gy);
        cv_line(orient, Hg, Wg, hpx, hpy, tpx, tpy, 0.5);
        long cxp = hpx < 0 ? 0 : (hpx > Wg - 1 ? Wg - 1 : hpx);
        long cyp = hpy < 0 ? 0 : (hpy > Hg - 1 ? Hg - 1 : hpy);
        orient[cyp * Wg + cxp] = 1.0;
    }
    /* ego crops (occupancy_map.py:112-140,379-410,492-521,557-587) */
    int wx = (int)(sx * m2gx);
    int wy = (int)((sy + P->vshift) * m2gy);
    double g2m = P->map_h / (double)Hg;
    for (int li = 0; li < LH; li++)
        for (int lj = 0; lj < LW; lj++) {
            int gi = (int)((double)(li + wy) - ((double)LH / 2));
            int gj = (int)((double)(lj + wx) - ((double)LW / 2));
            int inb = !(gi < 0 || gi >= Hg || gj < 0 || gj >= Wg);
            double f = 0.0, e = 1.0, o = 0.0, oc = 0.0;
            if (inb) {
                f = foot[(size_t)gi * Wg + gj];
                double d = P->goal_y - gi * g2m;
                if (d < 0) d = 0;
                e = d / P->goal_y;
                o = orient[(size_t)gi * Wg + gj];
                oc = occ[(size_t)gi * bw + gj];
            }
            size_t px = (size_t)li * LW + lj, pl = (size_t)LH * LW;
            obs[0 * pl + px] = (uint8_t)(f * 255);
            obs[1 * pl + px] = (uint8_t)(e * 255);
            obs[2 * pl + px] = (uint8_t)(o * 255);
            obs[3 * pl + px] = (uint8_t)(oc * 255);
        }
}

/* ---- test hooks for the numpy-only golden vectors (tests/golden/) ------------------------------------- */
/* total_work_done (evaluation/metrics.py:96-113) on explicit polygon lists; verts concatenated [sum(counts)][2] */
double orc_total_work(int npoly, const int *counts, const double *a, const double *b)
{
    double work = 0.0;
    int off = 0;
    for (int p = 0; p < npoly; p++) {
        int n = counts[p];
        double area = poly_area_np(n, a + 2 * off);
        double ax, ay, bx, by;
        poly_centroid_np(n, a + 2 * off, &ax, &ay);
        poly_centroid_np(n, b + 2 * off, &bx, &by);
        work += sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) * area;
        off += n;
    }
    return work;
}
/* ego crop of an injected global map gmap[Hg][Wg] (occupancy_map.py:112-140 index arithmetic), oob_val outside */
void orc_crop(const orc_params *P, const double *state, const double *gmap, double oob_val, uint8_t *out)
{
    double grid = 1.0 / P->m_to_pix;
    int Wg = (int)(P->map_w / grid), Hg = (int)(P->map_h / grid);
    int LH = (int)(P->local_h * P->m_to_pix), LW = (int)(P->local_w * P->m_to_pix);
    double m2gx = (double)Wg / P->map_w, m2gy = (double)Hg / P->map_h;
    int wx = (int)(state[0] * m2gx);
    int wy = (int)((state[1] + P->vshift) * m2gy);
    for (int li = 0; li < LH; li++)
        for (int lj = 0; lj < LW; lj++) {
            int gi = (int)((double)(li + wy) - ((double)LH / 2));
            int gj = (int)((double)(lj + wx) - ((double)LW / 2));
            double v = oob_val;
            if (!(gi < 0 || gi >= Hg || gj < 0 || gj >= Wg)) v = gmap[(size_t)gi * Wg + gj];
            out[(size_t)li * LW + lj] = (uint8_t)(v * 255);
        }
}

/* ---- CPU baseline driver (bench.py cpu_baseline leg): nenv oracle envs stepped nsteps times on nthreads cores ----- */
/* Trials in the packed layout of benchpush_amd.scenario.pack_trials.  Env e plays trial e % T with actions from a
 * 64-bit LCG in [-1, 1).  Only the env.step() loop is timed (resets, incl. their 1000 settle sub-steps, are not).
 * Returns wall seconds of the step loop; *out_steps = env-steps executed. */
double orc_bench(const orc_params *P, int T, int F, int V, const double *verts, const int *counts, const double *centres,
                 const double *starts, const int *nfloes, int nsv, const double *ship_verts, const double *head,
                 const double *tail, int nenv, int nsteps, int nthreads, int with_obs, long *out_steps)
{
    orc_env **envs = (orc_env **)calloc((size_t)nenv, sizeof(orc_env *));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < nenv; e++) {
        int t = e % T;
        int nf = nfloes[t];
        double *fv = (double *)malloc(sizeof(double) * 2 * (size_t)V * (size_t)(nf > 0 ? nf : 1));
        int off = 0;
        for (int f = 0; f < nf; f++) {
            int n = counts[(size_t)t * F + f];
            memcpy(fv + 2 * (size_t)off, verts + ((size_t)t * F + f) * V * 2, sizeof(double) * 2 * (size_t)n);
            off += n;
        }
        envs[e] = orc_create(P);
        orc_reset(envs[e], nf, fv, counts + (size_t)t * F, centres + (size_t)t * F * 2, nsv, ship_verts, head, tail, starts + 3 * t);
        free(fv);
    }
    long total = 0;
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int e = 0; e < nenv; e++) {
        uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(e + 1);
        uint8_t *obs = with_obs ? (uint8_t *)malloc(4 * 150 * 150) : NULL;
        for (int k = 0; k < nsteps; k++) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            double a = (double)(float)(((double)(st >> 11) / 9007199254740992.0) * 2.0 - 1.0);
            double r; int term;
            orc_step(envs[e], a, obs, &r, &term, NULL);
            total++;
            if (term) break;
        }
        free(obs);
    }
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    for (int e = 0; e < nenv; e++) orc_destroy(envs[e]);
    free(envs);
    if (out_steps) *out_steps = total;
    return t1 - t0;
}

/* =============================================================================================================== */
/* maze-NAMO-v0 (benchpush/environments/maze_NAMO/maze_NAMO_env.py), on the same restated Chipmunk step.            */
/* Shapes: 0 = robot body (type 1), 1..4 = wheels (type 0), all on the KINEMATIC body 0 (robot.py:77-118);            */
/*         5..4+nbox = boxes (type 2, create_polygon); then one static Segment(radius 0.5) per wall (type 3,          */
/*         sim_utils.py:174-181), represented as a 2-vertex hull whose two planes are +-seg.n.                        */
/* =============================================================================================================== */
static void draw_polygon(int n, const double *r, const double *c, int H, int W, int clip, double *img, double val);

/* compute_occ_img_walls (occupancy_map.py:67-94) */
static void maze_wall_raster(orc_env *E, double *img, int H, int W)
{
    double m2p = (double)H / E->P.map_h;
    double wr = E->P.wall_radius;
    for (int w = 0; w < E->nwalls; w++) {
        vec a = V(E->walls[w][0], E->walls[w][1]), b = V(E->walls[w][2], E->walls[w][3]);
        vec d = vsub(b, a);
        double len = sqrt(d.x * d.x + d.y * d.y);
        vec u = V(d.x / len, d.y / len);
        vec p = V(-u.y, u.x);
        double vx[4], vy[4];
        vx[0] = (a.x + wr * p.x - wr * u.x); vy[0] = (a.y + wr * p.y - wr * u.y);
        vx[1] = (a.x - wr * p.x - wr * u.x); vy[1] = (a.y - wr * p.y - wr * u.y);
        vx[2] = (b.x - wr * p.x + wr * u.x); vy[2] = (b.y - wr * p.y + wr * u.y);
        vx[3] = (b.x + wr * p.x + wr * u.x); vy[3] = (b.y + wr * p.y + wr * u.y);
        double r[4], c[4];
        for (int i = 0; i < 4; i++) { c[i] = vx[i] * m2p; r[i] = vy[i] * m2p; }
        draw_polygon(4, r, c, H, W, 1, img, 1.0);
    }
}
/* global_goal_point_dist_transform (occupancy_map.py:435-485): 8-connected wavefront, +1 per hop, goal = 1 */
static void maze_goal_map(orc_env *E)
{
    int H = E->map_h, W = E->map_w;
    double g2m = E->P.map_w / (double)W;
    int gx = (int)(E->P.goal_x / g2m), gy = (int)(E->P.goal_y / g2m);
    double *edt = E->dist_raw;
    memset(edt, 0, sizeof(double) * (size_t)H * W);
    memset(E->wall_map, 0, sizeof(double) * (size_t)H * W);
    maze_wall_raster(E, E->wall_map, H, W);
    unsigned char *vis = (unsigned char *)calloc((size_t)H * W, 1);
    int *q = (int *)malloc(sizeof(int) * (size_t)H * W);
    int qh = 0, qt = 0;
    edt[gy * W + gx] = 1.0; vis[gy * W + gx] = 1; q[qt++] = gy * W + gx;
    static const int dy[8] = {0, 0, 1, -1, 1, 1, -1, -1}, dx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    double mx = 1.0;
    while (qh < qt) {
        int cur = q[qh++];
        int y = cur / W, x = cur % W;
        for (int k = 0; k < 8; k++) {
            int ny = y + dy[k], nx = x + dx[k];
            if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
            if (vis[ny * W + nx]) continue;
            if (E->wall_map[ny * W + nx] == 1.0) continue;
            vis[ny * W + nx] = 1;
            edt[ny * W + nx] = edt[cur] + 1;
            if (edt[ny * W + nx] > mx) mx = edt[ny * W + nx];
            q[qt++] = ny * W + nx;
        }
    }
    for (int i = 0; i < H * W; i++) {
        E->dist_map[i] = edt[i] / mx;
        if (E->wall_map[i] == 1.0) E->dist_map[i] = 1.0;
    }
    free(vis); free(q);
}

/* reset(): boxes given by centre (squares +-size), robot verts (8) + 4 wheels x 4 verts, walls [nw][4] = ax, ay, bx, by */
int orc_maze_reset(orc_env *E, int nbox, const double *centres, double size, const double *robot_verts, int nrv,
                   const double *wheel_verts, int nwheels, const double *walls, int nwalls, const double *start)
{
    free(E->bodies); free(E->shapes); free(E->order); free(E->prev_wv); free(E->used); E->used = NULL;
    int nb = 1 + nbox + nwalls, ns = 1 + nwheels + nbox + nwalls;
    E->bodies = (body_t *)calloc((size_t)nb, sizeof(body_t));
    E->shapes = (shape_t *)calloc((size_t)ns, sizeof(shape_t));
    E->order = (int *)calloc((size_t)ns, sizeof(int));
    E->prev_wv = (double *)calloc((size_t)ns * ORC_MAXV * 2, sizeof(double));
    E->narb = 0; E->nactive = 0; E->stamp = 0; E->curr_dt = 0.0;
    E->total_work = 0.0; E->total_ke = 0.0; E->total_impulse = 0.0;
    E->n_post_solve = E->n_contact_pts = E->n_first_contact = 0;
    E->wall_collision = 0; E->have_prev_dist = 0; E->prev_dist = 0.0;
    E->nwalls = nwalls;
    for (int w = 0; w < nwalls; w++) for (int k = 0; k < 4; k++) E->walls[w][k] = walls[4 * w + k];
    E->robot_nv = nrv;
    for (int i = 0; i < nrv; i++) E->robot_verts[i] = V(robot_verts[2 * i], robot_verts[2 * i + 1]);
    /* robot: KINEMATIC body with 5 Poly(radius 0.02) shapes; friction left at pymunk's default 0, elasticity 0.01 */
    {
        body_t *b = &E->bodies[0];
        b->type = BODY_KINEMATIC; b->m = b->i = INFINITY; b->m_inv = b->i_inv = 0.0;
        b->p = V(start[0], start[1]); b->a = start[2]; b->cog = V(0, 0);
        body_set_transform(b);
        for (int k = 0; k <= nwheels; k++) {
            vec tmp[ORC_MAXV], hull[ORC_MAXV];
            int n = (k == 0) ? nrv : 4;
            const double *src = (k == 0) ? robot_verts : wheel_verts + 8 * (k - 1);
            for (int i = 0; i < n; i++) tmp[i] = V(src[2 * i], src[2 * i + 1]);
            int hn = convex_hull(n, tmp, hull);
            shape_t *s = &E->shapes[k];
            s->body = 0; s->r = E->P.poly_radius; s->e = E->P.elasticity; s->u = 0.0; s->ctype = (k == 0) ? 1 : 0;
            shape_set_verts(s, hn, hull);
        }
    }
    /* boxes: generate_obstacles + generate_sim_obs (maze_NAMO_env.py:313-321, sim_utils.py:136-163) */
    for (int k = 0; k < nbox; k++) {
        double ox = centres[2 * k], oy = centres[2 * k + 1];
        double raw[8] = {ox + size, oy + size, ox - size, oy + size, ox - size, oy - size, ox + size, oy - size};
        vec tmp[4], hull[4];
        for (int i = 0; i < 4; i++) tmp[i] = V(raw[2 * i] - ox, raw[2 * i + 1] - oy);
        int hn = convex_hull(4, tmp, hull);
        vec cog = centroid_for_poly(hn, hull);
        for (int i = 0; i < 4; i++) tmp[i] = V(tmp[i].x - cog.x, tmp[i].y - cog.y);
        hn = convex_hull(4, tmp, hull);
        int bi = 1 + k, si = 1 + nwheels + k;
        shape_t *s = &E->shapes[si];
        body_t *b = &E->bodies[bi];
        s->body = bi; s->r = E->P.poly_radius; s->e = E->P.elasticity; s->u = E->P.friction; s->ctype = 2;
        shape_set_verts(s, hn, hull);
        vec scog = centroid_for_poly(hn, hull);
        double area = area_for_poly(hn, hull, s->r);
        double m = E->P.density * area;
        double ipm = moment_for_poly(1.0, hn, hull, vneg(scog));
        b->type = BODY_DYNAMIC;
        double bm = 0.0, bI = 0.0; vec bc = V(0, 0);
        double msum = bm + m;
        bI += m * ipm + vdot(vsub(bc, scog), vsub(bc, scog)) * (m * bm) / msum;
        bc = vlerp(bc, scog, m / msum);
        bm = msum;
        b->m = bm; b->i = bI; b->cog = bc; b->m_inv = 1.0 / bm; b->i_inv = 1.0 / bI;
        b->a = 0.0;
        b->p = vadd(V(bc.x * 1.0 - bc.y * 0.0, bc.x * 0.0 + bc.y * 1.0), V(ox, oy));
        body_set_transform(b);
    }
    /* walls: STATIC body + Segment(a, b, 0.5), elasticity 0.5, friction 0.5, type 3 */
    for (int w = 0; w < nwalls; w++) {
        int bi = 1 + nbox + w, si = 1 + nwheels + nbox + w;
        body_t *b = &E->bodies[bi];
        b->type = BODY_STATIC; b->m = b->i = INFINITY; b->m_inv = b->i_inv = 0.0; b->p = V(0, 0); b->a = 0.0; b->cog = V(0, 0);
        body_set_transform(b);
        shape_t *s = &E->shapes[si];
        s->body = bi; s->r = E->P.wall_radius; s->e = 0.5; s->u = 0.5; s->ctype = 3; s->n = 2;
        vec a = V(walls[4 * w], walls[4 * w + 1]), bb = V(walls[4 * w + 2], walls[4 * w + 3]);
        vec sn = vrperp(vnormalize(vsub(bb, a))); /* cpSegmentShapeInit: n = rperp(normalize(b - a)) */
        s->lv[0] = a; s->lv[1] = bb; s->ln[0] = vneg(sn); s->ln[1] = sn;
    }
    E->nb = nb; E->ns = ns;
    for (int i = 0; i < E->ns; i++) E->order[i] = i;
    for (int s = 0; s < E->ns; s++) shape_cache_bb(&E->shapes[s], &E->bodies[E->shapes[s].body]);
    /* global distance map (static per maze) */
    int H = (int)(E->P.map_h * E->P.m_to_pix), W = (int)(E->P.map_w * E->P.m_to_pix);
    if (E->map_h != H || E->map_w != W || !E->dist_map) {
        free(E->dist_map); free(E->wall_map); free(E->dist_raw);
        E->dist_map = (double *)calloc((size_t)H * W, sizeof(double));
        E->wall_map = (double *)calloc((size_t)H * W, sizeof(double));
        E->dist_raw = (double *)calloc((size_t)H * W, sizeof(double));
        E->map_h = H; E->map_w = W;
    }
    maze_goal_map(E);
    double dts = E->P.dt / E->P.steps;
    for (int k = 0; k < E->P.settle_steps; k++) space_step(E, dts);
    E->wall_collision = 0; /* reset() clears the flag after init (maze_NAMO_env.py:338) */
    /* prev_obs snapshot of the box shapes */
    for (int s = 0; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double *dst = E->prev_wv + (size_t)s * ORC_MAXV * 2;
        for (int i = 0; i < sh->n; i++) { dst[2 * i] = sh->wv[i].x; dst[2 * i + 1] = sh->wv[i].y; }
    }
    return ns;
}

/* scipy.ndimage.rotate(img, deg, reshape=False, order=1, mode='constant', cval) restated (affine_transform ->
 * NI_GeometricTransform, spline order 1); angle passed as (cos, sin) of the rotation. img, out: [n][n] doubles. */
static void nd_rotate_order1(const double *img, int n, double c, double s, double cval, double *out)
{
    double ctr = ((double)n - 1) / 2;
    /* rot_matrix = [[c, s], [-s, c]]; offset = in_center - rot_matrix @ out_center */
    double oc0 = c * ctr + s * ctr, oc1 = -s * ctr + c * ctr;
    double off0 = ctr - oc0, off1 = ctr - oc1;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double c0 = 0.0, c1 = 0.0;
            c0 += (double)i * c; c0 += (double)j * s; c0 += off0;
            c1 += (double)i * -s; c1 += (double)j * c; c1 += off1;
            double t;
            if (c0 < 0 || c0 > n - 1 || c1 < 0 || c1 > n - 1) t = cval;
            else {
                int s0 = (int)floor(c0), s1 = (int)floor(c1);
                double x0 = c0 - s0, x1 = c1 - s1;
                double w0[2] = {1.0 - x0, x0}, w1[2] = {1.0 - x1, x1};
                t = 0.0;
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 2; b++) {
                        int ii = s0 + a, jj = s1 + b;
                        double coeff = (ii > n - 1 || jj > n - 1) ? cval : img[ii * n + jj];
                        coeff *= w0[a];
                        coeff *= w1[b];
                        t += coeff;
                    }
            }
            out[i * n + j] = t;
        }
}
void orc_nd_rotate(const double *img, int n, double c, double s, double cval, double *out) { nd_rotate_order1(img, n, c, s, cval, out); }

/* generate_observation -> OccupancyGrid.ego_view_map_maze (occupancy_map.py:142-202): u8 [4][LH][LW] */
void orc_maze_observe(orc_env *E, uint8_t *obs)
{
    const orc_params *P = &E->P;
    const body_t *rb = &E->bodies[0];
    int H = E->map_h, W = E->map_w;
    int LH = (int)(P->local_h * P->m_to_pix), LW = (int)(P->local_w * P->m_to_pix);
    double *gobs = (double *)calloc((size_t)H * W, sizeof(double));
    double *gfoot = (double *)calloc((size_t)H * W, sizeof(double));
    /* compute_occ_img over every box (no range culling), occupancy_map.py:37-65 */
    for (int s = 0; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        if (sh->ctype != 2) continue;
        double r[ORC_MAXV], c[ORC_MAXV];
        for (int i = 0; i < sh->n; i++) { c[i] = sh->wv[i].x * P->m_to_pix; r[i] = sh->wv[i].y * P->m_to_pix; }
        draw_polygon(sh->n, r, c, H, W, 1, gobs, 1.0);
    }
    /* _compute_global_footprint_maze (occupancy_map.py:340-376): robot body outline, vertices outside dropped */
    double sx = rb->p.x, sy = rb->p.y, sa = rb->a;
    {
        double ch, shh;
        bp_sincos(sa, &shh, &ch);
        double m2gx = (double)W / P->map_w, m2gy = (double)H / P->map_h;
        double r[ORC_MAXV], c[ORC_MAXV]; int cnt = 0;
        for (int i = 0; i < E->robot_nv; i++) {
            double vx = E->robot_verts[i].x * ch + E->robot_verts[i].y * -shh + sx;
            double vy = E->robot_verts[i].x * shh + E->robot_verts[i].y * ch + sy;
            double gx = vx * m2gx, gy = vy * m2gy;
            if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
            r[cnt] = gy; c[cnt] = gx; cnt++;
        }
        if (cnt > 0) draw_polygon(cnt, r, c, H, W, 0, gfoot, 1.0);
    }
    int wx = (int)(sx * P->m_to_pix), wy = (int)(sy * P->m_to_pix);
    int infl = (LW > LH ? LW : LH) / 2;
    int IH = LH + infl, IW = LW + infl; /* 288 x 288 */
    size_t ip = (size_t)IH * IW;
    double *loc = (double *)malloc(sizeof(double) * ip * 4), *rot = (double *)malloc(sizeof(double) * ip);
    for (size_t i = 0; i < ip; i++) { loc[i] = 0.0; loc[ip + i] = 0.0; loc[2 * ip + i] = 0.0; loc[3 * ip + i] = 1.0; }
    for (int li = 0; li < IH; li++)
        for (int lj = 0; lj < IW; lj++) {
            int gi = (int)((double)(li + wy) - ((double)IH / 2));
            int gj = (int)((double)(lj + wx) - ((double)IW / 2));
            if (gi < 0 || gi >= H || gj < 0 || gj >= W) continue;
            size_t l = (size_t)li * IW + lj, g = (size_t)gi * W + gj;
            loc[l] = gfoot[g]; loc[ip + l] = gobs[g]; loc[2 * ip + l] = E->wall_map[g]; loc[3 * ip + l] = E->dist_map[g];
        }
    /* rotate by (heading - pi/2); cos/sin from the deterministic bp_sincos instead of scipy's cosdg/sindg */
    double rs, rc;
    bp_sincos(sa - M_PI / 2, &rs, &rc);
    int half = infl / 2;
    size_t pl = (size_t)LH * LW;
    for (int ch = 0; ch < 4; ch++) {
        nd_rotate_order1(loc + ip * ch, IH, rc, rs, ch == 3 ? 1.0 : 0.0, rot);
        for (int i = 0; i < LH; i++)
            for (int j = 0; j < LW; j++) obs[pl * ch + (size_t)i * LW + j] = (uint8_t)(rot[(size_t)(half + i) * IW + (half + j)] * 255);
    }
    free(gobs); free(gfoot); free(loc); free(rot);
}

enum { ORC_MI_X = 0, ORC_MI_Y, ORC_MI_THETA, ORC_MI_TOTAL_WORK, ORC_MI_WORK, ORC_MI_COLL_REWARD, ORC_MI_SCALED_COLL, ORC_MI_DIST_INC,
       ORC_MI_SUCCESS, ORC_MI_BOUNDARY, ORC_MI_WALL, ORC_MI_KE, ORC_MI_IMPULSE, ORC_MI_NPOST, ORC_MI_NCONTACT, ORC_MI_NFIRST, ORC_MI_COUNT };

/* MazeNAMO.step (maze_NAMO_env.py:402-485) */
void orc_maze_step(orc_env *E, double action, uint8_t *obs, double *reward, int *terminated, double *info)
{
    const orc_params *P = &E->P;
    body_t *rb = &E->bodies[0];
    double sn, cs;
    bp_sincos(rb->a, &sn, &cs);
    rb->v = V(cs * P->target_speed + -sn * 0.0, sn * P->target_speed + cs * 0.0);
    rb->w = action * P->max_yaw_rate;
    int boundary = 0;
    double dts = P->dt / P->steps;
    for (int k = 0; k < P->steps; k++) {
        space_step(E, dts);
        if (rb->p.x < 0.0 || rb->p.x > P->map_w) boundary = 1;
    }
    double work = 0.0;
    for (int s = 0; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        if (sh->ctype != 2) continue;
        double now[ORC_MAXV * 2];
        for (int i = 0; i < sh->n; i++) { now[2 * i] = sh->wv[i].x; now[2 * i + 1] = sh->wv[i].y; }
        double *prev = E->prev_wv + (size_t)s * ORC_MAXV * 2;
        double area = poly_area_np(sh->n, prev);
        double ax, ay, bx, by;
        poly_centroid_np(sh->n, prev, &ax, &ay);
        poly_centroid_np(sh->n, now, &bx, &by);
        work += sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) * area;
        memcpy(prev, now, sizeof(double) * 2 * (size_t)sh->n);
    }
    E->total_work += work;
    double gd = sqrt((rb->p.x - P->goal_x) * (rb->p.x - P->goal_x) + (rb->p.y - P->goal_y) * (rb->p.y - P->goal_y));
    int goal = gd <= P->goal_reach;
    int term = goal || E->wall_collision;
    int px = (int)(rb->p.x * P->m_to_pix), py = (int)(rb->p.y * P->m_to_pix);
    if (px < 0) px = 0; if (px > E->map_w - 1) px = E->map_w - 1; if (py < 0) py = 0; if (py > E->map_h - 1) py = E->map_h - 1;
    double dist_value = E->dist_map[(size_t)py * E->map_w + px];
    double dinc = 0.0;
    if (rb->p.x != P->goal_x || rb->p.y != P->goal_y) {
        if (E->have_prev_dist) dinc = (E->prev_dist - dist_value) * P->k_increment;
        E->prev_dist = dist_value; E->have_prev_dist = 1;
    }
    double coll = -work;
    double r = P->beta * coll + dinc;
    if (boundary || E->wall_collision) r += P->boundary_penalty;
    int success = 0;
    if (term && !E->wall_collision) { r += P->terminal_reward; success = 1; }
    *reward = r; *terminated = term;
    if (info) {
        info[ORC_MI_X] = rb->p.x; info[ORC_MI_Y] = rb->p.y; info[ORC_MI_THETA] = rb->a; info[ORC_MI_TOTAL_WORK] = E->total_work;
        info[ORC_MI_WORK] = work; info[ORC_MI_COLL_REWARD] = coll; info[ORC_MI_SCALED_COLL] = coll * P->beta; info[ORC_MI_DIST_INC] = dinc;
        info[ORC_MI_SUCCESS] = success; info[ORC_MI_BOUNDARY] = boundary; info[ORC_MI_WALL] = E->wall_collision;
        info[ORC_MI_KE] = E->total_ke; info[ORC_MI_IMPULSE] = E->total_impulse;
        info[ORC_MI_NPOST] = (double)E->n_post_solve; info[ORC_MI_NCONTACT] = (double)E->n_contact_pts; info[ORC_MI_NFIRST] = (double)E->n_first_contact;
    }
    if (obs) orc_maze_observe(E, obs);
}
/* per-shape pose/velocity of the owning body: [ns][9] */
void orc_get_shape_states(const orc_env *E, double *out)
{
    for (int s = 0; s < E->ns; s++) {
        const body_t *b = &E->bodies[E->shapes[s].body];
        double *o = out + 9 * (size_t)s;
        o[0] = b->p.x; o[1] = b->p.y; o[2] = b->a; o[3] = b->v.x; o[4] = b->v.y; o[5] = b->w; o[6] = b->vb.x; o[7] = b->vb.y; o[8] = b->wb;
    }
}
/* test hooks: the restated skimage.draw.polygon / cv2.line rasterisers on a caller-owned image (tests/golden/make_golden_obs_pipeline.py
 * plugs them into the reference's OccupancyGrid in place of the absent libraries) */
void orc_draw_polygon(int n, const double *r, const double *c, int H, int W, double *img, double val) { draw_polygon(n, r, c, H, W, 1, img, val); }
void orc_cv_line(double *img, int H, int W, long x1, long y1, long x2, long y2, double val) { cv_line(img, H, W, x1, y1, x2, y2, val); }
/* test hook: replace the normalised goal map (tests/test_step_logic_golden.py injects the same map into the reference class) */
void orc_maze_set_dist_map(orc_env *E, const double *m) { memcpy(E->dist_map, m, sizeof(double) * (size_t)E->map_h * E->map_w); }
void orc_maze_maps(const orc_env *E, double *dist_norm, double *dist_raw, double *wall)
{
    size_t n = (size_t)E->map_h * E->map_w;
    if (dist_norm) memcpy(dist_norm, E->dist_map, n * sizeof(double));
    if (dist_raw) memcpy(dist_raw, E->dist_raw, n * sizeof(double));
    if (wall) memcpy(wall, E->wall_map, n * sizeof(double));
}

/* ---- global (planner) observation, egocentric_obs: false (ship_ice_env.py:96-99,394-406): uint8 [2][H/5][W/5] ------- */
/* ch0 = 5x5 block mean of the full 25 px/m occupancy raster (compute_occ_img without range culling + compute_con_gridmap,
 * occupancy_map.py:37-65,97-109); ch1 = compute_ship_footprint_planner on the 0.2 m grid (:253-296). */
void orc_observe_global(orc_env *E, double grid_m, uint8_t *obs)
{
    const orc_params *P = &E->P;
    const body_t *ship = &E->bodies[0];
    int Wg = (int)(P->map_w / grid_m), Hg = (int)(P->map_h / grid_m);
    int bw = (int)(P->map_w * P->m_to_pix), bh = (int)(P->map_h * P->m_to_pix);
    int by = (int)(grid_m * P->m_to_pix), bx = (int)(grid_m * P->m_to_pix);
    double *occ = (double *)calloc((size_t)bh * bw, sizeof(double));
    double *foot = (double *)calloc((size_t)Hg * Wg, sizeof(double));
    for (int s = 1; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double r[ORC_MAXV], c[ORC_MAXV];
        for (int i = 0; i < sh->n; i++) { c[i] = sh->wv[i].x * P->m_to_pix; r[i] = sh->wv[i].y * P->m_to_pix; }
        draw_polygon(sh->n, r, c, bh, bw, 1, occ, 1.0);
    }
    {
        double ch, shh;
        bp_sincos(ship->a, &shh, &ch);
        double m2gx = (double)Wg / P->map_w, m2gy = (double)Hg / P->map_h;
        double r[ORC_MAXV + 8], c[ORC_MAXV + 8]; int cnt = 0;
        for (int i = 0; i < E->ship_nv; i++) {
            double vx = E->ship_verts[i].x * ch + E->ship_verts[i].y * -shh + ship->p.x;
            double vy = E->ship_verts[i].x * shh + E->ship_verts[i].y * ch + ship->p.y;
            double gx = vx * m2gx, gy = vy * m2gy;
            if (gy < 0 || gy >= Hg || gx < 0 || gx >= Wg) continue;
            r[cnt] = gy; c[cnt] = gx; cnt++;
        }
        if (cnt > 0) draw_polygon(cnt, r, c, Hg, Wg, 0, foot, 1.0);
    }
    size_t pl = (size_t)Hg * Wg;
    for (int gi = 0; gi < Hg; gi++)
        for (int gj = 0; gj < Wg; gj++) {
            double sum = 0.0;
            for (int a = 0; a < by; a++)
                for (int b = 0; b < bx; b++) sum += occ[(size_t)(gi * by + a) * bw + (gj * bx + b)];
            double mean = sum / (double)(by * bx);
            obs[(size_t)gi * Wg + gj] = (uint8_t)(mean * 255);
            obs[pl + (size_t)gi * Wg + gj] = (uint8_t)(foot[(size_t)gi * Wg + gj] * 255);
        }
    free(occ); free(foot);
}

/* ---- planner cost map: CostMap.__init__ / boundary_cost / update / populate_costmap (common/cost_map.py:27-126,284-287) over the
 * environment's obstacles (info['obs'], cost_map.py:275-281).  out: [int(m*scale)][int(n*scale)] float64.
 * horizon <= 0: no horizon (`horizon=None`).  The reference's `** 0.5` / `** 2` are evaluated as sqrt / products here. */
#define ORC_MAX_COST 1e10
void orc_costmap(const orc_env *E, double scale, int m, int n, double alpha, double ship_mass, double horizon, int margin,
                 double ship_pos_y, double vs, double *out)
{
    const int H = (int)(m * scale), W = (int)(n * scale);
    memset(out, 0, (size_t)H * W * sizeof(double));
    if (margin) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < margin && j < W; j++) { out[(size_t)i * W + j] = ORC_MAX_COST; out[(size_t)i * W + (W - 1 - j)] = ORC_MAX_COST; }
    }
    const double hz = horizon > 0 ? horizon * scale : 0.0;
    for (int s = 1; s < E->ns; s++) {
        const shape_t *sh = &E->shapes[s];
        double ox[ORC_MAXV], oy[ORC_MAXV];
        int nn = sh->n;
        for (int i = 0; i < nn; i++) { ox[i] = sh->wv[i].x * scale; oy[i] = sh->wv[i].y * scale; }
        if (hz != 0.0) {
            int all = 1;
            for (int i = 0; i < nn; i++) if (!(oy[i] > (ship_pos_y + hz) || oy[i] < ship_pos_y)) all = 0;
            if (all) continue;
        }
        /* resample_vertices(decimals=0): drop a vertex whose rounded coordinates repeat an earlier vertex's */
        double c[ORC_MAXV], r[ORC_MAXV];
        int k = 0;
        for (int i = 0; i < nn; i++) {
            int dup = 0;
            for (int j = 0; j < i; j++) if (rint(ox[j]) == rint(ox[i]) && rint(oy[j]) == rint(oy[i])) dup = 1;
            if (!dup) { c[k] = ox[i]; r[k] = oy[i]; k++; }
        }
        /* skimage.draw.polygon(ob[:,1], ob[:,0], shape) */
        double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
        for (int i = 1; i < k; i++) { rmin = fmin(rmin, r[i]); rmax = fmax(rmax, r[i]); cmin = fmin(cmin, c[i]); cmax = fmax(cmax, c[i]); }
        long minr = (long)fmax(0.0, rmin), maxr = (long)ceil(rmax), minc = (long)fmax(0.0, cmin), maxc = (long)ceil(cmax);
        if (maxr > H - 1) maxr = H - 1;
        if (maxc > W - 1) maxc = W - 1;
        long cnt = 0, sr = 0, sc = 0;
        for (long ri = minr; ri <= maxr; ri++)
            for (long ci = minc; ci <= maxc; ci++)
                if (point_in_polygon(k, c, r, (double)ci, (double)ri)) { cnt++; sr += ri; sc += ci; }
        if (cnt == 0) continue;
        const double cx = (double)sc / (double)cnt, cy = (double)sr / (double)cnt;
        double rad = 0.0;   /* poly_radius (geometry/polygon.py:20-22) */
        for (int i = 0; i < k; i++) {
            const double d = sqrt((c[i] - cx) * (c[i] - cx) + (r[i] - cy) * (r[i] - cy));
            if (i == 0 || d > rad) rad = d;
        }
        double xy[2 * ORC_MAXV];
        for (int i = 0; i < k; i++) { xy[2 * i] = c[i] / scale; xy[2 * i + 1] = r[i] / scale; }
        const double mi = poly_area_np(k, xy);
        const double norm = alpha * ((vs * vs) * (mi * mi)) / (2 * (ship_mass + mi));
        for (long ri = minr; ri <= maxr; ri++)
            for (long ci = minc; ci <= maxc; ci++)
                if (point_in_polygon(k, c, r, (double)ci, (double)ri)) {
                    const double dist = sqrt(((double)ri - cy) * ((double)ri - cy) + ((double)ci - cx) * ((double)ci - cx));
                    const double nc = fmax(0.0, (rad * rad - dist * dist) / (rad * rad));
                    double *o = &out[(size_t)ri * W + ci];
                    *o = fmin(ORC_MAX_COST, fmax(nc * norm, *o));
                }
    }
}

#include "bp_oracle_bd.c"
