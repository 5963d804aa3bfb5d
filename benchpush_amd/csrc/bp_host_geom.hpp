// Host-side body/shape construction for bp_load_scenarios (product code, C++).
//
// Follows what pymunk does when the reference builds its sim objects:
//   create_polygon  benchpush/common/utils/sim_utils.py:136-153  (Poly(None, verts).center_of_gravity, recentre,
//                   Poly(body, vs, radius=0.02), shape.density = 0.001)
//   Ship.sim        benchpush/common/ship.py:77-98               (KINEMATIC body, Poly(radius=0.02))
// i.e. Chipmunk2D 7.0.3's cpConvexHull (QuickHull, tol 0), cpCentroidForPoly, cpAreaForPoly (with radius),
// cpMomentForPoly and cpBodyAccumulateMassFromShapes.  The hull's vertex ORDER is part of the contract: vertex
// indices feed the contact ids and every per-vertex sum runs in this order.
#pragma once
#include <cmath>
#include <cfloat>
#include <utility>
#include <vector>

namespace bpgeom {

struct P2 { double x, y; };
static inline P2 sub(P2 a, P2 b) { return {a.x - b.x, a.y - b.y}; }
static inline P2 add(P2 a, P2 b) { return {a.x + b.x, a.y + b.y}; }
static inline double cross(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
static inline double dot(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }
static inline double len(P2 a) { return std::sqrt(dot(a, a)); }

// Splits pts[0..n) into those strictly left of a->b (moved to the front, farthest first) and the rest.
static int hull_partition(P2 *pts, int n, P2 a, P2 b, double tol)
{
    if (n == 0) return 0;
    double best = 0;
    int pivot = 0;
    const P2 d = sub(b, a);
    const double vtol = tol * len(d);
    int head = 0;
    int tail = n - 1;
    while (head <= tail) {
        const double val = cross(sub(pts[head], a), d);
        if (val > vtol) {
            if (val > best) { best = val; pivot = head; }
            ++head;
        } else {
            std::swap(pts[head], pts[tail]);
            --tail;
        }
    }
    if (pivot != 0) std::swap(pts[0], pts[pivot]);
    return head;
}

static int hull_reduce(double tol, P2 *pts, int n, P2 a, P2 pivot, P2 b, P2 *out)
{
    if (n < 0) return 0;
    if (n == 0) { out[0] = pivot; return 1; }
    // the pivot of an empty side is never used (count - 1 < 0 returns at once), and with nl == n the element pts[nl] does not exist: Chipmunk's QHullReduce
    // evaluates verts[0] as an argument regardless; here it is not read (found by the ASan pass of tests/test_host_sanitize.py)
    const int nl = hull_partition(pts, n, a, pivot, tol);
    int k = hull_reduce(tol, pts + 1, nl - 1, a, nl > 0 ? pts[0] : pivot, pivot, out);
    out[k++] = pivot;
    const int nr = hull_partition(pts + nl, n - nl, pivot, b, tol);
    return k + hull_reduce(tol, pts + nl + 1, nr - 1, pivot, nr > 0 ? pts[nl] : pivot, b, out + k);
}

// cpConvexHull(count, verts, result, NULL, 0.0)
static std::vector<P2> convex_hull(const std::vector<P2> &in)
{
    const int n = (int)in.size();
    std::vector<P2> work(in), out(n + 1);
    int lo = 0, hi = 0;
    P2 mn = in[0], mx = in[0];
    for (int i = 1; i < n; ++i) {
        const P2 v = in[i];
        if (v.x < mn.x || (v.x == mn.x && v.y < mn.y)) { mn = v; lo = i; }
        else if (v.x > mx.x || (v.x == mx.x && v.y > mx.y)) { mx = v; hi = i; }
    }
    if (lo == hi) { return std::vector<P2>{in[0]}; }
    std::swap(work[0], work[lo]);
    std::swap(work[1], work[hi == 0 ? lo : hi]);
    const P2 a = work[0], b = work[1];
    // result aliases the work buffer in Chipmunk (result == verts copy); emulate with the same in-place layout
    std::vector<P2> buf(work);
    int cnt = hull_reduce(0.0, buf.data() + 2, n - 2, a, b, a, buf.data() + 1) + 1;
    buf.resize(cnt);
    return buf;
}

static P2 centroid(const std::vector<P2> &v)
{
    const int n = (int)v.size();
    double sum = 0.0;
    P2 acc{0, 0};
    for (int i = 0; i < n; ++i) {
        const P2 p = v[i], q = v[(i + 1) % n];
        const double c = cross(p, q);
        sum += c;
        const P2 s = add(p, q);
        acc = add(acc, P2{s.x * c, s.y * c});
    }
    const double f = 1.0 / (3.0 * sum);
    return {acc.x * f, acc.y * f};
}

static double area_with_radius(const std::vector<P2> &v, double r)
{
    const int n = (int)v.size();
    double a2 = 0.0, per = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 p = v[i], q = v[(i + 1) % n];
        a2 += cross(p, q);
        per += len(sub(p, q));
    }
    return r * (M_PI * std::fabs(r) + per) + a2 / 2.0;
}

static double moment_per_unit_mass(const std::vector<P2> &v, P2 off)
{
    const int n = (int)v.size();
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 p = add(v[i], off), q = add(v[(i + 1) % n], off);
        const double a = cross(q, p);
        const double b = dot(p, p) + dot(p, q) + dot(q, q);
        s1 += a * b;
        s2 += a;
    }
    return (1.0 * s1) / (6.0 * s2);
}

// numpy shoelace of the raw polygon (geometry/polygon.py:25-29) used by the zero-area filter, ship_ice_env.py:206
static double shoelace_area(const double *xy, int n)
{
    double d1 = 0.0, d2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const int p = (i - 1 + n) % n;
        d1 += xy[2 * i] * xy[2 * p + 1];
        d2 += xy[2 * i + 1] * xy[2 * p];
    }
    return 0.5 * std::fabs(d1 - d2);
}

struct Shape {
    std::vector<P2> verts;   // hull, body-local (COG recentred)
    std::vector<P2> normals; // plane i: edge verts[i-1] -> verts[i]
    double m_inv = 0, i_inv = 0;
    P2 cog{0, 0};
    P2 p{0, 0};              // world position of the centre of gravity
    double angle = 0;
    double radius = 0, e = 0, u = 0; // shape radius / elasticity / friction
    int kind = 0;            // collision_type | group << 8 | body_type << 16
};

static void set_planes(Shape &s)
{
    const int n = (int)s.verts.size();
    s.normals.resize(n);
    for (int i = 0; i < n; ++i) {
        const P2 a = s.verts[(i - 1 + n) % n], b = s.verts[i];
        const P2 e = sub(b, a);
        const P2 rp{e.y, -e.x};
        const double inv = 1.0 / (len(rp) + DBL_MIN);
        s.normals[i] = {rp.x * inv, rp.y * inv};
    }
}

// floe: returns false if dropped by the zero-area filter
static bool build_floe(const double *raw_xy, int n, double cx, double cy, double density, double radius, Shape &out)
{
    if (shoelace_area(raw_xy, n) == 0.0) return false;
    std::vector<P2> loc(n);
    for (int i = 0; i < n; ++i) loc[i] = {raw_xy[2 * i] - cx, raw_xy[2 * i + 1] - cy};
    const std::vector<P2> h0 = convex_hull(loc);
    const P2 c0 = centroid(h0);
    for (int i = 0; i < n; ++i) loc[i] = {loc[i].x - c0.x, loc[i].y - c0.y};
    out.verts = convex_hull(loc);
    set_planes(out);
    const P2 sc = centroid(out.verts);
    const double area = area_with_radius(out.verts, radius);
    const double m = density * area;
    const double ipm = moment_per_unit_mass(out.verts, P2{-sc.x, -sc.y});
    // cpBodyAccumulateMassFromShapes with a single shape on an empty body
    double bm = 0.0, bi = 0.0;
    P2 bc{0, 0};
    const double msum = bm + m;
    const P2 dc = sub(bc, sc);
    bi += m * ipm + dot(dc, dc) * (m * bm) / msum;
    const double t = m / msum;
    bc = P2{bc.x * (1.0 - t) + sc.x * t, bc.y * (1.0 - t) + sc.y * t};
    bm = msum;
    out.m_inv = 1.0 / bm;
    out.i_inv = 1.0 / bi;
    out.cog = bc;
    out.angle = 0.0;
    out.p = P2{(bc.x * 1.0 - bc.y * 0.0) + cx, (bc.x * 0.0 + bc.y * 1.0) + cy};
    return true;
}

static void build_ship(const double (*sv)[2], int n, double x, double y, double theta, Shape &out)
{
    std::vector<P2> loc(n);
    for (int i = 0; i < n; ++i) loc[i] = {sv[i][0], sv[i][1]};
    out.verts = convex_hull(loc);
    set_planes(out);
    out.m_inv = 0.0;
    out.i_inv = 0.0;
    out.cog = {0, 0};
    out.p = {x, y};
    out.angle = theta;
}


// ---- maze-NAMO-v0 host construction -------------------------------------------------------------------------------
// kinematic part of the robot (Robot.sim, robot.py:77-118): Poly(body, verts, radius=0.02) -> convex hull, no recentring
static void build_kinematic_part(const double (*v)[2], int n, double x, double y, double theta, Shape &out)
{
    std::vector<P2> loc(n);
    for (int i = 0; i < n; ++i) loc[i] = {v[i][0], v[i][1]};
    out.verts = convex_hull(loc);
    set_planes(out);
    out.m_inv = 0.0; out.i_inv = 0.0; out.cog = {0, 0}; out.p = {x, y}; out.angle = theta;
}
// static Segment(a, b, radius) (generate_sim_maze, sim_utils.py:174-181) as a 2-vertex hull: planes -n / +n with
// n = rperp(normalize(b - a)) (cpSegmentShapeInit)
static void build_wall(double ax, double ay, double bx, double by, Shape &out)
{
    out.verts = {P2{ax, ay}, P2{bx, by}};
    const P2 d = sub(P2{bx, by}, P2{ax, ay});
    const double inv = 1.0 / (len(d) + DBL_MIN);
    const P2 u{d.x * inv, d.y * inv};
    const P2 n{u.y, -u.x};
    out.normals = {P2{-n.x, -n.y}, n};
    out.m_inv = 0.0; out.i_inv = 0.0; out.cog = {0, 0}; out.p = {0, 0}; out.angle = 0.0;
}

// skimage.draw.polygon(r, c, shape) restated (point_in_polygon crossing rule incl. its 1e-12 vertex tolerance)
static bool pip(const std::vector<double> &xp, const std::vector<double> &yp, double x, double y)
{
    const int n = (int)xp.size();
    const double eps = 1e-12;
    unsigned lc = 0, rc = 0;
    double x1 = xp[n - 1] - x, y1 = yp[n - 1] - y;
    for (int i = 0; i < n; ++i) {
        const double x0 = xp[i] - x, y0 = yp[i] - y;
        if ((-eps < x0 && x0 < eps) && (-eps < y0 && y0 < eps)) return true;
        if ((y0 > 0) != (y1 > 0)) { if (((x0 * y1 - x1 * y0) / (y1 - y0)) > 0) rc++; }
        if ((y0 < 0) != (y1 < 0)) { if (((x0 * y1 - x1 * y0) / (y1 - y0)) < 0) lc++; }
        x1 = x0; y1 = y0;
    }
    if ((rc & 1) != (lc & 1)) return true;
    return (rc & 1) != 0;
}
static void fill_polygon(const std::vector<double> &r, const std::vector<double> &c, int H, int W, std::vector<unsigned char> &img)
{
    double rmin = r[0], rmax = r[0], cmin = c[0], cmax = c[0];
    for (size_t i = 1; i < r.size(); ++i) { rmin = std::fmin(rmin, r[i]); rmax = std::fmax(rmax, r[i]); cmin = std::fmin(cmin, c[i]); cmax = std::fmax(cmax, c[i]); }
    long minr = (long)std::fmax(0.0, rmin), maxr = (long)std::ceil(rmax), minc = (long)std::fmax(0.0, cmin), maxc = (long)std::ceil(cmax);
    if (maxr > H - 1) maxr = H - 1;
    if (maxc > W - 1) maxc = W - 1;
    for (long ri = minr; ri <= maxr; ++ri)
        for (long ci = minc; ci <= maxc; ++ci)
            if (pip(c, r, (double)ci, (double)ri)) img[(size_t)ri * W + ci] = 1;
}
// compute_occ_img_walls (occupancy_map.py:67-94) + global_goal_point_dist_transform (:435-485)
static void maze_maps(const double *walls, int nwalls, double wall_radius, double map_w, double map_h, int H, int W, double goal_x,
                      double goal_y, std::vector<unsigned char> &wall, std::vector<double> &norm, std::vector<double> &raw)
{
    wall.assign((size_t)H * W, 0);
    const double m2p = (double)H / map_h;
    for (int w = 0; w < nwalls; ++w) {
        const P2 a{walls[4 * w], walls[4 * w + 1]}, b{walls[4 * w + 2], walls[4 * w + 3]};
        const P2 d = sub(b, a);
        const double l = std::sqrt(d.x * d.x + d.y * d.y);
        const P2 u{d.x / l, d.y / l}, p{-u.y, u.x};
        const double wr = wall_radius;
        const double vx[4] = {(a.x + wr * p.x - wr * u.x), (a.x - wr * p.x - wr * u.x), (b.x - wr * p.x + wr * u.x), (b.x + wr * p.x + wr * u.x)};
        const double vy[4] = {(a.y + wr * p.y - wr * u.y), (a.y - wr * p.y - wr * u.y), (b.y - wr * p.y + wr * u.y), (b.y + wr * p.y + wr * u.y)};
        std::vector<double> r(4), c(4);
        for (int i = 0; i < 4; ++i) { c[i] = vx[i] * m2p; r[i] = vy[i] * m2p; }
        fill_polygon(r, c, H, W, wall);
    }
    const double g2m = map_w / (double)W;
    const int gx = (int)(goal_x / g2m), gy = (int)(goal_y / g2m);
    raw.assign((size_t)H * W, 0.0);
    std::vector<unsigned char> vis((size_t)H * W, 0);
    std::vector<int> q; q.reserve((size_t)H * W);
    raw[(size_t)gy * W + gx] = 1.0; vis[(size_t)gy * W + gx] = 1; q.push_back(gy * W + gx);
    static const int dy[8] = {0, 0, 1, -1, 1, 1, -1, -1}, dx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    double mx = 1.0;
    for (size_t qh = 0; qh < q.size(); ++qh) {
        const int cur = q[qh], y = cur / W, x = cur % W;
        for (int k = 0; k < 8; ++k) {
            const int ny = y + dy[k], nx = x + dx[k];
            if (ny < 0 || ny >= H || nx < 0 || nx >= W) continue;
            const size_t id = (size_t)ny * W + nx;
            if (vis[id] || wall[id]) continue;
            vis[id] = 1;
            raw[id] = raw[cur] + 1;
            if (raw[id] > mx) mx = raw[id];
            q.push_back((int)id);
        }
    }
    norm.assign((size_t)H * W, 0.0);
    for (size_t i = 0; i < norm.size(); ++i) norm[i] = wall[i] ? 1.0 : raw[i] / mx;
}

} // namespace bpgeom
