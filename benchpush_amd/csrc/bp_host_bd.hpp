// Host-side (load-time) construction for box-delivery-v0: shapes with a heading, and the static per-layout rasters of
// BoxDeliveryEnv.update_configuration_space / create_global_shortest_path_to_receptacle_map (box_delivery_env.py:1115-1175):
// cv2.fillPoly of the obstacle polygons, skimage disk dilation, scipy's nearest-free-cell indices, spfa from the receptacle.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "bp_host_geom.hpp"

namespace bpgeom {

// same deterministic sin/cos as the device code (bp_device.hpp: bp_sincos)
static void host_sincos(double x, double &sn, double &cs)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = std::rint(x * invpio2);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double y0 = r - w;
    if (std::fabs(y0) < std::fabs(x) * 7.62939453125e-06) {
        const double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
    }
    const double y1 = (r - y0) - w;
    const int q = (int)((long long)fn & 3);
    const double z = y0 * y0, v = z * y0;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double kc = 1.0 - (0.5 * z - (z * rc - y0 * y1));
    if (q == 0) { sn = ks; cs = kc; }
    else if (q == 1) { sn = kc; cs = -ks; }
    else if (q == 2) { sn = -ks; cs = -kc; }
    else { sn = -kc; cs = ks; }
}

// create_polygon with a heading (sim_utils.py:120-160): square box, COG recentred, density mass
static bool build_box(double cx, double cy, double heading, double half, double density, double radius, Shape &out)
{
    const double raw[8] = {cx + half, cy + half, cx - half, cy + half, cx - half, cy - half, cx + half, cy - half};
    if (!build_floe(raw, 4, cx, cy, density, radius, out)) return false;
    double sn, cs;
    host_sincos(heading, sn, cs);
    out.angle = heading;
    out.p = P2{(out.cog.x * cs - out.cog.y * sn) + cx, (out.cog.x * sn + out.cog.y * cs) + cy};
    return true;
}
// create_agent main shape (sim_utils.py:20-38): vertices recentred on the hull's centre of gravity, KINEMATIC body
static void build_agent_main(const double (*v)[2], int n, double x, double y, double theta, Shape &out)
{
    std::vector<P2> loc(n);
    for (int i = 0; i < n; ++i) loc[i] = {v[i][0], v[i][1]};
    const P2 c0 = centroid(convex_hull(loc));
    for (int i = 0; i < n; ++i) loc[i] = {v[i][0] - c0.x, v[i][1] - c0.y};
    out.verts = convex_hull(loc);
    set_planes(out);
    out.m_inv = 0.0; out.i_inv = 0.0; out.cog = {0, 0}; out.p = {x, y}; out.angle = theta;
}
// create_static / create_corners (sim_utils.py:75-135): STATIC body at (x, y, angle) with a convex polygon
static void build_static_poly(const double *xy, int n, double x, double y, double angle, Shape &out)
{
    std::vector<P2> loc(n);
    for (int i = 0; i < n; ++i) loc[i] = {xy[2 * i], xy[2 * i + 1]};
    out.verts = convex_hull(loc);
    set_planes(out);
    out.m_inv = 0.0; out.i_inv = 0.0; out.cog = {0, 0}; out.p = {x, y}; out.angle = angle;
}
// world vertices of a shape at its load-time pose (cpTransformPoint)
static std::vector<P2> world_verts(const Shape &s)
{
    double sn, cs;
    host_sincos(s.angle, sn, cs);
    const double tx = s.p.x - (s.cog.x * cs - s.cog.y * sn), ty = s.p.y - (s.cog.x * sn + s.cog.y * cs);
    std::vector<P2> w(s.verts.size());
    for (size_t i = 0; i < s.verts.size(); ++i)
        w[i] = P2{(cs * s.verts[i].x + (-sn) * s.verts[i].y) + tx, (sn * s.verts[i].x + cs * s.verts[i].y) + ty};
    return w;
}

// ---- rasters --------------------------------------------------------------------------------------------------------
static bool bd_clip_line(long long W, long long H, long long &x1, long long &y1, long long &x2, long long &y2)
{
    const long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1)); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
        if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1)); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) { a = c1 == 1 ? 0 : right; y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1)); x1 = a; c1 = 0; }
            if (c2) { a = c2 == 1 ? 0 : right; y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1)); x2 = a; c2 = 0; }
        }
    }
    return (c1 | c2) == 0;
}
// cv2.fillPoly for one polygon with integer vertices: 8-connected outline (cv::LineIterator) + scanline fill (16.16 edges)
static void bd_fill_poly(std::vector<unsigned char> &img, int H, int W, const long long *px, const long long *py, int n, unsigned char val)
{
    for (int e = 0; e < n; e++) {
        const int j = (e + n - 1) % n;
        long long x1 = px[j], y1 = py[j], x2 = px[e], y2 = py[e];
        if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
            (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
            if (!bd_clip_line(W, H, x1, y1, x2, y2)) continue;
        }
        long long dx = x2 - x1, dy = y2 - y1, sx0 = x1, sy0 = y1;
        if (dx < 0) { dx = -dx; dy = -dy; sx0 = x2; sy0 = y2; }
        long long sy = 1;
        if (dy < 0) { dy = -dy; sy = -1; }
        const bool vert = dy > dx;
        const long long dmaj = vert ? dy : dx, dmin = vert ? dx : dy;
        for (long long t = 0; t <= dmaj; t++) {
            const long long mt = dmaj == 0 ? 0 : (2 * dmin * t + dmaj - 1) / (2 * dmaj);
            const long long x = vert ? sx0 + mt : sx0 + t, y = vert ? sy0 + sy * t : sy0 + sy * mt;
            if (x >= 0 && x < W && y >= 0 && y < H) img[(size_t)y * W + x] = val;
        }
    }
    long long ymin = LLONG_MAX, ymax = LLONG_MIN;
    for (int i = 0; i < n; i++) { ymin = std::min(ymin, py[i]); ymax = std::max(ymax, py[i]); }
    if (ymax > H) ymax = H;
    for (long long y = ymin; y < ymax; y++) {
        long long xs[16]; int cnt = 0;
        for (int i = 0; i < n && cnt < 16; i++) {
            const int j = (i + n - 1) % n;
            const long long x0 = px[j] * 65536, x1 = px[i] * 65536, y0 = py[j], y1 = py[i];   // 16.16 fixed point (a left shift of a negative value is undefined before C++20)
            if (y0 == y1) continue;
            const long long edx = (x1 - x0) / (y1 - y0);
            long long ex, ey0, ey1;
            if (y0 < y1) { ey0 = y0; ey1 = y1; ex = x0; } else { ey0 = y1; ey1 = y0; ex = x1; }
            if (y < ey0 || y >= ey1) continue;
            xs[cnt++] = ex + (y - ey0) * edx;
        }
        if (y < 0) continue;
        std::sort(xs, xs + cnt);
        for (int a = 0; a + 1 < cnt; a += 2) {
            long long xl = (xs[a] + 65535) >> 16, xr = xs[a + 1] >> 16;
            if (xl < W && xr >= 0) {
                if (xl < 0) xl = 0;
                if (xr >= W) xr = W - 1;
                for (long long x = xl; x <= xr; x++) img[(size_t)y * W + x] = val;
            }
        }
    }
}
static void bd_dilate_disk(const std::vector<unsigned char> &img, int H, int W, int r, std::vector<unsigned char> &out)
{
    out.assign((size_t)H * W, 0);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            if (!img[(size_t)i * W + j]) continue;
            const bool interior = i > 0 && j > 0 && i < H - 1 && j < W - 1 && img[(size_t)(i - 1) * W + j] && img[(size_t)(i + 1) * W + j] &&
                                  img[(size_t)i * W + j - 1] && img[(size_t)i * W + j + 1];
            if (interior) { out[(size_t)i * W + j] = 1; continue; }
            for (int di = -r; di <= r; di++)
                for (int dj = -r; dj <= r; dj++) {
                    if (di * di + dj * dj > r * r) continue;
                    const int y = i + di, x = j + dj;
                    if (y < 0 || x < 0 || y >= H || x >= W) continue;
                    out[(size_t)y * W + x] = 1;
                }
        }
}
// nearest free cell of every cell (scipy.ndimage.distance_transform_edt(return_indices=True)); ties: smallest column, then row
static void bd_edt_indices(const std::vector<unsigned char> &freec, int H, int W, std::vector<int> &ii, std::vector<int> &jj)
{
    static const int DI[8] = {-1, -1, -1, 0, 1, 1, 1, 0}, DJ[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    std::vector<int> ci, cj;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            if (!freec[(size_t)i * W + j]) continue;
            bool edge = false;
            for (int k = 0; k < 8 && !edge; k++) {
                const int y = i + DI[k], x = j + DJ[k];
                if (y < 0 || x < 0 || y >= H || x >= W) continue;
                if (!freec[(size_t)y * W + x]) edge = true;
            }
            if (edge) { ci.push_back(i); cj.push_back(j); }
        }
    ii.assign((size_t)H * W, 0); jj.assign((size_t)H * W, 0);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const size_t v = (size_t)i * W + j;
            if (freec[v] || ci.empty()) { ii[v] = i; jj[v] = j; continue; }
            long long bd = LLONG_MAX; int bi = -1, bj = -1;
            for (size_t k = 0; k < ci.size(); k++) {
                const long long dy = ci[k] - i, dx = cj[k] - j, d = dy * dy + dx * dx;
                if (bi < 0 || d < bd || (d == bd && (cj[k] < bj || (cj[k] == bj && ci[k] < bi)))) { bd = d; bi = ci[k]; bj = cj[k]; }
            }
            ii[v] = bi; jj[v] = bj;
        }
}
// spfa.spfa distances (least fixed point of float32 relaxations over the 8-neighbour free-cell graph); unreachable -> 0
static void bd_spfa(const std::vector<unsigned char> &freec, int H, int W, int si, int sj, std::vector<float> &dist)
{
    static const int DI[8] = {-1, -1, -1, 0, 1, 1, 1, 0}, DJ[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    const float SQ2 = std::sqrt(2.0f);
    const size_t N = (size_t)H * W;
    const float inf = 2.0f * (float)N;
    dist.assign(N, inf);
    std::vector<int> queue(1, si * W + sj);
    std::vector<unsigned char> inq(N, 0);
    dist[(size_t)si * W + sj] = 0.0f;
    size_t head = 0;
    while (head < queue.size()) {
        const int u = queue[head++];
        inq[u] = 0;
        if (!freec[u]) continue;
        const int ui = u / W, uj = u % W;
        for (int k = 0; k < 8; k++) {
            const int vi = ui + DI[k], vj = uj + DJ[k];
            if (vi < 0 || vj < 0 || vi >= H || vj >= W) continue;
            const int v = vi * W + vj;
            if (!freec[v]) continue;
            const float nd = dist[u] + ((k & 1) ? 1.0f : SQ2);
            if (nd < dist[v]) { dist[v] = nd; if (!inq[v]) { inq[v] = 1; queue.push_back(v); } }
        }
        if (head > (1u << 20) && head * 2 > queue.size()) { queue.erase(queue.begin(), queue.begin() + head); head = 0; }
    }
    for (size_t i = 0; i < N; i++) if (!(dist[i] < inf - 1e-6f)) dist[i] = 0.0f;
}

struct BdMaps {
    int H, W, SH, SW, si0, sj0;
    std::vector<unsigned> free_bits, thin_bits;      // window
    std::vector<unsigned short> edt;                 // window [SH*SW][2] (window coordinates)
    std::vector<float> recept;                       // window, scaled
    int out_r = 0;                                   // dilation radius (area-clearing: closed form of the map outside the window)
    std::vector<unsigned char> small_free;           // window
};

// obstacle polygons in world coordinates (walls, columns, dividers, corner triangles) -> all static rasters of one layout
// area-clearing extras of bd_build_maps (task 1): boundary polygons, goal points, DISTANCE_SCALE_MAX
struct AcGeom { int nbd = 0, nob = 0, ngoal = 0; const double (*bd)[2] = nullptr; const double (*ob)[2] = nullptr; const double (*goals)[2] = nullptr; double scale_max = 0.5; };
static int ac_orient(const double (*p)[2], int n)
{
    double a = 0.0;
    for (int i = 0; i < n; i++) { const int j = (i + 1) % n; a += p[i][0] * p[j][1] - p[j][0] * p[i][1]; }
    return a > 0 ? 1 : -1;
}
static bool ac_contains_point(const double (*p)[2], int n, double x, double y) // shapely Polygon.contains(Point): strictly inside (convex)
{
    const int o = ac_orient(p, n);
    for (int i = 0; i < n; i++) {
        const int j = (i + 1) % n;
        const double cr = (p[j][0] - p[i][0]) * (y - p[i][1]) - (p[j][1] - p[i][1]) * (x - p[i][0]);
        if (!(cr * o > 0)) return false;
    }
    return true;
}

static bool bd_build_maps(const std::vector<std::vector<P2>> &obstacles, double room_length, double room_width, double ppm, int local_px,
                          double local_w, double robot_radius, double robot_half_width, double recept_x, double recept_y,
                          double sp_channel_scale, BdMaps &M, int task = 0, const AcGeom *G = nullptr, bool invert_recept = false)
{
    const double pad = (double)local_px * std::sqrt(2.0);
    M.H = (int)(2 * std::ceil((room_width * ppm + pad) / 2));
    M.W = (int)(2 * std::ceil((room_length * ppm + pad) / 2));
    M.SH = M.SW = local_px + 20;
    M.si0 = (int)((double)M.H / 2 - (double)M.SH / 2);
    M.sj0 = (int)((double)M.W / 2 - (double)M.SW / 2);
    const int H = M.H, W = M.W, SH = M.SH, SW = M.SW;
    std::vector<unsigned char> small((size_t)SH * SW, 0);
    const int off = (int)(local_w * ppm / 2) + 10;
    for (const auto &poly : obstacles) {
        long long px[8], py[8];
        const int n = (int)poly.size();
        if (n > 8) return false;
        for (int i = 0; i < n; i++) {
            long long ix = (long long)(int)(poly[i].x * ppm), iy = (long long)(int)(poly[i].y * ppm);
            ix += off; iy += off;
            iy = SH - iy;
            px[i] = ix; py[i] = iy;
        }
        bd_fill_poly(small, SH, SW, px, py, n, 1);
    }
    // box-delivery pads the room with obstacle, area-clearing with free space (area_clearing.py:1092)
    std::vector<unsigned char> obst((size_t)H * W, task == 1 ? 0 : 1), dil, freec((size_t)H * W), thin((size_t)H * W);
    for (int i = 0; i < SH; i++) memcpy(&obst[(size_t)(M.si0 + i) * W + M.sj0], &small[(size_t)i * SW], (size_t)SW);
    int rad = (int)std::floor(robot_radius * ppm), rad_thin = (int)std::floor(robot_half_width * ppm);
    if (task == 1) { const int rpw = (int)(2 * robot_radius * ppm); rad = rad_thin = (int)std::floor((double)rpw / 4); } // :1112-1117
    M.out_r = rad;
    bd_dilate_disk(obst, H, W, rad, dil);
    for (size_t i = 0; i < freec.size(); i++) freec[i] = !dil[i];
    bd_dilate_disk(obst, H, W, rad_thin, dil);
    for (size_t i = 0; i < thin.size(); i++) thin[i] = !dil[i];
    if (task == 1) {
        // the padding is free but cut off from the room by the outer walls: the window must be sealed by a ring of blocked cells
        for (int i = 0; i < SH; i++)
            for (int j = 0; j < SW; j++)
                if ((i == 0 || j == 0 || i == SH - 1 || j == SW - 1) && (freec[(size_t)(M.si0 + i) * W + M.sj0 + j] || thin[(size_t)(M.si0 + i) * W + M.sj0 + j])) return false;
    } else {
        // every free cell lies inside the window (the padding outside it is obstacle), so the window holds all that is needed
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                if ((freec[(size_t)i * W + j] || thin[(size_t)i * W + j]) && (i < M.si0 || i >= M.si0 + SH || j < M.sj0 || j >= M.sj0 + SW)) return false;
    }
    std::vector<int> ii, jj;
    bd_edt_indices(freec, H, W, ii, jj);
    const int words = (SH * SW + 31) / 32;
    M.free_bits.assign(words, 0u); M.thin_bits.assign(words, 0u);
    M.edt.assign((size_t)SH * SW * 2, 0);
    M.small_free.assign((size_t)SH * SW, 0);
    for (int i = 0; i < SH; i++)
        for (int j = 0; j < SW; j++) {
            const size_t g = (size_t)(M.si0 + i) * W + (M.sj0 + j);
            const int w = i * SW + j;
            if (freec[g]) M.free_bits[w >> 5] |= 1u << (w & 31);
            if (thin[g]) M.thin_bits[w >> 5] |= 1u << (w & 31);
            int a = ii[g] - M.si0, b = jj[g] - M.sj0;
            a = a < 0 ? 0 : (a > SH - 1 ? SH - 1 : a); b = b < 0 ? 0 : (b > SW - 1 ? SW - 1 : b);
            M.edt[(size_t)w * 2] = (unsigned short)a; M.edt[(size_t)w * 2 + 1] = (unsigned short)b;
            M.small_free[w] = small[w] ? 0 : 1;
        }
    if (task == 1) {
        // create_global_shortest_path_to_goal_points (area_clearing.py:1055-1082), float32 arithmetic as numpy does it
        const size_t N = (size_t)H * W;
        std::vector<float> g(N, INFINITY), img;
        const float ppm32 = (float)ppm;
        for (int q = 0; q < G->ngoal; q++) {
            long long gi = (long long)std::floor((double)H / 2 - G->goals[q][1] * ppm), gj = (long long)std::floor((double)W / 2 + G->goals[q][0] * ppm);
            gi = gi < 0 ? 0 : (gi > H - 1 ? H - 1 : gi); gj = gj < 0 ? 0 : (gj > W - 1 ? W - 1 : gj);
            bd_spfa(freec, H, W, ii[(size_t)gi * W + gj], jj[(size_t)gi * W + gj], img);
            for (size_t k = 0; k < N; k++) { const float v = img[k] / ppm32; if (v < g[k]) g[k] = v; }
        }
        const double div2 = (std::sqrt(2.0) * (double)local_px) / ppm;
        float mx = -INFINITY, mn = INFINITY;
        for (size_t k = 0; k < N; k++) { g[k] = (float)((double)g[k] / div2); if (g[k] > mx) mx = g[k]; if (g[k] < mn) mn = g[k]; }
        const float scale = (float)G->scale_max;
        for (size_t k = 0; k < N; k++) g[k] = (g[k] - mn) / (mx - mn) * scale;
        M.recept.assign((size_t)SH * SW, 0.0f);
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                const double x = ((double)j - (double)W / 2) / ppm, y = ((double)H / 2 - (double)i) / ppm;
                const size_t k = (size_t)i * W + j;
                if (!ac_contains_point(G->bd, G->nbd, x, y)) g[k] = 0.0f;
                if (!ac_contains_point(G->ob, G->nob, x, y)) g[k] = 1.0f;
                g[k] = g[k] + (1.0f - (freec[k] ? 1.0f : 0.0f));
                const bool inwin = i >= M.si0 && i < M.si0 + SH && j >= M.sj0 && j < M.sj0 + SW;
                if (inwin) M.recept[(size_t)(i - M.si0) * SW + (j - M.sj0)] = g[k];
                else {   // the kernels use a closed form outside the window: 1, and 2 within the dilation radius of the (all-wall) border
                    const int wi = i - M.si0, wj = j - M.sj0;
                    const int dy = wi < 0 ? -wi : (wi > SH - 1 ? wi - (SH - 1) : 0), dx = wj < 0 ? -wj : (wj > SW - 1 ? wj - (SW - 1) : 0);
                    const float expect = (dx * dx + dy * dy <= rad * rad) ? 2.0f : 1.0f;
                    if (g[k] != expect) return false;
                }
            }
        return true;
    }
    // receptacle map: spfa from the (snapped) receptacle cell, float32 scaling as numpy does it (box_delivery_env.py:1115-1129)
    long long ri = (long long)std::floor((double)H / 2 - recept_y * ppm), rj = (long long)std::floor((double)W / 2 + recept_x * ppm);
    ri = ri < 0 ? 0 : (ri > H - 1 ? H - 1 : ri); rj = rj < 0 ? 0 : (rj > W - 1 ? W - 1 : rj);
    const int s_i = ii[(size_t)ri * W + rj], s_j = jj[(size_t)ri * W + rj];
    std::vector<float> dist;
    bd_spfa(freec, H, W, s_i, s_j, dist);
    const float ppm32 = (float)ppm, scale32 = (float)sp_channel_scale;
    const double div2 = (std::sqrt(2.0) * (double)local_px) / ppm;
    M.recept.assign((size_t)SH * SW, 0.0f);
    for (int i = 0; i < SH; i++)
        for (int j = 0; j < SW; j++) {
            float v = dist[(size_t)(M.si0 + i) * W + (M.sj0 + j)] / ppm32;
            v = (float)((double)v / div2);
            v = v * scale32;
            if (invert_recept) { // cfg.env.invert_receptacle_map (box_delivery_env.py:1126-1128): + (1 - cspace), and cells equal to (1 - cspace) become 1
                const float inv = 1.0f - (freec[(size_t)(M.si0 + i) * W + (M.sj0 + j)] ? 1.0f : 0.0f);
                v = v + inv;
                if (v == inv) v = 1.0f;
            }
            M.recept[(size_t)i * SW + j] = v;
        }
    return true;
}

} // namespace bpgeom
