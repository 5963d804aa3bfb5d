/*
 * bp_oracle_bd.c -- CPU restatement ("oracle") of box-delivery-v0's env.step() path.
 * Included at the end of bp_oracle.c (one translation unit: it uses the Chipmunk restatement there).
 *
 * TEST INFRASTRUCTURE ONLY (see the header of bp_oracle.c).
 *
 * PARITY STATUS: "parity unpinned" for everything that lives in third-party packages absent from /root/reference and from
 * this image: pymunk/Chipmunk2D (physics, point queries), spfa (IvanIZ/spfa submodule, empty directory), cv2.fillPoly,
 * skimage.draw.line, skimage.measure.approximate_polygon, skimage.morphology.disk/binary_dilation.  Their published
 * algorithms are restated below from memory of the upstream sources.  PINNED against the real library in this container
 * (tests/test_bd_cpu.py): scipy.ndimage.rotate(order=0, reshape=True), scipy.ndimage.distance_transform_edt(return_indices),
 * scipy.ndimage.binary_dilation (same operator as skimage's), numpy float32 arithmetic of the observation channels; and against
 * golden vectors produced by the reference's own classes (tests/golden/make_golden_controller.py, tests/test_controller_golden.py):
 * DP.ideal_control / TargetCourse set-point logic, PositionController.get_waypoints_to_spatial_action (straight-line branch); and
 * by the reference's BoxDeliveryEnv / AreaClearingEnv classes themselves (tests/golden/make_golden_env_methods.py): episode
 * generators, robot_state_channel, get_local_map, execute_robot_path in free space.
 *
 * Reference call sites restated (paths relative to /root/reference/benchpush):
 *   BoxDeliveryEnv.step / reset                 environments/box_delivery/box_delivery_env.py:578-830
 *   collision handlers, boundary push-out       box_delivery_env.py:208-229,294-311
 *   execute_robot_path, until-still             box_delivery_env.py:891-1023
 *   PositionController                          common/controller/position_controller.py:56-181
 *   DP.ideal_control / TargetCourse             common/controller/dp.py:67-120,217-248
 *   configuration space, overhead map, obs      box_delivery_env.py:1045-1207
 *   shortest_path(+_distance)                   box_delivery_env.py:1209-1271
 *   bodies and shapes                           common/utils/sim_utils.py:20-160
 *
 * Deliberate, documented choices:
 *   - libm calls of the reference (np.arctan2, np.sin, np.cos) use the deterministic bp_atan2 / bp_sincos so that the
 *     GPU can match bit for bit; np.hypot and np.linalg.norm are sqrt(dx*dx + dy*dy).
 *   - spfa.spfa: distances are the least fixed point of float32 relaxations (what any SPFA order converges to);
 *     parents[v] = first neighbour u (spfa's direction order) with (float)(dist[u] + w) == dist[v].  The queue order of
 *     the C++ original (FIFO + small-label-first) can pick a different equal-cost parent; bd_spfa_queue() keeps that
 *     variant for comparison in tests.
 *   - heading action -> pixel index arithmetic is done in binary64 (the reference does it on a float32 numpy array).
 */

/* ---------------------------------------------------------------------------------------------
 * deterministic atan / atan2 (fdlibm s_atan.c / e_atan2.c restated; finite non-NaN inputs only)
 * ------------------------------------------------------------------------------------------- */
#include <limits.h>
static inline uint32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }

static double bp_atan(double x)
{
    static const double atanhi[] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double atanlo[] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double aT[] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                                9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                                4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    uint32_t hx = hi_word(x), ix = hx & 0x7fffffffu;
    int neg = (hx >> 31) != 0, id;
    if (ix >= 0x44100000u) return neg ? -(atanhi[3] + atanlo[3]) : (atanhi[3] + atanlo[3]);
    if (ix < 0x3fdc0000u) {
        if (ix < 0x3e200000u) return x;
        id = -1;
    } else {
        x = fabs(x);
        if (ix < 0x3ff30000u) {
            if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else {
            if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
            else { id = 3; x = -1.0 / x; }
        }
    }
    double z = x * x, w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return neg ? -z : z;
}

static double bp_atan2(double y, double x)
{
    static const double pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    uint32_t hx = hi_word(x), hy = hi_word(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (x == 1.0) return bp_atan(y);
    int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);
    if ((iy | lo_word(y)) == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi; default: return -pi; }
    }
    if ((ix | lo_word(x)) == 0) return (hy >> 31) ? -pi_o_2 : pi_o_2;
    int k = ((int)iy - (int)ix) >> 20;
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;
    else if ((hx >> 31) && k < -60) z = 0.0;
    else z = bp_atan(fabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

/* exact fmod for 0 <= a, 0 < b (every partial subtraction is exact), then python's sign rule: np.mod(a, b) */
static double bp_pymod(double a, double b)
{
    double r = fabs(a);
    while (r >= b) {
        double t = b;
        while (t + t <= r) t = t + t;
        r = r - t;
    }
    if (a < 0) r = -r;
    if (r != 0.0 && r < 0) r = r + b;
    return r;
}
static double bd_restrict_heading(double h) { return bp_pymod(h + M_PI, 2 * M_PI) - M_PI; } /* box_delivery_env.py:1319-1320 */
static double bd_heading_diff(double h1, double h2) { return bd_restrict_heading(h1 - h2); }
static double bd_dist2(double ax, double ay, double bx, double by)
{
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

/* ---------------------------------------------------------------------------------------------
 * raster / grid primitives
 * ------------------------------------------------------------------------------------------- */
/* cv2.fillPoly(img, [pts], color) for one polygon with int32 vertices (x, y), 8-connected outline + scanline fill
 * (OpenCV drawing.cpp: CollectPolyEdges draws every edge with Line(), FillEdgeCollection fills [ceil(xl), floor(xr)] with
 * 16.16 fixed-point edge x advanced by dx = (dx << 16) / dy per scanline). */
static void cv_line_f(float *img, int H, int W, long x1, long y1, long x2, long y2, float val)
{
    /* cv::LineIterator, 8-connected (same stepping as cv_line in bp_oracle.c, float image) */
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
static void bd_fill_poly(float *img, int H, int W, int n, const long *px, const long *py, float color)
{
    /* outline: CollectPolyEdges draws edge (v[i-1], v[i]) with Line() */
    for (int i = 0; i < n; i++) {
        int j = (i + n - 1) % n;
        cv_line_f(img, H, W, px[j], py[j], px[i], py[i], color);
    }
    /* interior: even-odd pairing of the active edges on every scanline y in [y0, y1) of each edge */
    long ymin = LONG_MAX, ymax = LONG_MIN;
    for (int i = 0; i < n; i++) { if (py[i] < ymin) ymin = py[i]; if (py[i] > ymax) ymax = py[i]; }
    if (ymax > H) ymax = H;
    for (long y = ymin; y < ymax; y++) {
        long long xs[16]; int cnt = 0;
        for (int i = 0; i < n && cnt < 16; i++) {
            int j = (i + n - 1) % n;
            long long x0 = (long long)px[j] << 16, x1 = (long long)px[i] << 16;
            long y0 = py[j], y1 = py[i];
            if (y0 == y1) continue;
            long long ex; long ey0, ey1; long long edx = (x1 - x0) / (long long)(y1 - y0);
            if (y0 < y1) { ey0 = y0; ey1 = y1; ex = x0; } else { ey0 = y1; ey1 = y0; ex = x1; }
            if (y < ey0 || y >= ey1) continue;
            xs[cnt++] = ex + (long long)(y - ey0) * edx;
        }
        if (y < 0) continue;
        for (int a = 1; a < cnt; a++) { long long v = xs[a]; int b = a - 1; while (b >= 0 && xs[b] > v) { xs[b + 1] = xs[b]; b--; } xs[b + 1] = v; }
        for (int a = 0; a + 1 < cnt; a += 2) {
            long xl = (long)((xs[a] + 65535) >> 16), xr = (long)(xs[a + 1] >> 16);
            if (xl < W && xr >= 0) {
                if (xl < 0) xl = 0;
                if (xr >= W) xr = W - 1;
                for (long x = xl; x <= xr; x++) img[y * W + x] = color;
            }
        }
    }
}

/* skimage.draw.line(r0, c0, r1, c1) (skimage/draw/_draw.pyx:_line): returns the number of points */
static int bd_sk_line(long r0, long c0, long r1, long c1, long *rr, long *cc)
{
    int steep = 0;
    long r = r0, c = c0, dr = labs(r1 - r0), dc = labs(c1 - c0);
    long sc = (c1 - c) > 0 ? 1 : -1, sr = (r1 - r) > 0 ? 1 : -1;
    if (dr > dc) { steep = 1; long t = c; c = r; r = t; t = dc; dc = dr; dr = t; t = sc; sc = sr; sr = t; }
    long d = 2 * dr - dc;
    for (long i = 0; i < dc; i++) {
        if (steep) { rr[i] = c; cc[i] = r; } else { rr[i] = r; cc[i] = c; }
        while (d >= 0) { r += sr; d -= 2 * dc; }
        c += sc; d += 2 * dr;
    }
    rr[dc] = r1; cc[dc] = c1;
    return (int)dc + 1;
}

/* skimage.measure.approximate_polygon(coords, tolerance) (Douglas-Peucker, skimage/measure/_polygon.py): keep[] flags */
static void bd_approx_polygon(int n, const long *cr, const long *cc, double tol, unsigned char *keep)
{
    memset(keep, 0, (size_t)n);
    if (n == 0) return;
    keep[0] = 1; keep[n - 1] = 1;
    int *stack = (int *)malloc(sizeof(int) * 2 * (size_t)(n + 2));
    int sp = 0;
    stack[sp++] = 0; stack[sp++] = n - 1;
    while (sp > 0) {
        int end = stack[--sp], start = stack[--sp];
        long r0 = cr[start], c0 = cc[start], r1 = cr[end], c1 = cc[end];
        long dr = r1 - r0, dc = c1 - c0;
        double ang = -bp_atan2((double)dr, (double)dc);
        double sa, ca; bp_sincos(ang, &sa, &ca);
        double seg_dist = (double)c0 * sa + (double)r0 * ca;
        double best = -1.0; int arg = -1; int any = 0;
        for (int k = start + 1; k < end; k++) {
            long dr0 = cr[k] - r0, dc0 = cc[k] - c0, dr1 = cr[k] - r1, dc1 = cc[k] - c1;
            long pl0 = dr0 * dr + dc0 * dc, pl1 = -dr1 * dr - dc1 * dc;
            double d;
            if (pl0 > 0 && pl1 > 0) d = fabs(((double)cr[k] * ca + (double)cc[k] * sa) - seg_dist);
            else d = fmin(sqrt((double)(dc0 * dc0 + dr0 * dr0)), sqrt((double)(dc1 * dc1 + dr1 * dr1)));
            if (d > tol) any = 1;
            if (d > best) { best = d; arg = k; }
        }
        if (any) {
            stack[sp++] = arg; stack[sp++] = end;
            stack[sp++] = start; stack[sp++] = arg;
            keep[arg] = 1;
        }
    }
    free(stack);
}

/* spfa.spfa(map, source): 8-neighbour grid graph over cells with map != 0, float32 edge lengths (1, sqrtf(2)).
 * dist = least fixed point of float32 relaxations; unreachable cells report 0 (the original multiplies by
 * (dist < inf - eps)); parents by the local rule in the header, -1 for none.  Returns dist of the farthest cell. */
static const int BD_DI[8] = {-1, -1, -1, 0, 1, 1, 1, 0}, BD_DJ[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
static void bd_spfa_queue(const float *map, int H, int W, int si, int sj, float *dist, int *parent_queue_order)
{
    const float SQ2 = sqrtf(2.0f);
    const float len[8] = {SQ2, 1, SQ2, 1, SQ2, 1, SQ2, 1};
    size_t N = (size_t)H * W;
    const float inf = 2.0f * (float)N;
    for (size_t i = 0; i < N; i++) { dist[i] = inf; if (parent_queue_order) parent_queue_order[i] = -1; }
    size_t qcap = 8 * N + 2;
    int *queue = (int *)malloc(sizeof(int) * qcap);
    unsigned char *inq = (unsigned char *)calloc(N, 1);
    size_t head = 0, tail = 0;
    int s = si * W + sj;
    dist[s] = 0;
    queue[++tail] = s; inq[s] = 1;
    while (head < tail) {
        int u = queue[++head];
        inq[u] = 0;
        int ui = u / W, uj = u % W;
        if (map[u] == 0.0f) continue;
        for (int k = 0; k < 8; k++) {
            int vi = ui + BD_DI[k], vj = uj + BD_DJ[k];
            if (vi < 0 || vj < 0 || vi >= H || vj >= W) continue;
            int v = vi * W + vj;
            if (map[v] == 0.0f) continue;
            float nd = dist[u] + len[k];
            if (nd < dist[v]) {
                dist[v] = nd;
                if (parent_queue_order) parent_queue_order[v] = u;
                if (!inq[v]) {
                    queue[++tail] = v; inq[v] = 1;
                    if (dist[queue[tail]] < dist[queue[head + 1]]) { int t = queue[tail]; queue[tail] = queue[head + 1]; queue[head + 1] = t; }
                }
            }
        }
    }
    for (size_t i = 0; i < N; i++) if (!(dist[i] < inf - 1e-6f)) dist[i] = 0.0f;
    free(queue); free(inq);
}
static void bd_spfa_parents(const float *map, int H, int W, int si, int sj, const float *dist, int *parent)
{
    const float SQ2 = sqrtf(2.0f);
    const float len[8] = {SQ2, 1, SQ2, 1, SQ2, 1, SQ2, 1};
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            int v = i * W + j;
            parent[v] = -1;
            if (map[v] == 0.0f || (i == si && j == sj)) continue;
            if (dist[v] == 0.0f) continue; /* unreachable */
            for (int k = 0; k < 8; k++) {
                int ui = i - BD_DI[k], uj = j - BD_DJ[k];
                if (ui < 0 || uj < 0 || ui >= H || uj >= W) continue;
                int u = ui * W + uj;
                if (map[u] == 0.0f) continue;
                if (!(ui == si && uj == sj) && dist[u] == 0.0f) continue;
                if ((float)(dist[u] + len[k]) == dist[v]) { parent[v] = u; break; }
            }
        }
}
void orc_bd_spfa(const float *map, int H, int W, int si, int sj, float *dist, int *parent, int *parent_queue)
{
    bd_spfa_queue(map, H, W, si, sj, dist, parent_queue);
    if (parent) bd_spfa_parents(map, H, W, si, sj, dist, parent);
}

/* skimage.morphology.binary_dilation(img, disk(r)) == scipy.ndimage.binary_dilation(img, structure=disk) (border 0) */
static void bd_dilate_disk(const float *img, int H, int W, int r, float *out)
{
    memset(out, 0, sizeof(float) * (size_t)H * W);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            if (img[(size_t)i * W + j] == 0.0f) continue;
            /* only boundary cells of the set need to stamp the disk */
            int interior = i > 0 && j > 0 && i < H - 1 && j < W - 1 && img[(size_t)(i - 1) * W + j] != 0.0f && img[(size_t)(i + 1) * W + j] != 0.0f &&
                           img[(size_t)i * W + j - 1] != 0.0f && img[(size_t)i * W + j + 1] != 0.0f;
            if (interior) { out[(size_t)i * W + j] = 1.0f; continue; }
            for (int di = -r; di <= r; di++)
                for (int dj = -r; dj <= r; dj++) {
                    if (di * di + dj * dj > r * r) continue;
                    int y = i + di, x = j + dj;
                    if (y < 0 || x < 0 || y >= H || x >= W) continue;
                    out[(size_t)y * W + x] = 1.0f;
                }
        }
}
void orc_bd_dilate(const float *img, int H, int W, int r, float *out) { bd_dilate_disk(img, H, W, r, out); }

/* scipy.ndimage.distance_transform_edt(1 - cspace, return_distances=False, return_indices=True): for every cell the
 * (row, col) of the nearest free cell (cspace != 0); ties resolved like scipy's feature transform (see bd_edt_better). */
static int bd_edt_better(long d, int i, int j, long bd, int bi, int bj)
{
    if (d != bd) return d < bd;
    if (j != bj) return j < bj; /* ties: smallest column, then smallest row (pinned against scipy 1.15 on random maps) */
    return i < bi;
}
static void bd_edt_indices(const float *cspace, int H, int W, int *idx_i, int *idx_j)
{
    /* candidates: free cells with a non-free 8-neighbour (the nearest free cell of a blocked cell is always one of them) */
    int nc = 0, cap = 1024;
    int *ci = (int *)malloc(sizeof(int) * cap), *cj = (int *)malloc(sizeof(int) * cap);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            if (cspace[(size_t)i * W + j] == 0.0f) continue;
            int edge = 0;
            for (int k = 0; k < 8 && !edge; k++) {
                int y = i + BD_DI[k], x = j + BD_DJ[k];
                if (y < 0 || x < 0 || y >= H || x >= W) continue;
                if (cspace[(size_t)y * W + x] == 0.0f) edge = 1;
            }
            if (!edge) continue;
            if (nc == cap) { cap *= 2; ci = (int *)realloc(ci, sizeof(int) * cap); cj = (int *)realloc(cj, sizeof(int) * cap); }
            ci[nc] = i; cj[nc] = j; nc++;
        }
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            size_t v = (size_t)i * W + j;
            if (cspace[v] != 0.0f || nc == 0) { idx_i[v] = i; idx_j[v] = j; continue; }
            long bd = LONG_MAX; int bi = -1, bj = -1;
            for (int k = 0; k < nc; k++) {
                long dy = ci[k] - i, dx = cj[k] - j, d = dy * dy + dx * dx;
                if (bi < 0 || bd_edt_better(d, ci[k], cj[k], bd, bi, bj)) { bd = d; bi = ci[k]; bj = cj[k]; }
            }
            idx_i[v] = bi; idx_j[v] = bj;
        }
    free(ci); free(cj);
}
void orc_bd_edt(const float *cspace, int H, int W, int *idx_i, int *idx_j) { bd_edt_indices(cspace, H, W, idx_i, idx_j); }

/* scipy.ndimage.rotate(img, angle_deg, order=0) (reshape=True, mode='constant', cval=0) restated: affine_transform with
 * rot = [[c, s], [-s, c]]; output shape = int(ptp(rot @ corners) + 0.5); nearest sample floor(x + 0.5).
 * (c, s) = cos/sin of the angle, taken from bp_sincos by the caller.  Returns the output shape in *oh, *ow. */
static void bd_rotate0_shape(int ih, int iw, double c, double s, int *oh, int *ow)
{
    /* out_bounds = rot @ [[0, 0, iy, iy], [0, ix, 0, ix]] */
    double b0[4] = {c * 0 + s * 0, c * 0 + s * iw, c * ih + s * 0, c * ih + s * iw};
    double b1[4] = {-s * 0 + c * 0, -s * 0 + c * iw, -s * ih + c * 0, -s * ih + c * iw};
    double mn0 = b0[0], mx0 = b0[0], mn1 = b1[0], mx1 = b1[0];
    for (int k = 1; k < 4; k++) { if (b0[k] < mn0) mn0 = b0[k]; if (b0[k] > mx0) mx0 = b0[k]; if (b1[k] < mn1) mn1 = b1[k]; if (b1[k] > mx1) mx1 = b1[k]; }
    *oh = (int)((mx0 - mn0) + 0.5);
    *ow = (int)((mx1 - mn1) + 0.5);
}
static inline float bd_rotate0_sample(const float *img, int ih, int iw, double c, double s, int oh, int ow, int oi, int oj)
{
    double oc0 = ((double)oh - 1) / 2, oc1 = ((double)ow - 1) / 2;
    double ic0 = ((double)ih - 1) / 2, ic1 = ((double)iw - 1) / 2;
    double off0 = ic0 - (c * oc0 + s * oc1), off1 = ic1 - (-s * oc0 + c * oc1);
    double c0 = 0.0, c1 = 0.0;
    c0 += (double)oi * c; c0 += (double)oj * s; c0 += off0;
    c1 += (double)oi * -s; c1 += (double)oj * c; c1 += off1;
    /* NI_GeometricTransform, order 0, mode constant: coordinate outside [0, n-1] -> cval (pinned against scipy 1.15) */
    if (c0 < 0 || c0 > ih - 1 || c1 < 0 || c1 > iw - 1) return 0.0f;
    long s0 = (long)floor(c0 + 0.5), s1 = (long)floor(c1 + 0.5);
    return img[(size_t)s0 * iw + s1];
}
void orc_bd_rotate0(const float *img, int ih, int iw, double c, double s, int *oh, int *ow, float *out)
{
    bd_rotate0_shape(ih, iw, c, s, oh, ow);
    if (!out) return;
    for (int i = 0; i < *oh; i++)
        for (int j = 0; j < *ow; j++) out[(size_t)i * *ow + j] = bd_rotate0_sample(img, ih, iw, c, s, *oh, *ow, i, j);
}

/* ---------------------------------------------------------------------------------------------
 * environment
 * ------------------------------------------------------------------------------------------- */
#define BD_MAXWP 64
typedef struct bd_params {
    double room_length, room_width;        /* config.yaml env.room_length / room_width_small|large */
    double recept_x, recept_y, recept_size;/* get_receptacle_position_and_size, box_delivery_env.py:322-324 */
    double ppm;                            /* local_map_pixels_per_meter = 224 / 10 */
    int    local_px;                       /* local_map_pixel_width 224 */
    double local_w;                        /* local_map_width 10 */
    double robot_radius, robot_half_width; /* box_delivery_env.py:122-123 */
    double step_size;                      /* agent.step_size 1.75 */
    double target_speed, ctrl_dt;          /* controller.target_speed 0.3, controller.dt 0.2 */
    int    steps;                          /* sim.steps 100 -> sub-step dt = ctrl_dt / steps */
    double partial_rewards_scale, goal_reward, collision_penalty, non_movement_penalty, correct_direction_reward_scale;
    int    use_correct_direction_reward;
    int    inactivity_cutoff;
    double ministep_size;
    double sp_channel_scale;               /* env.shortest_path_channel_scale 0.25 */
    int    invert_receptacle_map;
    int    num_boxes;
    int    step_limit;                     /* STEP_LIMIT 10000 */
    int    action_type;                    /* agent.action_type: 0 heading, 1 position, 2 velocity */
    /* shared controller knobs (box-delivery: 3, 2, 0; area-clearing: 0.5, 5, controller.Lfc) */
    int    task;                           /* 0 box-delivery-v0, 1 area-clearing-v0 (environments/area_clearing/area_clearing.py) */
    double omega_scale, v_scale, lfc;      /* apply_controller factors (box_delivery_env.py:887-889 / area_clearing.py:903-906), DP Lfc */
    /* area-clearing only */
    double yaw_rate_step;                  /* max_yaw_rate_step (pi/2)/15, area_clearing.py:198 */
    int    t_max;                          /* sim.t_max */
    double boundary_penalty, box_cleared_reward, box_putback_penalty, truncation_penalty, terminal_reward, pushing_mult, distance_scale_max;
} bd_params;

typedef struct bd_env {
    orc_env *E;
    bd_params B;
    int H, W, SH, SW;                      /* padded room / small map sizes */
    float *small_obstacle_free;            /* self.small_obstacle_map (1 = free, 0 = wall) */
    float *cspace, *cspace_thin;
    int *edt_i, *edt_j;
    float *recept_map;                     /* create_global_shortest_path_to_receptacle_map() */
    float *overhead, *robot_map, *tmpf; int *tmpp;
    float *robot_state_channel;
    int nbox, first_box_shape, nstatic, first_static_shape;
    int *box_alive;                        /* self.boxes membership per box index */
    int *box_order; int nalive;            /* self.boxes list order (indices into boxes) */
    double *prev_boxes;                    /* world verts of prev_boxes list [nalive_prev][4][2] */
    int nprev;
    double *box_dist;                      /* cached final_box_distances per box idx */
    int inactivity, cum_boxes; double cum_distance, cum_reward, total_work;
    int *cleared;
    /* last step diagnostics */
    long last_substeps; int last_nwp; double last_wp[BD_MAXWP][3];
    /* area-clearing */
    int nbd, nob, ngoal; double bd_poly[16][2], ob_poly[16][2], goals[128][2], footprint[4][2];
    int cleared_count, t; double *sprop;
} bd_env;

enum { BD_I_X = 0, BD_I_Y, BD_I_THETA, BD_I_CUM_DIST, BD_I_CUM_BOXES, BD_I_CUM_REWARD, BD_I_TOTAL_WORK, BD_I_MINISTEPS, BD_I_INACTIVITY,
       BD_I_HIT, BD_I_SUBSTEPS, BD_I_ROBOT_DIST, BD_I_BOXES_DIST, BD_I_NWP, BD_I_NALIVE, BD_I_WORK, BD_I_COUNT };

static int bd_round_up_even(double x) { return (int)(ceil(x / 2) * 2); }

bd_env *orc_bd_create(const orc_params *P, const bd_params *B)
{
    bd_env *D = (bd_env *)calloc(1, sizeof(bd_env));
    D->E = orc_create(P);
    D->E->kind = 2;
    D->B = *B;
    /* create_padded_room_zeros (box_delivery_env.py:1103-1107) */
    double pad = (double)B->local_px * sqrt(2.0);
    D->H = (int)(2 * ceil((B->room_width * B->ppm + pad) / 2));
    D->W = (int)(2 * ceil((B->room_length * B->ppm + pad) / 2));
    D->SH = D->SW = B->local_px + 20;
    size_t N = (size_t)D->H * D->W;
    D->small_obstacle_free = (float *)calloc((size_t)D->SH * D->SW, sizeof(float));
    D->cspace = (float *)calloc(N, sizeof(float)); D->cspace_thin = (float *)calloc(N, sizeof(float));
    D->edt_i = (int *)calloc(N, sizeof(int)); D->edt_j = (int *)calloc(N, sizeof(int));
    D->recept_map = (float *)calloc(N, sizeof(float)); D->overhead = (float *)calloc(N, sizeof(float));
    D->robot_map = (float *)calloc(N, sizeof(float)); D->tmpf = (float *)calloc(N, sizeof(float)); D->tmpp = (int *)calloc(N, sizeof(int));
    /* robot_state_channel: circular mask (box_delivery_env.py:124-131) */
    int lp = B->local_px;
    D->robot_state_channel = (float *)calloc((size_t)lp * lp, sizeof(float));
    int rpw = (int)(2 * B->robot_radius * B->ppm);
    int start = (int)floor((double)lp / 2 - (double)rpw / 2);
    for (int i = start; i < start + rpw; i++)
        for (int j = start; j < start + rpw; j++) {
            double a = ((double)i + 0.5) - (double)lp / 2, b = ((double)j + 0.5) - (double)lp / 2;
            if (sqrt(a * a + b * b) < (double)rpw / 2) D->robot_state_channel[(size_t)i * lp + j] = 1.0f;
        }
    return D;
}
void orc_bd_destroy(bd_env *D)
{
    if (!D) return;
    orc_destroy(D->E);
    free(D->small_obstacle_free); free(D->cspace); free(D->cspace_thin); free(D->edt_i); free(D->edt_j); free(D->recept_map);
    free(D->overhead); free(D->robot_map); free(D->tmpf); free(D->tmpp); free(D->robot_state_channel);
    free(D->box_alive); free(D->box_order); free(D->prev_boxes); free(D->box_dist); free(D->cleared);
    free(D);
}
void orc_bd_dims(const bd_env *D, int *out) { out[0] = D->H; out[1] = D->W; out[2] = D->SH; out[3] = D->SW; }

/* position_to_pixel_indices / pixel_indices_to_position (box_delivery_env.py:1325-1335) */
static void bd_pos_to_pix(const bd_env *D, double x, double y, int *pi, int *pj)
{
    long i = (long)floor((double)D->H / 2 - y * D->B.ppm), j = (long)floor((double)D->W / 2 + x * D->B.ppm);
    if (i < 0) i = 0; if (i > D->H - 1) i = D->H - 1;
    if (j < 0) j = 0; if (j > D->W - 1) j = D->W - 1;
    *pi = (int)i; *pj = (int)j;
}
static void bd_pix_to_pos(const bd_env *D, int i, int j, double *x, double *y)
{
    *x = ((double)j - (double)D->W / 2) / D->B.ppm;
    *y = ((double)D->H / 2 - (double)i) / D->B.ppm;
}

/* world vertices of a shape's hull, local_to_world on get_vertices() -> int pixel coords of the small map
 * (box_delivery_env.py:1151-1158,1188-1195) */
static void bd_box_world_verts(const orc_env *E, const shape_t *sh, vec *out);
static void bd_shape_px(const bd_env *D, const shape_t *sh, long *px, long *py)
{
    int off = (int)(D->B.local_w * D->B.ppm / 2) + 10;
    vec wv[ORC_MAXV];
    bd_box_world_verts(D->E, sh, wv);
    for (int i = 0; i < sh->n; i++) {
        double vx = wv[i].x * D->B.ppm, vy = wv[i].y * D->B.ppm;
        long ix = (long)(int)vx, iy = (long)(int)vy; /* astype(np.int32): truncation toward zero */
        ix += off; iy += off;
        iy = D->SH - iy;
        px[i] = ix; py[i] = iy;
    }
}

/* update_configuration_space (box_delivery_env.py:1140-1175) */
static void bd_update_cspace(bd_env *D)
{
    orc_env *E = D->E;
    size_t N = (size_t)D->H * D->W, SN = (size_t)D->SH * D->SW;
    float *small = (float *)calloc(SN, sizeof(float));
    float *obst = D->tmpf;
    /* box-delivery pads with obstacle (create_padded_room_ones), area-clearing with free space (area_clearing.py:1092) */
    for (size_t i = 0; i < N; i++) obst[i] = D->B.task == 1 ? 0.0f : 1.0f;
    for (int k = 0; k < D->nstatic; k++) {
        const shape_t *sh = &E->shapes[D->first_static_shape + k];
        if (sh->ctype != 3) continue;
        long px[ORC_MAXV], py[ORC_MAXV];
        bd_shape_px(D, sh, px, py);
        bd_fill_poly(small, D->SH, D->SW, sh->n, px, py, 1.0f);
    }
    int si = (int)((double)D->H / 2 - (double)D->SH / 2), sj = (int)((double)D->W / 2 - (double)D->SW / 2);
    for (int i = 0; i < D->SH; i++)
        for (int j = 0; j < D->SW; j++) obst[(size_t)(si + i) * D->W + (sj + j)] = small[(size_t)i * D->SW + j];
    float *dil = (float *)calloc(N, sizeof(float));
    int rad = (int)floor(D->B.robot_radius * D->B.ppm), rad_thin = (int)floor(D->B.robot_half_width * D->B.ppm);
    if (D->B.task == 1) { /* area_clearing.py:1112-1117: disk(floor(robot_pixel_width / 4)) for both */
        int rpw = (int)(2 * D->B.robot_radius * D->B.ppm);
        rad = rad_thin = (int)floor((double)rpw / 4);
    }
    bd_dilate_disk(obst, D->H, D->W, rad, dil);
    for (size_t i = 0; i < N; i++) D->cspace[i] = 1.0f - dil[i];
    bd_dilate_disk(obst, D->H, D->W, rad_thin, dil);
    for (size_t i = 0; i < N; i++) D->cspace_thin[i] = 1.0f - dil[i];
    bd_edt_indices(D->cspace, D->H, D->W, D->edt_i, D->edt_j);
    for (size_t i = 0; i < SN; i++) D->small_obstacle_free[i] = 1.0f - small[i];
    free(small); free(dil);
}

/* shortest_path (box_delivery_env.py:1209-1264 == position_controller.py:125-178): returns the number of waypoints
 * written to wp[][2] (<= BD_MAXWP) */
static int bd_shortest_path(bd_env *D, double sx, double sy, double tx, double ty, int check_straight, double wp[][2])
{
    int H = D->H, W = D->W;
    int si, sj, ti, tj;
    bd_pos_to_pix(D, sx, sy, &si, &sj);
    bd_pos_to_pix(D, tx, ty, &ti, &tj);
    long *rr = (long *)malloc(sizeof(long) * (size_t)(H + W + 4)), *cc = (long *)malloc(sizeof(long) * (size_t)(H + W + 4));
    if (check_straight) {
        int n = bd_sk_line(si, sj, ti, tj, rr, cc);
        int blocked = 0;
        for (int k = 0; k < n; k++) if (D->cspace_thin[(size_t)rr[k] * W + cc[k]] == 0.0f) { blocked = 1; break; }
        if (!blocked) { wp[0][0] = sx; wp[0][1] = sy; wp[1][0] = tx; wp[1][1] = ty; free(rr); free(cc); return 2; }
    }
    { int a = D->edt_i[(size_t)si * W + sj], b = D->edt_j[(size_t)si * W + sj]; si = a; sj = b; }
    { int a = D->edt_i[(size_t)ti * W + tj], b = D->edt_j[(size_t)ti * W + tj]; ti = a; tj = b; }
    float *dist = (float *)malloc(sizeof(float) * (size_t)H * W);
    int *par = D->tmpp;
    bd_spfa_queue(D->cspace, H, W, si, sj, dist, NULL);
    bd_spfa_parents(D->cspace, H, W, si, sj, dist, par);
    /* recover dense path target -> source */
    int cap = 1024, n = 0;
    long *pr = (long *)malloc(sizeof(long) * cap), *pc = (long *)malloc(sizeof(long) * cap);
    int i = ti, j = tj;
    pr[n] = i; pc[n] = j; n++;
    while (!(i == si && j == sj)) {
        int p = par[(size_t)i * W + j];
        if (p < 0) break; /* parents_ij = [-1, -1] -> i + j < 0 */
        i = p / W; j = p % W;
        if (n == cap) { cap *= 2; pr = (long *)realloc(pr, sizeof(long) * cap); pc = (long *)realloc(pc, sizeof(long) * cap); }
        pr[n] = i; pc[n] = j; n++;
    }
    unsigned char *keep = (unsigned char *)malloc((size_t)n);
    bd_approx_polygon(n, pr, pc, 1.0, keep);
    int m = 0;
    for (int k = 0; k < n; k++) if (keep[k]) { pr[m] = pr[k]; pc[m] = pc[k]; m++; }
    /* remove unnecessary waypoints */
    long *nr = (long *)malloc(sizeof(long) * (size_t)(m + 1)), *ncl = (long *)malloc(sizeof(long) * (size_t)(m + 1));
    int q = 0;
    nr[q] = pr[0]; ncl[q] = pc[0]; q++;
    for (int k = 1; k < m - 1; k++) {
        int ln = bd_sk_line(nr[q - 1], ncl[q - 1], pr[k + 1], pc[k + 1], rr, cc);
        int blocked = 0;
        for (int t = 0; t < ln; t++) if (D->cspace[(size_t)rr[t] * W + cc[t]] == 0.0f) { blocked = 1; break; }
        if (blocked) { nr[q] = pr[k]; ncl[q] = pc[k]; q++; }
    }
    if (m > 1) { nr[q] = pr[m - 1]; ncl[q] = pc[m - 1]; q++; }
    int nw;
    if (q < 2) { wp[0][0] = sx; wp[0][1] = sy; wp[1][0] = tx; wp[1][1] = ty; nw = 2; }
    else {
        if (q > BD_MAXWP) q = BD_MAXWP; /* capacity of the restatement; never reached in the supported rooms */
        for (int k = 0; k < q; k++) bd_pix_to_pos(D, (int)nr[q - 1 - k], (int)ncl[q - 1 - k], &wp[k][0], &wp[k][1]);
        wp[0][0] = sx; wp[0][1] = sy; wp[q - 1][0] = tx; wp[q - 1][1] = ty;
        nw = q;
    }
    free(rr); free(cc); free(dist); free(pr); free(pc); free(keep); free(nr); free(ncl);
    return nw;
}
static double bd_shortest_path_distance(bd_env *D, double sx, double sy, double tx, double ty)
{
    double wp[BD_MAXWP][2];
    int n = bd_shortest_path(D, sx, sy, tx, ty, 0, wp);
    double sum = 0.0;
    for (int i = 1; i < n; i++) sum += bd_dist2(wp[i - 1][0], wp[i - 1][1], wp[i][0], wp[i][1]);
    return sum;
}
int orc_bd_shortest_path(bd_env *D, const double *s, const double *t, int check_straight, double *out)
{
    double wp[BD_MAXWP][2];
    int n = bd_shortest_path(D, s[0], s[1], t[0], t[1], check_straight, wp);
    for (int i = 0; i < n; i++) { out[2 * i] = wp[i][0]; out[2 * i + 1] = wp[i][1]; }
    return n;
}

/* global shortest-path maps (box_delivery_env.py:1115-1138): float32 arithmetic as numpy does it in this image
 * (float32 array op python float -> float32; float32 array op np.float64 scalar -> computed in binary64, stored float32) */
static void bd_scale_sp_map(const bd_env *D, float *m)
{
    size_t N = (size_t)D->H * D->W;
    float ppm32 = (float)D->B.ppm;
    double div2 = (sqrt(2.0) * (double)D->B.local_px) / D->B.ppm;
    float scale32 = (float)D->B.sp_channel_scale;
    for (size_t i = 0; i < N; i++) {
        float v = m[i] / ppm32;
        v = (float)((double)v / div2);
        if (D->B.task != 1) v = v * scale32; /* area-clearing has no shortest_path_channel_scale (area_clearing.py:1046-1053) */
        m[i] = v;
    }
}
static void bd_recept_map(bd_env *D)
{
    int i, j;
    bd_pos_to_pix(D, D->B.recept_x, D->B.recept_y, &i, &j);
    { int a = D->edt_i[(size_t)i * D->W + j], b = D->edt_j[(size_t)i * D->W + j]; i = a; j = b; }
    bd_spfa_queue(D->cspace, D->H, D->W, i, j, D->recept_map, NULL);
    bd_scale_sp_map(D, D->recept_map);
    if (D->B.invert_receptacle_map) {
        size_t N = (size_t)D->H * D->W;
        for (size_t k = 0; k < N; k++) {
            float inv = 1.0f - D->cspace[k];
            float v = D->recept_map[k] + inv;
            if (v == inv) v = 1.0f;
            D->recept_map[k] = v;
        }
    }
}

/* shapely predicates for the convex polygons of the shipped area-clearing layouts (restated; shapely/GEOS is absent):
 * Polygon.contains(Point): strictly inside; Polygon.intersects(Polygon): the closed sets share a point (separating axis) */
static int ac_orient(int n, const double (*p)[2])
{
    double a = 0.0;
    for (int i = 0; i < n; i++) { int j = (i + 1) % n; a += p[i][0] * p[j][1] - p[j][0] * p[i][1]; }
    return a > 0 ? 1 : -1;
}
static int ac_contains_point(int n, const double (*p)[2], double x, double y)
{
    int o = ac_orient(n, p);
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n;
        double cr = (p[j][0] - p[i][0]) * (y - p[i][1]) - (p[j][1] - p[i][1]) * (x - p[i][0]);
        if (!(cr * o > 0)) return 0;
    }
    return 1;
}
static int ac_sep_axis(int na, const double (*a)[2], int nb, const double (*b)[2])
{
    int o = ac_orient(na, a);
    for (int i = 0; i < na; i++) { /* is some edge of a a separating line: all of b strictly outside */
        int j = (i + 1) % na;
        int all_out = 1;
        for (int k = 0; k < nb && all_out; k++) {
            double cr = (a[j][0] - a[i][0]) * (b[k][1] - a[i][1]) - (a[j][1] - a[i][1]) * (b[k][0] - a[i][0]);
            if (!(cr * o < 0)) all_out = 0;
        }
        if (all_out) return 1;
    }
    return 0;
}
static int ac_intersects(int na, const double (*a)[2], int nb, const double (*b)[2])
{
    return !(ac_sep_axis(na, a, nb, b) || ac_sep_axis(nb, b, na, a));
}

/* create_global_shortest_path_to_goal_points (area_clearing.py:1055-1082) into D->recept_map */
static void ac_goal_map(bd_env *D)
{
    size_t N = (size_t)D->H * D->W;
    float *g = D->recept_map;
    for (size_t k = 0; k < N; k++) g[k] = INFINITY;
    float *img = (float *)malloc(sizeof(float) * N);
    float ppm32 = (float)D->B.ppm;
    for (int q = 0; q < D->ngoal; q++) {
        int i, j;
        bd_pos_to_pix(D, D->goals[q][0], D->goals[q][1], &i, &j);
        { int a = D->edt_i[(size_t)i * D->W + j], b = D->edt_j[(size_t)i * D->W + j]; i = a; j = b; }
        bd_spfa_queue(D->cspace, D->H, D->W, i, j, img, NULL);
        for (size_t k = 0; k < N; k++) { float v = img[k] / ppm32; if (v < g[k]) g[k] = v; }
    }
    double div2 = (sqrt(2.0) * (double)D->B.local_px) / D->B.ppm;
    float mx = -INFINITY, mn = INFINITY;
    for (size_t k = 0; k < N; k++) { g[k] = (float)((double)g[k] / div2); if (g[k] > mx) mx = g[k]; if (g[k] < mn) mn = g[k]; }
    float scale = (float)D->B.distance_scale_max;
    for (size_t k = 0; k < N; k++) g[k] = (g[k] - mn) / (mx - mn) * scale;
    for (int i = 0; i < D->H; i++)
        for (int j = 0; j < D->W; j++) {
            double x, y; bd_pix_to_pos(D, i, j, &x, &y);
            size_t k = (size_t)i * D->W + j;
            if (!ac_contains_point(D->nbd, D->bd_poly, x, y)) g[k] = 0.0f;
            if (!ac_contains_point(D->nob, D->ob_poly, x, y)) g[k] = 1.0f;
        }
    for (size_t k = 0; k < N; k++) g[k] = g[k] + (1.0f - D->cspace[k]);
    free(img);
}
void orc_ac_set_geometry(bd_env *D, int nbd, const double *bd, int nob, const double *ob, int ng, const double *goals, const double *footprint,
                         const double *sprop)
{
    D->nbd = nbd; D->nob = nob; D->ngoal = ng;
    for (int i = 0; i < nbd; i++) { D->bd_poly[i][0] = bd[2 * i]; D->bd_poly[i][1] = bd[2 * i + 1]; }
    for (int i = 0; i < nob; i++) { D->ob_poly[i][0] = ob[2 * i]; D->ob_poly[i][1] = ob[2 * i + 1]; }
    for (int i = 0; i < ng; i++) { D->goals[i][0] = goals[2 * i]; D->goals[i][1] = goals[2 * i + 1]; }
    for (int i = 0; i < 4; i++) { D->footprint[i][0] = footprint[2 * i]; D->footprint[i][1] = footprint[2 * i + 1]; }
    free(D->sprop); D->sprop = NULL;
    (void)sprop;
}

/* area-clearing update_global_overhead_map (area_clearing.py:968-1026) */
static void bd_shape_px(const bd_env *D, const shape_t *sh, long *px, long *py);
static void ac_world_px(const bd_env *D, int n, const double (*w)[2], long *px, long *py)
{
    int off = (int)(D->B.local_w * D->B.ppm / 2) + 10;
    for (int i = 0; i < n; i++) {
        long ix = (long)(int)(w[i][0] * D->B.ppm), iy = (long)(int)(w[i][1] * D->B.ppm);
        ix += off; iy += off; iy = D->SH - iy;
        px[i] = ix; py[i] = iy;
    }
}
static void ac_update_overhead(bd_env *D)
{
    orc_env *E = D->E;
    size_t SN = (size_t)D->SH * D->SW;
    float *small = (float *)malloc(sizeof(float) * SN);
    memcpy(small, D->small_obstacle_free, sizeof(float) * SN);
    for (size_t q = 0; q < SN; q++) if (small[q] == 1.0f) small[q] = 1.0f / 8.0f;
    long px[ORC_MAXV], py[ORC_MAXV];
    double il = fabs(D->bd_poly[0][0]) * 2, iw = fabs(D->bd_poly[0][1]) * 2;
    double th = fabs(D->ob_poly[0][0]) - fabs(D->bd_poly[0][0]);
    double rects[4][4] = {{-il / 2 - th / 2, 0, th, iw}, {il / 2 + th / 2, 0, th, iw}, {0, -iw / 2 - th / 2, il + 2 * th, th}, {0, iw / 2 + th / 2, il + 2 * th, th}};
    for (int r = 0; r < 4; r++) {
        double x = rects[r][0], y = rects[r][1], l = rects[r][2], w = rects[r][3];
        double poly[4][2] = {{x - l / 2, y - w / 2}, {x + l / 2, y - w / 2}, {x + l / 2, y + w / 2}, {x - l / 2, y + w / 2}};
        ac_world_px(D, 4, poly, px, py);
        bd_fill_poly(small, D->SH, D->SW, 4, px, py, 3.0f / 8.0f);
    }
    for (int k = 0; k < D->nbox; k++) {
        const shape_t *sh = &E->shapes[D->first_box_shape + k];
        bd_shape_px(D, sh, px, py);
        bd_fill_poly(small, D->SH, D->SW, sh->n, px, py, D->cleared[k] ? 7.0f / 8.0f : 4.0f / 8.0f);
    }
    {
        const body_t *b = &E->bodies[0];
        double poly[4][2];
        for (int i = 0; i < 4; i++) {
            poly[i][0] = b->ta * D->footprint[i][0] + b->tc * D->footprint[i][1] + b->tx;
            poly[i][1] = b->tb * D->footprint[i][0] + b->td * D->footprint[i][1] + b->ty;
        }
        ac_world_px(D, 4, poly, px, py);
        bd_fill_poly(small, D->SH, D->SW, 4, px, py, 5.0f / 8.0f);
    }
    int si = (int)((double)D->H / 2 - (double)D->SH / 2), sj = (int)((double)D->W / 2 - (double)D->SW / 2);
    for (int i = 0; i < D->SH; i++)
        for (int j = 0; j < D->SW; j++) D->overhead[(size_t)(si + i) * D->W + (sj + j)] = small[(size_t)i * D->SW + j];
    free(small);
}

/* update_global_overhead_map (box_delivery_env.py:1177-1207) */
static void bd_update_overhead(bd_env *D)
{
    orc_env *E = D->E;
    size_t SN = (size_t)D->SH * D->SW;
    float *small = (float *)malloc(sizeof(float) * SN);
    memcpy(small, D->small_obstacle_free, sizeof(float) * SN);
    int drew = 0;
#define BD_FLOOR() do { for (size_t q = 0; q < SN; q++) if (small[q] == 1.0f) small[q] = 1.0f / 8.0f; } while (0)
    long px[ORC_MAXV], py[ORC_MAXV];
    for (int k = 0; k < D->nstatic; k++) {
        const shape_t *sh = &E->shapes[D->first_static_shape + k];
        if (sh->ctype != 4) continue;
        BD_FLOOR(); drew = 1;
        bd_shape_px(D, sh, px, py);
        bd_fill_poly(small, D->SH, D->SW, sh->n, px, py, 3.0f / 8.0f);
    }
    for (int q = 0; q < D->nalive; q++) {
        const shape_t *sh = &E->shapes[D->first_box_shape + D->box_order[q]];
        BD_FLOOR(); drew = 1;
        bd_shape_px(D, sh, px, py);
        bd_fill_poly(small, D->SH, D->SW, sh->n, px, py, 4.0f / 8.0f);
    }
    {
        const shape_t *sh = &E->shapes[0];
        BD_FLOOR(); drew = 1;
        bd_shape_px(D, sh, px, py);
        bd_fill_poly(small, D->SH, D->SW, sh->n, px, py, 6.0f / 8.0f);
    }
    (void)drew;
    int si = (int)((double)D->H / 2 - (double)D->SH / 2), sj = (int)((double)D->W / 2 - (double)D->SW / 2);
    for (int i = 0; i < D->SH; i++)
        for (int j = 0; j < D->SW; j++) D->overhead[(size_t)(si + i) * D->W + (sj + j)] = small[(size_t)i * D->SW + j];
    free(small);
}

/* get_local_map (box_delivery_env.py:1078-1096): local [lp][lp] float32 */
static void bd_local_map(const bd_env *D, const float *gmap, double rx, double ry, double rh, float *local)
{
    int lp = D->B.local_px;
    int cw = bd_round_up_even((double)lp * sqrt(2.0));
    double rot_deg = 90 - rh * (180.0 / M_PI); /* np.degrees */
    (void)rot_deg;
    int pi = (int)floor(-ry * D->B.ppm + (double)D->H / 2), pj = (int)floor(rx * D->B.ppm + (double)D->W / 2);
    /* crop = global_map[pi - cw//2 : pi + cw//2, pj - cw//2 : pj + cw//2] (numpy slicing clamps at the array bounds) */
    int i0 = pi - cw / 2, i1 = pi + cw / 2, j0 = pj - cw / 2, j1 = pj + cw / 2;
    if (i0 < 0) i0 = 0; if (j0 < 0) j0 = 0; if (i1 > D->H) i1 = D->H; if (j1 > D->W) j1 = D->W;
    int ch = i1 - i0, cwid = j1 - j0;
    if (ch < 0) ch = 0; if (cwid < 0) cwid = 0;
    float *crop = (float *)malloc(sizeof(float) * (size_t)(ch > 0 ? ch : 1) * (size_t)(cwid > 0 ? cwid : 1));
    for (int i = 0; i < ch; i++) memcpy(crop + (size_t)i * cwid, gmap + (size_t)(i0 + i) * D->W + j0, sizeof(float) * (size_t)cwid);
    /* rotation angle in radians = radians(90 - degrees(h)); the deterministic restatement uses (pi/2 - h) directly */
    double s, c;
    bp_sincos(M_PI / 2 - rh, &s, &c);
    int oh, ow;
    bd_rotate0_shape(ch, cwid, c, s, &oh, &ow);
    int a0 = oh / 2 - lp / 2, b0 = ow / 2 - lp / 2;
    for (int i = 0; i < lp; i++)
        for (int j = 0; j < lp; j++) {
            int oi = a0 + i, oj = b0 + j;
            float v = 0.0f;
            if (oi >= 0 && oi < oh && oj >= 0 && oj < ow) v = bd_rotate0_sample(crop, ch, cwid, c, s, oh, ow, oi, oj);
            local[(size_t)i * lp + j] = v;
        }
    free(crop);
}

/* generate_observation (box_delivery_env.py:1045-1059): uint8 [lp][lp][4], channels last */
void orc_bd_observe(bd_env *D, uint8_t *obs)
{
    orc_env *E = D->E;
    const body_t *rb = &E->bodies[0];
    int lp = D->B.local_px;
    size_t LN = (size_t)lp * lp;
    if (D->B.task == 1) ac_update_overhead(D); else bd_update_overhead(D);
    float *ch = (float *)malloc(sizeof(float) * LN);
    /* 0: overhead */
    bd_local_map(D, D->overhead, rb->p.x, rb->p.y, rb->a, ch);
    for (size_t k = 0; k < LN; k++) obs[4 * k + 0] = (uint8_t)(ch[k] * 255.0f);
    /* 1: robot state */
    for (size_t k = 0; k < LN; k++) obs[4 * k + 1] = (uint8_t)(D->robot_state_channel[k] * 255.0f);
    /* 2: shortest path from the robot */
    {
        int i, j;
        bd_pos_to_pix(D, rb->p.x, rb->p.y, &i, &j);
        { int a = D->edt_i[(size_t)i * D->W + j], b = D->edt_j[(size_t)i * D->W + j]; i = a; j = b; }
        bd_spfa_queue(D->cspace, D->H, D->W, i, j, D->robot_map, NULL);
        bd_scale_sp_map(D, D->robot_map);
        bd_local_map(D, D->robot_map, rb->p.x, rb->p.y, rb->a, ch);
        float mn = ch[0];
        for (size_t k = 1; k < LN; k++) if (ch[k] < mn) mn = ch[k];
        for (size_t k = 0; k < LN; k++) obs[4 * k + 2] = (uint8_t)((int)((ch[k] - mn) * 255.0f) & 0xFF); /* numpy astype(uint8): truncate, wrap mod 256 */
    }
    /* 3: shortest path to the receptacle */
    {
        bd_local_map(D, D->recept_map, rb->p.x, rb->p.y, rb->a, ch);
        float mn = ch[0];
        for (size_t k = 1; k < LN; k++) if (ch[k] < mn) mn = ch[k];
        for (size_t k = 0; k < LN; k++) obs[4 * k + 3] = (uint8_t)((int)((ch[k] - mn) * 255.0f) & 0xFF);
    }
    free(ch);
}

/* cpPolyShapePointQuery + cpSpacePointQuery(maxDistance 0): is p strictly inside shape (distance - r < 0) */
static int bd_point_in_shape(const shape_t *sh, vec p)
{
    if (!(sh->bl <= p.x && p.x <= sh->br && sh->bb <= p.y && p.y <= sh->bt)) return 0; /* cpBBNewForCircle(p, 0) vs shape bb */
    int count = sh->n;
    vec v0 = sh->wv[count - 1];
    double minDist = INFINITY; int outside = 0;
    for (int i = 0; i < count; i++) {
        vec v1 = sh->wv[i];
        outside = outside || (vdot(sh->wn[i], vsub(p, v1)) > 0.0);
        /* cpClosetPointOnSegment(p, v0, v1) */
        vec delta = vsub(v0, v1);
        double t = fclamp01(vdot(delta, vsub(p, v1)) / vdot(delta, delta));
        vec closest = vadd(v1, vmult(delta, t));
        double d = vlength(vsub(p, closest));
        if (d < minDist) minDist = d;
        v0 = v1;
    }
    double dist = outside ? minDist : -minDist;
    return (dist - sh->r) < 0.0;
}

static void bd_box_world_verts(const orc_env *E, const shape_t *sh, vec *out)
{
    const body_t *b = &E->bodies[sh->body];
    for (int i = 0; i < sh->n; i++) /* body.local_to_world(v): cpTransformPoint */
        out[i] = V(b->ta * sh->lv[i].x + b->tc * sh->lv[i].y + b->tx, b->tb * sh->lv[i].x + b->td * sh->lv[i].y + b->ty);
}

/* reset: robot (1 body, 6 shapes), boxes, statics.
 * robot_verts[4][2], wheel_verts[4][4][2], bumper_verts[4][2]; boxes[nbox][3] = x, y, heading; half = box_size / 2;
 * statics: sverts[ns][4][2] (count scount[ns] in {3,4}), spose[ns][3] (body x, y, angle), srad[ns], stype[ns] (3 | 4) */
int orc_bd_reset(bd_env *D, const double *start, const double *robot_verts, const double *wheel_verts, const double *bumper_verts,
                 int nbox, const double *boxes, double half, double density,
                 int ns, const double *sverts, const int *scount, const double *spose, const double *srad, const int *stype)
{
    orc_env *E = D->E;
    free(E->bodies); free(E->shapes); free(E->order); free(E->prev_wv); free(E->used); free(E->removed); E->used = NULL;
    int nb = 1 + nbox + ns, nsh = 6 + nbox + ns;
    E->bodies = (body_t *)calloc((size_t)nb, sizeof(body_t));
    E->shapes = (shape_t *)calloc((size_t)nsh, sizeof(shape_t));
    E->order = (int *)calloc((size_t)nsh, sizeof(int));
    E->prev_wv = NULL;
    E->removed = (unsigned char *)calloc((size_t)nsh, 1);
    E->narb = 0; E->nactive = 0; E->stamp = 0; E->curr_dt = 0.0; E->nevents = 0; E->robot_hit = 0;
    E->total_work = 0.0;
    D->nbox = nbox; D->first_box_shape = 6; D->nstatic = ns; D->first_static_shape = 6 + nbox;
    /* robot: create_agent (sim_utils.py:20-73) */
    {
        body_t *b = &E->bodies[0];
        b->type = BODY_KINEMATIC; b->m = b->i = INFINITY; b->m_inv = b->i_inv = 0.0;
        b->p = V(start[0], start[1]); b->a = start[2]; b->cog = V(0, 0);
        body_set_transform(b);
        vec tmp[8], hull[8];
        for (int i = 0; i < 4; i++) tmp[i] = V(robot_verts[2 * i], robot_verts[2 * i + 1]);
        int hn = convex_hull(4, tmp, hull);
        vec cog = centroid_for_poly(hn, hull);
        for (int i = 0; i < 4; i++) tmp[i] = V(robot_verts[2 * i] - cog.x, robot_verts[2 * i + 1] - cog.y);
        hn = convex_hull(4, tmp, hull);
        shape_t *s = &E->shapes[0];
        s->body = 0; s->r = 0.0; s->e = 0.01; s->u = 1.0; s->ctype = 1;
        shape_set_verts(s, hn, hull);
        for (int k = 0; k < 5; k++) {
            const double *src = (k < 4) ? wheel_verts + 8 * k : bumper_verts;
            for (int i = 0; i < 4; i++) tmp[i] = V(src[2 * i], src[2 * i + 1]);
            hn = convex_hull(4, tmp, hull);
            s = &E->shapes[1 + k];
            s->body = 0; s->r = 0.02; s->e = 0.01; s->u = 0.0; s->ctype = 0;
            shape_set_verts(s, hn, hull);
        }
    }
    /* boxes: generate_sim_boxes -> create_polygon (sim_utils.py:120-160) */
    for (int k = 0; k < nbox; k++) {
        double ox = boxes[3 * k], oy = boxes[3 * k + 1], oh = boxes[3 * k + 2];
        double raw[8] = {ox + half, oy + half, ox - half, oy + half, ox - half, oy - half, ox + half, oy - half};
        vec tmp[4], hull[4];
        for (int i = 0; i < 4; i++) tmp[i] = V(raw[2 * i] - ox, raw[2 * i + 1] - oy);
        int hn = convex_hull(4, tmp, hull);
        vec cog = centroid_for_poly(hn, hull);
        for (int i = 0; i < 4; i++) tmp[i] = V(tmp[i].x - cog.x, tmp[i].y - cog.y);
        hn = convex_hull(4, tmp, hull);
        int bi = 1 + k, si = 6 + k;
        shape_t *s = &E->shapes[si];
        body_t *b = &E->bodies[bi];
        s->body = bi; s->r = 0.02; s->e = 0.01; s->u = 1.0; s->ctype = 2;
        shape_set_verts(s, hn, hull);
        vec scog = centroid_for_poly(hn, hull);
        double area = area_for_poly(hn, hull, s->r);
        double m = density * area;
        double ipm = moment_for_poly(1.0, hn, hull, vneg(scog));
        b->type = BODY_DYNAMIC;
        double bm = 0.0, bI = 0.0; vec bc = V(0, 0);
        double msum = bm + m;
        bI += m * ipm + vdot(vsub(bc, scog), vsub(bc, scog)) * (m * bm) / msum;
        bc = vlerp(bc, scog, m / msum);
        bm = msum;
        b->m = bm; b->i = bI; b->cog = bc; b->m_inv = 1.0 / bm; b->i_inv = 1.0 / bI;
        b->a = oh;
        double sn, cs; bp_sincos(oh, &sn, &cs);
        b->p = vadd(V(bc.x * cs - bc.y * sn, bc.x * sn + bc.y * cs), V(ox, oy));
        body_set_transform(b);
    }
    /* statics: create_static / create_corners (sim_utils.py:75-135) */
    for (int k = 0; k < ns; k++) {
        int bi = 1 + nbox + k, si = 6 + nbox + k;
        body_t *b = &E->bodies[bi];
        b->type = BODY_STATIC; b->m = b->i = INFINITY; b->m_inv = b->i_inv = 0.0; b->cog = V(0, 0);
        b->p = V(spose[3 * k], spose[3 * k + 1]); b->a = spose[3 * k + 2];
        body_set_transform(b);
        vec tmp[4], hull[4];
        for (int i = 0; i < scount[k]; i++) tmp[i] = V(sverts[(size_t)k * 8 + 2 * i], sverts[(size_t)k * 8 + 2 * i + 1]);
        int hn = convex_hull(scount[k], tmp, hull);
        shape_t *s = &E->shapes[si];
        s->body = bi; s->r = srad[k]; s->e = 0.01; s->u = 1.0; s->ctype = stype[k];
        if (D->B.task == 1) { s->e = 0.0; s->u = 0.99; } /* area_clearing.py:446-448,472-474: pymunk default elasticity 0, friction 0.99 */
        shape_set_verts(s, hn, hull);
    }
    E->nb = nb; E->ns = nsh;
    for (int i = 0; i < E->ns; i++) E->order[i] = i;
    for (int s = 0; s < E->ns; s++) shape_cache_bb(&E->shapes[s], &E->bodies[E->shapes[s].body]);
    /* episode state */
    free(D->box_alive); free(D->box_order); free(D->prev_boxes); free(D->box_dist); free(D->cleared);
    D->box_alive = (int *)calloc((size_t)nbox + 1, sizeof(int)); D->box_order = (int *)calloc((size_t)nbox + 1, sizeof(int));
    D->prev_boxes = (double *)calloc((size_t)(nbox + 1) * 8, sizeof(double)); D->box_dist = (double *)calloc((size_t)nbox + 1, sizeof(double));
    D->cleared = (int *)calloc((size_t)nbox + 1, sizeof(int));
    for (int k = 0; k < nbox; k++) { D->box_alive[k] = 1; D->box_order[k] = k; }
    D->nalive = nbox;
    D->inactivity = 0; D->cum_boxes = 0; D->cum_distance = 0.0; D->cum_reward = 0.0; D->total_work = 0.0;
    bd_update_cspace(D);
    if (D->B.task == 1) ac_goal_map(D); else bd_recept_map(D);
    D->cleared_count = 0; D->t = 0;
    double dts = D->B.ctrl_dt / D->B.steps;
    for (int k = 0; k < 1000; k++) space_step(E, dts); /* box_delivery_env.py:284-285 */
    /* prev_boxes = CostMap.get_obs_from_poly(self.boxes) */
    D->nprev = D->nalive;
    for (int q = 0; q < D->nalive; q++) {
        const shape_t *sh = &E->shapes[D->first_box_shape + D->box_order[q]];
        vec wv[4]; bd_box_world_verts(E, sh, wv);
        for (int i = 0; i < 4; i++) { D->prev_boxes[8 * q + 2 * i] = wv[i].x; D->prev_boxes[8 * q + 2 * i + 1] = wv[i].y; }
    }
    return nsh;
}

/* test hook: robot pose after every sim step of execute_robot_path (from step index orc_bd_trace_from on) */
double *orc_bd_trace = NULL; long orc_bd_trace_from = 0, orc_bd_trace_n = 0, orc_bd_trace_cap = 0;
void orc_bd_set_trace(double *buf, long from, long cap) { orc_bd_trace = buf; orc_bd_trace_from = from; orc_bd_trace_n = 0; orc_bd_trace_cap = cap; }
long orc_bd_trace_count(void) { return orc_bd_trace_n; }

/* DP controller state (dp.py): only what ideal_control / advance use */
typedef struct { int valid; double cx[2], cy[2]; double plen; double al; double spx, spy; } bd_dp;

static void bd_dp_init(bd_dp *dp, double x, double y, const double wp[][3], double lfc)
{
    dp->valid = 1;
    dp->cx[0] = wp[0][0]; dp->cx[1] = wp[1][0]; dp->cy[0] = wp[0][1]; dp->cy[1] = wp[1][1];
    double dx = wp[1][0] - wp[0][0], dy = wp[1][1] - wp[0][1];
    dp->plen = sqrt(dx * dx + dy * dy); /* path_length(cumsum=True) of two points: one element */
    /* TargetCourse.init_setpoint: nearest of the two points (np.hypot + argmin), Lfc = 0 */
    double d0 = bd_dist2(x, y, dp->cx[0], dp->cy[0]), d1 = bd_dist2(x, y, dp->cx[1], dp->cy[1]);
    int ind = (d1 < d0) ? 1 : 0;
    /* look-ahead (dp.py:78-83); with Lfc == 0 (box-delivery) the loop never runs */
    while (lfc > bd_dist2(x, y, dp->cx[ind], dp->cy[ind])) { if (ind + 1 >= 2) break; ind++; }
    dp->al = dp->plen; /* path_length[min(len - 1, ind)] with len == 1 */
    dp->spx = dp->cx[ind]; dp->spy = dp->cy[ind];
}

/* PositionController.get_waypoints_to_spatial_action (position_controller.py:56-123): waypoints + headings for a local-map pixel */
static int bd_plan(bd_env *D, int x_pixel, int y_pixel, double ix, double iy, double ih, double wpp[][2], double *wph, double *move_sign_out)
{
    const bd_params *B = &D->B;
    int nwp; double move_sign;
        double xm = -B->local_w / 2 + (double)x_pixel / B->ppm;
        double ym = B->local_w / 2 - (double)y_pixel / B->ppm;
        double sld = sqrt(xm * xm + ym * ym);
        double turn = bp_atan2(-xm, ym);
        double slh = bd_restrict_heading(ih + turn);
        double sh_, ch_; bp_sincos(slh, &sh_, &ch_);
        double tx = ix + sld * ch_, ty = iy + sld * sh_;
        double dfx = tx - ix, dfy = ty - iy;
        double ratio_x = 1, ratio_y = 1;
        double sgx = (tx > 0) - (tx < 0), sgy = (ty > 0) - (ty < 0);
        double bound_x = sgx * B->room_length / 2, bound_y = sgy * B->room_width / 2;
        if (fabs(tx) > fabs(bound_x)) ratio_x = (bound_x - ix) / (tx - ix);
        if (fabs(ty) > fabs(bound_y)) ratio_y = (bound_y - iy) / (ty - iy);
        double ratio = ratio_y < ratio_x ? ratio_y : ratio_x; /* python min(a, b): b if b < a else a */
        tx = ix + ratio * dfx; ty = iy + ratio * dfy;
        nwp = bd_shortest_path(D, ix, iy, tx, ty, 1, wpp);
        wph[0] = 0.0; /* None */
        for (int i = 1; i < nwp; i++)
            wph[i] = bd_restrict_heading(bp_atan2(wpp[i][1] - wpp[i - 1][1], wpp[i][0] - wpp[i - 1][0]));
        double dte = bd_dist2(wpp[nwp - 2][0], wpp[nwp - 2][1], wpp[nwp - 1][0], wpp[nwp - 1][1]);
        double signed_dist = dte - B->robot_radius;
        move_sign = (signed_dist > 0) - (signed_dist < 0);
        if (nwp > 2 && signed_dist < 0) {
            wpp[nwp - 2][0] = wpp[nwp - 1][0]; wpp[nwp - 2][1] = wpp[nwp - 1][1];
            wph[nwp - 2] = bd_restrict_heading(bp_atan2(wpp[nwp - 2][1] - wpp[nwp - 3][1], wpp[nwp - 2][0] - wpp[nwp - 3][0]));
            move_sign = 1;
        }
    *move_sign_out = move_sign;
    return nwp;
}
int orc_bd_plan(bd_env *D, int x_pixel, int y_pixel, const double *pose, double *out, double *move_sign)
{
    double wpp[BD_MAXWP][2], wph[BD_MAXWP];
    int n = bd_plan(D, x_pixel, y_pixel, pose[0], pose[1], pose[2], wpp, wph, move_sign);
    for (int i = 0; i < n; i++) { out[3 * i] = wpp[i][0]; out[3 * i + 1] = wpp[i][1]; out[3 * i + 2] = wph[i]; }
    return n;
}

/* DP.ideal_control + get_setpoint (dp.py:217-248,194-204,104-107) for the current pose; returns omega and the global velocity */
static void bd_dp_control(bd_dp *dp, double x, double y, double h, double target_speed, double ctrl_dt, double *omega, double *gvx, double *gvy)
{
    double theta_d = bp_atan2(dp->spy - y, dp->spx - x);
    double theta_e = theta_d - h;
    double se, ce; bp_sincos(theta_e, &se, &ce);
    theta_e = bp_atan2(se, ce);
    double om = 1.0 * theta_e;
    om = om / ctrl_dt;
    double sy_, cy_; bp_sincos(h, &sy_, &cy_);
    *gvx = cy_ * target_speed + -sy_ * 0.0; *gvy = sy_ * target_speed + cy_ * 0.0;
    *omega = om;
    dp->al += target_speed * ctrl_dt; /* TargetCourse.advance(target_speed, dt) */
    { int ind = (dp->plen < dp->al) ? 1 : 0; dp->spx = dp->cx[ind]; dp->spy = dp->cy[ind]; }
}
/* test hook: run the controller over a recorded pose sequence; out[n][5] = omega, vx, vy, setpoint x, setpoint y (after the call) */
void orc_bd_controller_trace(const double *wp2, double lfc, double target_speed, double ctrl_dt, int n, const double *poses, double *out)
{
    bd_dp dp; dp.valid = 0;
    double two[2][3] = {{wp2[0], wp2[1], 0}, {wp2[2], wp2[3], 0}};
    for (int k = 0; k < n; k++) {
        double x = poses[3 * k], y = poses[3 * k + 1], h = poses[3 * k + 2];
        if (!dp.valid) bd_dp_init(&dp, x, y, two, lfc);
        bd_dp_control(&dp, x, y, h, target_speed, ctrl_dt, &out[5 * k], &out[5 * k + 1], &out[5 * k + 2]);
        out[5 * k + 3] = dp.spx; out[5 * k + 4] = dp.spy;
    }
}

/* execute_robot_path (box_delivery_env.py:891-988 / area_clearing.py:800-901): returns the distance credited to the robot */
static double bd_execute_path(bd_env *D, double wpp[][2], const double *wph, int nwp, double ix, double iy, double ih, long *total_sub_io)
{
    orc_env *E = D->E;
    const bd_params *B = &D->B;
    body_t *rb = &E->bodies[0];
    double dts = B->ctrl_dt / B->steps;
    double robot_distance = 0.0;
    long total_sub = *total_sub_io;
        double px = ix, py = iy, ph = ih;
        int wi = 1, path0 = 0; /* path0: index of self.path[0] in the waypoint list (self.path = self.path[1:]) */
        double pwx = wpp[0][0], pwy = wpp[0][1];
        double prev_hd = 0.0; int done_turning = 0; long sim_steps = 0;
        bd_dp dp; dp.valid = 0;
        for (;;) {
            double prevx = px, prevy = py, prevh = ph;
            double hd = bd_heading_diff(ph, wph[wi]);
            if (fabs(hd) > 15 * (M_PI / 180.0) && fabs(hd - prev_hd) > 0.001) { /* TURN_STEP_SIZE = np.radians(15) */ }
            else done_turning = 1;
            /* controller (box_delivery_env.py:867-885) */
            if (!dp.valid) {
                double two[2][3] = {{wpp[path0][0], wpp[path0][1], 0}, {wpp[path0 + 1][0], wpp[path0 + 1][1], 0}};
                bd_dp_init(&dp, prevx, prevy, two, B->lfc);
            }
            double omega, gvx, gvy;
            bd_dp_control(&dp, prevx, prevy, prevh, B->target_speed, B->ctrl_dt, &omega, &gvx, &gvy);
            /* apply_controller */
            rb->w = omega * B->omega_scale;   /* omega*3 | omega/2 */
            if (!done_turning) rb->v = V((gvx * 0) * B->v_scale, (gvy * 0) * B->v_scale);
            else rb->v = V(gvx * B->v_scale, gvy * B->v_scale);
            space_step(E, dts);
            total_sub++;
            px = rb->p.x; py = rb->p.y; ph = bd_restrict_heading(rb->a);
            if (orc_bd_trace && sim_steps >= orc_bd_trace_from && orc_bd_trace_n < orc_bd_trace_cap) {
                double *o = orc_bd_trace + 4 * (size_t)orc_bd_trace_n++;
                o[0] = px; o[1] = py; o[2] = rb->a; o[3] = (double)E->nactive;
            }
            prev_hd = hd;
            if (bd_dist2(pwx, pwy, px, py) > 0.05) { /* MOVE_STEP_SIZE */
                if (E->robot_hit) break;
            }
            if (bd_dist2(px, py, wpp[wi][0], wpp[wi][1]) < 0.6 && fabs(ph - wph[wi]) < 10 * (M_PI / 180.0)) {
                robot_distance += bd_dist2(pwx, pwy, px, py);
                if (wi == nwp - 1) break;
                wi++;
                pwx = wpp[wi - 1][0]; pwy = wpp[wi - 1][1];
                done_turning = 0; dp.valid = 0; path0++;
            }
            sim_steps++;
            if (sim_steps > B->step_limit) break;
        }
    *total_sub_io = total_sub;
    return robot_distance;
}
/* test hook: run the path execution for given waypoints [n][3] = x, y, heading from the robot's current pose;
 * out[6] = robot_distance, turn angle, x, y, angle, sim steps */
void orc_bd_execute_path(bd_env *D, int n, const double *wp, double *out)
{
    orc_env *E = D->E;
    body_t *rb = &E->bodies[0];
    double wpp[BD_MAXWP][2], wph[BD_MAXWP];
    for (int i = 0; i < n; i++) { wpp[i][0] = wp[3 * i]; wpp[i][1] = wp[3 * i + 1]; wph[i] = wp[3 * i + 2]; }
    double ix = rb->p.x, iy = rb->p.y, ih = bd_restrict_heading(rb->a);
    long sub = 0;
    E->robot_hit = 0;
    out[0] = bd_execute_path(D, wpp, wph, n, ix, iy, ih, &sub);
    out[1] = bd_heading_diff(ih, bd_restrict_heading(rb->a));
    out[2] = rb->p.x; out[3] = rb->p.y; out[4] = rb->a; out[5] = (double)sub;
}
/* test hook: get_local_map of an arbitrary padded-room image (float32 [H][W]) for a pose */
void orc_bd_local_map(bd_env *D, const float *gmap, double x, double y, double h, float *out) { bd_local_map(D, gmap, x, y, h, out); }

/* AreaClearingEnv.step after the movement (area_clearing.py:691-778): `steps` more sim steps, completion test, rewards */
enum { AC_I_X = 0, AC_I_Y, AC_I_THETA, AC_I_TOTAL_WORK, AC_I_COLL_REWARD, AC_I_DIFF_REWARD, AC_I_BOX_REWARD, AC_I_BOX_COUNT, AC_I_MINISTEPS,
       AC_I_HIT, AC_I_SUBSTEPS, AC_I_ROBOT_DIST, AC_I_T, AC_I_NWP, AC_I_WORK, AC_I_PUSH_REWARD, AC_I_COUNT };
static void ac_finish_step(bd_env *D, double robot_distance, double ih, long total_sub, int nwp, uint8_t *obs, double *reward,
                           int *terminated, int *truncated, double *info)
{
    orc_env *E = D->E;
    const bd_params *B = &D->B;
    body_t *rb = &E->bodies[0];
    double dts = B->ctrl_dt / B->steps;
    for (int k = 0; k < B->steps; k++) { space_step(E, dts); total_sub++; }
    double collision_penalty = E->robot_hit ? B->boundary_penalty : 0;
    /* updated_obstacles = CostMap.get_obs_from_poly(self.box_shapes) */
    double (*now)[4][2] = malloc(sizeof(double) * 8 * (size_t)(D->nbox + 1));
    for (int k = 0; k < D->nbox; k++) {
        const shape_t *sh = &E->shapes[D->first_box_shape + k];
        vec wv[4]; bd_box_world_verts(E, sh, wv);
        for (int i = 0; i < 4; i++) { now[k][i][0] = wv[i].x; now[k][i][1] = wv[i].y; }
    }
    /* boxes_completed (area_clearing.py:1122-1140) */
    int num_completed = 0;
    for (int k = 0; k < D->nbox; k++) {
        int inter = ac_intersects(D->nbd, D->bd_poly, 4, now[k]);
        if (!inter) num_completed++;
        D->cleared[k] = !inter;
    }
    int all_completed = num_completed == D->nbox;
    /* obs_to_goal_difference (metrics.py:73-94) */
    double diff_reward = 0.0;
    for (int k = 0; k < D->nbox; k++) {
        const double (*pa)[2] = (const double (*)[2])(D->prev_boxes + 8 * (size_t)k);
        if (!ac_intersects(D->nbd, D->bd_poly, 4, pa)) continue;
        double ax, ay, bx, by;
        poly_centroid_np(4, D->prev_boxes + 8 * (size_t)k, &ax, &ay);
        poly_centroid_np(4, &now[k][0][0], &bx, &by);
        double min_a = INFINITY, min_b = INFINITY;
        for (int g = 0; g < D->ngoal; g++) {
            double da = bd_dist2(ax, ay, D->goals[g][0], D->goals[g][1]), db = bd_dist2(bx, by, D->goals[g][0], D->goals[g][1]);
            if (da < min_a) min_a = da;
            if (db < min_b) min_b = db;
        }
        diff_reward += min_a - min_b;
    }
    double pushing_reward = diff_reward * B->pushing_mult;
    double box_reward;
    if (num_completed > D->cleared_count) { box_reward = fabs((double)(num_completed - D->cleared_count)) * B->box_cleared_reward; D->t = 0; }
    else box_reward = fabs((double)(num_completed - D->cleared_count)) * B->box_putback_penalty;
    D->cleared_count = num_completed;
    double nonmovement_penalty = 0; /* NONMOVEMENT_PENALTY = 0 */
    double work = 0.0;
    for (int k = 0; k < D->nbox; k++) {
        const double *prev = D->prev_boxes + 8 * (size_t)k;
        double area = poly_area_np(4, prev), ax, ay, bx, by;
        poly_centroid_np(4, prev, &ax, &ay);
        poly_centroid_np(4, &now[k][0][0], &bx, &by);
        work += sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) * area;
    }
    D->total_work += work;
    memcpy(D->prev_boxes, now, sizeof(double) * 8 * (size_t)D->nbox);
    int term = all_completed;
    double r = box_reward + collision_penalty + pushing_reward + nonmovement_penalty;
    int trunc = D->t >= B->t_max;
    if (trunc) r += B->truncation_penalty;
    else if (term) r += B->terminal_reward;
    *reward = r; *terminated = term; *truncated = trunc;
    if (info) {
        info[AC_I_X] = rb->p.x; info[AC_I_Y] = rb->p.y; info[AC_I_THETA] = rb->a; info[AC_I_TOTAL_WORK] = D->total_work; info[AC_I_COLL_REWARD] = -work;
        info[AC_I_DIFF_REWARD] = diff_reward; info[AC_I_BOX_REWARD] = box_reward; info[AC_I_BOX_COUNT] = num_completed;
        info[AC_I_MINISTEPS] = (B->action_type == 2) ? 1.0 : robot_distance / 2.5; info[AC_I_HIT] = E->robot_hit; info[AC_I_SUBSTEPS] = (double)total_sub;
        info[AC_I_ROBOT_DIST] = robot_distance; info[AC_I_T] = D->t; info[AC_I_NWP] = nwp; info[AC_I_WORK] = work; info[AC_I_PUSH_REWARD] = pushing_reward;
    }
    (void)ih;
    if (obs) orc_bd_observe(D, obs);
    E->robot_hit = 0; /* area_clearing.py:776 */
    D->last_substeps = total_sub;
    free(now);
}

/* BoxDeliveryEnv.step (box_delivery_env.py:634-830); action2 is the angular speed of 'velocity' actions */
void orc_bd_step2(bd_env *D, double action, double action2, uint8_t *obs, double *reward, int *terminated, int *truncated, double *info);
void orc_bd_step(bd_env *D, double action, uint8_t *obs, double *reward, int *terminated, int *truncated, double *info)
{
    orc_bd_step2(D, action, 0.0, obs, reward, terminated, truncated, info);
}
void orc_bd_step2(bd_env *D, double action, double action2, uint8_t *obs, double *reward, int *terminated, int *truncated, double *info)
{
    orc_env *E = D->E;
    const bd_params *B = &D->B;
    body_t *rb = &E->bodies[0];
    double dts = B->ctrl_dt / B->steps;
    if (B->task == 1) D->t += 1; else E->robot_hit = 0;
    int robot_boxes = 0; double robot_reward = 0.0;
    double ix = rb->p.x, iy = rb->p.y, ih = bd_restrict_heading(rb->a);
    /* initial box distances */
    double *init_d = (double *)malloc(sizeof(double) * (size_t)(D->nbox + 1));
    for (int q = 0; q < (B->task == 1 ? 0 : D->nalive); q++) {
        int k = D->box_order[q];
        const body_t *bb = &E->bodies[1 + k];
        init_d[k] = bd_shortest_path_distance(D, bb->p.x, bb->p.y, B->recept_x, B->recept_y);
    }
    double robot_distance = 0.0;
    long total_sub = 0;
    int nwp = 0;
    if (B->action_type == 2 && B->task == 1) {
        /* area-clearing velocity control (area_clearing.py:660-667): set once, the common `steps` sim steps follow */
        double sn_, cs_; bp_sincos(rb->a, &sn_, &cs_);
        rb->w = B->yaw_rate_step * action2 / 2;
        double sv = B->target_speed * action;
        rb->v = V(cs_ * sv + -sn_ * 0.0, sn_ * sv + cs_ * 0.0);
        D->last_nwp = 0;
    } else if (B->action_type == 2) {
        /* velocity control (box_delivery_env.py:672-703) */
        double lin = action, angv = action2;
        if (fabs(lin) >= B->target_speed) lin = B->target_speed * (double)((lin > 0) - (lin < 0));
        rb->w = angv;
        for (int k = 0; k < B->steps; k++) {
            double sn_, cs_; bp_sincos(rb->a, &sn_, &cs_);
            rb->v = V(cs_ * lin + -sn_ * 0.0, sn_ * lin + cs_ * 0.0);
            space_step(E, dts);
            total_sub++;
            if (E->robot_hit) break;
        }
        robot_distance = bd_dist2(ix, iy, rb->p.x, rb->p.y);
        D->last_nwp = 0;
    } else {
    /* heading action -> spatial action index (box_delivery_env.py:706-723); a position action is the index itself */
    double angle = (action + 1) * M_PI + M_PI / 2;
    double sa, ca; bp_sincos(angle, &sa, &ca);
    double x_movement = B->step_size * ca, y_movement = B->step_size * sa;
    int x_pixel = (int)((double)B->local_px / 2 + x_movement * B->ppm);
    int y_pixel = (int)((double)B->local_px / 2 - y_movement * B->ppm);
    if (B->action_type == 1) { long idx = (long)action; y_pixel = (int)(idx / B->local_px); x_pixel = (int)(idx % B->local_px); }
    double wpp[BD_MAXWP][2]; double wph[BD_MAXWP]; double move_sign;
    nwp = bd_plan(D, x_pixel, y_pixel, ix, iy, ih, wpp, wph, &move_sign);
    (void)move_sign; /* only feeds robot_new_position, which the reference computes and never uses */
    D->last_nwp = nwp;
    for (int i = 0; i < nwp && i < BD_MAXWP; i++) { D->last_wp[i][0] = wpp[i][0]; D->last_wp[i][1] = wpp[i][1]; D->last_wp[i][2] = wph[i]; }
    robot_distance = bd_execute_path(D, wpp, wph, nwp, ix, iy, ih, &total_sub);
    } /* action_type */
    if (B->task == 1) { free(init_d); ac_finish_step(D, robot_distance, ih, total_sub, nwp, obs, reward, terminated, truncated, info); return; }
    /* step_simulation_until_still (box_delivery_env.py:990-1023) */
    {
        int np_ = 0; double *prevp = (double *)malloc(sizeof(double) * 2 * (size_t)(D->nbox + 2));
        long sim_steps = 0; int done = 0;
        while (!done) {
            for (int q = 0; q < D->nalive; q++) {
                int k = D->box_order[q];
                const shape_t *sh = &E->shapes[D->first_box_shape + k];
                body_t *bb = &E->bodies[1 + k];
                vec wv[4]; bd_box_world_verts(E, sh, wv);
                int stuck = 0;
                for (int i = 0; i < sh->n && !stuck; i++)
                    for (int s2 = 0; s2 < D->nstatic && !stuck; s2++) {
                        const shape_t *st = &E->shapes[D->first_static_shape + s2];
                        if (st->ctype == 3 && bd_point_in_shape(st, wv[i])) stuck = 1;
                    }
                if (stuck) {
                    int pi, pj; bd_pos_to_pix(D, bb->p.x, bb->p.y, &pi, &pj);
                    int ni = D->edt_i[(size_t)pi * D->W + pj], nj = D->edt_j[(size_t)pi * D->W + pj];
                    double nx, ny; bd_pix_to_pos(D, ni, nj, &nx, &ny);
                    bb->p = V(nx, ny); body_set_transform(bb);
                    bb->v = V(0, 0);
                }
            }
            int n = D->nalive + 1;
            double *cur = (double *)malloc(sizeof(double) * 2 * (size_t)n);
            for (int q = 0; q < D->nalive; q++) { const body_t *bb = &E->bodies[1 + D->box_order[q]]; cur[2 * q] = bb->p.x; cur[2 * q + 1] = bb->p.y; }
            cur[2 * D->nalive] = rb->p.x; cur[2 * D->nalive + 1] = rb->p.y;
            if (np_ > 0) {
                done = 1;
                for (int i = 0; i < n; i++)
                    if (bd_dist2(prevp[2 * i], prevp[2 * i + 1], cur[2 * i], cur[2 * i + 1]) > 0.005) {
                        done = 0;
                        if (orc_bd_trace && orc_bd_trace_n < orc_bd_trace_cap) {
                            double *o = orc_bd_trace + 4 * (size_t)orc_bd_trace_n++;
                            o[0] = -1000 - i; o[1] = cur[2 * i]; o[2] = cur[2 * i + 1]; o[3] = bd_dist2(prevp[2 * i], prevp[2 * i + 1], cur[2 * i], cur[2 * i + 1]);
                        }
                        break;
                    }
            }
            memcpy(prevp, cur, sizeof(double) * 2 * (size_t)n); np_ = n;
            free(cur);
            space_step(E, dts);
            total_sub++;
            sim_steps++;
            if (sim_steps > B->step_limit) break;
        }
        free(prevp);
    }
    D->last_substeps = total_sub;
    /* final distances, rewards, removal (box_delivery_env.py:736-767) */
    double boxes_distance = 0.0;
    int nrem = 0; int *rem = (int *)malloc(sizeof(int) * (size_t)(D->nbox + 1));
    for (int q = 0; q < D->nalive; q++) {
        int k = D->box_order[q];
        const body_t *bb = &E->bodies[1 + k];
        double fin = bd_shortest_path_distance(D, bb->p.x, bb->p.y, B->recept_x, B->recept_y);
        double moved = init_d[k] - fin;
        boxes_distance += fabs(moved);
        if (B->use_correct_direction_reward && moved > 0) moved *= B->correct_direction_reward_scale;
        robot_reward += B->partial_rewards_scale * moved;
        const shape_t *sh = &E->shapes[D->first_box_shape + k];
        vec wv[4]; bd_box_world_verts(E, sh, wv);
        int inside = 1;
        for (int i = 0; i < sh->n && inside; i++) {
            int any = 0;
            for (int s2 = 0; s2 < D->nstatic && !any; s2++) {
                const shape_t *st = &E->shapes[D->first_static_shape + s2];
                if (st->ctype == 4 && bd_point_in_shape(st, wv[i])) any = 1;
            }
            if (!any) inside = 0;
        }
        if (inside) {
            rem[nrem++] = k; D->cleared[k] = 1; D->inactivity = 0; robot_boxes += 1; robot_reward += B->goal_reward;
        }
    }
    for (int r = 0; r < nrem; r++) {
        int k = rem[r], s = D->first_box_shape + k;
        E->removed[s] = 1; E->bodies[1 + k].type = BODY_STATIC; D->box_alive[k] = 0;
        int w = 0; /* cpSpaceFilterArbiters */
        for (int a = 0; a < E->narb; a++) { if (E->arbs[a].sa == s || E->arbs[a].sb == s) continue; if (w != a) E->arbs[w] = E->arbs[a]; w++; }
        E->narb = w;
        int w2 = 0;
        for (int a = 0; a < E->nactive; a++) { uint32_t key = (uint32_t)E->active[a]; if ((int)(key >> 16) == s || (int)(key & 0xffff) == s) continue; E->active[w2++] = E->active[a]; }
        E->nactive = w2;
        int w3 = 0;
        for (int q = 0; q < D->nalive; q++) if (D->box_order[q] != k) D->box_order[w3++] = D->box_order[q];
        D->nalive = w3;
    }
    free(rem); free(init_d);
    if (E->robot_hit) robot_reward -= B->collision_penalty;
    double rh = bd_restrict_heading(rb->a);
    double turn_angle = bd_heading_diff(ih, rh);
    if (robot_distance < 0.05 && fabs(turn_angle) < 0.05 * (M_PI / 180.0)) robot_reward -= B->non_movement_penalty;
    D->cum_distance += robot_distance; D->cum_boxes += robot_boxes; D->cum_reward += robot_reward;
    /* work: total_work_done(prev_boxes, updated_boxes) zips the two lists by position (metrics.py:96-113) */
    double work = 0.0;
    {
        int n = D->nprev < D->nalive ? D->nprev : D->nalive;
        for (int q = 0; q < n; q++) {
            const shape_t *sh = &E->shapes[D->first_box_shape + D->box_order[q]];
            double now[8];
            vec wv[4]; bd_box_world_verts(E, sh, wv);
            for (int i = 0; i < 4; i++) { now[2 * i] = wv[i].x; now[2 * i + 1] = wv[i].y; }
            const double *prev = D->prev_boxes + 8 * (size_t)q;
            double area = poly_area_np(4, prev), ax, ay, bx, by;
            poly_centroid_np(4, prev, &ax, &ay);
            poly_centroid_np(4, now, &bx, &by);
            work += sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) * area;
        }
        for (int q = 0; q < D->nalive; q++) {
            const shape_t *sh = &E->shapes[D->first_box_shape + D->box_order[q]];
            vec wv[4]; bd_box_world_verts(E, sh, wv);
            for (int i = 0; i < 4; i++) { D->prev_boxes[8 * q + 2 * i] = wv[i].x; D->prev_boxes[8 * q + 2 * i + 1] = wv[i].y; }
        }
        D->nprev = D->nalive;
    }
    D->total_work += work;
    if (robot_boxes == 0) D->inactivity += 1;
    int term = 0, trunc = 0;
    if (D->cum_boxes == B->num_boxes) term = 1;
    if (D->inactivity >= B->inactivity_cutoff) { term = 1; trunc = 1; }
    *reward = robot_reward; *terminated = term; *truncated = trunc;
    if (info) {
        info[BD_I_X] = rb->p.x; info[BD_I_Y] = rb->p.y; info[BD_I_THETA] = rb->a;
        info[BD_I_CUM_DIST] = D->cum_distance; info[BD_I_CUM_BOXES] = D->cum_boxes; info[BD_I_CUM_REWARD] = D->cum_reward;
        info[BD_I_TOTAL_WORK] = D->total_work; info[BD_I_MINISTEPS] = robot_distance / B->ministep_size; info[BD_I_INACTIVITY] = D->inactivity;
        info[BD_I_HIT] = E->robot_hit; info[BD_I_SUBSTEPS] = (double)total_sub; info[BD_I_ROBOT_DIST] = robot_distance;
        info[BD_I_BOXES_DIST] = boxes_distance; info[BD_I_NWP] = nwp; info[BD_I_NALIVE] = D->nalive; info[BD_I_WORK] = work;
    }
    if (obs) orc_bd_observe(D, obs);
}

/* accessors for tests */
void orc_bd_get_maps(const bd_env *D, float *cspace, float *cspace_thin, int *edt_i, int *edt_j, float *recept, float *small_free, float *overhead)
{
    size_t N = (size_t)D->H * D->W;
    if (cspace) memcpy(cspace, D->cspace, N * sizeof(float));
    if (cspace_thin) memcpy(cspace_thin, D->cspace_thin, N * sizeof(float));
    if (edt_i) memcpy(edt_i, D->edt_i, N * sizeof(int));
    if (edt_j) memcpy(edt_j, D->edt_j, N * sizeof(int));
    if (recept) memcpy(recept, D->recept_map, N * sizeof(float));
    if (small_free) memcpy(small_free, D->small_obstacle_free, (size_t)D->SH * D->SW * sizeof(float));
    if (overhead) memcpy(overhead, D->overhead, N * sizeof(float));
}
orc_env *orc_bd_physics(bd_env *D) { return D->E; }
/* current world vertices of every shape (body.local_to_world of the hull vertices): out[ns][4][2], counts[ns] */
void orc_bd_world_verts(bd_env *D, double *out, int *counts)
{
    orc_env *E = D->E;
    for (int s2 = 0; s2 < E->ns; s2++) {
        const shape_t *sh = &E->shapes[s2];
        vec wv[ORC_MAXV]; bd_box_world_verts(E, sh, wv);
        counts[s2] = sh->n;
        for (int i = 0; i < sh->n && i < 4; i++) { out[(size_t)s2 * 8 + 2 * i] = wv[i].x; out[(size_t)s2 * 8 + 2 * i + 1] = wv[i].y; }
    }
}
/* test hook: all-free configuration space (the straight-line branch of the position controller, tests/test_controller_golden.py) */
void orc_bd_set_all_free(bd_env *D) { size_t N = (size_t)D->H * D->W; for (size_t i = 0; i < N; i++) { D->cspace[i] = 1.0f; D->cspace_thin[i] = 1.0f; D->edt_i[i] = (int)(i / D->W); D->edt_j[i] = (int)(i % D->W); } }
int orc_bd_num_alive(const bd_env *D) { return D->nalive; }
void orc_bd_get_alive(const bd_env *D, int *out) { for (int k = 0; k < D->nbox; k++) out[k] = D->box_alive[k]; }
int orc_bd_last_waypoints(const bd_env *D, double *out)
{
    for (int i = 0; i < D->last_nwp; i++) { out[3 * i] = D->last_wp[i][0]; out[3 * i + 1] = D->last_wp[i][1]; out[3 * i + 2] = D->last_wp[i][2]; }
    return D->last_nwp;
}
double orc_atan2(double y, double x) { return bp_atan2(y, x); }
double orc_pymod(double a, double b) { return bp_pymod(a, b); }
void orc_bd_fill_poly(float *img, int H, int W, int n, const long *px, const long *py, float color) { bd_fill_poly(img, H, W, n, px, py, color); }
int orc_bd_sk_line(long r0, long c0, long r1, long c1, long *rr, long *cc) { return bd_sk_line(r0, c0, r1, c1, rr, cc); }
void orc_bd_approx_polygon(int n, const long *cr, const long *cc, double tol, unsigned char *keep) { bd_approx_polygon(n, cr, cc, tol, keep); }
int orc_bd_point_in_shape(bd_env *D, int shape, double x, double y) { return bd_point_in_shape(&D->E->shapes[shape], V(x, y)); }
