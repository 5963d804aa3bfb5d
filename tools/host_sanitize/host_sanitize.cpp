// CPU sanitizer harness for the HOST-side product code: benchpush_amd/csrc/bp_host_geom.hpp + bp_host_bd.hpp -- hulls, mass
// properties, fillPoly, disk dilation, EDT indices, spfa, maze maps -- are normally compiled only inside the .hip translation unit, where no sanitizer
// runs (GPU ASan is not available on the pool).  This program includes the same two headers, is built with -fsanitize=address,undefined by
// tests/test_host_sanitize.py, and replays the load-time sequences of bp_load_scenarios / bp_load_maze / bp_bd_load (bp_capi.hip) on case files that the
// test writes from the shipped configurations: the arguments are exactly the arrays the Python envs hand to those entry points.
//
//   host_sanitize <case-file>      prints one line per record and "host-sanitize-ok <records>" at the end; exit code 0 unless a loader refused a case
//
// File format (little endian): records of  int32 tag ; payload
//   tag 1 ship-ice : bp_config ; int32 T, F, V ; double verts[T][F][V][2] ; int32 counts[T][F] ; double centres[T][F][2] ; double starts[T][3] ; int32 nfloes[T]
//   tag 2 maze     : bp_config ; int32 T, nbox, nwalls, grid_h, grid_w ; double centres[T][nbox][2] ; double walls[nwalls][4] ; double start[T][3]
//   tag 3 box / area: bp_bd_config ; int32 T, nbox, ns ; double starts[T][3] ; double boxes[T][nbox][3] ; double sverts[T][ns][4][2] ; int32 scount[T][ns] ;
//                     double spose[T][ns][3] ; double srad[T][ns] ; int32 stype[T][ns]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/benchpush_amd.h"
#include "../../benchpush_amd/csrc/bp_host_bd.hpp"

using namespace bpgeom;

static FILE *g_f;
template <typename T> static std::vector<T> rd(size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, g_f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}
template <typename T> static T rd1() { return rd<T>(1)[0]; }

static double checksum(const std::vector<std::vector<Shape>> &trials)
{
    double s = 0.0;
    for (const auto &b : trials)
        for (const Shape &sh : b) {
            s += sh.m_inv + sh.i_inv + sh.cog.x + sh.cog.y + sh.p.x + sh.p.y + sh.angle;
            if (sh.verts.size() != sh.normals.size()) { fprintf(stderr, "planes != vertices\n"); exit(4); }
            for (size_t i = 0; i < sh.verts.size(); i++) s += sh.verts[i].x + sh.verts[i].y + sh.normals[i].x + sh.normals[i].y;
        }
    return s;
}

static int ship_ice()
{
    const bp_config cf = rd1<bp_config>();
    const int T = rd1<int32_t>(), F = rd1<int32_t>(), V = rd1<int32_t>();
    const auto verts = rd<double>((size_t)T * F * V * 2);
    const auto counts = rd<int32_t>((size_t)T * F);
    const auto centres = rd<double>((size_t)T * F * 2);
    const auto starts = rd<double>((size_t)T * 3);
    const auto nfloes = rd<int32_t>(T);
    std::vector<std::vector<Shape>> trials(T);
    size_t nb = 0;
    for (int t = 0; t < T; t++) {                       // bp_load_scenarios
        Shape ship;
        build_ship(cf.ship_verts, cf.num_ship_verts, starts[3 * t], starts[3 * t + 1], starts[3 * t + 2], ship);
        trials[t].push_back(ship);
        if (nfloes[t] > F) return 1;
        for (int f = 0; f < nfloes[t]; f++) {
            const int n = counts[(size_t)t * F + f];
            if (n < 3) continue;
            if (n > V) return 1;
            Shape s;
            if (!build_floe(verts.data() + ((size_t)t * F + f) * V * 2, n, centres[((size_t)t * F + f) * 2], centres[((size_t)t * F + f) * 2 + 1], cf.density,
                            cf.poly_radius, s))
                continue;
            if ((int)s.verts.size() > BP_MAXV) return 1;
            trials[t].push_back(s);
        }
        nb += trials[t].size();
    }
    printf("ship-ice: %d trials, %zu bodies, checksum %.17g\n", T, nb, checksum(trials));
    return 0;
}

static int maze()
{
    const bp_config cf = rd1<bp_config>();
    const int T = rd1<int32_t>(), nbox = rd1<int32_t>(), nwalls = rd1<int32_t>(), gh = rd1<int32_t>(), gw = rd1<int32_t>();
    const auto centres = rd<double>((size_t)T * nbox * 2);
    const auto walls = rd<double>((size_t)nwalls * 4);
    const auto start = rd<double>((size_t)T * 3);
    std::vector<std::vector<Shape>> trials(T);
    for (int t = 0; t < T; t++) {                       // bp_load_maze
        for (int k = 0; k <= cf.num_wheels; k++) {
            Shape s;
            const double *st3 = start.data() + 3 * (size_t)t;
            if (k == 0) build_kinematic_part(cf.ship_verts, cf.num_ship_verts, st3[0], st3[1], st3[2], s);
            else build_kinematic_part(cf.wheel_verts[k - 1], 4, st3[0], st3[1], st3[2], s);
            trials[t].push_back(s);
        }
        for (int b = 0; b < nbox; b++) {
            const double ox = centres[((size_t)t * nbox + b) * 2], oy = centres[((size_t)t * nbox + b) * 2 + 1], sz = cf.obstacle_size;
            const double raw[8] = {ox + sz, oy + sz, ox - sz, oy + sz, ox - sz, oy - sz, ox + sz, oy - sz};
            Shape s;
            if (!build_floe(raw, 4, ox, oy, cf.density, cf.poly_radius, s)) continue;
            trials[t].push_back(s);
        }
        for (int w = 0; w < nwalls; w++) {
            Shape s;
            build_wall(walls[4 * w], walls[4 * w + 1], walls[4 * w + 2], walls[4 * w + 3], s);
            trials[t].push_back(s);
        }
    }
    std::vector<unsigned char> wall;
    std::vector<double> norm, raw;
    maze_maps(walls.data(), nwalls, cf.wall_radius, cf.map_w, cf.map_h, gh, gw, cf.goal_x, cf.goal_y, wall, norm, raw);
    size_t nwall = 0;
    double mx = 0.0;
    for (size_t i = 0; i < wall.size(); i++) { nwall += wall[i]; if (raw[i] > mx) mx = raw[i]; }
    printf("maze: %d layouts x %d boxes, %d walls, grid %dx%d: %zu wall cells, longest wavefront %.0f, checksum %.17g\n", T, nbox, nwalls, gh, gw, nwall, mx,
           checksum(trials));
    return 0;
}

static int box()
{
    const bp_bd_config cf = rd1<bp_bd_config>();
    const int T = rd1<int32_t>(), nbox = rd1<int32_t>(), ns = rd1<int32_t>();
    const auto starts = rd<double>((size_t)T * 3);
    const auto boxes = rd<double>((size_t)T * nbox * 3);
    const auto sverts = rd<double>((size_t)T * ns * 8);
    const auto scount = rd<int32_t>((size_t)T * ns);
    const auto spose = rd<double>((size_t)T * ns * 3);
    const auto srad = rd<double>((size_t)T * ns);
    const auto stype = rd<int32_t>((size_t)T * ns);
    std::vector<std::vector<Shape>> trials(T);
    std::vector<std::vector<std::vector<P2>>> map_keys;
    size_t free_cells = 0;
    for (int t = 0; t < T; t++) {                       // bp_bd_load
        std::vector<Shape> &bodies = trials[t];
        const double sx = starts[3 * t], sy = starts[3 * t + 1], sh = starts[3 * t + 2];
        {
            Shape s;
            build_agent_main(cf.robot_verts, 4, sx, sy, sh, s);
            bodies.push_back(s);
            for (int k = 0; k < 5; k++) {
                Shape w;
                build_kinematic_part(k < 4 ? cf.wheel_verts[k] : cf.bumper_verts, 4, sx, sy, sh, w);
                bodies.push_back(w);
            }
        }
        for (int b = 0; b < nbox; b++) {
            const double *bx = boxes.data() + ((size_t)t * nbox + b) * 3;
            Shape s;
            if (!build_box(bx[0], bx[1], bx[2], cf.box_half, cf.box_density, 0.02, s)) return 1;
            bodies.push_back(s);
        }
        std::vector<std::vector<P2>> obstacles;
        int nrec = 0;
        for (int k = 0; k < ns; k++) {
            const size_t o = (size_t)t * ns + k;
            const int n = scount[o];
            if (n == 0) continue;
            if (n < 3 || n > 4) return 1;
            Shape s;
            build_static_poly(sverts.data() + o * 8, n, spose[o * 3], spose[o * 3 + 1], spose[o * 3 + 2], s);
            s.radius = srad[o];
            if (stype[o] == 4) { if (s.verts.size() != 4) return 1; nrec++; (void)world_verts(s); continue; }
            bodies.push_back(s);
            obstacles.push_back(world_verts(s));
        }
        if (cf.task == 0 && nrec != 1) return 1;
        if (cf.task == 1 && nrec != 0) return 1;
        bool seen = false;
        for (const auto &mk : map_keys) {
            bool same = mk.size() == obstacles.size();
            for (size_t q = 0; same && q < obstacles.size(); q++) {
                same = mk[q].size() == obstacles[q].size();
                for (size_t v = 0; same && v < obstacles[q].size(); v++) same = mk[q][v].x == obstacles[q][v].x && mk[q][v].y == obstacles[q][v].y;
            }
            seen = seen || same;
        }
        if (!seen) {
            BdMaps M;
            AcGeom G;
            G.nbd = cf.num_boundary_verts; G.nob = cf.num_outer_verts; G.ngoal = cf.num_goal_points;
            G.bd = cf.boundary; G.ob = cf.outer_boundary; G.goals = cf.goal_points; G.scale_max = cf.distance_scale_max;
            if (!bd_build_maps(obstacles, cf.room_length, cf.room_width, cf.ppm, cf.local_px, cf.local_w, cf.robot_radius, cf.robot_half_width, cf.recept_x,
                               cf.recept_y, cf.sp_channel_scale, M, cf.task, &G, cf.task == 0 && cf.invert_receptacle_map != 0))
                return 1;
            for (unsigned w : M.free_bits) free_cells += (size_t)__builtin_popcount(w);
            map_keys.push_back(obstacles);
        }
    }
    printf("%s: %d trials x %d boxes, %zu distinct layouts, %zu free cells, checksum %.17g\n", cf.task == 1 ? "area-clearing" : "box-delivery", T, nbox,
           map_keys.size(), free_cells, checksum(trials));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: host_sanitize <case-file>\n"); return 2; }
    g_f = fopen(argv[1], "rb");
    if (!g_f) { perror(argv[1]); return 2; }
    int records = 0, rc = 0;
    int32_t tag;
    while (fread(&tag, sizeof(tag), 1, g_f) == 1) {
        rc = tag == 1 ? ship_ice() : tag == 2 ? maze() : tag == 3 ? box() : 9;
        if (rc) { fprintf(stderr, "record %d (tag %d) refused: rc %d\n", records, tag, rc); return 1; }
        records++;
    }
    fclose(g_f);
    printf("host-sanitize-ok %d\n", records);
    return 0;
}
