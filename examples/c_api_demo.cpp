// The C ABI of include/benchpush_amd.h driven from a plain C++ program: no Python, no torch -- hipMalloc'ed buffers and a HIP stream.
//   hipcc -O2 -Iinclude examples/c_api_demo.cpp -Lbenchpush_amd -lbenchpush_hip -Wl,-rpath,$PWD/benchpush_amd -o /tmp/c_api_demo
//   /tmp/c_api_demo scenario.bin result.bin
// scenario.bin (written by tests/test_gpu_c_api.py from the same data the Python path uses): int32 E, T, F, V, steps; the bp_config
// bytes; verts f64[T][F][V][2]; counts i32[T][F]; centres f64[T][F][2]; starts f64[T][3]; nfloes i32[T]; actions f64[steps][E].
// result.bin: per step reward f64[E], terminated u8[E], info f64[E][16], then the final observation u8[E][4][150][150].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "benchpush_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, h ? bp_last_error(h) : ""); return 2; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

template <class T> static std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n); if (fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(4); } return v; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s scenario.bin result.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    const std::vector<int32_t> hd = rd<int32_t>(f, 5);
    const int E = hd[0], T = hd[1], F = hd[2], V = hd[3], steps = hd[4];
    bp_handle *h = nullptr;
    if (bp_sizeof_config() != (int)sizeof(bp_config)) { fprintf(stderr, "bp_config size mismatch\n"); return 1; }
    const std::vector<unsigned char> cfgb = rd<unsigned char>(f, sizeof(bp_config));
    const bp_config *cfg = (const bp_config *)cfgb.data();
    const std::vector<double> verts = rd<double>(f, (size_t)T * F * V * 2);
    const std::vector<int32_t> counts = rd<int32_t>(f, (size_t)T * F);
    const std::vector<double> centres = rd<double>(f, (size_t)T * F * 2), starts = rd<double>(f, (size_t)T * 3);
    const std::vector<int32_t> nfloes = rd<int32_t>(f, T);
    const std::vector<double> actions = rd<double>(f, (size_t)steps * E);
    fclose(f);

    CHECK(bp_create(cfg, E, 0, 0, &h));
    CHECK(bp_load_scenarios(h, T, F, V, verts.data(), counts.data(), centres.data(), starts.data(), nfloes.data()));
    const int H = bp_obs_height(h), W = bp_obs_width(h);
    const size_t obs_bytes = (size_t)E * BP_OBS_C * H * W;
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    uint8_t *d_obs, *d_term, *d_trunc;
    double *d_rew, *d_info, *d_act;
    HIP(hipMalloc(&d_obs, obs_bytes)); HIP(hipMalloc(&d_term, E)); HIP(hipMalloc(&d_trunc, E));
    HIP(hipMalloc(&d_rew, E * sizeof(double))); HIP(hipMalloc(&d_info, (size_t)E * BP_INFO_COUNT * sizeof(double))); HIP(hipMalloc(&d_act, E * sizeof(double)));
    CHECK(bp_reset(h, nullptr, d_obs, d_info, st));
    FILE *o = fopen(argv[2], "wb");
    std::vector<double> rew(E), info((size_t)E * BP_INFO_COUNT);
    std::vector<uint8_t> term(E), obs(obs_bytes);
    for (int t = 0; t < steps; t++) {
        HIP(hipMemcpyAsync(d_act, actions.data() + (size_t)t * E, E * sizeof(double), hipMemcpyHostToDevice, st));
        CHECK(bp_step(h, d_act, d_obs, d_rew, d_term, d_trunc, d_info, st));
        HIP(hipMemcpyAsync(rew.data(), d_rew, E * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(term.data(), d_term, E, hipMemcpyDeviceToHost, st));
        HIP(hipMemcpyAsync(info.data(), d_info, info.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP(hipStreamSynchronize(st));
        fwrite(rew.data(), sizeof(double), E, o); fwrite(term.data(), 1, E, o); fwrite(info.data(), sizeof(double), info.size(), o);
    }
    HIP(hipMemcpy(obs.data(), d_obs, obs_bytes, hipMemcpyDeviceToHost));
    fwrite(obs.data(), 1, obs_bytes, o);
    fclose(o);
    std::vector<int32_t> err(E);
    CHECK(bp_check_errors(h, err.data()));
    for (int e = 0; e < E; e++) if (err[e]) { fprintf(stderr, "capacity error in env %d: %d\n", e, err[e]); return 5; }
    printf("c_api_demo: %d envs x %d steps, reward[0] of the last step %.17g\n", E, steps, rew[0]);
    CHECK(bp_destroy(h));
    return 0;
}
