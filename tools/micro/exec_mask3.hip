// Threshold and instruction-class scan of the few-active-lanes slowdown (see exec_mask2.hip): dependent chains of f64 mul/add, f32 mul/add,
// u32 mad, and f64 with the idle lanes kept busy on dummy data ("padded").
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, unsigned long long *cyc, int nact, int iters)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 0.999999 + lane * 1e-12, c = 1e-7;
    float fa = 1.0f + lane * 1e-3f, fb = 0.9999f, fc = 1e-3f;
    unsigned ua = lane, ub = 3, uc = 7;
    const bool act = lane < nact;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 3) {        // padded: every lane computes, the lanes >= nact on their own (unused) values
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { a = a * b; a = a + c; }
        }
    } else if (act) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (MODE == 0) { a = a * b; a = a + c; }
                if (MODE == 1) { fa = fa * fb; fa = fa + fc; }
                if (MODE == 2) { ua = ua * ub; ua = ua + uc; }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a + fa + ua;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, double *out, unsigned long long *cyc)
{
    const int iters = 5000, blocks = 2048;
    for (int nact : {1, 8, 9, 10, 11, 12, 16, 64}) {
        float ms = 0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, nact, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        static unsigned long long h[4096]; (void)hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
        printf("%-10s active %2d (2 waves/SIMD): kernel %.3f ms | %.2f ticks/op\n", name, nact, ms, s / blocks / (iters * 32.0));
    }
}
int main()
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 8 * 4096);
    run<0>("f64", out, cyc); run<1>("f32", out, cyc); run<2>("u32", out, cyc); run<3>("f64 padded", out, cyc);
    return 0;
}
