// Wall-clock check of exec_mask.hip: kernel duration (hipEvents), shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime)
// for a dependent f64 chain under different numbers of active lanes, with every SIMD of the chip holding `wps` waves.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *cyc, int nact, int iters)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 0.999999 + lane * 1e-12, c = 1e-7;
    const bool act = lane < nact;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (act) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { a = a * b; a = a + c; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + lane] = a;
    if (lane == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
int main()
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 16 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2}) for (int nact : {1, 4, 8, 12, 15, 16, 20, 32, 64}) {
        const int blocks = 1024 * wps;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, nact, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            static unsigned long long h[2 * 4096]; (void)hipMemcpy(h, cyc, 16 * blocks, hipMemcpyDeviceToHost);
            double s = 0, r = 0; for (int i = 0; i < blocks; i++) { s += h[2 * i]; r += h[2 * i + 1]; }
            if (rep == 1)
                printf("waves/SIMD %d active %2d: kernel %.3f ms | %.2f s_memtime ticks/op | %.3f ns/op by s_memrealtime | in-kernel clock %.0f MHz\n",
                       wps, nact, ms, s / blocks / (iters * 32.0), r / blocks * 10.0 / (iters * 32.0), s / r * 100.0);
        }
    }
    return 0;
}
