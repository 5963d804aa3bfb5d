// exec_mask3 follow-up: is the few-active-lanes slowdown a latency effect (dependent chain only) or a throughput effect?  ILP = 1, 2, 4
// independent f64 mul/add chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k(double *out, unsigned long long *cyc, int nact, int iters)
{
    const int lane = threadIdx.x;
    double a[ILP];
    for (int q = 0; q < ILP; q++) a[q] = 1.0 + lane * 1e-9 + q;
    const double b = 0.999999 + lane * 1e-12, c = 1e-7;
    const bool act = lane < nact;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (act) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int q = 0; q < ILP; q++) a[q] = a[q] * b;
#pragma unroll
                for (int q = 0; q < ILP; q++) a[q] = a[q] + c;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int q = 0; q < ILP; q++) s += a[q];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP> void run(double *out, unsigned long long *cyc)
{
    const int iters = 4000, blocks = 2048;
    for (int nact : {1, 4, 8, 9, 16, 64}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, nact, iters); (void)hipDeviceSynchronize(); }
        static unsigned long long h[4096]; (void)hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
        printf("ILP %d active %2d (2 waves/SIMD): %.2f cycles per f64 op\n", ILP, nact, s / blocks / (iters * 16.0 * ILP));
    }
}
int main()
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 8 * 4096);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc);
    return 0;
}
