// Does a wave64 f64 VALU instruction cost less when only some 16-lane quarters have active lanes?  (gfx950 microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *cyc, int nact, int stride, int iters)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 0.999999 + lane * 1e-12, c = 1e-7;
    const bool act = (lane % stride == 0) && (lane / stride < nact);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (act) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { a = a * b; a = a + c; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 2000;
    int cfgs[][2] = {{1, 1}, {8, 1}, {16, 1}, {17, 1}, {32, 1}, {33, 1}, {48, 1}, {64, 1}, {4, 16}, {2, 32}, {16, 4}};
    for (auto &c : cfgs) {
        for (int blocks : {1, 1024}) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, c[0], c[1], iters);
            hipDeviceSynchronize();
            unsigned long long h[1024]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
            printf("active %2d stride %2d blocks %4d: %.2f cycles per dependent f64 op\n", c[0], c[1], blocks, s / blocks / (iters * 32.0));
        }
    }
    return 0;
}
