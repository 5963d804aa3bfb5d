// What does it cost to replace a finished workgroup by the next one of the grid?  Workgroups shaped like the step scheduler's (64 threads, 20 480 B of LDS, a
// full half of a SIMD's registers, optionally a scratch segment) spin for a given number of shader cycles and leave; the grid holds many rounds of the 2 048
// wave slots.  Launch time against the ideal rounds x spin gives the turnover gap per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o wg_turnover wg_turnover.hip && ./wg_turnover
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

extern __shared__ double2 smem[];

template <int SCRATCH>
__global__ __launch_bounds__(64, 2) void k_spin(const unsigned long long cycles, unsigned long long *out, int *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    volatile int scratch[SCRATCH > 0 ? SCRATCH : 1];
    if (SCRATCH > 0) for (int i = 0; i < SCRATCH; i++) scratch[i] = threadIdx.x + i;
    asm volatile("v_mov_b32 v250, 0" ::: "v250");   // claim the whole register budget of two waves per SIMD
    smem[threadIdx.x] = make_double2(1.0, 2.0);
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (SCRATCH > 0 && scratch[threadIdx.x % SCRATCH] == -1) sink[0] = 1;
    if (threadIdx.x == 0 && out) atomicAdd(out, 1ull);
}

template <int SCRATCH>
static void run(const char *name, int rounds, unsigned long long cycles, double clock_ghz)
{
    unsigned long long *d; int *s;
    hipMalloc(&d, 8); hipMalloc(&s, 4); hipMemset(d, 0, 8);
    hipFuncSetAttribute((const void *)k_spin<SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int slots = 2048, grid = slots * rounds;
    hipLaunchKernelGGL(k_spin<SCRATCH>, dim3(slots), dim3(64), 20480, 0, cycles, d, s);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin<SCRATCH>, dim3(grid), dim3(64), 20480, 0, cycles, d, s);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ideal = rounds * (double)cycles / (clock_ghz * 1e6);
    printf("%-14s spin %7llu cycles (%.1f us), %3d rounds of 2048 workgroups: %8.3f ms, ideal %8.3f ms -> %.1f us per workgroup turnover\n", name, cycles,
           cycles / (clock_ghz * 1e3), rounds, ms, ideal, (ms - ideal) * 1e3 / rounds);
    hipFree(d); hipFree(s);
}

// staggered: workgroup b spins for a pseudo-random 0.2 .. 2.4 ms (mean 1.3), a few "heavy" ones for 13 ms -- the step scheduler's mix; the ideal is the larger of
// the sum of all spins / 2048 slots and the longest spin, and the rest is what the dispatcher loses when slots free up one by one
__global__ __launch_bounds__(64, 2) void k_spin_mixed(const int heavy_every, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_mov_b32 v250, 0" ::: "v250");
    smem[threadIdx.x] = make_double2(1.0, 2.0);
    unsigned h = blockIdx.x * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    unsigned long long cycles = 20000ull + (h % 220000u);
    if (heavy_every > 0 && (blockIdx.x % heavy_every) == 0 && blockIdx.x < 2048) cycles = 1300000ull;
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicAdd(out, cycles);
}
// the same 24000 workgroups split over `nq` kernels on `nq` streams (one hardware queue each): does a second in-order dispatcher fill the slots the first leaves empty?
__global__ __launch_bounds__(64, 2) void k_spin_mixed_q(const int base, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_mov_b32 v250, 0" ::: "v250");
    smem[threadIdx.x] = make_double2(1.0, 2.0);
    unsigned h = (blockIdx.x + base) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned long long cycles = 20000ull + (h % 220000u);
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicAdd(out, cycles);
}
static void run_mixed_q(int nq)
{
    unsigned long long *d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipFuncSetAttribute((const void *)k_spin_mixed_q, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
    hipStream_t st[16]; hipEvent_t ev[16];
    for (int q = 0; q < nq; ++q) { hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking); hipEventCreate(&ev[q]); }
    hipDeviceSynchronize();
    const int grid = 24000, per = grid / nq;
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < nq; ++q) hipLaunchKernelGGL(k_spin_mixed_q, dim3(per), dim3(64), 20480, st[q], q * per, d);
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long sum; hipMemcpy(&sum, d, 8, hipMemcpyDeviceToHost);
    const double work = sum / 100.0 / 1e3 / 2048;
    printf("mixed spins, %d workgroups over %d queues: %.3f ms (host clock); sum / 2048 slots = %.3f ms -> %.1f %% over\n", grid, nq, ms, work, 100.0 * (ms - work) / work);
    hipFree(d);
}
static void run_mixed(int heavy_every)
{
    unsigned long long *d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipFuncSetAttribute((const void *)k_spin_mixed, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 24000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin_mixed, dim3(grid), dim3(64), 20480, 0, heavy_every, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long sum; hipMemcpy(&sum, d, 8, hipMemcpyDeviceToHost);
    const double work = sum / 100.0 / 1e3 / 2048;   // ms
    printf("mixed spins, %d workgroups%s: %.3f ms; sum of spins / 2048 slots = %.3f ms -> %.1f %% over\n", grid, heavy_every ? " + 13 ms ones among the first 2048" : "", ms, work,
           100.0 * (ms - work) / work);
    hipFree(d);
}

int main()
{
    run_mixed(0); run_mixed(0); run_mixed(16);
    run_mixed_q(1); run_mixed_q(1); run_mixed_q(2); run_mixed_q(4); run_mixed_q(8);
    const double ghz = 0.1;   // s_memrealtime: the 100 MHz reference clock
    for (unsigned long long c : {1000ull, 10000ull, 30000ull, 130000ull}) {     // 10 us, 100 us, 300 us, 1.3 ms (a task of the step scheduler)
        run<0>("no scratch", 12, c, ghz);
        run<20>("80 B scratch", 12, c, ghz);
    }
    return 0;
}
