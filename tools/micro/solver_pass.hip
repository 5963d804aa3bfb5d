// Floor of one colour pass of the sequential-impulse solver (substep() step 6d, cpArbiterApplyImpulse) in isolation.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -I../../benchpush_amd/csrc -o solver_pass solver_pass.hip
// The pass below is the source of the kernel's `iterate` lambda (no-bias copy and bias copy), run as the kernel runs it: one arbiter per lane in registers,
// `nact` warm arbiters spread over `ncol` colours, every colour pass under the EXEC mask (warm && level == lvl), 10 iterations per "sub-step".
//   MODE 0 = as in the kernel: body velocities gathered from / scattered to LDS velocity slots around every pass (three round trips: sv, sw, write-back)
//   MODE 1 = arithmetic only: the two bodies' velocities stay in registers (what a pass would cost if no body were shared between arbiters)
//   MODE 2 = LDS trips only: gather + scatter, no arithmetic
//   MODE 3 = side A in registers, side B through LDS (what a pass costs when one of its two bodies is unshared or of infinite mass)
// Since round 4's product change the no-bias LDS forms move only `w` of the (w, w_bias) slot (8-byte accesses), like the kernel.
// For every (mode, contacts, active lanes, colours) the program prints, at 1 / 2 / 3 / 4 wavefronts per SIMD (occupancy forced through the dynamic LDS size):
//   wave cycles per pass (s_memtime around the loop, mean over waves) and SIMD cycles per pass (= wave cycles / waves per SIMD: the throughput view).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bp_device.hpp"

struct Arb {
    double jn0, jt0, jn1, jt1;
    d2 n, r1_0, r2_0, r1_1, r2_1;
    double ma, ia, mb, ib, u;
    int slotA, slotB, count, level;
};

__device__ __forceinline__ void apply_ci(const Arb &A, int c, d2 &va, double &wa, d2 &vb, double &wb, d2 j)
{
    const d2 r1 = c ? A.r1_1 : A.r1_0, r2 = c ? A.r2_1 : A.r2_0;
    const d2 jn = vneg(j);
    va = vadd(va, vmul(jn, A.ma));
    wa += A.ia * vcross(r1, jn);
    vb = vadd(vb, vmul(j, A.mb));
    wb += A.ib * vcross(r2, j);
}

extern __shared__ double2 smem[];

template <int MODE, bool AB>
__global__ __launch_bounds__(64) void k_pass(double *out, unsigned long long *cyc, const double *__restrict__ init, const int nact, const int ncol, const int two, const int substeps)
{
    const int lane = threadIdx.x;
    d2 *sv = smem, *sw = smem + 97, *sb = smem + 194;
    for (int i = lane; i < 97; i += 64) { sv[i] = mk2(0.01 * i, -0.02 * i); sw[i] = mk2(1e-3 * i, 0.0); sb[i] = mk2(0.0, 0.0); }
    // every per-arbiter quantity comes from memory (per lane, unknown to the compiler), as in the kernel
    const double *in = init + lane * 32;
    Arb A;
    A.jn0 = in[0]; A.jt0 = in[1]; A.jn1 = in[2]; A.jt1 = in[3];
    A.n = mk2(in[4], in[5]); A.r1_0 = mk2(in[6], in[7]); A.r2_0 = mk2(in[8], in[9]); A.r1_1 = mk2(in[10], in[11]); A.r2_1 = mk2(in[12], in[13]);
    A.ma = in[14]; A.ia = in[15]; A.mb = in[16]; A.ib = in[17]; A.u = in[18];
    A.slotA = (int)in[19]; A.slotB = (int)in[20];
    A.count = (int)in[21 + (two ? 1 : 0)];
    const bool warm = lane < nact;
    A.level = 1 + (lane % ncol);
    const double nMass0 = in[23], tMass0 = in[24], nMass1 = in[25], tMass1 = in[26], bias0 = in[27], bias1 = in[28], bounce0 = in[29], bounce1 = in[30];
    double jBias0 = 0.0, jBias1 = 0.0;
    unsigned lvlmask = 0;
    for (int l = 1; l <= ncol; l++) if (ballot(warm && A.level == l)) lvlmask |= 1u << l;
    const int wA = (A.ma != 0.0) ? A.slotA : 96, wB = (A.mb != 0.0) ? A.slotB : 96;
    d2 rva = sv[A.slotA], rvb = sv[A.slotB], rwa = sw[A.slotA], rwb = sw[A.slotB], rba = mk2(0, 0), rbb = mk2(0, 0);
    lds_sync();
    int chg_all = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int ss = 0; ss < substeps; ss++) {
        for (int it = 0; it < 10; it++) {
            int chg = 0;
            for (unsigned lm = lvlmask; lm; lm &= lm - 1u) {
                const int lvl = __ffs((int)lm) - 1;
                if (warm && A.level == lvl) {
                    d2 va, vb, wa2, wb2, vba = mk2(0.0, 0.0), vbb = mk2(0.0, 0.0);
                    if (MODE == 1) { va = rva; vb = rvb; wa2 = rwa; wb2 = rwb; vba = rba; vbb = rbb; }
                    else {
                        if (MODE == 3) { va = rva; wa2 = rwa; vba = rba; }
                        else { va = sv[A.slotA]; if (AB) { wa2 = sw[A.slotA]; vba = sb[A.slotA]; } else { wa2 = mk2(0.0, 0.0); wa2.x = sw[A.slotA].x; } }
                        vb = sv[A.slotB]; if (AB) { wb2 = sw[A.slotB]; vbb = sb[A.slotB]; } else { wb2 = mk2(0.0, 0.0); wb2.x = sw[A.slotB].x; }
                    }
                    const d2 n = A.n;
                    if (MODE != 2) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            if (c == 0 || A.count > 1) {
                                const d2 r1 = c ? A.r1_1 : A.r1_0, r2 = c ? A.r2_1 : A.r2_0;
                                const double nMass = c ? nMass1 : nMass0, tMass = c ? tMass1 : tMass0;
                                const double bias = c ? bias1 : bias0, bounce = c ? bounce1 : bounce0;
                                const d2 v1 = vadd(va, vmul(vperp(r1), wa2.x));
                                const d2 v2 = vadd(vb, vmul(vperp(r2), wb2.x));
                                const d2 vr = vsub(v2, v1);
                                const double vrn = vdot(vr, n);
                                const double vrt = vdot(vr, vperp(n));
                                const double jbnOld = c ? jBias1 : jBias0;
                                double jBias = jbnOld;
                                if (AB) {
                                    const d2 vb1 = vadd(vba, vmul(vperp(r1), wa2.y));
                                    const d2 vb2 = vadd(vbb, vmul(vperp(r2), wb2.y));
                                    const double vbn = vdot(vsub(vb2, vb1), n);
                                    const double jbn = (bias - vbn) * nMass;
                                    jBias = fmax(jbnOld + jbn, 0.0);
                                }
                                const double jn = -(bounce + vrn) * nMass;
                                const double jnOld = c ? A.jn1 : A.jn0;
                                const double jnAcc = fmax(jnOld + jn, 0.0);
                                const double jtMax = A.u * jnAcc;
                                const double jt = -vrt * tMass;
                                const double jtOld = c ? A.jt1 : A.jt0;
                                const double jtAcc = fclampd(jtOld + jt, -jtMax, jtMax);
                                if (c) { jBias1 = jBias; A.jn1 = jnAcc; A.jt1 = jtAcc; }
                                else   { jBias0 = jBias; A.jn0 = jnAcc; A.jt0 = jtAcc; }
                                if (AB) {
                                    const double djb = jBias - jbnOld;
                                    chg |= __double2loint(djb) | (__double2hiint(djb) & 0x7FFFFFFF);
                                    const d2 jb = vmul(n, djb);
                                    const d2 jbneg = vneg(jb);
                                    vba = vadd(vba, vmul(jbneg, A.ma));
                                    wa2.y += A.ia * vcross(r1, jbneg);
                                    vbb = vadd(vbb, vmul(jb, A.mb));
                                    wb2.y += A.ib * vcross(r2, jb);
                                }
                                const double djn = jnAcc - jnOld, djt = jtAcc - jtOld;
                                chg |= __double2loint(djn) | __double2loint(djt) | ((__double2hiint(djn) | __double2hiint(djt)) & 0x7FFFFFFF);
                                const d2 j = vrotate(n, mk2(djn, djt));
                                apply_ci(A, c, va, wa2.x, vb, wb2.x, j);
                            }
                        }
                    }
                    if (MODE == 1) { rva = va; rvb = vb; rwa = wa2; rwb = wb2; rba = vba; rbb = vbb; }
                    else {
                        if (MODE == 3) { rva = va; rwa = wa2; rba = vba; }
                        else { sv[wA] = va; if (AB) { sw[wA] = wa2; sb[wA] = vba; } else sw[wA].x = wa2.x; }
                        sv[wB] = vb; if (AB) { sw[wB] = wb2; sb[wB] = vbb; } else sw[wB].x = wb2.x;
                    }
                }
                lds_sync();
            }
            chg_all |= chg;
            // the kernel's fixed-point test (never taken here: the perturbation below keeps the impulses moving)
            if (__builtin_expect(!ballot(warm && chg != 0) && ss < 0, 0)) break;
        }
        // next "sub-step": perturb the warm-start impulses so that the iteration does not sit on its fixed point
        A.jn0 = A.jn0 * 0.999 + 1e-3; A.jt0 = A.jt0 * 0.5; A.jn1 = A.jn1 * 0.999 + 1e-3; A.jt1 = A.jt1 * 0.5;
        if (MODE == 1) { rva = mk2(0.01 * lane, -0.02); rvb = mk2(0.0, 0.0); rwa = mk2(1e-3, 0.0); rwb = mk2(0.0, 0.0); }
        else if (lane < 48) { sv[2 * lane + 1] = mk2(0.0, 0.0); sw[2 * lane + 1] = mk2(0.0, 0.0); }
        lds_sync();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = A.jn0 + A.jt0 + A.jn1 + A.jt1 + jBias0 + jBias1 + rva.x + rvb.y + rwa.x + rwb.x + rba.x + rbb.x + sv[lane].x + (double)chg_all;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

static double *g_init;
template <int MODE, bool AB>
static void run_cfg(double *out, unsigned long long *cyc, int nact, int ncol, int two)
{
    const int substeps = 40;
    printf("mode %d bias %d contacts %d lanes %2d colours %d :", MODE, (int)AB, two ? 2 : 1, nact, ncol);
    for (int W = 1; W <= 4; W++) {
        const int blocks = 256 * 4 * W;
        const size_t lds = (160 * 1024) / (4 * W) - 64;   // exactly 4 W single-wave workgroups per CU
        (void)hipFuncSetAttribute((const void *)k_pass<MODE, AB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float ms = 0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k_pass<MODE, AB>), dim3(blocks), dim3(64), lds, 0, out, cyc, g_init, nact, ncol, two, substeps);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; i++) s += (double)h[i];
        const double passes = (double)substeps * 10 * ncol;
        const double wave = s / blocks / passes;
        printf("  W%d wave %7.1f simd %7.1f (%.3f ms)", W, wave, wave / W, ms);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\n");
}

int main()
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 64 * 8 * 4096); (void)hipMalloc(&cyc, 8 * 4096);
    {
        std::vector<double> h(64 * 32);
        for (int l = 0; l < 64; l++) {
            double *a = &h[l * 32];
            a[0] = 0.1 + l * 1e-3; a[1] = 0.01; a[2] = 0.2; a[3] = -0.01; a[4] = 0.6; a[5] = 0.8;
            a[6] = 0.3 + l * 1e-2; a[7] = -0.2; a[8] = -0.25; a[9] = 0.15; a[10] = 0.31; a[11] = 0.22; a[12] = -0.2; a[13] = -0.1;
            a[14] = (l & 1) ? 0.0 : 0.5; a[15] = (l & 1) ? 0.0 : 0.8; a[16] = 0.7; a[17] = 1.1; a[18] = 1.0;
            a[19] = (l * 2) % 96; a[20] = (l * 2 + 1) % 96; a[21] = 1; a[22] = 2;
            a[23] = 0.9; a[24] = 0.8; a[25] = 0.95; a[26] = 0.85; a[27] = 1e-3; a[28] = 2e-3; a[29] = 1e-4; a[30] = 2e-4;
        }
        (void)hipMalloc(&g_init, h.size() * 8);
        (void)hipMemcpy(g_init, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    printf("# cycles per colour pass: wave = one wavefront's view, simd = wave / waves per SIMD (throughput view); W = waves per SIMD\n");
    // the mean env of c2: 3.2 warm arbiters in 2.1 colours, 23 %% of the passes with a second contact
    for (int two = 0; two <= 1; two++) {
        run_cfg<0, false>(out, cyc, 3, 2, two);
        run_cfg<1, false>(out, cyc, 3, 2, two);
        run_cfg<0, true>(out, cyc, 3, 2, two);
        run_cfg<1, true>(out, cyc, 3, 2, two);
        run_cfg<3, false>(out, cyc, 3, 2, two);
        run_cfg<3, true>(out, cyc, 3, 2, two);
    }
    run_cfg<2, false>(out, cyc, 3, 2, 0);
    run_cfg<2, true>(out, cyc, 3, 2, 0);
    // lane-count dependence (the sparse-EXEC step of exec_mask*.hip), one colour
    for (int nact : {1, 2, 8, 9, 16, 64}) run_cfg<0, false>(out, cyc, nact, 1, 0);
    for (int nact : {1, 2, 8, 9, 16, 64}) run_cfg<1, false>(out, cyc, nact, 1, 0);
    // the heaviest env: 11 warm arbiters in 3-4 colours, two contacts
    run_cfg<0, false>(out, cyc, 11, 4, 1);
    run_cfg<1, false>(out, cyc, 11, 4, 1);
    return 0;
}
