// Latency / issue interval of the fp64 instructions on the chain's critical path, ONE wave on a
// SIMD (the chain wave's situation).  Cycles are shader cycles: s_memtime deltas, cross-checked
// against s_memrealtime (100 MHz) to get the clock the chip actually held.
// build: hipcc -O2 --offload-arch=gfx950 scripts/microbench/fp64_chain.hip -o scripts/microbench/fp64_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// MODE 0: dependent v_fma_f64 chain        1: 4 independent chains   2: dependent v_mul_f64
//      3: dependent v_add_f64              4: v_rcp_f64 dependent     5: v_rndne dependent
//      6: v_ldexp dependent                7: dependent fma with an SGPR operand
//      8: dependent v_cndmask_b32 pair     9: dependent fma, 16 lanes active
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double a, double b, int iters)
{
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int ei = (int)(a * 0.0);
    if (MODE == 9 && threadIdx.x >= 16) return;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 9) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
        if (MODE == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));) }
        if (MODE == 2) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x0) : "v"(a));) }
        if (MODE == 3) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(b));) }
        if (MODE == 4) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(x0));) }
        if (MODE == 5) { REP64(asm volatile("v_rndne_f64 %0, %0" : "+v"(x0));) }
        if (MODE == 6) { REP64(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x0) : "v"(ei));) }
        if (MODE == 7) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "s"(a), "v"(b));) }
        if (MODE == 8) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ei) : "v"(ei));) }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    out[threadIdx.x] = x0 + x1 + x2 + x3 + ei;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

// two waves on ONE SIMD (threads 0..63 and 256..319 of a 320-thread block land on SIMD 0 twice?
// no: use a 512-thread block, waves w and w + 4 share a SIMD): wave 0 runs the dependent FMA chain,
// wave 4 runs back-to-back fp64 MFMAs (MFMA = 1) or independent fp64 FMAs (MFMA = 0)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int PARTNER>
__global__ void k2(double* out, unsigned long long* cyc, double a, double b, int iters)
{
    const int w = threadIdx.x >> 6;
    double x0 = threadIdx.x * 1e-3 + 1.0;
    d4 acc = {0, 0, 0, 0};
    unsigned long long t0, t1;
    if (w == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    } else if (w == 4) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters * (PARTNER == 1 ? 4 : 1); ++i) {
            if (PARTNER == 1) { REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);) }
            if (PARTNER == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));) }
            if (PARTNER == 3) { REP64(asm volatile("ds_read_b64 %0, %1" : "=v"(x0) : "v"(0));) }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (threadIdx.x == 256) cyc[2] = t1 - t0;
    }
    out[threadIdx.x] = x0 + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE> void run(const char* what, double* out, unsigned long long* cyc)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 0.999, 1e-3, iters);
    unsigned long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per instruction   (clock %.0f MHz)\n", what, (double)h[0] / (64.0 * iters),
           (double)h[0] / (double)h[1] * 100.0);
}
template <int P> void run2(const char* what, double* out, unsigned long long* cyc)
{
    const int iters = 2000;
    hipMemset(cyc, 0, 32);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k2<P>, dim3(1), dim3(512), 0, 0, out, cyc, 0.999, 1e-3, iters);
    unsigned long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s chain %6.2f cycles per FMA; partner %8.2f cycles per instruction\n", what,
           (double)h[0] / (64.0 * iters), (double)h[2] / ((P == 1 ? 64.0 : 64.0) * iters));
}
int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    run<0>("dependent v_fma_f64", out, cyc);
    run<1>("4 independent v_fma_f64 chains", out, cyc);
    run<2>("dependent v_mul_f64", out, cyc);
    run<3>("dependent v_add_f64", out, cyc);
    run<4>("dependent v_rcp_f64", out, cyc);
    run<5>("dependent v_rndne_f64", out, cyc);
    run<6>("dependent v_ldexp_f64", out, cyc);
    run<7>("dependent v_fma_f64, SGPR operand", out, cyc);
    run<8>("dependent v_cndmask_b32", out, cyc);
    run<9>("dependent v_fma_f64, 16 lanes active", out, cyc);
    run2<0>("chain alone (512-thread block, others idle)", out, cyc);
    run2<1>("chain + fp64 MFMA stream on the same SIMD", out, cyc);
    run2<2>("chain + dependent FMA chain on the same SIMD", out, cyc);
    run2<3>("chain + ds_read stream on the same SIMD", out, cyc);
    return 0;
}
