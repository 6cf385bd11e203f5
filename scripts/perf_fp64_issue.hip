// Issue cadence of fp64 VALU instructions for ONE wave on a SIMD (the chain wave's situation):
// N dependent FMAs against the same count spread over 2 / 4 / 8 independent chains, and the same
// with two waves on the SIMD.   build: hipcc -O2 --offload-arch=gfx950 scripts/perf_fp64_issue.hip -o scripts/bin/perf_fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k(double* out, long long* cyc, double a, double b, int iters)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
void run(int threads, double* out, long long* cyc)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.999, 1e-3, iters);
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.999, 1e-3, iters);
    long long h;
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %4d (%d wave(s) per SIMD)  %d independent chain(s): %.2f cycles per FMA instruction per wave\n",
           threads, threads / 256 ? threads / 256 : 1, CH, (double)h / (64.0 * iters));
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    for (int threads : {64, 256, 512}) {
        run<1>(threads, out, cyc); run<2>(threads, out, cyc); run<4>(threads, out, cyc); run<8>(threads, out, cyc);
    }
    return 0;
}
