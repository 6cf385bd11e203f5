// Microbenchmark (not part of the product): issue rate of the fp64 MFMA / FMA instructions on
// gfx950, cycles per instruction per SIMD, for 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(double* out, unsigned long long* cyc, int iters)
{
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        } else if (MODE == 2) {
            s0 = fma(a, b, s0); s1 = fma(a, b, s1); s2 = fma(a, b, s2); s3 = fma(a, b, s3);
            s4 = fma(a, b, s4); s5 = fma(a, b, s5); s6 = fma(a, b, s6); s7 = fma(a, b, s7);
        } else {  // single dependent accumulator chain of 16x16x4
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] =
        c0.x + c1.y + c2.z + c3.w + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, double flops_per_inst)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {
        const int thr = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(thr), 0, 0, out, cyc, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(thr), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double cy = 0; for (auto v : h) cy += v; cy /= 256;
        double ninst_simd = (double)iters * per_iter * wps;  // instructions issued on one SIMD
        double tf = flops_per_inst * ninst_simd * 4 * 256 / (ms * 1e-3) / 1e12;
        printf("%-22s waves/SIMD=%d  cycles/inst/SIMD=%6.2f  ms=%7.3f  TFLOP/s=%6.1f  clk=%.2f GHz\n",
               name, wps, cy / ninst_simd, ms, tf, cy / (ms * 1e6));
    }
}

int main()
{
    run<0>("mfma_f64_16x16x4", 4, 2048.0);
    run<3>("mfma_f64_16x16x4 dep", 4, 2048.0);
    run<1>("mfma_f64_4x4x4", 4, 512.0);
    run<2>("v_fma_f64", 8, 128.0);
    return 0;
}
