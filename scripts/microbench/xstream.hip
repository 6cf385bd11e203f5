// Microbenchmark (not part of the product): one wave streams MFMA A operands from global memory
// through a register ring (16-byte loads = 2 operands) and feeds T dependent-chain MFMAs per operand.
// Reports cycles per operand-pair for ring depths, T, footprint and workgroup counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NR2, int T>
__global__ void __launch_bounds__(256) k(const double2* __restrict__ X, size_t pairs_per_wave_pass,
                                         int passes, size_t wave_stride, double* out,
                                         unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double2* base = X + (size_t)s * wave_stride + lane;
    d4 acc[T];
    for (int t = 0; t < T; ++t) acc[t] = d4{0, 0, 0, 0};
    double b = 1.0 + lane * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int ps = 0; ps < passes; ++ps) {
        const double2* xp = base;
        double2 ring[NR2];
#pragma unroll
        for (int i = 0; i < NR2; ++i) { ring[i] = *xp; xp += 64; }
        for (size_t g0 = 0; g0 < pairs_per_wave_pass; g0 += NR2) {
#pragma unroll
            for (int i = 0; i < NR2; ++i) {
                const double2 a = ring[i];
                ring[i] = *xp; xp += 64;   // reads NR2 pairs past the end: buffer is padded
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b, acc[t], 0, 0, 0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0; for (int t = 0; t < T; ++t) r += acc[t].x + acc[t].w;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NR2, int T>
void run(const double2* X, size_t pairs, int passes, int nwg, double* out, unsigned long long* cyc)
{
    size_t wave_stride = (pairs + 64) * 64;
    hipLaunchKernelGGL((k<NR2, T>), dim3(nwg), dim3(256), 0, 0, X, pairs, passes, wave_stride, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR2, T>), dim3(nwg), dim3(256), 0, 0, X, pairs, passes, wave_stride, out, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= nwg;
    double per_pair = c / ((double)pairs * passes);
    double bytes = (double)nwg * 4 * pairs * passes * 1024.0;
    printf("ring=%2d pairs T=%d footprint/wave=%6zu KB nwg=%3d: %7.1f cyc/pair (ideal %d)  %6.2f TB/s  %5.1f B/clk/CU\n",
           NR2, T, pairs, nwg, per_pair, 128 * T, bytes / (ms * 1e-3) / 1e12, 4 * 1024.0 / per_pair);
}

int main()
{
    const size_t maxpairs = 1 << 16;  // 64 MB per wave slot x 4
    double2* X; hipMalloc(&X, 4 * (maxpairs + 64) * 64 * sizeof(double2));
    hipMemset(X, 0, 4 * (maxpairs + 64) * 64 * sizeof(double2));
    double* out; hipMalloc(&out, 256 * 256 * 8);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
    for (int nwg : {8, 256}) {
        for (size_t pairs : {(size_t)512, (size_t)32768}) {   // 2 MB (L2) vs 128 MB total footprint
            int passes = (int)(65536 / pairs);
            run<4, 1>(X, pairs, passes, nwg, out, cyc);
            run<8, 1>(X, pairs, passes, nwg, out, cyc);
            run<16, 1>(X, pairs, passes, nwg, out, cyc);
            run<32, 1>(X, pairs, passes, nwg, out, cyc);
            run<4, 2>(X, pairs, passes, nwg, out, cyc);
            run<8, 2>(X, pairs, passes, nwg, out, cyc);
            run<16, 2>(X, pairs, passes, nwg, out, cyc);
        }
    }
    return 0;
}
