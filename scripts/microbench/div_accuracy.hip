// Accuracy of num / den, den in [1, 2], num in [0, 1], from v_rcp_f64 with 0, 1, 2 Newton steps and
// the quotient's residual correction (fastmath.h: div_1to2), against the correctly rounded quotient.
// build: hipcc --offload-arch=gfx950 -O2 scripts/microbench/div_accuracy.hip -o scripts/microbench/div_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NEWTON>
__device__ double div_n(double num, double den)
{
    double rcp = __builtin_amdgcn_rcp(den);
#pragma unroll
    for (int i = 0; i < NEWTON; ++i) rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    const double qv = num * rcp;
    return fma(fma(-den, qv, num), rcp, qv);
}

__global__ void run(const double* num, const double* den, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __builtin_amdgcn_rcp(den[i]);
    out[n + i] = div_n<0>(num[i], den[i]);
    out[2 * n + i] = div_n<1>(num[i], den[i]);
    out[3 * n + i] = div_n<2>(num[i], den[i]);
}

int main()
{
    const int n = 1 << 24;
    std::vector<double> hn(n), hd(n), ho(4 * (size_t)n);
    srand48(7);
    for (int i = 0; i < n; ++i) {
        hd[i] = 1.0 + drand48();
        hn[i] = (i & 1) ? 1.0 : hd[i] - 1.0;   // the two numerators of the sigmoid: 1 and e = den - 1
        if ((i & 7) == 7) hn[i] = drand48();
    }
    double *dn, *dd, *dout;
    hipMalloc(&dn, n * 8); hipMalloc(&dd, n * 8); hipMalloc(&dout, 4 * (size_t)n * 8);
    hipMemcpy(dn, hn.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dd, hd.data(), n * 8, hipMemcpyHostToDevice);
    run<<<(n + 255) / 256, 256>>>(dn, dd, dout, n);
    hipMemcpy(ho.data(), dout, 4 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* names[4] = {"v_rcp_f64 alone (1/den)", "0 Newton + correction", "1 Newton + correction",
                            "2 Newton + correction"};
    for (int v = 0; v < 4; ++v) {
        double worst = 0.0;
        long wrong = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = (v == 0) ? 1.0L / (long double)hd[i] : (long double)hn[i] / (long double)hd[i];
            const double got = ho[(size_t)v * n + i];
            const double r = (double)ref;                      // correctly rounded
            if (got != r) ++wrong;
            const double ulp = std::ldexp(1.0, std::ilogb(r == 0 ? 1e-300 : r) - 52);
            const double e = (double)fabsl((long double)got - ref) / ulp;
            if (e > worst) worst = e;
        }
        printf("%-26s max error %.3e ulp, not correctly rounded: %ld of %d\n", names[v], worst, wrong, n);
    }
    return 0;
}
