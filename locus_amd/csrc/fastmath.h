// Branch-free fp64 helpers of the element-wise spike-and-slab update (both sweep kernels).
#pragma once
#include <hip/hip_runtime.h>

namespace locus {

// exp(x) for x <= 0, to < 1 ulp of truncation error and a few ulp of rounding: k = rint(x log2 e),
// r = x - k ln 2 (two-term Cody-Waite), degree-12 Taylor polynomial in Horner form, ldexp.
// 20 instructions with no branches and no table; at one wave per SIMD every fp64 VALU
// instruction costs 8 issue cycles, so the instruction COUNT of the chain is what matters.
__device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -800.0);  // exp underflows to 0 below -745.2; keeps k in int range
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;               // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);           // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);           // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);           // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);           // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);           // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);           // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);           // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);           // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// num / den for den in [1, 2], num in [0, 1]: v_rcp_f64 + two Newton steps + one residual
// correction (no scaling / fix-up needed in this range).
__device__ __forceinline__ double div_1to2(double num, double den)
{
    double rcp = __builtin_amdgcn_rcp(den);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    const double qv = num * rcp;
    return fma(fma(-den, qv, num), rcp, qv);
}

}  // namespace locus
