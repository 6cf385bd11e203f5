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
    // exp underflows to 0 below -745.2; the clamp keeps k in int range.  Written as a select, not
    // fmax: fmax(NaN, -800) = -800 would turn a NaN log-odds into gam = 1, whereas the reference
    // propagates it (std::exp in src/utils.cpp:3-12); with the select a NaN stays a NaN.
    x = (x < -800.0) ? -800.0 : x;
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

// 2^(j/64), j = 0 .. 63, correctly rounded (scripts/microbench/exp_table_accuracy.c): the table of the
// chain wave's exponential in mfma_sweep1_kernel.h, copied to LDS once per workgroup.
//     exp(x) = 2^k 2^(j/64) exp(r),  N = 64 k + j = round(64 x / ln 2),  |r| <= ln 2 / 128,
// exp(r) - 1 by a degree-5 polynomial (truncation 3.5e-17 relative): 6 instructions after r where
// the table-free form needs 14.  Max error 1.03 ulp over [-708, 705], 1.56 ulp for the resulting
// 1 / (1 + exp(a)) against long double on 2e7 points (same file).
static __device__ const double kExp2Tab64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};

// The same function with a SHORT dependent path, for the chain wave that runs alone on its SIMD
// (mfma_sweep1_kernel.h: there every dependent fp64 instruction costs its full ~8 cycle latency):
// p(r) = 1 + r q(r), q of degree 11 in Estrin form (pairs, then r^2, then r^4) -- 14 instructions,
// 5 dependent levels after r instead of 13.  The final 1 + r q keeps the last rounding where
// Horner has it: max error over [-745, 0] 2.5 ulp against 2.3 ulp for exp_nonpos (measured
// against long double, 2 * 10^7 points).
__device__ __forceinline__ double exp_nonpos_short(double x)
{
    x = (x < -800.0) ? -800.0 : x;  // a select, not fmax: see exp_nonpos
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    const double r2 = r * r;
    const double q0 = fma(r, 0.5, 1.0);
    const double q1 = fma(r, 4.16666666666666666667e-02, 1.66666666666666666667e-01);
    const double q2 = fma(r, 1.38888888888888888889e-03, 8.33333333333333333333e-03);
    const double q3 = fma(r, 2.48015873015873015873e-05, 1.98412698412698412698e-04);
    const double q4 = fma(r, 2.75573192239858906526e-07, 2.75573192239858906526e-06);
    const double q5 = fma(r, 2.08767569878680989792e-09, 2.50521083854417187751e-08);
    const double r4 = r2 * r2;
    const double s0 = fma(q1, r2, q0);
    const double s1 = fma(q3, r2, q2);
    const double s2 = fma(q5, r2, q4);
    const double q = fma(fma(s2, r4, s1), r4, s0);
    return __builtin_ldexp(fma(r, q, 1.0), (int)k);
}

// num / den for den in [1, 2], num in [0, 1]: v_rcp_f64 (2^-24.4 relative on gfx950) + ONE Newton
// step (2^-49) + the quotient's residual correction, which squares the error once more (no scaling
// / fix-up needed in this range).  scripts/microbench/div_accuracy.hip, 1.7e7 quotients: 0.5 ulp
// at worst and the same 4107 results off the correctly rounded one as with two Newton steps.
__device__ __forceinline__ double div_1to2(double num, double den)
{
    double rcp = __builtin_amdgcn_rcp(den);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    const double qv = num * rcp;
    return fma(fma(-den, qv, num), rcp, qv);
}

// 1 / x for positive, finite, normal x with 1 / x normal too: v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double rcp_pos_normal(double x)
{
    double rcp = __builtin_amdgcn_rcp(x);
    rcp = fma(fma(-x, rcp, 1.0), rcp, rcp);
    return fma(fma(-x, rcp, 1.0), rcp, rcp);
}

// log(x) for positive, finite, normal x (no special cases): fdlibm's e_log.c scheme -- x = 2^e m,
// m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log m = f - hfsq + s (hfsq + R(s^2)) --
// with v_frexp_* for the split and div_1to2-style Newton steps for the quotient (2 + f in
// [1.7, 2.42]).  < 1 ulp of truncation error; about half the instructions of the library log.
__device__ __forceinline__ double log_pos_normal(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);       // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double rcp = __builtin_amdgcn_rcp(den);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    double sq = f * rcp;
    sq = fma(fma(-den, sq, f), rcp, sq);             // s = f / (2 + f)
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                              3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    // fdlibm: dk*ln2_hi - ((hfsq - (s*(hfsq+R) + dk*ln2_lo)) - f)
    return fma(dk, 6.93147180369123816490e-01,
               -((hfsq - fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

}  // namespace locus
