/* Accuracy of the table-driven exponential and of gam = 1 / (1 + exp(a)) as the chain wave of
 * mfma_sweep1_kernel.h evaluates them (same constants, same operation order; fma() is exact on the
 * host as v_fma_f64 is on the GPU; the reciprocal seed is modelled as a 2^-24-accurate value).
 * Reference: long double (64-bit significand).  build: gcc -O2 -o /tmp/exp_acc exp_table_accuracy.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

static const double kTab[64] = {
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

static double exp_tab(double x)   /* x in [-708, 700] */
{
    const double MAGIC = 6755399441055744.0;              /* 1.5 * 2^52 */
    const double kf0 = fma(x, 92.33248261689366, MAGIC);  /* low word = N = round(64 x / ln 2) */
    int64_t bits;
    memcpy(&bits, &kf0, 8);
    const int32_t N = (int32_t)(uint32_t)bits;
    const double nf = kf0 - MAGIC;
    double r = fma(nf, -6.93147180369123816490e-01 / 64, x);
    r = fma(nf, -1.90821492927058770002e-10 / 64, r);
    const double r2 = r * r;
    const double A = fma(r, 1.66666666666666666667e-01, 0.5);
    const double B = fma(r, 8.33333333333333333333e-03, 4.16666666666666666667e-02);
    const double C = fma(B, r2, A);
    const double D = fma(C, r2, r);                        /* exp(r) - 1 */
    const double T = kTab[N & 63];
    return ldexp(fma(T, D, T), N >> 6);
}

static double gam_of(double a, uint64_t* seed)
{
    double x = fmax(a, -708.0);
    x = fmin(x, 700.0);
    const double E = exp_tab(x);
    const double den = 1.0 + E;
    /* v_rcp_f64: 2^-24.4 on gfx950; perturb the exact reciprocal by up to 2^-24 */
    *seed = *seed * 6364136223846793005ull + 1442695040888963407ull;
    const double pert = ((double)(int32_t)(*seed >> 32)) * (1.0 / 2147483648.0) * 5.9e-8;
    double rcp = (1.0 / den) * (1.0 + pert);
    double tt = fma(-den, rcp, 1.0);
    rcp = fma(tt, rcp, rcp);
    const double qv = 1.0 * rcp;
    tt = fma(-den, qv, 1.0);
    return fma(tt, rcp, qv);
}

int main(void)
{
    uint64_t seed = 12345;
    double max_e = 0, max_g = 0, at_e = 0, at_g = 0;
    const long NPT = 20000000;
    for (long i = 0; i < NPT; ++i) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(seed >> 11) * (1.0 / 9007199254740992.0);
        double a;
        switch (i & 3) {
        case 0: a = -708.0 + u * 1408.0; break;       /* whole range */
        case 1: a = -40.0 + u * 80.0; break;          /* where gam is neither 0 nor 1 */
        case 2: a = -2.0 + u * 4.0; break;
        default: a = (u - 0.5) * 1e-3; break;
        }
        const long double le = expl((long double)a);
        const double e = exp_tab(a);
        const double ue = (double)fabsl(((long double)e - le) / le) * 0x1p52;
        if (ue > max_e) { max_e = ue; at_e = a; }
        const long double lg = 1.0L / (1.0L + le);
        const double g = gam_of(a, &seed);
        const double ug = (double)fabsl(((long double)g - lg) / lg) * 0x1p52;
        if (ug > max_g) { max_g = ug; at_g = a; }
    }
    printf("exp_tab: max rel. error %.3f x 2^-52 at %.17g\n", max_e, at_e);
    printf("gam    : max rel. error %.3f x 2^-52 at %.17g\n", max_g, at_g);
    /* edge cases */
    uint64_t s2 = 1;
    const double edge[] = {-1e300, -745.0, -708.0, -0.0, 0.0, 700.0, 745.0, 1e300, INFINITY, -INFINITY};
    for (unsigned k = 0; k < sizeof edge / sizeof *edge; ++k)
        printf("gam(%g) = %.17g\n", edge[k], gam_of(edge[k], &s2));
    return 0;
}
