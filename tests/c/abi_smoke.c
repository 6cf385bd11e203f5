/* The C ABI from a plain C program (no Python, no torch): what an Rcpp stub sees.
 * Runs coreLoop through the stateless entry point and through the resident context on the same
 * inputs and checks that both give the same four matrices and consistent reductions.
 *   gcc -O1 -I include tests/c/abi_smoke.c -o abi_smoke -L locus_amd -llocus_hip -lm
 * (tests/test_resident_gpu.py builds and runs it on the GPU box). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "locus_hip.h"

static uint64_t s_ = 88172645463325252ull;
static double unif(void)
{
    s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17;
    return (double)(s_ >> 11) / 9007199254740992.0;
}
static double gauss(void) { return sqrt(-2.0 * log(unif() + 1e-300)) * cos(6.283185307179586 * unif()); }

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != 0) {                                                               \
            printf("FAIL %s -> %d: %s\n", #call, rc_, locus_hip_last_error());        \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(void)
{
    const int64_t n = 120, p = 70, d = 13;
    double *X = malloc(sizeof(double) * n * p), *Y = malloc(sizeof(double) * n * d);
    double *gam = malloc(sizeof(double) * p * d), *mu = malloc(sizeof(double) * p * d);
    double *beta = malloc(sizeof(double) * p * d), *M = calloc(n * d, sizeof(double));
    double *lo = malloc(sizeof(double) * p), *l1 = malloc(sizeof(double) * p), *ind = malloc(sizeof(double) * p);
    double *ltau = malloc(sizeof(double) * d), *s2 = malloc(sizeof(double) * d), *tau = malloc(sizeof(double) * d);
    for (int64_t j = 0; j < p; ++j) {          /* standardised columns (R: scale(X)) */
        double m = 0, v = 0;
        for (int64_t i = 0; i < n; ++i) { X[j * n + i] = gauss(); m += X[j * n + i]; }
        m /= n;
        for (int64_t i = 0; i < n; ++i) { X[j * n + i] -= m; v += X[j * n + i] * X[j * n + i]; }
        v = sqrt(v / (n - 1));
        for (int64_t i = 0; i < n; ++i) X[j * n + i] /= v;
        lo[j] = -3.0 + 0.1 * gauss(); l1[j] = -0.05; ind[j] = (double)j;
    }
    for (int64_t j = p - 1; j > 0; --j) {      /* a permutation, as sample(0:(p-1)) */
        int64_t k = (int64_t)(unif() * (j + 1)); double t = ind[j]; ind[j] = ind[k]; ind[k] = t;
    }
    for (int64_t e = 0; e < n * d; ++e) Y[e] = gauss();
    for (int64_t e = 0; e < p * d; ++e) { gam[e] = pow(unif(), 6.0); mu[e] = 0.3 * gauss(); beta[e] = gam[e] * mu[e]; }
    for (int64_t k = 0; k < d; ++k)            /* mat_x_m1 = X beta */
        for (int64_t j = 0; j < p; ++j)
            for (int64_t i = 0; i < n; ++i) M[k * n + i] += X[j * n + i] * beta[k * p + j];
    const double sig2_inv = 0.7;
    for (int64_t k = 0; k < d; ++k) { tau[k] = 0.5 + unif(); ltau[k] = log(tau[k]) - 0.01; s2[k] = 1.0 / ((n - 1 + sig2_inv) * tau[k]); }

    if (locus_hip_device_count() < 1) { printf("FAIL no device: %s\n", locus_hip_last_error()); return 1; }

    /* resident context first (the inputs are still untouched) */
    locus_hip_ctx* ctx = NULL;
    double *g2 = malloc(sizeof(double) * p * d), *m2 = malloc(sizeof(double) * p * d);
    double *b2 = malloc(sizeof(double) * p * d), *M2 = malloc(sizeof(double) * n * d);
    double *rs = malloc(sizeof(double) * p), *cs = malloc(sizeof(double) * 4 * d), ent = 0.0;
    CHECK(locus_hip_ctx_create(X, Y, n, p, d, &ctx));
    CHECK(locus_hip_ctx_set_state(ctx, gam, mu, NULL));
    CHECK(locus_hip_ctx_sweep(ctx, lo, l1, log(sig2_inv), ltau, s2, tau, ind, 1.0, 2.220446049250313e-16,
                              rs, cs, &ent));
    CHECK(locus_hip_ctx_get_state(ctx, g2, m2, b2, M2));
    CHECK(locus_hip_ctx_destroy(ctx));

    /* stateless drop-in: mutates gam, beta, M, mu in place like src/coreLoop.cpp */
    CHECK(locus_hip_core_loop(X, Y, gam, lo, l1, log(sig2_inv), ltau, beta, M, mu, s2, tau, ind, 1.0, n, p, d));

    double worst = 0.0;
    for (int64_t e = 0; e < p * d; ++e) {
        worst = fmax(worst, fabs(g2[e] - gam[e]));
        worst = fmax(worst, fabs(m2[e] - mu[e]) / (1e-3 + fabs(mu[e])));
        worst = fmax(worst, fabs(b2[e] - beta[e]));
    }
    for (int64_t e = 0; e < n * d; ++e) worst = fmax(worst, fabs(M2[e] - M[e]));
    double rs_err = 0.0, cs_err = 0.0;
    for (int64_t j = 0; j < p; ++j) {
        double s = 0; for (int64_t k = 0; k < d; ++k) s += gam[k * p + j];
        rs_err = fmax(rs_err, fabs(s - rs[j]));
    }
    for (int64_t k = 0; k < d; ++k) {
        double s = 0, r = 0;
        for (int64_t j = 0; j < p; ++j) s += gam[k * p + j];
        for (int64_t i = 0; i < n; ++i) r += (Y[k * n + i] - M[k * n + i]) * (Y[k * n + i] - M[k * n + i]);
        cs_err = fmax(cs_err, fabs(s - cs[k]));
        cs_err = fmax(cs_err, fabs(r - cs[3 * d + k]) / r);
    }
    /* a bad permutation is refused and leaves the arguments untouched */
    ind[0] = ind[1];
    double g0 = gam[0];
    int rc = locus_hip_core_loop(X, Y, gam, lo, l1, log(sig2_inv), ltau, beta, M, mu, s2, tau, ind, 1.0, n, p, d);
    printf("stateless vs resident: %.3e  rowSums: %.3e  colSums/rss: %.3e  ent: %.6f  bad perm rc: %d\n",
           worst, rs_err, cs_err, ent, rc);
    if (worst > 1e-10 || rs_err > 1e-10 || cs_err > 1e-10 || rc != LOCUS_HIP_EPERM || gam[0] != g0 ||
        !(ent < 0.0)) {
        printf("FAIL\n");
        return 1;
    }
    printf("OK\n");
    return 0;
}
