// Internal launchers of liblocus_hip (device pointers; async on `st`).
#pragma once
#include "common.h"

namespace locus {

int pack_x(const double* X, double* Xp, double* xtx, int n, int p, int ld, hipStream_t st);
int build_steps(const double* shuffled_ind, const double* lo, const double* l1, const double* xtx,
                StepRec* steps, int* visited, int* err, int p, int nsteps, int ntotal,
                hipStream_t st);
int residual_transform(const double* Y, const double* A, const double* Zmu, const double* W2,
                       double* out, size_t len, int to_residual, hipStream_t st);
int transpose(const double* in, double* out, int rows, int cols, hipStream_t st);
int gamma_rows(const double* gam, double* rs_gam, double* ent, int p, int d, double eps,
               hipStream_t st);
int state_cols(double* state, const double* s2k, const double* s2jk, double* st_partial, int p,
               int d, int nchunks, hipStream_t st);

int batch_update(double* state, const double* lo, const double* l1, const double* fac,
                 const double* hinv, const double* cstk, double log_sig2_inv, double nm1,
                 double cfac, int probit, int p, int d, hipStream_t st);

// binary-response drivers (link.hip)
int m2_minus_b2(const double* state, const double* s2jk, double s2c, double* D, size_t pd,
                hipStream_t st);
int s2_from_xwx(const double* xwx, double sig2_inv, double* s2, size_t len, hipStream_t st);
int half_log(const double* in, double* out, size_t len, hipStream_t st);
int logit_chi_psi(const double* Q, const double* M, const double* Zmu, const double* Y,
                  const double* Z, const double* s2a, const double* chi_old, const double* W2_old,
                  double* chi_new, double* W2_new, double* ey, int n, int d, int q, hipStream_t st);
int logit_sig2_alpha(const double* Z, const double* W2, const double* zeta2_inv, double* s2a,
                     int n, int d, int q, hipStream_t st);
int alpha_step(const double* Yw, const double* W2, const double* M, double* Zmu, const double* Zi,
               const double* s2a_i, double* alpha_i, int n, int d, hipStream_t st);
int probit_w(const double* Y, const double* M, const double* Zmu, const double* Q, const double* zs,
             double* W, double* ey, int n, int d, hipStream_t st);
int bin_rows(const double* state, const double* s2jk, double s2c, double* rs_gam, double* ent,
             double* gls, double* m2, int p, int d, double eps, hipStream_t st);

// C (m x n) = alpha * op(A) * op(B) + beta * C with arbitrary element strides;
// fp64 MFMA (v_mfma_f64_16x16x4_f64).  sqA squares A's elements on load (X^2 products
// of update_vb.R:107-111,163-167 without materialising X^2).
struct GemmDesc {
    int m, n, k;
    const double* A; long sa_m, sa_k;
    const double* B; long sb_k, sb_n;
    double* C; long sc_m, sc_n;
    double alpha, beta;
    int sqA, sqB;
    // split-K launches (set by gemm_f64 itself): workgroup z covers k in [z kchunk, (z + 1) kchunk)
    // and writes its partial product to C + z * cz
    int kchunk = 0;
    long cz = 0;
    int xcd_order = 0;  // 1: XCD-contiguous tile order (LOCUS_HIP_GEMM_XCD, measurements only)
};
int gemm_f64(const GemmDesc& g, hipStream_t st);
int identity_tail(const locus_hip_tail_args& a, int phase, hipStream_t st);

}  // namespace locus
