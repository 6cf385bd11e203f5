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
                 int probit, int p, int d, hipStream_t st);

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
};
int gemm_f64(const GemmDesc& g, hipStream_t st);

}  // namespace locus
