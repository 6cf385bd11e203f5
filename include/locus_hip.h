/*
 * locus_hip.h -- C ABI of liblocus_hip.so, the MI355X (gfx950) implementation of the
 * CAVI-sweep hot path of the `locus` R package.
 *
 * Conventions (same as the reference, src/locus_types.h:8-13):
 *   - every matrix is `double`, column-major, unit inner stride, no alignment
 *     requirement beyond 8 bytes;  A[r, c] = A[r + c * nrow];
 *   - `shuffled_ind` holds 0-based covariate indices stored as doubles
 *     (R/locus_core.R:88, src/coreLoop.cpp:31,38) and must be a permutation of 0..p-1;
 *   - gam_vb, beta_vb, mat_x_m1, mu_beta_vb are updated IN PLACE
 *     (src/coreLoop.cpp:21,26-28); everything else is read-only;
 *   - functions return 0 on success, a negative LOCUS_HIP_E* code or a positive
 *     hipError_t otherwise; locus_hip_last_error() gives the text (thread-local).
 *
 * Section 1 mirrors, one to one, the eight `.Call` routines the reference registers
 * (src/RcppExports.cpp:191-201) with the shapes R passes implicitly made explicit
 * (n = nrow(X), p = ncol(X), d = ncol(Y)).  Host pointers; stateless; synchronous.
 *
 * Section 2 is the device-resident API the drivers use (device pointers, asynchronous
 * on the given hipStream_t, passed as void*; NULL = default stream).  It has no
 * counterpart in the reference, which has no device; it exists so that the p x d
 * state never crosses PCIe between CAVI iterations.
 */
#ifndef LOCUS_HIP_H
#define LOCUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOCUS_HIP_EINVAL (-1)       /* bad argument (shape, null pointer, ...)   */
#define LOCUS_HIP_EUNSUPPORTED (-2) /* shape outside what the kernels cover      */
#define LOCUS_HIP_EPERM (-3)        /* shuffled_ind is not a permutation of 0..p-1 */
#define LOCUS_HIP_ETIMEOUT (-4)     /* a sweep whose workgroups cooperate in clusters (rows of a
                                       response tile cut over 2-3 CUs) gave up waiting for a
                                       partner workgroup: the cluster was not resident at once
                                       (another process or a CU mask held CUs, or two such sweeps
                                       overlapped on one GPU).  The sweep stopped part-way:
                                       gam_vb / mu_beta_vb / beta_vb rows of the covariates
                                       visited so far hold new values, the residual and the
                                       statistics are those from before the sweep.  Device-
                                       resident callers re-initialise the residual from the state
                                       (mat_x_m1 = X beta_vb) or restart; the stateless entry
                                       points of section 1 return without touching the caller's
                                       buffers.  The reference's sweep cannot fail half-way
                                       (src/coreLoop.cpp:36-52).  Serialise cluster sweeps per GPU. */

const char* locus_hip_last_error(void);
int locus_hip_version(void);      /* 100 * major + minor */
int locus_hip_device_count(void); /* visible HIP devices, 0 if none / no driver */

/* ------------------------------------------------------------------------- */
/* 1. Stateless mirrors of the reference's .Call routines (host pointers)    */
/* ------------------------------------------------------------------------- */

/* replaces coreLoop, src/coreLoop.cpp:19-54 (R wrapper R/RcppExports.R:16-18) */
int locus_hip_core_loop(const double* X, const double* Y, double* gam_vb,
                        const double* log_om_vb, const double* log_1_min_om_vb,
                        double log_sig2_inv_vb, const double* log_tau_vb, double* beta_vb,
                        double* mat_x_m1, double* mu_beta_vb, const double* sig2_beta_vb,
                        const double* tau_vb, const double* shuffled_ind, double c,
                        int64_t n, int64_t p, int64_t d);

/* replaces coreZLoop, src/coreLoop.cpp:60-96 (R/RcppExports.R:20-22) */
int locus_hip_core_z_loop(const double* X, const double* Y, double* gam_vb,
                          const double* log_om_vb, const double* log_1_min_om_vb,
                          double log_sig2_inv_vb, const double* log_tau_vb, double* beta_vb,
                          double* mat_x_m1, const double* mat_z_mu, double* mu_beta_vb,
                          const double* sig2_beta_vb, const double* tau_vb,
                          const double* shuffled_ind, double c,
                          int64_t n, int64_t p, int64_t d);

/* replaces coreLogitLoop, src/coreLoop.cpp:101-133 (R/RcppExports.R:24-26);
 * sig2_beta_vb is p x d, Y already has 1/2 subtracted */
int locus_hip_core_logit_loop(const double* X, const double* Y, double* gam_vb,
                              const double* log_om_vb, const double* log_1_min_om_vb,
                              double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                              const double* mat_z_mu, double* mu_beta_vb, const double* psi_vb,
                              const double* sig2_beta_vb, const double* shuffled_ind,
                              int64_t n, int64_t p, int64_t d);

/* replaces coreProbitLoop, src/coreLoop.cpp:139-171 (R/RcppExports.R:28-30);
 * scalar sig2_beta_vb */
int locus_hip_core_probit_loop(const double* X, const double* W, double* gam_vb,
                               const double* log_om_vb, const double* log_1_min_om_vb,
                               double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                               const double* mat_z_mu, double* mu_beta_vb, double sig2_beta_vb,
                               const double* shuffled_ind, int64_t n, int64_t p, int64_t d);

/* replaces coreStructLoop, src/coreStructLoop.cpp:18-52 (R/RcppExports.R:32-34) */
int locus_hip_core_struct_loop(const double* X, const double* Y, double* gam_vb,
                               const double* log_Phi_mu_theta_vb,
                               const double* log_1_min_Phi_mu_theta_vb, double log_sig2_inv_vb,
                               const double* log_tau_vb, double* beta_vb, double* mat_x_m1,
                               double* mu_beta_vb, const double* sig2_beta_vb,
                               const double* tau_vb, const double* shuffled_ind,
                               int64_t n, int64_t p, int64_t d);

/* replaces coreBatch, src/coreBatch.cpp:19-42 (R/RcppExports.R:4-6) */
int locus_hip_core_batch(const double* X, const double* Y, double* gam_vb,
                         const double* log_om_vb, const double* log_1_min_om_vb,
                         double log_sig2_inv_vb, const double* log_tau_vb, double* beta_vb,
                         double* mat_x_m1, double* mu_beta_vb, const double* sig2_beta_vb,
                         const double* tau_vb, int64_t n, int64_t p, int64_t d);

/* replaces coreZBatch, src/coreBatch.cpp:48-73 (R/RcppExports.R:8-10) */
int locus_hip_core_z_batch(const double* X, const double* Y, double* gam_vb,
                           const double* log_om_vb, const double* log_1_min_om_vb,
                           double log_sig2_inv_vb, const double* log_tau_vb, double* beta_vb,
                           double* mat_x_m1, const double* mat_z_mu, double* mu_beta_vb,
                           const double* sig2_beta_vb, const double* tau_vb,
                           int64_t n, int64_t p, int64_t d);

/* replaces coreProbitBatch, src/coreBatch.cpp:78-100 (R/RcppExports.R:12-14) */
int locus_hip_core_probit_batch(const double* X, const double* W, double* gam_vb,
                                const double* log_om_vb, const double* log_1_min_om_vb,
                                double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                                const double* mat_z_mu, double* mu_beta_vb, double sig2_beta_vb,
                                int64_t n, int64_t p, int64_t d);

/* ------------------------------------------------------------------------- */
/* 2. Device-resident API (device pointers, async on `stream`)                */
/* ------------------------------------------------------------------------- */

/* Leading dimension (doubles) of the packed X for n rows: n rounded up to 128. */
int64_t locus_hip_packed_ld(int64_t n);

/* X (n x p column-major) -> Xp [p][ld] zero-padded columns and xtx[j] = X[,j]'X[,j].
 * Done once per data set (the reference reads X in place, src/coreLoop.cpp:40). */
int locus_hip_pack_x_dev(const double* X, int64_t n, int64_t p, double* Xp, double* xtx,
                         void* stream);

/* Bytes of scratch a sweep over p covariates of n rows needs (step records, permutation
 * check, and the per-sweep block-packed copies of X the matrix-core kernel reads). */
int64_t locus_hip_sweep_workspace_bytes(int64_t n, int64_t p);

typedef struct locus_hip_sweep_args {
    int64_t n, p, d;            /* d = responses resident on this device            */
    const double* Xp;           /* [p][ld] from locus_hip_pack_x_dev                 */
    const double* xtx;          /* [p]                                              */
    const double* shuffled_ind; /* [p]  0-based indices as doubles                   */
    const double* lo;           /* [p]  log_om_vb        | log_Phi_mu_theta_vb       */
    const double* l1;           /* [p]  log_1_min_om_vb  | log_1_min_Phi_mu_theta_vb */
    /* identity family (coreLoop / coreZLoop / coreStructLoop / coreProbitLoop):      */
    const double* scale;        /* [d]  (c * sig2_beta_k) * tau_k   (probit: sig2_beta) */
    const double* hinv;         /* [d]  1 / (2 sig2_beta_k)                          */
    const double* cst;          /* [d]  -(log_tau_k + log_sig2_inv + log sig2_beta_k)/2 */
    const double* s2k;          /* [d]  sig2_beta_k                                  */
    double cfac;                /* annealing factor c (1 if none)                    */
    /* logit (coreLogitLoop):                                                         */
    const double* s2;           /* [p][d] sig2_beta_vb, response-contiguous          */
    const double* xwx;          /* [p][d] sum_i X_ij^2 * 2 psi_ik                    */
    const double* W2;           /* n x d column-major, 2 * psi_vb                    */
    double half_lsi;            /* log_sig2_inv_vb / 2                               */
    /* in / out                                                                       */
    double* R;                  /* n x d column-major residual                       */
    double* state;              /* [3][p][d]: gam_vb, mu_beta_vb, beta_vb planes     */
    double* stats;              /* [4][d] or NULL: sum_j gam, sum_j m2_beta, sum_j beta^2,
                                   sum_i R^2 after the sweep                         */
    void* workspace;            /* >= locus_hip_sweep_workspace_bytes(n, p)          */
    int64_t workspace_bytes;
    int64_t nsteps;             /* covariates to visit (0 or p = full sweep)          */
    int32_t cfg[4];             /* NPL, C, S, G kernel geometry; all 0 = automatic;
                                   NPL = -1: matrix-core kernel (identity family) with C = tiles
                                   of 16 responses per workgroup (1, 2; 0 = planned from d);
                                   C = 1: S = workgroups sharing the rows of a tile (1, 2, 3;
                                   0 = 1 for n <= 384, else 3 (n > 768) or 2 when that many
                                   workgroups per tile fit on the CUs); G = 16-row tiles
                                   per matrix wave (0 = smallest that fits n);
                                   NPL = -2: streaming kernel (any n; what "automatic" picks
                                   above 4096 rows), C, S, G ignored;
                                   NPL = -3: logit look-ahead kernel, C = responses per
                                   workgroup (1, 2, 4; 0 = from d), G = rows per lane of its
                                   six row waves (1..6, 8; 0 = smallest that covers n <= 3072) */
    const double* ls2;          /* logit: [p][d] log(sig2_beta_vb) / 2 (locus_hip_half_log_dev),
                                   the last term of src/coreLoop.cpp:125 tabulated outside the
                                   latency-bound sweep; NULL = the kernel takes the logarithm */
    int32_t prepacked;          /* 1: locus_hip_sweep_prepack_dev has run on this workspace with the
                                   same n, p, d, shuffled_ind, nsteps and cfg (see there); 0: the
                                   sweep packs its operands itself */
    int32_t pad_;
} locus_hip_sweep_args;

/* One CAVI sweep over the covariates in `shuffled_ind` order; logit = 0 runs the
 * identity-family update, logit = 1 the coreLogitLoop update.  A non-permutation
 * makes the sweep a no-op and is reported by locus_hip_sweep_status_dev(). */
int locus_hip_sweep_dev(const locus_hip_sweep_args* args, int logit, void* stream);

/* The part of the next sweep that depends only on X and on the order of the covariates, done
 * ahead of it: the permutation check and the block-packed operand copies of X for every launch
 * geometry the sweep over d responses will use (identity family, matrix-core sweep: n <= 2016;
 * otherwise LOCUS_HIP_EUNSUPPORTED and nothing is done -- sweep as usual).  Reads args->n, p, d,
 * Xp, xtx, shuffled_ind, nsteps, cfg, workspace (>= locus_hip_sweep_prepack_workspace_bytes(n, p,
 * d): the geometries lie side by side); lo, l1, R, state and the per-response constants are not
 * touched, so this can run on another stream while the current sweep -- on ANOTHER workspace --
 * and the iteration's reductions are still in flight (the drivers do: ~3 % of a C3 iteration,
 * ~7 % at C2).  The sweep proper is then locus_hip_sweep_dev with args->prepacked = 1 on the same
 * workspace, ordered after this call.  NO COUNTERPART IN THE REFERENCE (src/coreLoop.cpp:36-52 reads
 * X in place). */
int locus_hip_sweep_prepack_dev(const locus_hip_sweep_args* args, void* stream);
/* Bytes of workspace for that; 0 when (n, d) is not swept by the matrix-core kernels. */
int64_t locus_hip_sweep_prepack_workspace_bytes(int64_t n, int64_t p, int64_t d);

/* Synchronises `stream` and returns 0, LOCUS_HIP_EPERM if the last sweep that used
 * `workspace` (sized for p covariates) was given an invalid shuffled_ind (that sweep was a
 * no-op), or LOCUS_HIP_ETIMEOUT if one of its row-split clusters timed out (state as described
 * at LOCUS_HIP_ETIMEOUT).  workspace == NULL: the static workspace of the current device (the
 * last locus_hip_sweep_static_dev there). */
int locus_hip_sweep_status_dev(const void* workspace, int64_t p, void* stream);

/* locus_hip_sweep_dev for callers that cannot allocate and have no workspace to lend: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE
 * REFERENCE (which has no device); same arguments and results as locus_hip_sweep_dev, but
 * args->workspace / workspace_bytes are ignored: step records, the permutation check and the
 * block-packed operand copies go to a fixed array of LOCUS_HIP_STATIC_WORKSPACE_BYTES that is
 * part of the library's code object (each GPU the library runs on holds it for the life of the
 * process).  Calls no hipMalloc / hipFree or allocating routine.  Limits: returns
 * LOCUS_HIP_EUNSUPPORTED when locus_hip_sweep_workspace_bytes(n, p) exceeds the array.  For
 * n <= 2016 that size covers the block-packed X operands and exchange buffers of the matrix-core
 * sweep whatever the family: p <= 4768 at n = 200, 1648 at n = 1000, 880 at n = 2000; above 2016
 * rows only ~28 p bytes are needed (p up to ~2.3 million).  All calls on one GPU share that GPU's array:
 * two such sweeps that overlap on one GPU (different streams or threads) race -- serialise them.
 * Set-up: the first use on a device looks the array's address up (hipGetSymbolAddress; the HIP
 * runtime may materialise the array at that moment).  A caller that must not trigger even that
 * inside its hot region calls locus_hip_sweep_status_dev(NULL, 1, stream) once per device first. */
#define LOCUS_HIP_STATIC_WORKSPACE_BYTES (64ll << 20)
int locus_hip_sweep_static_dev(const locus_hip_sweep_args* args, int logit, void* stream);

/* The launches the matrix-core sweep of the identity family makes for d responses of n rows on the
 * current device: up to max_rows rows of 8 ints {T = tiles of 16
 * responses per cluster, CS = workgroups per cluster (0: the 32-responses-per-workgroup kernel),
 * KT, ODD = row tiling, k_begin, k_end = responses [k_begin, k_end), region = which packed operand
 * set, capacity = responses one such launch can take}.  Returns the number of launches (0: this
 * shape is not swept by those kernels) or a negative error.  Introspection only. */
int locus_hip_sweep_plan(int64_t n, int64_t d, int32_t* rows, int max_rows);

/* Geometry the library would pick for (n, d): fills cfg[4] = NPL, C, S, G
 * ({-1, 0, 0, 0}: the matrix-core sweep, launches planned from d; {-3, 0, 0, 0}: the logit
 * look-ahead kernel, responses per workgroup and rows per lane set at launch; {-2, 0, 0, 0}: the
 * streaming kernel, n > 4096). */
int locus_hip_sweep_config(int64_t n, int64_t d, int logit, int32_t* cfg);

/* out = Y - A - Zmu                       (W2 == NULL; same formula both directions)
 * out = Y - W2 * (A + Zmu)                (W2 != NULL, to_residual = 1)
 * out = (Y - A) / W2 - Zmu                (W2 != NULL, to_residual = 0)
 * all n x d column-major; Zmu may be NULL. */
int locus_hip_residual_dev(const double* Y, const double* A, const double* Zmu,
                           const double* W2, double* out, int64_t n, int64_t d,
                           int to_residual, void* stream);

/* out[cols][rows] = transpose(in[rows][cols]) (row-major).  An R p x d matrix is a
 * row-major [d][p]; transpose(rows = d, cols = p) gives the [p][d] state plane. */
int locus_hip_transpose_dev(const double* in, double* out, int64_t rows, int64_t cols,
                            void* stream);

/* rs_gam[j] = sum_k gam[j][k]  (rowSums(gam_vb), R/locus_core.R:95) and, if ent != NULL,
 * ent[j] = sum_k gam log(gam + eps) + (1 - gam) log(1 - gam + eps)  (R/elbo.R:52). */
int locus_hip_gamma_rows_dev(const double* gam, double* rs_gam, double* ent, int64_t p,
                             int64_t d, double eps, void* stream);

/* Column sums over covariates of the state without a sweep (initialisation,
 * R/locus_core.R:36-42): writes beta = gam * mu into plane 2 and
 * st_partial[chunk][3][d] = partial sums of gam, gam (mu^2 + s2), beta^2.
 * s2 comes from s2k[d] (identity family) or s2jk[p][d] (logit); one may be NULL. */
int locus_hip_state_cols_dev(double* state, const double* s2k, const double* s2jk,
                             double* st_partial, int64_t p, int64_t d, int64_t nchunks,
                             void* stream);

/* C[m x n] = alpha * op(A) op(B) + beta * C on the fp64 matrix cores; element (i, k) of
 * op(A) is A[i * sa_m + k * sa_k] (squared first if sqA), likewise B and C. */
int locus_hip_gemm_dev(int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                       int64_t sa_m, int64_t sa_k, int sqA, const double* B, int64_t sb_k,
                       int64_t sb_n, int sqB, double beta, double* C, int64_t sc_m,
                       int64_t sc_n, void* stream);

/* Element-wise part of coreBatch / coreZBatch / coreProbitBatch on [p][d] planes
 * (src/coreBatch.cpp:32-39):  T = X'(Y - M [- Zmu]) on entry in `mu` plane;
 *   mu = (T + (n-1) beta) * fac_k ; gam = exp(-log1pexp(...)) ; beta = mu * gam.
 * fac, hinv, cstk are [d]: fac = sig2_beta_k tau_k (probit: sig2_beta),
 * hinv = 1/(2 sig2_beta_k), cstk = log_tau_k/2 + log(sig2_beta_k)/2 (probit: log(sig2_beta)/2
 * with probit = 1, which also changes the order of the additions, :94-95).  cfac = 1 for
 * coreBatch; the driver's internal scheme "x" (R/locus_core.R:104-111) anneals with
 * cfac = c multiplying the argument of log_one_plus_exp_ (fac then carries c as well). */
int locus_hip_batch_update_dev(double* state, const double* lo, const double* l1,
                               const double* fac, const double* hinv, const double* cstk,
                               double log_sig2_inv, double nm1, double cfac, int probit,
                               int64_t p, int64_t d, void* stream);

/* ---- binary-response drivers (logit / probit): device side of R/locus_logistic_core.R and
 * R/locus_probit_core.R around the sweep.  n x d matrices column-major, [p][d] / [q][d]
 * planes response-contiguous, Z n x q column-major. ---- */

/* D = m2_beta - beta_vb^2 = gam (mu^2 + s2) - (gam mu)^2 on [p][d] planes; s2 from s2jk[p][d]
 * (logit) or the scalar s2c when s2jk == NULL (probit).  Right-hand factor of the X^2
 * contractions of update_chi_vb_ (R/update_vb.R:163-167), e_y_logit_ (R/elbo.R:210-218),
 * e_y_probit_ (R/elbo.R:221-237). */
int locus_hip_m2_minus_b2_dev(const double* state, const double* s2jk, double s2c, double* D,
                              int64_t p, int64_t d, void* stream);

/* s2 = 1 / (xwx + sig2_inv): update_sig2_beta_logit_vb_ (R/update_vb.R:107-111) from
 * xwx = crossprod(X^2, 2 psi_vb). */
int locus_hip_s2_from_xwx_dev(const double* xwx, double sig2_inv, double* s2, int64_t len,
                              void* stream);

/* out = log(in) / 2, element-wise (locus_hip_sweep_args.ls2 from sig2_beta_vb). */
int locus_hip_half_log_dev(const double* in, double* out, int64_t len, void* stream);

/* update_chi_vb_ + update_psi_logit_vb_ (R/update_vb.R:163-174) fused with e_y_logit_
 * (R/elbo.R:210-218): c2 = Q + M^2 + Z^2 s2a + Zmu^2 + 2 M Zmu with Q = X^2 D; chi_new =
 * sqrt(c2), W2_new = 2 psi(chi_new); ey[k] = column sums of e_y_logit_ evaluated with
 * chi_old / W2_old (skipped when ey == NULL).  s2a is [q][d]. */
int locus_hip_logit_chi_psi_dev(const double* Q, const double* M, const double* Zmu,
                                const double* Y, const double* Z, const double* s2a,
                                const double* chi_old, const double* W2_old, double* chi_new,
                                double* W2_new, double* ey, int64_t n, int64_t d, int64_t q,
                                void* stream);

/* update_sig2_alpha_logit_vb_ (R/update_vb.R:52-56): s2a[i][k] = 1 / (sum_r Z_ri^2 W2_rk +
 * zeta2_inv[i]); zeta2_inv is a device vector [q]. */
int locus_hip_logit_sig2_alpha_dev(const double* Z, const double* W2, const double* zeta2_inv,
                                   double* s2a, int64_t n, int64_t d, int64_t q, void* stream);

/* One covariate i of the alpha loop (R/locus_logistic_core.R:79-85 with W2 = 2 psi_vb and
 * Yw = Y; R/locus_probit_core.R:76-84 with W2 == NULL and Yw = W): updates alpha_i[d] and
 * Zmu in place.  Zi = Z[, i] (n), s2a_i = sig2_alpha_vb[i, ] (d). */
int locus_hip_alpha_step_dev(const double* Yw, const double* W2, const double* M, double* Zmu,
                             const double* Zi, const double* s2a_i, double* alpha_i, int64_t n,
                             int64_t d, void* stream);

/* update_W_probit_ (R/update_vb.R:382-386, inv_mills_ratio_matrix_ R/utils.R:178-198) fused
 * with the column sums of e_y_probit_ (R/elbo.R:221-237); Q = X^2 D, zs = Z^2 sig2_alpha (n).
 * Q == NULL: W only. */
int locus_hip_probit_w_dev(const double* Y, const double* M, const double* Zmu, const double* Q,
                           const double* zs, double* W, double* ey, int64_t n, int64_t d,
                           void* stream);

/* Iteration tail of locus_core_ on the device, on REDUCED statistics: the ELBO of the iteration
 * that ended (elbo_, R/locus_core.R:261-292; e_y_, e_beta_gamma_, e_tau_, e_sig2_inv_, e_omega_ of
 * R/elbo.R:43-54,114-143,189-200) and the updates at the top of the next one
 * (R/locus_core.R:57-84; R/update_vb.R:181-237,275).  Two phases around the caller's all-reduce of
 * [rowSums(gam_vb) | shard2] over the response shards:
 *   phase 0 (after the sweep and gamma_rows): shard2[0] = sum_k tau_k colSums(m2_beta)_k,
 *           shard2[1] = shard-local ELBO terms (0 unless with_elbo);
 *   phase 1 (after the all-reduce): scal[0] = ELBO (with_elbo), scal[1] = sum(gam_vb),
 *           scal[2] = sum_k tau_k colSums(m2_beta)_k, scal[4] / scal[5] = sig2_inv_vb /
 *           log_sig2_inv_vb of the next iteration at annealing factor c_next; lo / l1 =
 *           log_om_vb / log_1_min_om_vb, and tau_next, s2_next, scale, hinv, cst = tau_vb,
 *           sig2_beta_vb and the per-response constants locus_hip_sweep_dev takes (NULL: skipped).
 *   phase 2 = phase 1 for models whose covariate-wise terms are not those of the Beta prior on
 *           omega (structured prior, R/locus_struct_core.R:201-221): the caller fills part[0..2] =
 *           {sum(gam_vb), covariate terms of e_beta_gamma_, the other replicated ELBO terms} and
 *           zeroes part[3..191]; a, b, rs_gam, lo, l1 are not used.
 * *si_in is sig2_inv_vb of the iteration that ended (read by both phases; it may be scal[4] of
 * the previous call's scal buffer). */
typedef struct locus_hip_tail_args {
    int64_t n, p, d_loc, d_total;
    const double* a;            /* [p] hyper-parameters of omega                       */
    const double* b;
    const double* eta;          /* [d_loc] hyper-parameters of tau                     */
    const double* kappa;
    double lam, nu;             /* hyper-parameters of sig2_inv                        */
    const double* stats;        /* [4][d_loc] column statistics of the sweep epilogue  */
    const double* tau;          /* [d_loc] tau_vb the sweep used                       */
    const double* s2;           /* [d_loc] sig2_beta_vb the sweep used                 */
    const double* rs_gam;       /* [p] rowSums(gam_vb) over ALL shards (phase 1)       */
    const double* ent;          /* [p] entropy partials of gamma_rows (phase 0) / NULL */
    const double* si_in;        /* sig2_inv_vb the sweep used                          */
    double* shard2;             /* [2] see above                                       */
    double* scal;               /* [8] see above                                       */
    int32_t with_elbo;
    int32_t pad_;
    double c_next;
    double* lo;                 /* [p] or NULL                                         */
    double* l1;
    double* tau_next;           /* [d_loc] or NULL (then s2_next, scale, hinv, cst too) */
    double* s2_next;
    double* scale;
    double* hinv;
    double* cst;
    double* part;               /* [192] scratch of phase 1                            */
} locus_hip_tail_args;
int locus_hip_identity_tail_dev(const locus_hip_tail_args* args, int phase, void* stream);

/* Per-covariate sums over the local responses: rs_gam, entropy term (as gamma_rows),
 * gls[j] = sum_k gam log(s2), m2[j] = sum_k gam (mu^2 + s2): the O(p d) parts of
 * e_beta_gamma_bin_ (R/elbo.R:77-87) and update_nu_bin_vb_ (R/update_vb.R:210). */
int locus_hip_bin_rows_dev(const double* state, const double* s2jk, double s2c, double* rs_gam,
                           double* ent, double* gls, double* m2, int64_t p, int64_t d, double eps,
                           void* stream);

/* ------------------------------------------------------------------------- */
/* 3. Resident context (host pointers; state kept on the GPU between calls)    */
/* ------------------------------------------------------------------------- */
/* For a host that cannot hold device memory itself -- the R session of the reference.  The loop
 * body of locus_core_ / locus_z_core_ / locus_struct_core_ (R/locus_core.R:57-211) becomes
 *     ctx_create(X, Y)                               once   (X, Y as passed to coreLoop)
 *     ctx_set_state(gam_vb, mu_beta_vb, mat_z_mu)    once   (R/locus_core.R:36-42)
 *     per iteration: the R updates of the length-p / length-d quantities, then
 *       [ctx_set_response(W, mat_z_mu)  coreZLoop / coreProbitLoop / mix drivers only: the new
 *                         alpha_vb's Z alpha_vb and / or the new latent W, R/locus_z_core.R:100-107,
 *                         R/locus_probit_core.R:76-88]
 *       ctx_sweep(...)  = coreLoop (R/locus_core.R:90-95) + rowSums(gam_vb) + the column sums
 *                         update_eta/kappa/lambda/nu_vb_ and elbo_ need (R/update_vb.R:198-237)
 *     ctx_get_state(...)                              once, at the end
 * so that X, Y and the three p x d matrices stay in HBM and 3 p + 4 d doubles go up, p + 4 d + 1
 * come back per iteration.  All calls are synchronous; a context belongs to the device that was
 * current at creation and is not thread-safe. */
typedef struct locus_hip_ctx locus_hip_ctx;

/* X n x p, Y n x d column-major, as given to coreLoop (src/coreLoop.cpp:19-20). */
int locus_hip_ctx_create(const double* X, const double* Y, int64_t n, int64_t p, int64_t d,
                         locus_hip_ctx** ctx);

/* (Re)starts the fit from gam_vb, mu_beta_vb (p x d column-major): uploads them over the
 * resident planes, forms beta_vb = gam_vb * mu_beta_vb and mat_x_m1 = X beta_vb
 * (R/locus_core.R:36-40).  mat_z_mu (n x d) or NULL: the term coreZLoop / coreProbitLoop subtract
 * (src/coreLoop.cpp:80).  Call it ONCE, before the loop; it is NOT the way to tell the context
 * about a new alpha_vb or W between iterations (it would reset the fit to the host's stale
 * gam_vb / mu_beta_vb): that is locus_hip_ctx_set_response. */
int locus_hip_ctx_set_state(locus_hip_ctx* ctx, const double* gam_vb, const double* mu_beta_vb,
                            const double* mat_z_mu);

/* Between iterations of locus_z_core_ / locus_probit_core_ / locus_mix_core_: replaces the
 * response matrix the sweep sees (Y_or_W: n x d, the probit / mix latent W of
 * R/update_vb.R:382-386; NULL = unchanged) and / or mat_z_mu = Z alpha_vb (n x d,
 * R/locus_z_core.R:100-107; NULL = unchanged) and rebuilds the residual on the device from the
 * resident mat_x_m1.  gam_vb, mu_beta_vb, beta_vb and mat_x_m1 keep their resident values.
 * No counterpart in the reference, where coreZLoop reads Y and mat_z_mu in place each call
 * (src/coreLoop.cpp:60-62,84). */
int locus_hip_ctx_set_response(locus_hip_ctx* ctx, const double* Y_or_W, const double* mat_z_mu);

/* One sweep with the arguments of coreLoop (probit: tau_vb = 1, log_tau_vb = 0, sig2_beta_vb
 * repeated; struct: log Phi / log(1 - Phi) in place of log_om / log_1_min_om, c = 1).  Outputs,
 * each may be NULL: rs_gam[p] = rowSums(gam_vb); col_stats[4][d] = colSums(gam_vb),
 * colSums(m2_beta), colSums(beta_vb^2), colSums((Y - mat_x_m1 - mat_z_mu)^2); *ent_sum =
 * sum gam log(gam + eps) + (1 - gam) log(1 - gam + eps) (R/elbo.R:46-52). */
int locus_hip_ctx_sweep(locus_hip_ctx* ctx, const double* log_om_vb,
                        const double* log_1_min_om_vb, double log_sig2_inv_vb,
                        const double* log_tau_vb, const double* sig2_beta_vb,
                        const double* tau_vb, const double* shuffled_ind, double c, double ent_eps,
                        double* rs_gam, double* col_stats, double* ent_sum);

/* The same three outputs for the CURRENT state (before the first sweep: the sums
 * R/locus_core.R:36-42 forms from the initial values; sig2_beta_vb[d] enters m2_beta). */
int locus_hip_ctx_stats(locus_hip_ctx* ctx, const double* sig2_beta_vb, double ent_eps,
                        double* rs_gam, double* col_stats, double* ent_sum);

/* The four matrices coreLoop mutates, in R's layout; each pointer may be NULL. */
int locus_hip_ctx_get_state(locus_hip_ctx* ctx, double* gam_vb, double* mu_beta_vb,
                            double* beta_vb, double* mat_x_m1);

int locus_hip_ctx_destroy(locus_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LOCUS_HIP_H */
