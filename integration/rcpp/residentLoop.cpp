// New file for the reference's src/: the resident context (include/locus_hip.h, section 3) as
// six exported functions.  X, Y and the p x d state stay in HBM behind an external pointer; the
// loop body of locus_core_ (R/locus_core.R:57-211) keeps its length-p / length-d updates and
// replaces coreLoop(...) (:90-92) by ctxSweep(ctx, ...); see INTEGRATION.md section 2.
#include <cfloat>
#include "locus_hip_glue.h"

static locus_hip_ctx* ctx_of(SEXP x) {
  locus_hip_ctx* c = (locus_hip_ctx*)R_ExternalPtrAddr(x);
  if (c == nullptr) Rcpp::stop("locus_hip: context already released");
  return c;
}
static void ctx_finalizer(SEXP x) {
  locus_hip_ctx_destroy((locus_hip_ctx*)R_ExternalPtrAddr(x));
  R_ClearExternalPtr(x);
}

// [[Rcpp::export]]
SEXP ctxCreate(const MapMat X, const MapMat Y) {
  locus_hip_ctx* c = nullptr;
  locus_hip_check(locus_hip_ctx_create(X.data(), Y.data(), X.rows(), X.cols(), Y.cols(), &c));
  SEXP x = PROTECT(R_MakeExternalPtr(c, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(x, ctx_finalizer, TRUE);
  UNPROTECT(1);
  return x;
}

// [[Rcpp::export]]
void ctxSetState(SEXP ctx, const MapArr2D gam_vb, const MapArr2D mu_beta_vb,
                 Rcpp::Nullable<Rcpp::NumericMatrix> mat_z_mu = R_NilValue) {
  const double* z = mat_z_mu.isNotNull() ? REAL(mat_z_mu.get()) : nullptr;
  locus_hip_check(locus_hip_ctx_set_state(ctx_of(ctx), gam_vb.data(), mu_beta_vb.data(), z));
}

// new W (probit / mix) and / or new Z alpha_vb (coreZLoop drivers) between iterations
// [[Rcpp::export]]
void ctxSetResponse(SEXP ctx, Rcpp::Nullable<Rcpp::NumericMatrix> Y_or_W = R_NilValue,
                    Rcpp::Nullable<Rcpp::NumericMatrix> mat_z_mu = R_NilValue) {
  const double* y = Y_or_W.isNotNull() ? REAL(Y_or_W.get()) : nullptr;
  const double* z = mat_z_mu.isNotNull() ? REAL(mat_z_mu.get()) : nullptr;
  locus_hip_check(locus_hip_ctx_set_response(ctx_of(ctx), y, z));
}

static Rcpp::List ctx_result(const Rcpp::NumericVector& rs_gam,
                             const Rcpp::NumericMatrix& col_stats, double ent) {
  return Rcpp::List::create(Rcpp::_["rs_gam"] = rs_gam, Rcpp::_["col_stats"] = col_stats,
                            Rcpp::_["ent"] = ent);
}

// coreLoop's arguments (R/RcppExports.R:16-18) minus the matrices the context holds
// [[Rcpp::export]]
Rcpp::List ctxSweep(SEXP ctx, const MapArr1D log_om_vb, const MapArr1D log_1_min_om_vb,
                    const double log_sig2_inv_vb, const MapArr1D log_tau_vb,
                    const MapArr1D sig2_beta_vb, const MapArr1D tau_vb,
                    const MapArr1D shuffled_ind, const double c = 1) {
  const int p = log_om_vb.size(), d = tau_vb.size();
  Rcpp::NumericVector rs_gam(p);
  Rcpp::NumericMatrix col_stats(d, 4);  // columns: colSums(gam), colSums(m2_beta), colSums(beta^2), RSS
  double ent = 0;
  locus_hip_check(locus_hip_ctx_sweep(
      ctx_of(ctx), log_om_vb.data(), log_1_min_om_vb.data(), log_sig2_inv_vb, log_tau_vb.data(),
      sig2_beta_vb.data(), tau_vb.data(), shuffled_ind.data(), c, DBL_EPSILON, rs_gam.begin(),
      col_stats.begin(), &ent));
  return ctx_result(rs_gam, col_stats, ent);
}

// [[Rcpp::export]]
Rcpp::List ctxStats(SEXP ctx, const MapArr1D sig2_beta_vb, const int p) {
  const int d = sig2_beta_vb.size();
  Rcpp::NumericVector rs_gam(p);
  Rcpp::NumericMatrix col_stats(d, 4);
  double ent = 0;
  locus_hip_check(locus_hip_ctx_stats(ctx_of(ctx), sig2_beta_vb.data(), DBL_EPSILON,
                                      rs_gam.begin(), col_stats.begin(), &ent));
  return ctx_result(rs_gam, col_stats, ent);
}

// fills the caller's matrices in place, like the reference's kernels (src/coreLoop.cpp:21,26-28)
// [[Rcpp::export]]
void ctxGetState(SEXP ctx, MapArr2D gam_vb, MapArr2D mu_beta_vb, MapMat beta_vb, MapMat mat_x_m1) {
  locus_hip_check(locus_hip_ctx_get_state(ctx_of(ctx), gam_vb.data(), mu_beta_vb.data(),
                                          beta_vb.data(), mat_x_m1.data()));
}
