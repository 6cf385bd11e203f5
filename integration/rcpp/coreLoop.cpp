// Replacement for the reference's src/coreLoop.cpp: same four exported functions, same
// signatures (src/coreLoop.cpp:19-32, 60-74, 101-113, 139-150), bodies forwarded to liblocus_hip.
// src/RcppExports.cpp and R/RcppExports.R are regenerated unchanged by Rcpp::compileAttributes().
// n = nrow(X), p = ncol(X), d = ncol(Y) are made explicit for the C ABI.
#include "locus_hip_glue.h"

// [[Rcpp::export]]
void coreLoop(const MapMat X, const MapMat Y, MapArr2D gam_vb, const MapArr1D log_om_vb,
              const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb,
              const MapArr1D log_tau_vb, MapMat beta_vb, MapMat mat_x_m1, MapArr2D mu_beta_vb,
              const MapArr1D sig2_beta_vb, const MapArr1D tau_vb, const MapArr1D shuffled_ind,
              const double c = 1) {
  locus_hip_check(locus_hip_core_loop(
      X.data(), Y.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, log_tau_vb.data(), beta_vb.data(), mat_x_m1.data(), mu_beta_vb.data(),
      sig2_beta_vb.data(), tau_vb.data(), shuffled_ind.data(), c, X.rows(), X.cols(), Y.cols()));
}

// [[Rcpp::export]]
void coreZLoop(const MapMat X, const MapMat Y, MapArr2D gam_vb, const MapArr1D log_om_vb,
               const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb,
               const MapArr1D log_tau_vb, MapMat beta_vb, MapMat mat_x_m1, MapMat mat_z_mu,
               MapArr2D mu_beta_vb, const MapArr1D sig2_beta_vb, const MapArr1D tau_vb,
               const MapArr1D shuffled_ind, const double c = 1) {
  locus_hip_check(locus_hip_core_z_loop(
      X.data(), Y.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, log_tau_vb.data(), beta_vb.data(), mat_x_m1.data(), mat_z_mu.data(),
      mu_beta_vb.data(), sig2_beta_vb.data(), tau_vb.data(), shuffled_ind.data(), c, X.rows(),
      X.cols(), Y.cols()));
}

// [[Rcpp::export]]
void coreLogitLoop(const MapMat X, const MapArr2D Y, MapArr2D gam_vb, const MapArr1D log_om_vb,
                   const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb, MapMat beta_vb,
                   MapArr2D mat_x_m1, MapArr2D mat_z_mu, MapArr2D mu_beta_vb,
                   const MapArr2D psi_vb, const MapArr2D sig2_beta_vb,
                   const MapArr1D shuffled_ind) {
  locus_hip_check(locus_hip_core_logit_loop(
      X.data(), Y.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, beta_vb.data(), mat_x_m1.data(), mat_z_mu.data(), mu_beta_vb.data(),
      psi_vb.data(), sig2_beta_vb.data(), shuffled_ind.data(), X.rows(), X.cols(), Y.cols()));
}

// [[Rcpp::export]]
void coreProbitLoop(const MapMat X, const MapMat W, MapArr2D gam_vb, const MapArr1D log_om_vb,
                    const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb, MapMat beta_vb,
                    MapMat mat_x_m1, MapMat mat_z_mu, MapArr2D mu_beta_vb,
                    const double sig2_beta_vb, const MapArr1D shuffled_ind) {
  locus_hip_check(locus_hip_core_probit_loop(
      X.data(), W.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, beta_vb.data(), mat_x_m1.data(), mat_z_mu.data(), mu_beta_vb.data(),
      sig2_beta_vb, shuffled_ind.data(), X.rows(), X.cols(), W.cols()));
}
