// Replacement for the reference's src/coreBatch.cpp (signatures :19-30, 48-60, 78-88).
#include "locus_hip_glue.h"

// [[Rcpp::export]]
void coreBatch(const MapMat X, const MapMat Y, MapArr2D gam_vb, const MapArr1D log_om_vb,
               const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb,
               const MapArr1D log_tau_vb, MapMat beta_vb, MapMat mat_x_m1, MapArr2D mu_beta_vb,
               const MapArr1D sig2_beta_vb, const MapArr1D tau_vb) {
  locus_hip_check(locus_hip_core_batch(
      X.data(), Y.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, log_tau_vb.data(), beta_vb.data(), mat_x_m1.data(), mu_beta_vb.data(),
      sig2_beta_vb.data(), tau_vb.data(), X.rows(), X.cols(), Y.cols()));
}

// [[Rcpp::export]]
void coreZBatch(const MapMat X, const MapMat Y, MapArr2D gam_vb, const MapArr1D log_om_vb,
                const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb,
                const MapArr1D log_tau_vb, MapMat beta_vb, MapMat mat_x_m1, MapMat mat_z_mu,
                MapArr2D mu_beta_vb, const MapArr1D sig2_beta_vb, const MapArr1D tau_vb) {
  locus_hip_check(locus_hip_core_z_batch(
      X.data(), Y.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, log_tau_vb.data(), beta_vb.data(), mat_x_m1.data(), mat_z_mu.data(),
      mu_beta_vb.data(), sig2_beta_vb.data(), tau_vb.data(), X.rows(), X.cols(), Y.cols()));
}

// [[Rcpp::export]]
void coreProbitBatch(const MapMat X, const MapMat W, MapArr2D gam_vb, const MapArr1D log_om_vb,
                     const MapArr1D log_1_min_om_vb, const double log_sig2_inv_vb, MapMat beta_vb,
                     MapMat mat_x_m1, MapMat mat_z_mu, MapArr2D mu_beta_vb,
                     const double sig2_beta_vb) {
  locus_hip_check(locus_hip_core_probit_batch(
      X.data(), W.data(), gam_vb.data(), log_om_vb.data(), log_1_min_om_vb.data(),
      log_sig2_inv_vb, beta_vb.data(), mat_x_m1.data(), mat_z_mu.data(), mu_beta_vb.data(),
      sig2_beta_vb, X.rows(), X.cols(), W.cols()));
}
