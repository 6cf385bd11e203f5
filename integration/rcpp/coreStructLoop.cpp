// Replacement for the reference's src/coreStructLoop.cpp (signature :18-30).
#include "locus_hip_glue.h"

// [[Rcpp::export]]
void coreStructLoop(const MapMat X, const MapMat Y, MapArr2D gam_vb,
                    const MapArr1D log_Phi_mu_theta_vb, const MapArr1D log_1_min_Phi_mu_theta_vb,
                    const double log_sig2_inv_vb, const MapArr1D log_tau_vb, MapMat beta_vb,
                    MapMat mat_x_m1, MapArr2D mu_beta_vb, const MapArr1D sig2_beta_vb,
                    const MapArr1D tau_vb, const MapArr1D shuffled_ind) {
  locus_hip_check(locus_hip_core_struct_loop(
      X.data(), Y.data(), gam_vb.data(), log_Phi_mu_theta_vb.data(),
      log_1_min_Phi_mu_theta_vb.data(), log_sig2_inv_vb, log_tau_vb.data(), beta_vb.data(),
      mat_x_m1.data(), mu_beta_vb.data(), sig2_beta_vb.data(), tau_vb.data(),
      shuffled_ind.data(), X.rows(), X.cols(), Y.cols()));
}
