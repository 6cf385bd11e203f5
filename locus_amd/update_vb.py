"""Closed-form variational updates of the drivers, host side (mirrors R/update_vb.R).

The reference evaluates these on the full p x d / n x d matrices; here the O(p d) and
O(n d) reductions they contain are done on the GPU (sweep epilogue, gamma_rows,
state_cols) and the functions below take the REDUCED statistics:

  cs_gam[k]  = colSums(gam_vb)[k]          cs_m2[k] = colSums(m2_beta)[k]
  cs_b2[k]   = colSums(beta_vb^2)[k]       rss[k]   = colSums((Y - mat_x_m1)^2)[k]
  rs_gam[j]  = rowSums(gam_vb)[j]          (summed over all shards)

Everything here is O(p + d) numpy on replicated / shard-local vectors.
"""
import numpy as np
from scipy.special import digamma


def update_a_vb(a, rs_gam, c=1.0):  # R/update_vb.R:181
    return c * (a + rs_gam) - c + 1


def update_b_vb(b, d, rs_gam, c=1.0):  # :184
    return c * (b - rs_gam + d) - c + 1


def update_log_om_vb(a, digam_sum, rs_gam, c=1.0):  # :187
    return digamma(c * (a + rs_gam) - c + 1) - digam_sum


def update_log_1_min_om_vb(b, d, digam_sum, rs_gam, c=1.0):  # :190
    return digamma(c * (b - rs_gam + d) - c + 1) - digam_sum


def update_lambda_vb_(lam, sum_gam, c=1.0):  # :198
    return c * (lam + sum_gam / 2) - c + 1


def update_nu_vb_(nu, sum_tau_m2, c=1.0):
    """:204 with crossprod(tau_vb, colSums(m2_beta)) already reduced (over all shards)."""
    return c * float(nu + sum_tau_m2 / 2)


def update_nu_bin_vb_(nu, sum_m2):  # :210
    return nu + sum_m2 / 2


def update_log_sig2_inv_vb_(lambda_vb, nu_vb):  # :213
    return digamma(lambda_vb) - np.log(nu_vb)


def update_eta_vb_(n, eta, cs_gam, c=1.0):  # :220
    return c * (eta + n / 2 + cs_gam / 2) - c + 1


def update_kappa_vb_(n, kappa, rss, cs_m2, cs_b2, sig2_inv_vb, c=1.0):
    """:229-237.  colSums(Y^2) - 2 colSums(Y * M) + colSums(M^2) is the residual sum of
    squares rss = colSums((Y - M)^2), which the sweep keeps on chip."""
    return c * (kappa + (rss + (n - 1 + sig2_inv_vb) * cs_m2 - (n - 1) * cs_b2) / 2)


def update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb=None, c=1.0):  # :93-104
    if tau_vb is None:
        return 1 / (c * (n - 1 + sig2_inv_vb))
    return 1 / (c * (n - 1 + sig2_inv_vb) * tau_vb)


def update_log_tau_vb_(eta_vb, kappa_vb):  # :275
    return digamma(eta_vb) - np.log(kappa_vb)


def update_m2_alpha_(alpha_vb, sig2_alpha_vb):  # :18-29 (sig2_alpha_vb of alpha_vb's shape)
    return sig2_alpha_vb + alpha_vb ** 2


def update_phi_z_vb_(phi, d, c=1.0):  # :407
    return c * (phi + d / 2) - c + 1


def update_xi_bin_vb_(xi, rs_m2_alpha):  # :413 (rowSums(m2_alpha) reduced over shards)
    return xi + rs_m2_alpha / 2


def update_sig2_alpha_vb_(n, zeta2_inv_vb, intercept=False, c=1.0):  # :32-49 (tau NULL)
    den = n - 1 + np.array(zeta2_inv_vb, dtype=np.float64)
    if intercept:
        den[0] = den[0] + 1
    return 1 / (c * den)


def get_annealing_ladder_(anneal):  # R/utils.R:108-146
    k_m = 1.0 / anneal[1]
    m = int(anneal[2])
    if anneal[0] == 1:  # geometric
        delta_k = k_m ** (1.0 / (1 - m)) - 1
        return (1 + delta_k) ** (1 - np.arange(m, 0, -1.0))
    if anneal[0] == 2:  # harmonic
        delta_k = (1 / k_m - 1) / (m - 1)
        return 1 / (1 + delta_k * (np.arange(m, 0, -1.0) - 1))
    delta_k = (1 - k_m) / (m - 1)  # linear
    return k_m + delta_k * (np.arange(1, m + 1.0) - 1)
