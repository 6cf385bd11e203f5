"""ELBO summands of the drivers on reduced statistics (mirrors R/elbo.R).

Terms that the reference sums over k (responses) are returned as SHARD-LOCAL sums so
that they can ride in the per-iteration all-reduce; terms over j (covariates) and the
scalar terms are replicated and must be added once.
"""
import numpy as np
from scipy.special import betaln, gammaln


def e_y_(n, kappa, kappa_vb, log_tau_vb, cs_m2, sig2_inv_vb, tau_vb):  # R/elbo.R:189-200
    arg = (-n / 2 * np.log(2 * np.pi) + n / 2 * log_tau_vb
           - tau_vb * (kappa_vb - cs_m2 * sig2_inv_vb / 2 - kappa))
    return arg.sum()


def e_beta_gamma_local_(cs_gam, cs_m2, ent, log_tau_vb, sig2_beta_vb, sig2_inv_vb, tau_vb):
    """Shard-local part of e_beta_gamma_ (R/elbo.R:43-54): the terms swept by response,
    and the entropy  sum gam log(gam + eps) + (1 - gam) log(1 - gam + eps)  (`ent`)."""
    return np.sum(cs_gam * log_tau_vb / 2 - cs_m2 * tau_vb * sig2_inv_vb / 2
                  + cs_gam * (np.log(sig2_beta_vb) + 1) / 2) - ent


def e_beta_gamma_global_(rs_gam, sum_gam, d, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb):
    """Replicated part of e_beta_gamma_: terms swept by covariate + the log_sig2_inv term."""
    return (log_sig2_inv_vb * sum_gam / 2 + np.sum(rs_gam * log_om_vb)
            + np.sum((d - rs_gam) * log_1_min_om_vb))


def e_beta_gamma_bin_local_(sum_gam_s2term, sum_m2, ent, sig2_inv_vb):
    """Shard-local part of e_beta_gamma_bin_ (R/elbo.R:77-87):
    sum gam (log sig2_beta + 1) / 2 - m2_beta sig2_inv / 2 - entropy."""
    return sum_gam_s2term - sum_m2 * sig2_inv_vb / 2 - ent


def e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb):  # :114-118
    return np.sum((a - a_vb) * log_om_vb + (b - b_vb) * log_1_min_om_vb
                  - betaln(a, b) + betaln(a_vb, b_vb))


def e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb):  # :125-130
    return ((lam - lambda_vb) * log_sig2_inv_vb - (nu - nu_vb) * sig2_inv_vb
            + lam * np.log(nu) - lambda_vb * np.log(nu_vb) - gammaln(lam) + gammaln(lambda_vb))


def e_tau_(eta, eta_vb, kappa, kappa_vb, log_tau_vb, tau_vb):  # :138-143
    return np.sum((eta - eta_vb) * log_tau_vb - (kappa - kappa_vb) * tau_vb
                  + eta * np.log(kappa) - eta_vb * np.log(kappa_vb)
                  - gammaln(eta) + gammaln(eta_vb))


def e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb):  # :244-249
    return np.sum((phi - phi_vb) * log_zeta2_inv_vb - (xi - xi_vb) * zeta2_inv_vb
                  + phi * np.log(xi) - phi_vb * np.log(xi_vb) - gammaln(phi) + gammaln(phi_vb))


def e_alpha_logit_(rs_m2_alpha, sum_log_sig2_alpha, q, d, log_zeta2_inv_vb, zeta2_inv_vb):
    """R/elbo.R:22-27 on reduced statistics: rs_m2_alpha = rowSums(m2_alpha) (q),
    sum_log_sig2_alpha = sum(log(sig2_alpha_vb)) over the q x d matrix."""
    return 0.5 * (-np.sum(rs_m2_alpha * zeta2_inv_vb) + d * np.sum(log_zeta2_inv_vb)
                  + sum_log_sig2_alpha + q * d)


def e_alpha_probit_(rs_m2_alpha, sig2_alpha_vb, q, d, log_zeta2_inv_vb, zeta2_inv_vb):
    """R/elbo.R:30-35 (sig2_alpha_vb is a length-q vector swept over responses)."""
    return 0.5 * (-np.sum(rs_m2_alpha * zeta2_inv_vb)
                  + d * np.sum(log_zeta2_inv_vb + np.log(sig2_alpha_vb)) + q * d)
