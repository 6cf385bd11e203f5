"""numpy restatement of the reference's R drivers -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/locus_oracle.c header): the reference holds no golden
vectors for this path and R is not installed, so nothing here could be checked
against the reference running.  The file follows the R sources statement by
statement (cited as file:line under /root/reference) and works, like R, on the
full p x d / n x d matrices on the host; the sweeps call the C oracle
(oracle/c_oracle.py).  Third-party arithmetic of R (digamma, lgamma, lbeta, pnorm)
is taken from scipy.special.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under locus_amd/ does.

Permutations: R draws `sample(0:(p-1))` inside the loop (R/locus_core.R:88).  Here
the caller passes `perm_fn(it) -> float64[p]` so that the GPU path and the oracle
see the same sweep order.
"""
import numpy as np
from scipy.special import betaln, digamma, gammaln, log_ndtr

from . import c_oracle

EPS_SQRT = np.finfo(np.float64).eps ** 0.5  # R/locus_core.R:31


# --------------------------------------------------------------------------- #
# R/utils.R
# --------------------------------------------------------------------------- #
def log_one_plus_exp_(x):  # R/utils.R:149-155
    x = np.asarray(x, dtype=np.float64)
    m = np.where(x < 0, 0.0, x)
    return np.log(np.exp(x - m) + np.exp(-m)) + m


def log_sigmoid_(chi):  # R/utils.R:158-162
    return -np.log(1 + np.exp(-chi))


def inv_mills_ratio_(y, U, log_1_pnorm_U, log_pnorm_U):  # R/utils.R:201-220
    if y == 1:
        m = np.exp(-U ** 2 / 2 - np.log(np.sqrt(2 * np.pi)) - log_pnorm_U)
        m = np.where(m < -U, -U, m)
    else:
        m = -np.exp(-U ** 2 / 2 - np.log(np.sqrt(2 * np.pi)) - log_1_pnorm_U)
        m = np.where(m > -U, -U, m)
    return m


def _dnorm_log(u):
    return -u ** 2 / 2 - 0.5 * np.log(2 * np.pi)


def inv_mills_ratio_matrix_(Y, U):  # R/utils.R:178-198
    m = np.empty_like(U)
    i1 = (Y == 1)
    U1 = U[i1]
    m1 = np.exp(_dnorm_log(U1) - log_ndtr(U1))
    m1 = np.where(m1 < -U1, -U1, m1)
    m[i1] = m1
    i0 = (Y == 0)
    U0 = U[i0]
    m0 = -np.exp(_dnorm_log(U0) - log_ndtr(-U0))
    m0 = np.where(m0 > -U0, -U0, m0)
    m[i0] = m0
    return m


def get_annealing_ladder_(anneal):  # R/utils.R:108-146
    k_m = 1.0 / anneal[1]
    m = int(anneal[2])
    if anneal[0] == 1:
        delta_k = k_m ** (1.0 / (1 - m)) - 1
        return (1 + delta_k) ** (1 - np.arange(m, 0, -1.0))
    if anneal[0] == 2:
        delta_k = (1 / k_m - 1) / (m - 1)
        return 1 / (1 + delta_k * (np.arange(m, 0, -1.0) - 1))
    delta_k = (1 - k_m) / (m - 1)
    return k_m + delta_k * (np.arange(1, m + 1.0) - 1)


# --------------------------------------------------------------------------- #
# R/update_vb.R (subset on the hot path)
# --------------------------------------------------------------------------- #
def update_beta_vb_(gam_vb, mu_beta_vb):  # :66
    return np.asfortranarray(gam_vb * mu_beta_vb)


def update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=False):  # :78-90
    if sweep:
        return (mu_beta_vb ** 2 + sig2_beta_vb[None, :]) * gam_vb
    return (mu_beta_vb ** 2 + sig2_beta_vb) * gam_vb


def update_m2_alpha_(alpha_vb, sig2_alpha_vb, sweep=False):  # :18-29
    if sweep:
        return alpha_vb ** 2 + np.asarray(sig2_alpha_vb).reshape(-1, 1)
    return sig2_alpha_vb + alpha_vb ** 2


def update_sig2_alpha_vb_(n, zeta2_inv_vb, intercept=False, c=1.0):  # :32-49 (tau NULL)
    den = n - 1 + np.array(zeta2_inv_vb, dtype=np.float64)
    if intercept:
        den[0] = den[0] + 1
    return 1 / (c * den)


def update_sig2_alpha_logit_vb_(Z, psi_vb, zeta2_inv_vb):  # :52-56
    return 1 / (2 * (Z ** 2).T @ psi_vb + np.asarray(zeta2_inv_vb).reshape(-1, 1))


def update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb=None, c=1.0):  # :93-104
    if tau_vb is None:
        return 1 / (c * (n - 1 + sig2_inv_vb))
    return 1 / (c * (n - 1 + sig2_inv_vb) * tau_vb)


def update_sig2_beta_logit_vb_(X, psi_vb, sig2_inv_vb):  # :107-111
    return np.asfortranarray(1 / (2 * (X ** 2).T @ psi_vb + sig2_inv_vb))


def update_chi_vb_(X, Z, beta_vb, m2_beta, mat_x_m1, mat_z_mu, sig2_alpha_vb):  # :163-167
    X2 = X ** 2
    return np.sqrt(X2 @ m2_beta + mat_x_m1 ** 2 - X2 @ beta_vb ** 2 + (Z ** 2) @ sig2_alpha_vb
                   + mat_z_mu ** 2 + 2 * mat_x_m1 * mat_z_mu)


def update_psi_logit_vb_(chi_vb):  # :170-174
    return np.exp(np.log(np.exp(log_sigmoid_(chi_vb)) - 1 / 2) - np.log(2 * chi_vb))


def update_a_vb(a, rs_gam, c=1.0):  # :181
    return c * (a + rs_gam) - c + 1


def update_b_vb(b, d, rs_gam, c=1.0):  # :184
    return c * (b - rs_gam + d) - c + 1


def update_log_om_vb(a, digam_sum, rs_gam, c=1.0):  # :187
    return digamma(c * (a + rs_gam) - c + 1) - digam_sum


def update_log_1_min_om_vb(b, d, digam_sum, rs_gam, c=1.0):  # :190
    return digamma(c * (b - rs_gam + d) - c + 1) - digam_sum


def update_lambda_vb_(lam, sum_gam, c=1.0):  # :198
    return c * (lam + sum_gam / 2) - c + 1


def update_nu_vb_(nu, m2_beta, tau_vb, c=1.0):  # :204
    return c * float(nu + tau_vb @ m2_beta.sum(axis=0) / 2)


def update_nu_bin_vb_(nu, m2_beta):  # :210
    return nu + m2_beta.sum() / 2


def update_log_sig2_inv_vb_(lambda_vb, nu_vb):  # :213
    return digamma(lambda_vb) - np.log(nu_vb)


def update_eta_vb_(n, eta, gam_vb, c=1.0):  # :220
    return c * (eta + n / 2 + gam_vb.sum(axis=0) / 2) - c + 1


def update_kappa_vb_(Y, kappa, mat_x_m1, beta_vb, m2_beta, sig2_inv_vb, c=1.0):  # :229-237
    n = Y.shape[0]
    return c * (kappa + ((Y ** 2).sum(axis=0) - 2 * (Y * mat_x_m1).sum(axis=0)
                         + (n - 1 + sig2_inv_vb) * m2_beta.sum(axis=0)
                         + (mat_x_m1 ** 2).sum(axis=0)
                         - (n - 1) * (beta_vb ** 2).sum(axis=0)) / 2)


def update_log_tau_vb_(eta_vb, kappa_vb):  # :275
    return digamma(eta_vb) - np.log(kappa_vb)


def update_phi_z_vb_(phi, d, c=1.0):  # :407
    return c * (phi + d / 2) - c + 1


def update_xi_bin_vb_(xi, m2_alpha):  # :413
    return xi + m2_alpha.sum(axis=1) / 2


def update_W_probit_(Y, mat_z_mu, mat_x_m1):  # :382-386
    return np.asfortranarray(mat_z_mu + mat_x_m1
                             + inv_mills_ratio_matrix_(Y, mat_z_mu + mat_x_m1))


def update_W_struct_(gam_vb, theta_vb):  # :389-399
    log_pnorm = log_ndtr(theta_vb)
    log_1_pnorm = log_ndtr(-theta_vb)
    imr0 = inv_mills_ratio_(0, theta_vb, log_1_pnorm, log_pnorm)
    imr1 = inv_mills_ratio_(1, theta_vb, log_1_pnorm, log_pnorm)
    return gam_vb * (imr1 - imr0)[:, None] + (theta_vb + imr0)[:, None]


def near_pd_corr(x, eig_tol=1e-6, conv_tol=1e-7, posd_tol=1e-8, maxit=100):
    """Matrix::nearPD(x, corr = TRUE, do2eigen = TRUE) (Matrix >= 1.2-2, DESCRIPTION:27;
    not vendored in the reference): Higham (2002) alternating projections with
    Dykstra's correction, then the positive-definiteness `do2eigen` polish.
    Restated from the published algorithm / documented defaults."""
    n = x.shape[1]
    D_S = np.zeros_like(x)
    X = x.copy()
    it = 0
    converged = False
    while it < maxit and not converged:
        Yk = X
        R = Yk - D_S
        d, Q = np.linalg.eigh(R)
        d = d[::-1]
        Q = Q[:, ::-1]
        pmask = d > eig_tol * d[0]
        if not pmask.any():
            raise ValueError("Matrix seems negative semi-definite")
        Qp = Q[:, pmask]
        X = (Qp * d[pmask][None, :]) @ Qp.T
        D_S = X - R
        X = X.copy()
        np.fill_diagonal(X, 1.0)
        conv = np.abs(Yk - X).sum(axis=1).max() / np.abs(Yk).sum(axis=1).max()  # "I" norm
        it += 1
        converged = conv <= conv_tol
    d, Q = np.linalg.eigh(X)
    d = d[::-1]
    Q = Q[:, ::-1]
    Eps = posd_tol * abs(d[0])
    if d[n - 1] < Eps:
        d = np.where(d < Eps, Eps, d)
        o_diag = np.diag(X).copy()
        X = Q @ (d[:, None] * Q.T)
        Dv = np.sqrt(np.maximum(Eps, o_diag) / np.diag(X))
        X = Dv[:, None] * X * Dv[None, :]
    X = X.copy()
    np.fill_diagonal(X, 1.0)
    return X


def update_sig2_theta_vb_(d, p, vec_fac_st, s02, X):  # :327-374 (list_struct non-NULL)
    bl_ids = list(dict.fromkeys(vec_fac_st.tolist()))  # unique(), first-occurrence order
    S0_inv, sig2_theta_vb, vec_sum_log_det = [], [], []
    for bl in bl_ids:
        Xb = X[:, vec_fac_st == bl]
        corX = np.corrcoef(Xb, rowvar=False).reshape(Xb.shape[1], Xb.shape[1])
        corX = near_pd_corr(corX)
        S = np.linalg.inv(corX) / s02
        V = np.linalg.inv(S + d * np.eye(S.shape[0]))
        S0_inv.append(S)
        sig2_theta_vb.append(V)
        s1, l1 = np.linalg.slogdet(S)
        s2, l2 = np.linalg.slogdet(V)
        vec_sum_log_det.append(s1 * l1 + s2 * l2)  # R/utils.R:164-176
    return S0_inv, sig2_theta_vb, np.array(vec_sum_log_det), bl_ids


def update_theta_vb_(W, m0, S0_inv, sig2_theta_vb, vec_fac_st, bl_ids):  # :282-324 (blocks)
    theta = np.empty(W.shape[0])
    rsW = W.sum(axis=1)
    for bl, S, V in zip(bl_ids, S0_inv, sig2_theta_vb):
        idx = (vec_fac_st == bl)
        theta[idx] = V @ (rsW[idx] + S @ m0[idx])
    return theta


# --------------------------------------------------------------------------- #
# R/elbo.R (subset on the hot path)
# --------------------------------------------------------------------------- #
def e_y_(n, kappa, kappa_vb, log_tau_vb, m2_beta, sig2_inv_vb, tau_vb):  # :189-200
    arg = (-n / 2 * np.log(2 * np.pi) + n / 2 * log_tau_vb
           - tau_vb * (kappa_vb - m2_beta.sum(axis=0) * sig2_inv_vb / 2 - kappa))
    return arg.sum()


def e_beta_gamma_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                  m2_beta, sig2_beta_vb, sig2_inv_vb, tau_vb):  # :43-54
    eps = np.finfo(np.float64).eps
    return np.sum(log_sig2_inv_vb * gam_vb / 2 + gam_vb * log_tau_vb[None, :] / 2
                  - m2_beta * tau_vb[None, :] * sig2_inv_vb / 2
                  + gam_vb * log_om_vb[:, None]
                  + (1 - gam_vb) * log_1_min_om_vb[:, None]
                  + 1 / 2 * gam_vb * (np.log(sig2_beta_vb) + 1)[None, :]
                  - gam_vb * np.log(gam_vb + eps) - (1 - gam_vb) * np.log(1 - gam_vb + eps))


def e_beta_gamma_bin_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, m2_beta,
                      sig2_beta_vb, sig2_inv_vb):  # :77-87
    eps = np.finfo(np.float64).eps
    return np.sum(gam_vb * log_sig2_inv_vb / 2 - m2_beta * sig2_inv_vb / 2
                  + gam_vb * log_om_vb[:, None]
                  + (1 - gam_vb) * log_1_min_om_vb[:, None]
                  + gam_vb * (np.log(sig2_beta_vb) + 1) / 2
                  - gam_vb * np.log(gam_vb + eps) - (1 - gam_vb) * np.log(1 - gam_vb + eps))


def e_beta_gamma_struct_(gam_vb, log_sig2_inv_vb, log_tau_vb, theta_vb, m2_beta,
                         sig2_beta_vb, list_sig2_theta_vb, sig2_inv_vb, tau_vb):  # :90-106
    eps = np.finfo(np.float64).eps ** 0.75
    diag_sig2_theta_vb = np.concatenate([np.diag(V) for V in list_sig2_theta_vb])
    return np.sum(log_sig2_inv_vb * gam_vb / 2 + gam_vb * log_tau_vb[None, :] / 2
                  - m2_beta * tau_vb[None, :] * sig2_inv_vb / 2
                  + gam_vb * log_ndtr(theta_vb)[:, None]
                  + ((1 - gam_vb) * log_ndtr(-theta_vb)[:, None]
                     - (diag_sig2_theta_vb / 2)[:, None])
                  + 1 / 2 * gam_vb * (np.log(sig2_beta_vb) + 1)[None, :]
                  - gam_vb * np.log(gam_vb + eps) - (1 - gam_vb) * np.log(1 - gam_vb + eps))


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


def e_theta_(m0, theta_vb, list_S0_inv, list_sig2_theta_vb, vec_fac_st, bl_ids,
             vec_sum_log_det):  # :151-182 (blocks)
    tot = 0.0
    for i, bl in enumerate(bl_ids):
        idx = (vec_fac_st == bl)
        dlt = theta_vb[idx] - m0[idx]
        S, V = list_S0_inv[i], list_sig2_theta_vb[i]
        tot += (vec_sum_log_det[i] - dlt @ (S @ dlt) - np.sum(S * V) + S.shape[1]) / 2
    return tot


def e_y_logit_(X, Y, Z, chi_vb, beta_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu, alpha_vb,
               psi_vb):  # :210-218
    X2, Z2 = X ** 2, Z ** 2
    return np.sum(log_sigmoid_(chi_vb) + Y * (mat_x_m1 + mat_z_mu) - chi_vb / 2
                  - psi_vb * (X2 @ m2_beta + mat_x_m1 ** 2 - X2 @ beta_vb ** 2
                              + Z2 @ m2_alpha + mat_z_mu ** 2 - Z2 @ alpha_vb ** 2
                              + 2 * mat_x_m1 * mat_z_mu - chi_vb ** 2))


def e_y_probit_(X, Y, Z, beta_vb, m2_beta, mat_x_m1, mat_z_mu, sig2_alpha_vb):  # :221-237
    U = mat_x_m1 + mat_z_mu
    arg = (Y * log_ndtr(U) + (1 - Y) * log_ndtr(-U) - (X ** 2) @ (m2_beta - beta_vb ** 2) / 2)
    arg = arg - ((Z ** 2) @ sig2_alpha_vb / 2)[:, None]
    return arg.sum()


def e_alpha_logit_(m2_alpha, log_zeta2_inv_vb, sig2_alpha_vb, zeta2_inv_vb):  # :22-27
    return 1 / 2 * np.sum(-m2_alpha * zeta2_inv_vb[:, None] + log_zeta2_inv_vb[:, None]
                          + np.log(sig2_alpha_vb) + 1)


def e_alpha_probit_(m2_alpha, log_zeta2_inv_vb, sig2_alpha_vb, zeta2_inv_vb):  # :30-35
    return 1 / 2 * np.sum(-m2_alpha * zeta2_inv_vb[:, None]
                          + (log_zeta2_inv_vb + np.log(sig2_alpha_vb))[:, None] + 1)


def e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb):  # :244-249
    return np.sum((phi - phi_vb) * log_zeta2_inv_vb - (xi - xi_vb) * zeta2_inv_vb
                  + phi * np.log(xi) - phi_vb * np.log(xi_vb) - gammaln(phi) + gammaln(phi_vb))


# --------------------------------------------------------------------------- #
# R/locus_core.R
# --------------------------------------------------------------------------- #
def elbo_(Y, a, a_vb, b, b_vb, beta_vb, eta, gam_vb, kappa, lam, nu, sig2_beta_vb,
          sig2_inv_vb, tau_vb, m2_beta, mat_x_m1, sum_gam):  # R/locus_core.R:261-292
    n = Y.shape[0]
    eta_vb = update_eta_vb_(n, eta, gam_vb)
    kappa_vb = update_kappa_vb_(Y, kappa, mat_x_m1, beta_vb, m2_beta, sig2_inv_vb)
    lambda_vb = update_lambda_vb_(lam, sum_gam)
    nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
    log_tau_vb = digamma(eta_vb) - np.log(kappa_vb)
    log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
    log_om_vb = digamma(a_vb) - digamma(a_vb + b_vb)
    log_1_min_om_vb = digamma(b_vb) - digamma(a_vb + b_vb)
    A = e_y_(n, kappa, kappa_vb, log_tau_vb, m2_beta, sig2_inv_vb, tau_vb)
    B = e_beta_gamma_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                      m2_beta, sig2_beta_vb, sig2_inv_vb, tau_vb)
    C = e_tau_(eta, eta_vb, kappa, kappa_vb, log_tau_vb, tau_vb)
    D = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    E = e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb)
    return A + B + C + D + E


def locus_core_(Y, X, list_hyper, gam_vb, mu_beta_vb, sig2_beta_vb, tau_vb, tol, maxit,
                perm_fn, anneal=None, batch="y", debug=True, full_output=True,
                trace=None, k_perm_fn=None, perm0_fn=None):
    """R/locus_core.R:8-255.  `trace`, if a list, receives (it, lb_new) per iteration.
    The internal schemes "x" and "0" sweep the responses in the order k_perm_fn(it) (sample(1:d),
    :102,137); "0" sweeps the covariates of response k in the order perm0_fn(it, k) (:142)."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    n, d = Y.shape
    p = X.shape[1]
    a, b = list_hyper["a"], list_hyper["b"]
    eta, kappa = list_hyper["eta"], list_hyper["kappa"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
    tau_vb = np.array(tau_vb, dtype=np.float64)

    if anneal is None:
        annealing, c = False, 1.0
        ladder = None
    else:
        annealing = True
        ladder = get_annealing_ladder_(anneal)
        c = float(ladder[0])

    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)                       # :36
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)                           # :39
    rs_gam = gam_vb.sum(axis=1)
    sum_gam = rs_gam.sum()
    converged, lb_new, it = False, -np.inf, 0
    lb_old = -np.inf
    om_vb = None

    while (not converged) and it < maxit:                               # :49
        lb_old = lb_new
        it += 1
        digam_sum = digamma(c * (a + b + d) - 2 * c + 2)                # :57
        lambda_vb = update_lambda_vb_(lam, sum_gam, c)
        nu_vb = update_nu_vb_(nu, m2_beta, tau_vb, c)
        sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = update_eta_vb_(n, eta, gam_vb, c)
        kappa_vb = update_kappa_vb_(Y, kappa, mat_x_m1, beta_vb, m2_beta, sig2_inv_vb, c)
        tau_vb = eta_vb / kappa_vb
        sig2_beta_vb = update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb, c)
        log_tau_vb = update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb = update_log_om_vb(a, digam_sum, rs_gam, c)
        log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam, c)

        if batch == "y":                                                # :81-95
            shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
            c_oracle.coreLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                              log_tau_vb, beta_vb, mat_x_m1, mu_beta_vb, sig2_beta_vb,
                              tau_vb, shuffled_ind, c)
        elif batch == "x-y":                                            # :121-133
            if annealing:
                raise RuntimeError("Annealing not implemented for this scheme. Exit.")
            c_oracle.coreBatch(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                               log_tau_vb, beta_vb, mat_x_m1, mu_beta_vb, sig2_beta_vb,
                               tau_vb)
        elif batch == "x":                                              # :97-117
            for k in k_perm_fn(it):
                k = int(k)
                mu_beta_vb[:, k] = c * sig2_beta_vb[k] * tau_vb[k] * (
                    (Y[:, k] - mat_x_m1[:, k]) @ X + (n - 1) * beta_vb[:, k])
                gam_vb[:, k] = np.exp(-log_one_plus_exp_(c * (
                    log_1_min_om_vb - log_om_vb - log_tau_vb[k] / 2 - log_sig2_inv_vb / 2
                    - mu_beta_vb[:, k] ** 2 / (2 * sig2_beta_vb[k]) - np.log(sig2_beta_vb[k]) / 2)))
                beta_vb[:, k] = mu_beta_vb[:, k] * gam_vb[:, k]
                mat_x_m1[:, k] = X @ beta_vb[:, k]
        elif batch == "0":                                              # :135-161
            for k in k_perm_fn(it):
                k = int(k)
                log_om_vb = update_log_om_vb(a, digam_sum, rs_gam, c)
                log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam, c)
                for j in perm0_fn(it, k):
                    j = int(j)
                    mat_x_m1[:, k] -= X[:, j] * beta_vb[j, k]
                    mu_beta_vb[j, k] = c * sig2_beta_vb[k] * tau_vb[k] * (
                        (Y[:, k] - mat_x_m1[:, k]) @ X[:, j])
                    gam_vb[j, k] = np.exp(-log_one_plus_exp_(c * (
                        log_1_min_om_vb[j] - log_om_vb[j] - log_tau_vb[k] / 2 - log_sig2_inv_vb / 2
                        - mu_beta_vb[j, k] ** 2 / (2 * sig2_beta_vb[k])
                        - np.log(sig2_beta_vb[k]) / 2)))
                    beta_vb[j, k] = mu_beta_vb[j, k] * gam_vb[j, k]
                    mat_x_m1[:, k] += X[:, j] * beta_vb[j, k]
                rs_gam = gam_vb.sum(axis=1)
        else:
            raise ValueError("Batch scheme not defined. Exit.")
        rs_gam = gam_vb.sum(axis=1)

        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)  # :169
        a_vb = update_a_vb(a, rs_gam, c)
        b_vb = update_b_vb(b, d, rs_gam, c)
        om_vb = a_vb / (a_vb + b_vb)
        sum_gam = rs_gam.sum()

        if annealing:                                                   # :177-190
            c = float(ladder[it]) if it < len(ladder) else 1.0
            if np.isclose(c, 1.0, rtol=1.5e-8, atol=0):
                annealing = False
        else:
            lb_new = elbo_(Y, a, a_vb, b, b_vb, beta_vb, eta, gam_vb, kappa, lam, nu,
                           sig2_beta_vb, sig2_inv_vb, tau_vb, m2_beta, mat_x_m1, sum_gam)
            if trace is not None:
                trace.append((it, float(lb_new)))
            if debug and lb_new + EPS_SQRT < lb_old:                    # :202-203
                raise RuntimeError("ELBO not increasing monotonically. Exit. ")
            converged = abs(lb_new - lb_old) < tol                      # :205

    out = dict(beta_vb=beta_vb, gam_vb=gam_vb, om_vb=om_vb, converged=converged, it=it,
               lb_opt=lb_new, diff_lb=abs(lb_new - lb_old))
    if full_output:
        out.update(mu_beta_vb=mu_beta_vb, sig2_beta_vb=sig2_beta_vb, sig2_inv_vb=sig2_inv_vb,
                   tau_vb=tau_vb, m2_beta=m2_beta, mat_x_m1=mat_x_m1, sum_gam=sum_gam)
    return out


# --------------------------------------------------------------------------- #
# R/locus_struct_core.R
# --------------------------------------------------------------------------- #
def elbo_struct_(Y, beta_vb, eta, gam_vb, kappa, lam, m0, theta_vb, nu, sig2_beta_vb,
                 list_S0_inv, list_sig2_theta_vb, sig2_inv_vb, tau_vb, m2_beta, mat_x_m1,
                 vec_fac_st, bl_ids, vec_sum_log_det):  # R/locus_struct_core.R:191-225
    n = Y.shape[0]
    eta_vb = update_eta_vb_(n, eta, gam_vb)
    kappa_vb = update_kappa_vb_(Y, kappa, mat_x_m1, beta_vb, m2_beta, sig2_inv_vb)
    lambda_vb = update_lambda_vb_(lam, gam_vb.sum())
    nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
    log_tau_vb = update_log_tau_vb_(eta_vb, kappa_vb)
    log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
    A = e_y_(n, kappa, kappa_vb, log_tau_vb, m2_beta, sig2_inv_vb, tau_vb)
    B = e_beta_gamma_struct_(gam_vb, log_sig2_inv_vb, log_tau_vb, theta_vb, m2_beta,
                             sig2_beta_vb, list_sig2_theta_vb, sig2_inv_vb, tau_vb)
    C = e_theta_(m0, theta_vb, list_S0_inv, list_sig2_theta_vb, vec_fac_st, bl_ids,
                 vec_sum_log_det)
    D = e_tau_(eta, eta_vb, kappa, kappa_vb, log_tau_vb, tau_vb)
    E = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    return A + B + C + D + E


def locus_struct_core_(Y, X, list_hyper, gam_vb, mu_beta_vb, sig2_beta_vb, tau_vb,
                       vec_fac_st, tol, maxit, perm_fn, debug=False, trace=None):
    """R/locus_struct_core.R:8-184."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    n, d = Y.shape
    p = X.shape[1]
    eta, kappa = list_hyper["eta"], list_hyper["kappa"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    m0, s02 = np.asarray(list_hyper["m0"], dtype=np.float64), list_hyper["s02"]
    vec_fac_st = np.asarray(vec_fac_st)
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
    tau_vb = np.array(tau_vb, dtype=np.float64)

    theta_vb = m0.copy()                                                # :25
    list_S0_inv, list_sig2_theta_vb, vec_sum_log_det, bl_ids = \
        update_sig2_theta_vb_(d, p, vec_fac_st, s02, X)                 # :30
    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)
    converged, lb_new, it = False, -np.inf, 0
    lb_old = -np.inf

    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        lambda_vb = update_lambda_vb_(lam, gam_vb.sum())                # :60
        nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
        sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = update_eta_vb_(n, eta, gam_vb)
        kappa_vb = update_kappa_vb_(Y, kappa, mat_x_m1, beta_vb, m2_beta, sig2_inv_vb)
        tau_vb = eta_vb / kappa_vb
        sig2_beta_vb = update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb)
        log_tau_vb = update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_Phi = log_ndtr(theta_vb)                                    # :83-84
        log_1_min_Phi = log_ndtr(-theta_vb)
        shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
        c_oracle.coreStructLoop(X, Y, gam_vb, log_Phi, log_1_min_Phi, log_sig2_inv_vb,
                                log_tau_vb, beta_vb, mat_x_m1, mu_beta_vb, sig2_beta_vb,
                                tau_vb, shuffled_ind)                    # :89-91
        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)  # :124
        W = update_W_struct_(gam_vb, theta_vb)                          # :127
        theta_vb = update_theta_vb_(W, m0, list_S0_inv, list_sig2_theta_vb, vec_fac_st,
                                    bl_ids)                              # :129
        lb_new = elbo_struct_(Y, beta_vb, eta, gam_vb, kappa, lam, m0, theta_vb, nu,
                              sig2_beta_vb, list_S0_inv, list_sig2_theta_vb, sig2_inv_vb,
                              tau_vb, m2_beta, mat_x_m1, vec_fac_st, bl_ids,
                              vec_sum_log_det)
        if trace is not None:
            trace.append((it, float(lb_new)))
        if debug and lb_new < lb_old:
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol

    return dict(beta_vb=beta_vb, gam_vb=gam_vb, theta_vb=theta_vb, converged=converged,
                it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old), mu_beta_vb=mu_beta_vb,
                mat_x_m1=mat_x_m1, tau_vb=tau_vb, sig2_beta_vb=sig2_beta_vb)


# --------------------------------------------------------------------------- #
# R/locus_logistic_core.R
# --------------------------------------------------------------------------- #
def elbo_logit_(Y, X, Z, a, a_vb, b, b_vb, beta_vb, chi_vb, gam_vb, lam, nu, phi, phi_vb,
                psi_vb, sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, xi, zeta2_inv_vb,
                alpha_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu,
                sum_gam):  # R/locus_logistic_core.R:271-303
    lambda_vb = update_lambda_vb_(lam, sum_gam)
    nu_vb = update_nu_bin_vb_(nu, m2_beta)
    xi_vb = update_xi_bin_vb_(xi, m2_alpha)
    log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
    log_zeta2_inv_vb = digamma(phi_vb) - np.log(xi_vb)
    log_om_vb = digamma(a_vb) - digamma(a_vb + b_vb)
    log_1_min_om_vb = digamma(b_vb) - digamma(a_vb + b_vb)
    A = e_y_logit_(X, Y, Z, chi_vb, beta_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu,
                   alpha_vb, psi_vb)
    B = e_beta_gamma_bin_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, m2_beta,
                          sig2_beta_vb, sig2_inv_vb)
    C = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    D = e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb)
    E = e_alpha_logit_(m2_alpha, log_zeta2_inv_vb, sig2_alpha_vb, zeta2_inv_vb)
    F = e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb)
    return A + B + C + D + E + F


def locus_logit_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                      sig2_beta_vb, tol, maxit, perm_fn, alpha_perm_fn=None, debug=False,
                      trace=None):
    """R/locus_logistic_core.R:8-265.  Y already has 1/2 subtracted (:13)."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    Z = np.asfortranarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    a, b = list_hyper["a"], list_hyper["b"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    phi, xi = np.atleast_1d(np.asarray(list_hyper["phi"], dtype=np.float64)), \
        np.atleast_1d(np.asarray(list_hyper["xi"], dtype=np.float64))
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    alpha_vb = np.array(alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_alpha_vb = np.array(sig2_alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_beta_vb = np.asfortranarray(sig2_beta_vb, dtype=np.float64).copy(order="F")

    m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)                # :23
    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)
    mat_z_mu = np.asfortranarray(Z @ alpha_vb)
    phi_vb = update_phi_z_vb_(phi, d)
    rs_gam = gam_vb.sum(axis=1)
    sum_gam = rs_gam.sum()
    digam_sum = digamma(a + b + d)                                      # :34
    converged, lb_new, it = False, -np.inf, 0
    lb_old = -np.inf
    om_vb = None

    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        chi_vb = update_chi_vb_(X, Z, beta_vb, m2_beta, mat_x_m1, mat_z_mu, sig2_alpha_vb)
        psi_vb = np.asfortranarray(update_psi_logit_vb_(chi_vb))        # :49-51
        xi_vb = update_xi_bin_vb_(xi, m2_alpha)
        zeta2_inv_vb = phi_vb / xi_vb
        lambda_vb = update_lambda_vb_(lam, sum_gam)
        nu_vb = update_nu_bin_vb_(nu, m2_beta)
        sig2_inv_vb = lambda_vb / nu_vb
        sig2_alpha_vb = update_sig2_alpha_logit_vb_(Z, psi_vb, zeta2_inv_vb)
        sig2_beta_vb = update_sig2_beta_logit_vb_(X, psi_vb, sig2_inv_vb)
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb = update_log_om_vb(a, digam_sum, rs_gam)
        log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam)

        order = range(q) if alpha_perm_fn is None else alpha_perm_fn(it)
        for i in order:                                                 # :79-85
            mat_z_mu = mat_z_mu - np.outer(Z[:, i], alpha_vb[i, :])
            alpha_vb[i, :] = sig2_alpha_vb[i, :] * (
                (Y - 2 * psi_vb * (mat_z_mu + mat_x_m1)).T @ Z[:, i])
            mat_z_mu = mat_z_mu + np.outer(Z[:, i], alpha_vb[i, :])
        mat_z_mu = np.asfortranarray(mat_z_mu)

        shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
        c_oracle.coreLogitLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                               beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, psi_vb,
                               sig2_beta_vb, shuffled_ind)               # :90-92
        rs_gam = gam_vb.sum(axis=1)
        sum_gam = rs_gam.sum()
        m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)            # :204-205
        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb)
        a_vb = update_a_vb(a, rs_gam)
        b_vb = update_b_vb(b, d, rs_gam)
        om_vb = a_vb / (a_vb + b_vb)
        lb_new = elbo_logit_(Y, X, Z, a, a_vb, b, b_vb, beta_vb, chi_vb, gam_vb, lam, nu,
                             phi, phi_vb, psi_vb, sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb,
                             xi, zeta2_inv_vb, alpha_vb, m2_alpha, m2_beta, mat_x_m1,
                             mat_z_mu, sum_gam)
        if trace is not None:
            trace.append((it, float(lb_new)))
        if debug and lb_new < lb_old:
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol

    return dict(beta_vb=beta_vb, gam_vb=gam_vb, om_vb=om_vb, alpha_vb=alpha_vb,
                converged=converged, it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old),
                mu_beta_vb=mu_beta_vb, mat_x_m1=mat_x_m1, mat_z_mu=mat_z_mu,
                sig2_beta_vb=sig2_beta_vb, psi_vb=psi_vb)


# --------------------------------------------------------------------------- #
# R/locus_probit_core.R
# --------------------------------------------------------------------------- #
def elbo_probit_(Y, X, Z, a, a_vb, b, b_vb, beta_vb, gam_vb, lam, nu, phi, phi_vb,
                 sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, xi, zeta2_inv_vb, alpha_vb,
                 m2_alpha, m2_beta, mat_x_m1, mat_z_mu,
                 sum_gam):  # R/locus_probit_core.R:257-288
    lambda_vb = update_lambda_vb_(lam, sum_gam)
    nu_vb = update_nu_bin_vb_(nu, m2_beta)
    xi_vb = update_xi_bin_vb_(xi, m2_alpha)
    log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
    log_zeta2_inv_vb = digamma(phi_vb) - np.log(xi_vb)
    log_om_vb = digamma(a_vb) - digamma(a_vb + b_vb)
    log_1_min_om_vb = digamma(b_vb) - digamma(a_vb + b_vb)
    A = e_y_probit_(X, Y, Z, beta_vb, m2_beta, mat_x_m1, mat_z_mu, sig2_alpha_vb)
    B = e_beta_gamma_bin_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, m2_beta,
                          sig2_beta_vb, sig2_inv_vb)
    C = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    D = e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb)
    E = e_alpha_probit_(m2_alpha, log_zeta2_inv_vb, sig2_alpha_vb, zeta2_inv_vb)
    F = e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb)
    return A + B + C + D + E + F


def locus_probit_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                       sig2_beta_vb, tol, maxit, perm_fn, alpha_perm_fn=None, debug=False,
                       trace=None):
    """R/locus_probit_core.R:8-250.  Y is 0/1; Z[:, 0] is the intercept (:13)."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    Z = np.asfortranarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    a, b = list_hyper["a"], list_hyper["b"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    phi, xi = np.atleast_1d(np.asarray(list_hyper["phi"], dtype=np.float64)), \
        np.atleast_1d(np.asarray(list_hyper["xi"], dtype=np.float64))
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    alpha_vb = np.array(alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_alpha_vb = np.atleast_1d(np.array(sig2_alpha_vb, dtype=np.float64))
    sig2_beta_vb = float(sig2_beta_vb)

    m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb, sweep=True)    # :23
    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)
    mat_z_mu = np.asfortranarray(Z @ alpha_vb)
    W = update_W_probit_(Y, mat_z_mu, mat_x_m1)                         # :30
    rs_gam = gam_vb.sum(axis=1)
    sum_gam = rs_gam.sum()
    digam_sum = digamma(a + b + d)
    phi_vb = update_phi_z_vb_(phi, d)
    converged, lb_new, it = False, -np.inf, 0
    lb_old = -np.inf
    om_vb = None

    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        xi_vb = update_xi_bin_vb_(xi, m2_alpha)                         # :51
        zeta2_inv_vb = phi_vb / xi_vb
        lambda_vb = update_lambda_vb_(lam, sum_gam)
        nu_vb = update_nu_bin_vb_(nu, m2_beta)
        sig2_inv_vb = lambda_vb / nu_vb
        sig2_alpha_vb = update_sig2_alpha_vb_(n, zeta2_inv_vb, intercept=True)  # :63
        sig2_beta_vb = float(update_sig2_beta_vb_(n, sig2_inv_vb))
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb = update_log_om_vb(a, digam_sum, rs_gam)
        log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam)

        order = range(q) if alpha_perm_fn is None else alpha_perm_fn(it)
        for i in order:                                                 # :76-84
            mat_z_mu = mat_z_mu - np.outer(Z[:, i], alpha_vb[i, :])
            alpha_vb[i, :] = sig2_alpha_vb[i] * ((W - mat_z_mu - mat_x_m1).T @ Z[:, i])
            mat_z_mu = mat_z_mu + np.outer(Z[:, i], alpha_vb[i, :])
        mat_z_mu = np.asfortranarray(mat_z_mu)

        shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
        c_oracle.coreProbitLoop(X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                                beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb,
                                shuffled_ind)                            # :89-91
        rs_gam = gam_vb.sum(axis=1)
        sum_gam = rs_gam.sum()
        m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb, sweep=True)  # :187
        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb)
        W = update_W_probit_(Y, mat_z_mu, mat_x_m1)                     # :190
        a_vb = update_a_vb(a, rs_gam)
        b_vb = update_b_vb(b, d, rs_gam)
        om_vb = a_vb / (a_vb + b_vb)
        lb_new = elbo_probit_(Y, X, Z, a, a_vb, b, b_vb, beta_vb, gam_vb, lam, nu, phi,
                              phi_vb, sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, xi,
                              zeta2_inv_vb, alpha_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu,
                              sum_gam)
        if trace is not None:
            trace.append((it, float(lb_new)))
        if debug and lb_new < lb_old:
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol

    return dict(beta_vb=beta_vb, gam_vb=gam_vb, om_vb=om_vb, alpha_vb=alpha_vb,
                converged=converged, it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old),
                mu_beta_vb=mu_beta_vb, mat_x_m1=mat_x_m1, mat_z_mu=mat_z_mu,
                sig2_beta_vb=sig2_beta_vb, sig2_alpha_vb=sig2_alpha_vb, sig2_inv_vb=sig2_inv_vb,
                zeta2_inv_vb=zeta2_inv_vb)


# --------------------------------------------------------------------------- #
# R/locus_z_core.R (identity link with fixed covariates Z)
# --------------------------------------------------------------------------- #
def update_sig2_alpha_z_vb_(n, zeta2_inv_vb, tau_vb, c=1.0):  # R/update_vb.R:32-49 (tau given)
    den = n - 1 + np.asarray(zeta2_inv_vb, dtype=np.float64)
    return 1 / (c * np.outer(den, tau_vb))


def update_eta_z_vb_(n, q, eta, gam_vb, c=1.0):  # R/update_vb.R:226
    return c * (eta + n / 2 + gam_vb.sum(axis=0) / 2 + q / 2) - c + 1


def update_kappa_z_vb_(Y, kappa, alpha_vb, beta_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu,
                       sig2_inv_vb, zeta2_inv_vb, c=1.0):  # R/update_vb.R:255-272
    n = Y.shape[0]
    return c * (kappa + ((Y ** 2).sum(axis=0) - 2 * (Y * (mat_x_m1 + mat_z_mu)).sum(axis=0)
                         + (n - 1 + sig2_inv_vb) * m2_beta.sum(axis=0)
                         + (mat_x_m1 ** 2).sum(axis=0) - (n - 1) * (beta_vb ** 2).sum(axis=0)
                         + (n - 1) * m2_alpha.sum(axis=0) + m2_alpha.T @ zeta2_inv_vb
                         + (mat_z_mu ** 2).sum(axis=0) - (n - 1) * (alpha_vb ** 2).sum(axis=0)
                         + 2 * (mat_x_m1 * mat_z_mu).sum(axis=0)) / 2)


def update_xi_z_vb_(xi, tau_vb, m2_alpha, c=1.0):  # R/update_vb.R:410
    return c * (xi + m2_alpha @ tau_vb / 2)


def e_alpha_(m2_alpha, log_tau_vb, log_zeta2_inv_vb, sig2_alpha_vb, tau_vb, zeta2_inv_vb):
    """R/elbo.R:11-19."""
    return 1 / 2 * np.sum(-(m2_alpha * tau_vb[None, :]) * zeta2_inv_vb[:, None]
                          + log_tau_vb[None, :] + log_zeta2_inv_vb[:, None]
                          + np.log(sig2_alpha_vb) + 1)


def elbo_z_(Y, Z, a, a_vb, b, b_vb, beta_vb, eta, gam_vb, kappa, lam, alpha_vb, nu, phi, phi_vb,
            sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, tau_vb, xi, zeta2_inv_vb, m2_alpha, m2_beta,
            mat_x_m1, mat_z_mu, sum_gam):
    """R/locus_z_core.R:299-342."""
    n, q = Y.shape[0], Z.shape[1]
    xi_vb = update_xi_z_vb_(xi, tau_vb, m2_alpha)
    eta_vb = update_eta_z_vb_(n, q, eta, gam_vb)
    kappa_vb = update_kappa_z_vb_(Y, kappa, alpha_vb, beta_vb, m2_alpha, m2_beta, mat_x_m1,
                                  mat_z_mu, sig2_inv_vb, zeta2_inv_vb)
    lambda_vb = update_lambda_vb_(lam, sum_gam)
    nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
    log_tau_vb = digamma(eta_vb) - np.log(kappa_vb)
    log_zeta2_inv_vb = digamma(phi_vb) - np.log(xi_vb)
    log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
    log_om_vb = digamma(a_vb) - digamma(a_vb + b_vb)
    log_1_min_om_vb = digamma(b_vb) - digamma(a_vb + b_vb)
    elbo_A = (e_y_(n, kappa, kappa_vb, log_tau_vb, m2_beta, sig2_inv_vb, tau_vb)
              + np.sum(tau_vb * (m2_alpha.T @ zeta2_inv_vb) / 2))       # R/elbo.R:195-196
    elbo_B = e_beta_gamma_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                           m2_beta, sig2_beta_vb, sig2_inv_vb, tau_vb)
    elbo_C = e_tau_(eta, eta_vb, kappa, kappa_vb, log_tau_vb, tau_vb)
    elbo_D = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    elbo_E = e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb)
    elbo_F = e_alpha_(m2_alpha, log_tau_vb, log_zeta2_inv_vb, sig2_alpha_vb, tau_vb,
                      zeta2_inv_vb)
    elbo_G = e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb)
    return elbo_A + elbo_B + elbo_C + elbo_D + elbo_E + elbo_F + elbo_G


def locus_z_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                  sig2_beta_vb, tau_vb, tol, maxit, perm_fn, alpha_perm_fn=None, anneal=None,
                  debug=True, trace=None):
    """R/locus_z_core.R:8-290, scheme batch = "y".  Y centred, X and Z standardised."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    Z = np.asfortranarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    a, b = list_hyper["a"], list_hyper["b"]
    eta, kappa = list_hyper["eta"], list_hyper["kappa"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    phi = np.atleast_1d(np.asarray(list_hyper["phi"], dtype=np.float64))
    xi = np.atleast_1d(np.asarray(list_hyper["xi"], dtype=np.float64))
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    alpha_vb = np.array(alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_alpha_vb = np.array(sig2_alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
    tau_vb = np.array(tau_vb, dtype=np.float64)

    if anneal is None:
        annealing, c, ladder = False, 1.0, None
    else:
        annealing, ladder = True, get_annealing_ladder_(anneal)
        c = float(ladder[0])

    m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)                # :32
    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)
    mat_z_mu = np.asfortranarray(Z @ alpha_vb)
    rs_gam = gam_vb.sum(axis=1)
    sum_gam = rs_gam.sum()
    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    om_vb = None

    while (not converged) and it < maxit:                               # :49
        lb_old = lb_new
        it += 1
        digam_sum = digamma(c * (a + b + d) - 2 * c + 2)
        phi_vb = update_phi_z_vb_(phi, d, c)                            # :60-63
        xi_vb = update_xi_z_vb_(xi, tau_vb, m2_alpha, c)
        zeta2_inv_vb = phi_vb / xi_vb
        lambda_vb = update_lambda_vb_(lam, sum_gam, c)                  # :67-70
        nu_vb = update_nu_vb_(nu, m2_beta, tau_vb, c)
        sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = update_eta_z_vb_(n, q, eta, gam_vb, c)                 # :74-79
        kappa_vb = update_kappa_z_vb_(Y, kappa, alpha_vb, beta_vb, m2_alpha, m2_beta, mat_x_m1,
                                      mat_z_mu, sig2_inv_vb, zeta2_inv_vb, c)
        tau_vb = eta_vb / kappa_vb
        sig2_alpha_vb = update_sig2_alpha_z_vb_(n, zeta2_inv_vb, tau_vb, c)   # :82-83
        sig2_beta_vb = update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb, c)
        log_tau_vb = update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb = update_log_om_vb(a, digam_sum, rs_gam, c)           # :93-94
        log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam, c)

        order = range(q) if alpha_perm_fn is None else alpha_perm_fn(it)
        for i in order:                                                 # :96-103
            mat_z_mu = mat_z_mu - np.outer(Z[:, i], alpha_vb[i, :])
            alpha_vb[i, :] = c * sig2_alpha_vb[i, :] * (tau_vb * ((Y - mat_z_mu - mat_x_m1).T
                                                                  @ Z[:, i]))
            mat_z_mu = mat_z_mu + np.outer(Z[:, i], alpha_vb[i, :])
        mat_z_mu = np.asfortranarray(mat_z_mu)

        shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
        c_oracle.coreZLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                           log_tau_vb, beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb,
                           tau_vb, shuffled_ind, c)                     # :108-110
        rs_gam = gam_vb.sum(axis=1)

        m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)            # :204-205
        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
        a_vb = update_a_vb(a, rs_gam, c)
        b_vb = update_b_vb(b, d, rs_gam, c)
        om_vb = a_vb / (a_vb + b_vb)
        sum_gam = rs_gam.sum()

        if annealing:                                                   # :214-228
            c = float(ladder[it]) if it < len(ladder) else 1.0
            if np.isclose(c, 1.0, rtol=1.5e-8, atol=0):
                annealing = False
        else:
            lb_new = elbo_z_(Y, Z, a, a_vb, b, b_vb, beta_vb, eta, gam_vb, kappa, lam, alpha_vb,
                             nu, phi, phi_vb, sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, tau_vb,
                             xi, zeta2_inv_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu, sum_gam)
            if trace is not None:
                trace.append((it, float(lb_new)))
            if debug and lb_new + EPS_SQRT < lb_old:                    # :238-239
                raise RuntimeError("ELBO not increasing monotonically. Exit. ")
            converged = abs(lb_new - lb_old) < tol

    return dict(beta_vb=beta_vb, gam_vb=gam_vb, om_vb=om_vb, alpha_vb=alpha_vb,
                converged=converged, it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old),
                mu_beta_vb=mu_beta_vb, mat_x_m1=mat_x_m1, mat_z_mu=mat_z_mu,
                sig2_alpha_vb=sig2_alpha_vb, sig2_beta_vb=sig2_beta_vb, tau_vb=tau_vb,
                zeta2_inv_vb=zeta2_inv_vb, sig2_inv_vb=sig2_inv_vb)


# --------------------------------------------------------------------------- #
# R/locus_mix_core.R (identity link for some responses, probit for the others)
# --------------------------------------------------------------------------- #
def elbo_mix_(Y_bin, Y_cont, ind_bin, X, Z, a, a_vb, b, b_vb, beta_vb, eta, gam_vb, kappa, lam,
              alpha_vb, nu, phi, phi_vb, sig2_alpha_vb, sig2_beta_vb, sig2_inv_vb, tau_vb,
              log_tau_vb, xi, zeta2_inv_vb, m2_alpha, m2_beta, mat_x_m1, mat_z_mu, sum_gam):
    """R/locus_mix_core.R:287-349; ind_bin 0-based, `cont` its complement."""
    n, q = Z.shape
    d = gam_vb.shape[1]
    cont = np.setdiff1d(np.arange(d), ind_bin)
    xi_vb = update_xi_z_vb_(xi, tau_vb, m2_alpha)
    eta_vb = update_eta_z_vb_(n, q, eta, gam_vb[:, cont])
    kappa_vb = (update_kappa_z_vb_(Y_cont, kappa, alpha_vb[:, cont], beta_vb[:, cont],
                                   m2_alpha[:, cont], m2_beta[:, cont], mat_x_m1[:, cont],
                                   mat_z_mu[:, cont], sig2_inv_vb, zeta2_inv_vb)
                + (m2_alpha[0, cont] - alpha_vb[0, cont] ** 2) / 2)     # intercept = TRUE
    lambda_vb = update_lambda_vb_(lam, sum_gam)
    nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
    log_tau_vb = log_tau_vb.copy()
    log_tau_vb[cont] = digamma(eta_vb) - np.log(kappa_vb)
    log_zeta2_inv_vb = digamma(phi_vb) - np.log(xi_vb)
    log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
    log_om_vb = digamma(a_vb) - digamma(a_vb + b_vb)
    log_1_min_om_vb = digamma(b_vb) - digamma(a_vb + b_vb)
    elbo_A_cont = (e_y_(n, kappa, kappa_vb, log_tau_vb[cont], m2_beta[:, cont], sig2_inv_vb,
                        tau_vb[cont])
                   + np.sum(tau_vb[cont] * (m2_alpha[:, cont].T @ zeta2_inv_vb) / 2))
    U = mat_x_m1[:, ind_bin] + mat_z_mu[:, ind_bin]                     # e_y_probit_, sweep = FALSE
    elbo_A_bin = np.sum(Y_bin * log_ndtr(U) + (1 - Y_bin) * log_ndtr(-U)
                        - (X ** 2) @ (m2_beta[:, ind_bin] - beta_vb[:, ind_bin] ** 2) / 2
                        - (Z ** 2) @ sig2_alpha_vb[:, ind_bin] / 2)
    elbo_B = e_beta_gamma_(gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                           m2_beta, sig2_beta_vb, sig2_inv_vb, tau_vb)
    elbo_C = e_tau_(eta, eta_vb, kappa, kappa_vb, log_tau_vb[cont], tau_vb[cont])
    elbo_D = e_sig2_inv_(lam, lambda_vb, log_sig2_inv_vb, nu, nu_vb, sig2_inv_vb)
    elbo_E = e_omega_(a, a_vb, b, b_vb, log_om_vb, log_1_min_om_vb)
    elbo_F = e_alpha_(m2_alpha, log_tau_vb, log_zeta2_inv_vb, sig2_alpha_vb, tau_vb,
                      zeta2_inv_vb)
    elbo_G = e_zeta2_inv_(log_zeta2_inv_vb, phi, phi_vb, xi, xi_vb, zeta2_inv_vb)
    return elbo_A_cont + elbo_A_bin + elbo_B + elbo_C + elbo_D + elbo_E + elbo_F + elbo_G


def locus_mix_core_(Y, X, Z, ind_bin, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                    sig2_beta_vb, tau_vb, tol, maxit, perm_fn, alpha_perm_fn=None, debug=False,
                    trace=None):
    """R/locus_mix_core.R:8-280, scheme batch = "y".  ind_bin: 0-based indices of the binary (0/1)
    responses; the continuous ones are centred; Z[:, 0] is the intercept; eta, kappa have one
    entry per continuous response; tau_vb has length d with 1 at the binary responses."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    Z = np.asfortranarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    ind_bin = np.asarray(ind_bin, dtype=np.int64)
    cont = np.setdiff1d(np.arange(d), ind_bin)
    a, b = list_hyper["a"], list_hyper["b"]
    eta, kappa = list_hyper["eta"], list_hyper["kappa"]
    lam, nu = list_hyper["lambda"], list_hyper["nu"]
    phi = np.atleast_1d(np.asarray(list_hyper["phi"], dtype=np.float64))
    xi = np.atleast_1d(np.asarray(list_hyper["xi"], dtype=np.float64))
    gam_vb = np.asfortranarray(gam_vb, dtype=np.float64).copy(order="F")
    mu_beta_vb = np.asfortranarray(mu_beta_vb, dtype=np.float64).copy(order="F")
    alpha_vb = np.array(alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_alpha_vb = np.array(sig2_alpha_vb, dtype=np.float64).reshape(q, d)
    sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
    tau_vb = np.array(tau_vb, dtype=np.float64)

    W = Y.copy(order="F")
    Y_bin, Y_cont = Y[:, ind_bin], Y[:, cont]
    m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)                # :27
    beta_vb = update_beta_vb_(gam_vb, mu_beta_vb)
    m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
    mat_x_m1 = np.asfortranarray(X @ beta_vb)
    mat_z_mu = np.asfortranarray(Z @ alpha_vb)
    W[:, ind_bin] = update_W_probit_(Y_bin, mat_z_mu[:, ind_bin], mat_x_m1[:, ind_bin])
    phi_vb = update_phi_z_vb_(phi, d)
    rs_gam = gam_vb.sum(axis=1)
    sum_gam = rs_gam.sum()
    digam_sum = digamma(a + b + d)
    log_tau_vb = np.zeros(d)
    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    om_vb = None
    den_add = np.zeros(q)
    den_add[0] = 1.0                                                    # intercept = TRUE

    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        xi_vb = update_xi_z_vb_(xi, tau_vb, m2_alpha)                   # :62-64
        zeta2_inv_vb = phi_vb / xi_vb
        lambda_vb = update_lambda_vb_(lam, sum_gam)
        nu_vb = update_nu_vb_(nu, m2_beta, tau_vb)
        sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = update_eta_z_vb_(n, q, eta, gam_vb[:, cont])           # :74-87
        kappa_vb = (update_kappa_z_vb_(Y_cont, kappa, alpha_vb[:, cont], beta_vb[:, cont],
                                       m2_alpha[:, cont], m2_beta[:, cont], mat_x_m1[:, cont],
                                       mat_z_mu[:, cont], sig2_inv_vb, zeta2_inv_vb)
                    + (m2_alpha[0, cont] - alpha_vb[0, cont] ** 2) / 2)
        tau_vb[cont] = eta_vb / kappa_vb
        sig2_alpha_vb = 1 / np.outer(n - 1 + zeta2_inv_vb + den_add, tau_vb)   # :90
        sig2_beta_vb = update_sig2_beta_vb_(n, sig2_inv_vb, tau_vb)
        log_tau_vb[cont] = update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb = update_log_om_vb(a, digam_sum, rs_gam)              # :101-102
        log_1_min_om_vb = update_log_1_min_om_vb(b, d, digam_sum, rs_gam)

        order = range(q) if alpha_perm_fn is None else alpha_perm_fn(it)
        for i in order:                                                 # :104-111
            mat_z_mu = mat_z_mu - np.outer(Z[:, i], alpha_vb[i, :])
            alpha_vb[i, :] = sig2_alpha_vb[i, :] * (tau_vb * ((W - mat_z_mu - mat_x_m1).T @ Z[:, i]))
            mat_z_mu = mat_z_mu + np.outer(Z[:, i], alpha_vb[i, :])
        mat_z_mu = np.asfortranarray(mat_z_mu)

        shuffled_ind = np.ascontiguousarray(perm_fn(it), dtype=np.float64)
        c_oracle.coreZLoop(X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                           log_tau_vb, beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb,
                           tau_vb, shuffled_ind)                        # :116-118
        rs_gam = gam_vb.sum(axis=1)

        m2_alpha = update_m2_alpha_(alpha_vb, sig2_alpha_vb)            # :215-221
        m2_beta = update_m2_beta_(gam_vb, mu_beta_vb, sig2_beta_vb, sweep=True)
        W[:, ind_bin] = update_W_probit_(Y_bin, mat_z_mu[:, ind_bin], mat_x_m1[:, ind_bin])
        a_vb = update_a_vb(a, rs_gam)
        b_vb = update_b_vb(b, d, rs_gam)
        om_vb = a_vb / (a_vb + b_vb)
        sum_gam = rs_gam.sum()
        lb_new = elbo_mix_(Y_bin, Y_cont, ind_bin, X, Z, a, a_vb, b, b_vb, beta_vb, eta, gam_vb,
                           kappa, lam, alpha_vb, nu, phi, phi_vb, sig2_alpha_vb, sig2_beta_vb,
                           sig2_inv_vb, tau_vb, log_tau_vb, xi, zeta2_inv_vb, m2_alpha, m2_beta,
                           mat_x_m1, mat_z_mu, sum_gam)
        if trace is not None:
            trace.append((it, float(lb_new)))
        if debug and lb_new < lb_old:
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol

    return dict(beta_vb=beta_vb, gam_vb=gam_vb, om_vb=om_vb, alpha_vb=alpha_vb,
                converged=converged, it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old),
                mu_beta_vb=mu_beta_vb, mat_x_m1=mat_x_m1, mat_z_mu=mat_z_mu,
                sig2_alpha_vb=sig2_alpha_vb, sig2_beta_vb=sig2_beta_vb, tau_vb=tau_vb,
                sig2_inv_vb=sig2_inv_vb, zeta2_inv_vb=zeta2_inv_vb)


# --------------------------------------------------------------------------- #
# Second, independent restatement of one sweep -- used only to cross-check the C
# oracle.  It is taken from the scalar R formulas of the "batch 0" scheme
# (R/locus_core.R:144-155), i.e. a different operand association from the C++
# (constants summed as -log_tau/2 - log_sig2_inv/2 - ... instead of `cst`).
# --------------------------------------------------------------------------- #
def sweep_identity_scalar_R(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb,
                            log_tau_vb, beta_vb, mat_x_m1, mu_beta_vb, sig2_beta_vb, tau_vb,
                            shuffled_ind, c=1.0):
    for jf in shuffled_ind:
        j = int(jf)
        xj = X[:, j]
        mat_x_m1 -= np.outer(xj, beta_vb[j, :])                         # :144
        mu = c * sig2_beta_vb * tau_vb * ((Y - mat_x_m1).T @ xj)        # :146
        mu_beta_vb[j, :] = mu
        gam_vb[j, :] = np.exp(-log_one_plus_exp_(c * (
            log_1_min_om_vb[j] - log_om_vb[j] - log_tau_vb / 2 - log_sig2_inv_vb / 2
            - mu ** 2 / (2 * sig2_beta_vb) - np.log(sig2_beta_vb) / 2)))  # :148-151
        beta_vb[j, :] = mu_beta_vb[j, :] * gam_vb[j, :]                 # :153
        mat_x_m1 += np.outer(xj, beta_vb[j, :])                         # :155


# --------------------------------------------------------------------------- #
# R/cross_validate.R
# --------------------------------------------------------------------------- #
def cross_validate_(Y, X, Z, list_cv, set_hyper_init, perm_fn, link="identity",
                    alpha_perm_fn=None, ind_bin=None):
    """R/cross_validate.R:177-418 for link = "identity" / "probit" / "mix" (full matrices, folds
    one after the other; ind_bin 0-based).  `set_hyper_init(Y_tr, p, pg, q)` returns (list_hyper, list_init) -- the defaults of
    R/set_hyper_init.R, whose random draws the caller controls.  Returns (p0_av_opt, lb_mat)."""
    n, d = Y.shape
    grid = np.asarray(list_cv["p0_av_grid"], dtype=np.float64)
    n_folds = list_cv["n_folds"]
    folds = np.arange(n) % n_folds                                       # :189
    lb_mat = np.zeros((n_folds, len(grid)))

    def scale_pair(A_tr, A_test):                                        # :200-209
        with np.errstate(invalid="ignore", divide="ignore"):
            S = (A_tr - A_tr.mean(axis=0)) / A_tr.std(axis=0, ddof=1)
        keep = ~np.isnan(S.sum(axis=0))
        return S[:, keep], A_test[:, keep]

    for k in range(n_folds):
        cur = folds == k
        X_tr, X_test = scale_pair(X[~cur], X[cur])
        p = X_tr.shape[1]
        if link == "identity":
            Y_tr = Y[~cur] - Y[~cur].mean(axis=0)                        # :214-215
            Y_test = Y[cur] - Y[cur].mean(axis=0)
        elif link == "mix":                                              # :217-222
            cont = np.setdiff1d(np.arange(d), ind_bin)
            Y_tr, Y_test = Y[~cur].copy(), Y[cur].copy()
            Y_tr[:, cont] -= Y_tr[:, cont].mean(axis=0)
            Y_test[:, cont] -= Y_test[:, cont].mean(axis=0)
        else:
            Y_tr, Y_test = Y[~cur], Y[cur]
        q = None
        if Z is not None:
            Z_tr, Z_test = scale_pair(Z[~cur], Z[cur])
            q = Z_tr.shape[1]
        for g, pg in enumerate(grid):
            hy, li = set_hyper_init(Y_tr, p, pg, q)
            kw = dict(tol=list_cv["tol_cv"], maxit=list_cv["maxit_cv"], perm_fn=perm_fn)
            a, b = np.broadcast_to(hy["a"], (p,)), np.broadcast_to(hy["b"], (p,))
            if link == "mix":                                            # :258-300, :363-383
                one_tr, one_te = np.ones((X_tr.shape[0], 1)), np.ones((X_test.shape[0], 1))
                if q is None:
                    Zi_tr, Zi_te = one_tr, one_te
                    phi = xi = np.array([1e-3])
                    al0, s2a0 = np.zeros((1, d)), np.ones((1, d))
                else:
                    Zi_tr, Zi_te = np.hstack([one_tr, Z_tr]), np.hstack([one_te, Z_test])
                    phi = np.concatenate([[1e-3], np.broadcast_to(hy["phi"], (q,))])
                    xi = np.concatenate([[1e-3], np.broadcast_to(hy["xi"], (q,))])
                    al0 = np.vstack([np.zeros((1, d)), li["alpha_vb"]])
                    s2a0 = np.vstack([np.ones((1, d)), li["sig2_alpha_vb"]])
                hy = dict(hy, phi=phi, xi=xi)
                vb = locus_mix_core_(Y_tr, X_tr, Zi_tr, ind_bin, hy, li["gam_vb"].copy(), al0,
                                     li["mu_beta_vb"].copy(), s2a0, li["sig2_beta_vb"],
                                     li["tau_vb"], alpha_perm_fn=alpha_perm_fn, **kw)
                rs = vb["gam_vb"].sum(axis=1)
                m2_beta = update_m2_beta_(vb["gam_vb"], vb["mu_beta_vb"], vb["sig2_beta_vb"],
                                          sweep=True)
                m2_alpha = update_m2_alpha_(vb["alpha_vb"], vb["sig2_alpha_vb"])
                lb_mat[k, g] = elbo_mix_(
                    Y_test[:, ind_bin], Y_test[:, cont], ind_bin, X_test, Zi_te, a,
                    update_a_vb(a, rs), b, update_b_vb(b, d, rs), vb["beta_vb"], hy["eta"],
                    vb["gam_vb"], hy["kappa"], hy["lambda"], vb["alpha_vb"], hy["nu"], phi,
                    update_phi_z_vb_(phi, d), vb["sig2_alpha_vb"], vb["sig2_beta_vb"],
                    vb["sig2_inv_vb"], vb["tau_vb"], np.zeros(d), xi, vb["zeta2_inv_vb"], m2_alpha,
                    m2_beta, X_test @ vb["beta_vb"], Zi_te @ vb["alpha_vb"],
                    float(vb["gam_vb"].sum()))
            elif link == "probit":                                       # :258-300, :345-361
                one_tr, one_te = np.ones((X_tr.shape[0], 1)), np.ones((X_test.shape[0], 1))
                if q is None:
                    Zi_tr, Zi_te = one_tr, one_te
                    phi = xi = np.array([1e-3])
                    al0, s2a0 = np.zeros((1, d)), np.array([1.0])
                else:
                    Zi_tr, Zi_te = np.hstack([one_tr, Z_tr]), np.hstack([one_te, Z_test])
                    phi = np.concatenate([[1e-3], np.broadcast_to(hy["phi"], (q,))])
                    xi = np.concatenate([[1e-3], np.broadcast_to(hy["xi"], (q,))])
                    al0 = np.vstack([np.zeros((1, d)), li["alpha_vb"]])
                    s2a0 = np.concatenate([[1.0], np.atleast_1d(li["sig2_alpha_vb"])])
                hy = dict(hy, phi=phi, xi=xi)
                vb = locus_probit_core_(Y_tr, X_tr, Zi_tr, hy, li["gam_vb"].copy(), al0,
                                        li["mu_beta_vb"].copy(), s2a0, li["sig2_beta_vb"],
                                        alpha_perm_fn=alpha_perm_fn, **kw)
                rs = vb["gam_vb"].sum(axis=1)
                m2_beta = update_m2_beta_(vb["gam_vb"], vb["mu_beta_vb"], vb["sig2_beta_vb"])
                m2_alpha = update_m2_alpha_(vb["alpha_vb"], vb["sig2_alpha_vb"], sweep=True)
                lb_mat[k, g] = elbo_probit_(
                    Y_test, X_test, Zi_te, a, update_a_vb(a, rs), b, update_b_vb(b, d, rs),
                    vb["beta_vb"], vb["gam_vb"], hy["lambda"], hy["nu"], phi,
                    update_phi_z_vb_(phi, d), vb["sig2_alpha_vb"], vb["sig2_beta_vb"],
                    vb["sig2_inv_vb"], xi, vb["zeta2_inv_vb"], vb["alpha_vb"], m2_alpha, m2_beta,
                    X_test @ vb["beta_vb"], Zi_te @ vb["alpha_vb"], float(vb["gam_vb"].sum()))
            elif q is None:                                              # :303-322
                vb = locus_core_(Y_tr, X_tr, hy, li["gam_vb"].copy(), li["mu_beta_vb"].copy(),
                                 li["sig2_beta_vb"], li["tau_vb"], full_output=True, **kw)
                rs = vb["gam_vb"].sum(axis=1)
                lb_mat[k, g] = elbo_(
                    Y_test, a, update_a_vb(a, rs), b, update_b_vb(b, d, rs), vb["beta_vb"],
                    hy["eta"], vb["gam_vb"], hy["kappa"], hy["lambda"], hy["nu"],
                    vb["sig2_beta_vb"], vb["sig2_inv_vb"], vb["tau_vb"], vb["m2_beta"],
                    X_test @ vb["beta_vb"], vb["sum_gam"])
            else:                                                        # :323-343
                vb = locus_z_core_(Y_tr, X_tr, Z_tr, hy, li["gam_vb"].copy(), li["alpha_vb"].copy(),
                                   li["mu_beta_vb"].copy(), li["sig2_alpha_vb"], li["sig2_beta_vb"],
                                   li["tau_vb"], **kw)
                rs = vb["gam_vb"].sum(axis=1)
                m2_beta = update_m2_beta_(vb["gam_vb"], vb["mu_beta_vb"], vb["sig2_beta_vb"],
                                          sweep=True)
                m2_alpha = update_m2_alpha_(vb["alpha_vb"], vb["sig2_alpha_vb"])
                phi = np.broadcast_to(np.asarray(hy["phi"], dtype=np.float64), (q,))
                xi = np.broadcast_to(np.asarray(hy["xi"], dtype=np.float64), (q,))
                lb_mat[k, g] = elbo_z_(
                    Y_test, Z_test, a, update_a_vb(a, rs), b, update_b_vb(b, d, rs), vb["beta_vb"],
                    hy["eta"], vb["gam_vb"], hy["kappa"], hy["lambda"], vb["alpha_vb"], hy["nu"],
                    phi, update_phi_z_vb_(phi, d), vb["sig2_alpha_vb"], vb["sig2_beta_vb"],
                    vb["sig2_inv_vb"], vb["tau_vb"], xi, vb["zeta2_inv_vb"], m2_alpha, m2_beta,
                    X_test @ vb["beta_vb"], Z_test @ vb["alpha_vb"], float(vb["gam_vb"].sum()))
    return float(grid[int(np.argmax(lb_mat.mean(axis=0)))]), lb_mat
