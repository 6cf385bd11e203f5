"""Synthetic SNP / molecular-trait data of the shape the reference's examples and tests
use (R/locus.R:152-187, tests/testthat/main.R:3-37), for tests and bench.py.

numpy's Generator replaces R's RNG, so the draws differ from R's for the same seed;
the distributions and the preprocessing are the reference's.
"""
import numpy as np


def simulate(n, p, d, p0, seed=123, link="identity", d0_frac=0.8, frac_assoc=0.2, sd_noise=0.5,
             maf=0.25, dtype=np.float64):
    """Return dict(X, Y, active) with X standardised and Y centred (identity), shifted by
    -1/2 (logit) or left 0/1 (probit), exactly as prepare_data_ leaves them
    (R/prepare_locus.R:109-141)."""
    rng = np.random.default_rng(seed)
    X_act = None
    # SNPs ~ Binomial(2, maf); redraw constant columns (the reference drops them,
    # R/prepare_locus.R:111-117; here the shape is kept fixed instead)
    X = rng.binomial(2, maf, size=(n, p)).astype(dtype)
    for _ in range(20):
        const = X.std(axis=0) == 0
        if not const.any():
            break
        X[:, const] = rng.binomial(2, maf, size=(n, int(const.sum())))
    d0 = max(1, int(np.floor(d0_frac * d)))
    ind_p0 = np.sort(rng.choice(p, size=p0, replace=False))
    ind_d0 = np.sort(rng.choice(d, size=d0, replace=False))
    beta = np.zeros((p0, d0))
    mask = rng.random((p0, d0)) < frac_assoc
    for r in range(p0):  # each active SNP is associated with at least one response
        if not mask[r].any():
            mask[r, rng.integers(d0)] = True
    beta[mask] = rng.normal(size=int(mask.sum()))
    X_act = X[:, ind_p0]
    lin = np.zeros((n, d))
    lin[:, ind_d0] = X_act @ beta
    Yc = lin + rng.normal(scale=sd_noise, size=(n, d))
    # scale(X): centre, divide by the (n-1)-denominator standard deviation
    X = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
    if link == "identity":
        Y = Yc - Yc.mean(axis=0)
    elif link == "logit":
        Y = (Yc > 0).astype(dtype) - 0.5
    elif link == "probit":
        Y = (Yc > 0).astype(dtype)
    else:
        raise ValueError(f"unknown link {link!r}")
    active = np.zeros((p, d), dtype=bool)
    act = np.zeros((p0, d), dtype=bool)
    act[:, ind_d0] = mask
    active[ind_p0, :] = act
    return dict(X=np.asfortranarray(X), Y=np.asfortranarray(Y), active=active,
                ind_p0=ind_p0, ind_d0=ind_d0)


def sweep_inputs(n, p, d, seed=0, variant="identity", c=1.0):
    """Random but well-scaled arguments for ONE call of a native sweep routine, laid out as
    R passes them.  Used by the kernel-level parity tests (the drivers build their own)."""
    rng = np.random.default_rng(seed)
    X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float64)
    X[0, X.std(axis=0) == 0] += 1.0
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    gam = np.asfortranarray(rng.beta(1.0, 8.0, size=(p, d)))
    mu = np.asfortranarray(rng.normal(scale=0.3, size=(p, d)))
    beta = np.asfortranarray(gam * mu)
    M = np.asfortranarray(X @ beta)
    lin = M + rng.normal(scale=0.7, size=(n, d))
    om = rng.beta(2.0, 20.0, size=p)
    out = dict(X=X, gam_vb=gam, mu_beta_vb=mu, beta_vb=beta, mat_x_m1=M,
               log_om_vb=np.log(om), log_1_min_om_vb=np.log1p(-om),
               log_sig2_inv_vb=float(np.log(0.7)),
               shuffled_ind=rng.permutation(p).astype(np.float64), c=float(c))
    if variant in ("identity", "z", "struct"):
        tau = rng.gamma(5.0, 0.4, size=d)
        out.update(Y=np.asfortranarray(lin - lin.mean(axis=0)), tau_vb=tau,
                   log_tau_vb=np.log(tau) - 0.01,
                   sig2_beta_vb=1.0 / (c * (n - 1 + 0.7) * tau))
        if variant == "z":
            out["mat_z_mu"] = np.asfortranarray(rng.normal(scale=0.2, size=(n, d)))
        if variant == "struct":
            theta = rng.normal(-1.5, 0.5, size=p)
            from scipy.special import log_ndtr
            out["log_om_vb"] = log_ndtr(theta)
            out["log_1_min_om_vb"] = log_ndtr(-theta)
    elif variant == "probit":
        out.update(Y=np.asfortranarray(lin), sig2_beta_vb=1.0 / (n - 1 + 0.7),
                   mat_z_mu=np.asfortranarray(np.outer(np.ones(n), rng.normal(scale=0.2, size=d))))
    elif variant == "logit":
        chi = np.abs(rng.normal(1.0, 0.5, size=(n, d))) + 0.05
        psi = (1.0 / (1.0 + np.exp(-chi)) - 0.5) / (2.0 * chi)
        out.update(Y=np.asfortranarray((lin > 0).astype(np.float64) - 0.5),
                   psi_vb=np.asfortranarray(psi),
                   sig2_beta_vb=np.asfortranarray(1.0 / (2.0 * (X ** 2).T @ psi + 0.7)),
                   mat_z_mu=np.asfortranarray(np.outer(np.ones(n), rng.normal(scale=0.2, size=d))))
    else:
        raise ValueError(variant)
    return out
