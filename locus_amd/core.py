"""Host-array front end of the eight native routines, named and ordered like the
reference's R wrappers (R/RcppExports.R:4-34).

Like the reference (src/coreLoop.cpp:21,26-28) the functions update `gam_vb`,
`beta_vb`, `mat_x_m1` and `mu_beta_vb` IN PLACE and return None; matrices must be
float64 and Fortran-ordered (R's column-major layout, src/locus_types.h:8-13).
Shapes are taken from X and Y (n = nrow(X), p = ncol(X), d = ncol(Y)).

Each call copies its operands to the GPU, runs the HIP kernels and copies the four
in/out matrices back (stateless, like a `.Call`).  The drivers in this package keep
the state on the device instead (locus_amd.device).
"""
import ctypes

import numpy as np

from . import _lib


def _m(a, shape, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError(f"{name}: float64 numpy array required")
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {a.shape}, expected {tuple(shape)}")
    if a.ndim == 2 and not a.flags.f_contiguous:
        raise ValueError(f"{name}: must be Fortran-ordered (column-major)")
    if a.ndim == 1 and not a.flags.c_contiguous:
        raise ValueError(f"{name}: must be contiguous")
    return ctypes.c_void_p(a.ctypes.data)


def _dims(X, Y):
    if X.ndim != 2 or Y.ndim != 2 or X.shape[0] != Y.shape[0]:
        raise ValueError("X must be n x p and Y n x d")
    return X.shape[0], X.shape[1], Y.shape[1]


def coreLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb, beta_vb,
             mat_x_m1, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, c=1.0):
    """src/coreLoop.cpp:19-54."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_loop(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(log_tau_vb, (d,), "log_tau_vb"), _m(beta_vb, (p, d), "beta_vb"),
        _m(mat_x_m1, (n, d), "mat_x_m1"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        _m(sig2_beta_vb, (d,), "sig2_beta_vb"), _m(tau_vb, (d,), "tau_vb"),
        _m(shuffled_ind, (p,), "shuffled_ind"), float(c), n, p, d))


def coreZLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb, beta_vb,
              mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, c=1.0):
    """src/coreLoop.cpp:60-96."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_z_loop(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(log_tau_vb, (d,), "log_tau_vb"), _m(beta_vb, (p, d), "beta_vb"),
        _m(mat_x_m1, (n, d), "mat_x_m1"), _m(mat_z_mu, (n, d), "mat_z_mu"),
        _m(mu_beta_vb, (p, d), "mu_beta_vb"), _m(sig2_beta_vb, (d,), "sig2_beta_vb"),
        _m(tau_vb, (d,), "tau_vb"), _m(shuffled_ind, (p,), "shuffled_ind"), float(c), n, p, d))


def coreLogitLoop(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, beta_vb, mat_x_m1,
                  mat_z_mu, mu_beta_vb, psi_vb, sig2_beta_vb, shuffled_ind):
    """src/coreLoop.cpp:101-133."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_logit_loop(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(beta_vb, (p, d), "beta_vb"), _m(mat_x_m1, (n, d), "mat_x_m1"),
        _m(mat_z_mu, (n, d), "mat_z_mu"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        _m(psi_vb, (n, d), "psi_vb"), _m(sig2_beta_vb, (p, d), "sig2_beta_vb"),
        _m(shuffled_ind, (p,), "shuffled_ind"), n, p, d))


def coreProbitLoop(X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, beta_vb, mat_x_m1,
                   mat_z_mu, mu_beta_vb, sig2_beta_vb, shuffled_ind):
    """src/coreLoop.cpp:139-171."""
    n, p, d = _dims(X, W)
    _lib.check(_lib.lib().locus_hip_core_probit_loop(
        _m(X, (n, p), "X"), _m(W, (n, d), "W"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(beta_vb, (p, d), "beta_vb"), _m(mat_x_m1, (n, d), "mat_x_m1"),
        _m(mat_z_mu, (n, d), "mat_z_mu"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        float(sig2_beta_vb), _m(shuffled_ind, (p,), "shuffled_ind"), n, p, d))


def coreStructLoop(X, Y, gam_vb, log_Phi_mu_theta_vb, log_1_min_Phi_mu_theta_vb, log_sig2_inv_vb,
                   log_tau_vb, beta_vb, mat_x_m1, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind):
    """src/coreStructLoop.cpp:18-52."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_struct_loop(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_Phi_mu_theta_vb, (p,), "log_Phi_mu_theta_vb"),
        _m(log_1_min_Phi_mu_theta_vb, (p,), "log_1_min_Phi_mu_theta_vb"),
        float(log_sig2_inv_vb), _m(log_tau_vb, (d,), "log_tau_vb"), _m(beta_vb, (p, d), "beta_vb"),
        _m(mat_x_m1, (n, d), "mat_x_m1"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        _m(sig2_beta_vb, (d,), "sig2_beta_vb"), _m(tau_vb, (d,), "tau_vb"),
        _m(shuffled_ind, (p,), "shuffled_ind"), n, p, d))


def coreBatch(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb, beta_vb,
              mat_x_m1, mu_beta_vb, sig2_beta_vb, tau_vb):
    """src/coreBatch.cpp:19-42."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_batch(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(log_tau_vb, (d,), "log_tau_vb"), _m(beta_vb, (p, d), "beta_vb"),
        _m(mat_x_m1, (n, d), "mat_x_m1"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        _m(sig2_beta_vb, (d,), "sig2_beta_vb"), _m(tau_vb, (d,), "tau_vb"), n, p, d))


def coreZBatch(X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb, beta_vb,
               mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb, tau_vb):
    """src/coreBatch.cpp:48-73."""
    n, p, d = _dims(X, Y)
    _lib.check(_lib.lib().locus_hip_core_z_batch(
        _m(X, (n, p), "X"), _m(Y, (n, d), "Y"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(log_tau_vb, (d,), "log_tau_vb"), _m(beta_vb, (p, d), "beta_vb"),
        _m(mat_x_m1, (n, d), "mat_x_m1"), _m(mat_z_mu, (n, d), "mat_z_mu"),
        _m(mu_beta_vb, (p, d), "mu_beta_vb"), _m(sig2_beta_vb, (d,), "sig2_beta_vb"),
        _m(tau_vb, (d,), "tau_vb"), n, p, d))


def coreProbitBatch(X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, beta_vb, mat_x_m1,
                    mat_z_mu, mu_beta_vb, sig2_beta_vb):
    """src/coreBatch.cpp:78-100."""
    n, p, d = _dims(X, W)
    _lib.check(_lib.lib().locus_hip_core_probit_batch(
        _m(X, (n, p), "X"), _m(W, (n, d), "W"), _m(gam_vb, (p, d), "gam_vb"),
        _m(log_om_vb, (p,), "log_om_vb"), _m(log_1_min_om_vb, (p,), "log_1_min_om_vb"),
        float(log_sig2_inv_vb), _m(beta_vb, (p, d), "beta_vb"), _m(mat_x_m1, (n, d), "mat_x_m1"),
        _m(mat_z_mu, (n, d), "mat_z_mu"), _m(mu_beta_vb, (p, d), "mu_beta_vb"),
        float(sig2_beta_vb), n, p, d))
