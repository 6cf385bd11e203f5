"""Structured-prior CAVI driver with device-resident state (mirrors R/locus_struct_core.R).

The sweep is coreStructLoop (src/coreStructLoop.cpp:18-52) = the identity-family kernel with
c = 1 and per-covariate terms log Phi(theta_j), log(1 - Phi(theta_j)).  Everything that
involves theta (W, the block matvecs, e_theta_) needs only rowSums(gam_vb), so it is
replicated host arithmetic on the all-reduced vector.
"""
import numpy as np
from scipy.special import log_ndtr

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F
from .locus_core import HostBase, IdentityEngine, _gather, _gather_vec, resolve_backend
from .struct import StructBlocks, row_sums_W_struct_

EPS_STRUCT = np.finfo(np.float64).eps ** 0.75  # R/elbo.R:94


class StructHost(HostBase):
    """Replicated / shard-local vector arithmetic of locus_struct_core_
    (R/locus_struct_core.R:51-145).  Same interface as IdentityHost."""

    ent_eps = EPS_STRUCT

    def __init__(self, n, p, d_total, hyper, tau_vb, sig2_beta_vb, blocks, comm=None):
        self.n, self.p, self.d = int(n), int(p), int(d_total)
        self.comm = comm or Comm()
        self.eta = np.asarray(hyper["eta"], dtype=np.float64)      # shard-local
        self.kappa = np.asarray(hyper["kappa"], dtype=np.float64)  # shard-local
        self.lam, self.nu = float(hyper["lambda"]), float(hyper["nu"])
        self.m0 = np.asarray(hyper["m0"], dtype=np.float64)
        self.blocks = blocks
        self.theta_vb = self.m0.copy()                             # :25
        self.tau_vb = np.array(tau_vb, dtype=np.float64)
        self.sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
        self.sig2_inv_vb = None
        self.rs_gam = self.sum_gam = self.sum_tau_m2 = None
        self.stats = None

    def set_initial(self, stats, rs_gam_local):
        self.stats = np.asarray(stats, dtype=np.float64)
        self.rs_gam, sc = self._all_reduce(rs_gam_local, [float(self.tau_vb @ self.stats[1])])
        self.sum_gam = float(self.rs_gam.sum())
        self.sum_tau_m2 = float(sc[0])

    def begin_iteration(self, c=1.0):                                     # :60-84
        assert c == 1.0, "annealing is not available for the structured model"
        n = self.n
        cs_gam, cs_m2, cs_b2, rss = self.stats
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        self.sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = U.update_eta_vb_(n, self.eta, cs_gam)
        kappa_vb = U.update_kappa_vb_(n, self.kappa, rss, cs_m2, cs_b2, self.sig2_inv_vb)
        self.tau_vb = eta_vb / kappa_vb
        self.sig2_beta_vb = U.update_sig2_beta_vb_(n, self.sig2_inv_vb, self.tau_vb)
        log_tau_vb = U.update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        s2 = self.sig2_beta_vb
        return dict(lo=log_ndtr(self.theta_vb), l1=log_ndtr(-self.theta_vb), c=1.0,
                    scale=s2 * self.tau_vb, hinv=1.0 / (2.0 * s2),
                    cst=-(log_tau_vb + log_sig2_inv_vb + np.log(s2)) / 2, s2k=s2)

    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True):
        n, d = self.n, self.d
        self.stats = np.asarray(stats, dtype=np.float64)
        cs_gam, cs_m2, cs_b2, rss = self.stats
        tau, s2, si = self.tau_vb, self.sig2_beta_vb, self.sig2_inv_vb
        # shard-local part of elbo_struct_ (R/locus_struct_core.R:201-221)
        eta_vb = U.update_eta_vb_(n, self.eta, cs_gam)
        kappa_vb = U.update_kappa_vb_(n, self.kappa, rss, cs_m2, cs_b2, si)
        log_tau_vb = U.update_log_tau_vb_(eta_vb, kappa_vb)
        loc = (E.e_y_(n, self.kappa, kappa_vb, log_tau_vb, cs_m2, si, tau)
               + E.e_beta_gamma_local_(cs_gam, cs_m2, ent_local, log_tau_vb, s2, si, tau)
               + E.e_tau_(self.eta, eta_vb, self.kappa, kappa_vb, log_tau_vb, tau))
        self.rs_gam, sc = self._all_reduce(rs_gam_local, [float(tau @ cs_m2), loc])
        self.sum_tau_m2 = float(sc[0])
        self.sum_gam = float(self.rs_gam.sum())
        # W and theta (:127-129): only rowSums(W) enters
        rs_W = row_sums_W_struct_(self.rs_gam, d, self.theta_vb)
        self.theta_vb = self.blocks.update_theta_vb_(rs_W, self.m0)
        # replicated part of elbo_struct_, with the NEW theta_vb (:132-135)
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        glob = (log_sig2_inv_vb * self.sum_gam / 2
                + np.sum(self.rs_gam * log_ndtr(self.theta_vb))
                + np.sum((d - self.rs_gam) * log_ndtr(-self.theta_vb))
                - d * np.sum(self.blocks.diag_sig2_theta) / 2)            # R/elbo.R:98-104
        return float(sc[1] + glob + self.blocks.e_theta_(self.m0, self.theta_vb)
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si))


def locus_struct_core_(Y, X, list_hyper, gam_vb, mu_beta_vb, sig2_beta_vb, tau_vb, list_struct,
                       tol, maxit, verbose=False, batch="y", full_output=False, debug=False,
                       perm_fn=None, comm=None, device=None, trace=None, _test_backend=None):
    """R/locus_struct_core.R:8-184 on the GPU; Y centred, X standardised.  `list_struct` is
    what locus_amd.struct.set_struct returns.  See locus_core_ for perm_fn / comm."""
    dv, device = resolve_backend(_test_backend, device)
    import torch
    if batch != "y":
        raise NotImplementedError("only the 'y' scheme (coreStructLoop) is implemented")
    comm = comm or Comm()
    Y = np.asarray(Y, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    n, d = Y.shape
    p = X.shape[1]
    k_lo, k_hi = comm.shard(d)
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731

    blocks = StructBlocks(d, list_struct["vec_fac_st"], list_hyper["s02"], X)   # :30
    hyper_loc = dict(list_hyper)
    for key in ("eta", "kappa"):
        hyper_loc[key] = np.broadcast_to(np.asarray(list_hyper[key], dtype=np.float64),
                                         (d,))[k_lo:k_hi]
    host = StructHost(n, p, d, hyper_loc, np.asarray(tau_vb)[k_lo:k_hi],
                      np.asarray(sig2_beta_vb)[k_lo:k_hi], blocks, comm)
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, k_lo:k_hi], device)
    state = torch.empty((3, p, k_hi - k_lo), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, slice(k_lo, k_hi), device)
    state[1] = columns_to_device(mu_beta_vb, slice(k_lo, k_hi), device)
    eng = IdentityEngine(px, Y_dev, state, host, device, dv=dv)
    eng.initialise()

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    while (not converged) and it < maxit:                                # :51
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        lb_new = eng.iterate(np.ascontiguousarray(perm_fn(it), dtype=np.float64), 1.0)
        if trace is not None:
            trace.append((it, lb_new))
        if verbose and (it == 1 or it % 5 == 0):
            print(f"ELBO = {lb_new}\n")
        if debug and lb_new < lb_old:                                    # :140-141
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol                           # :143
    if verbose and not converged:
        print("Maximal number of iterations reached before convergence. Exit.")

    loc = eng.download()
    out = dict(beta_vb=_gather(comm, loc["beta_vb"], d), gam_vb=_gather(comm, loc["gam_vb"], d),
               theta_vb=host.theta_vb, converged=converged, it=it, lb_opt=lb_new,
               diff_lb=abs(lb_new - lb_old))
    if full_output:                                                      # :169-175
        out.update(mu_beta_vb=_gather(comm, loc["mu_beta_vb"], d),
                   mat_x_m1=_gather(comm, loc["mat_x_m1"], d),
                   sig2_beta_vb=_gather_vec(comm, host.sig2_beta_vb, d),
                   tau_vb=_gather_vec(comm, host.tau_vb, d), sig2_inv_vb=host.sig2_inv_vb)
    return out
