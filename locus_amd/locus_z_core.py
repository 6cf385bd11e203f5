"""Identity-link CAVI driver with fixed covariates Z, device-resident (mirrors R/locus_z_core.R,
scheme batch = "y": the alpha loop :96-103, then coreZLoop :108-110).

Same split as locus_core.py.  What Z adds:
  * the residual the sweep works on is Y - mat_x_m1 - mat_z_mu (src/coreLoop.cpp:78-94 differs from
    coreLoop only there), so the sweep kernel and its fused statistics are reused unchanged:
    colSums(Y^2) - 2 colSums(Y (M + Zmu)) + colSums(M^2) + colSums(Zmu^2) + 2 colSums(M Zmu) of
    update_kappa_z_vb_ (R/update_vb.R:260-266) is the residual sum of squares;
  * the q covariate effects alpha_vb (q x d) are updated one coordinate at a time on the device
    (`alpha_step`: fused residual dot + mat_z_mu update), q is small;
  * one more exchange per iteration: m2_alpha %*% tau_vb (length q) rides in the same all-reduce.
"""
import numpy as np
from scipy.special import digamma

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F
from .locus_core import (EPS_SQRT, IdentityEngine, IdentityHost, _gather, _gather_vec,
                         resolve_backend)


class ZHost(IdentityHost):
    """Host arithmetic of locus_z_core_ on reduced statistics.  Shard-local: everything of
    IdentityHost plus alpha_vb / sig2_alpha_vb / m2_alpha (q x d_loc).  Replicated: phi, xi,
    zeta2_inv_vb (length q)."""

    def __init__(self, n, p, q, d_total, hyper, tau_vb, sig2_beta_vb, alpha_vb, sig2_alpha_vb,
                 comm=None):
        super().__init__(n, p, d_total, hyper, tau_vb, sig2_beta_vb, comm)
        self.q = int(q)
        self.phi = np.broadcast_to(np.asarray(hyper["phi"], dtype=np.float64), (q,)).copy()
        self.xi = np.broadcast_to(np.asarray(hyper["xi"], dtype=np.float64), (q,)).copy()
        self.sig2_alpha_vb = np.array(sig2_alpha_vb, dtype=np.float64).reshape(q, -1)
        self.m2_alpha = U.update_m2_alpha_(np.asarray(alpha_vb, dtype=np.float64).reshape(q, -1),
                                           self.sig2_alpha_vb)                       # :32
        self.cs_a2 = (np.asarray(alpha_vb, dtype=np.float64).reshape(q, -1) ** 2).sum(axis=0)
        self.xi_sum = None          # m2_alpha %*% tau_vb over ALL responses (length q)
        self.zeta2_inv_vb = self.phi_vb = None

    def set_initial(self, stats, rs_gam_local):
        self.stats = np.asarray(stats, dtype=np.float64)
        t_loc = float(self.tau_vb @ self.stats[1])
        sc = self._reduce_p(rs_gam_local, [t_loc] + list(self.m2_alpha @ self.tau_vb))
        self.sum_tau_m2 = float(sc[0])
        self.xi_sum = np.array(sc[1:1 + self.q])

    def _kappa_z(self, c, zeta2_inv_vb, sig2_inv_vb):
        """update_kappa_z_vb_ (R/update_vb.R:255-272, no intercept) on reduced statistics."""
        n = self.n
        cs_gam, cs_m2, cs_b2, rss = self.stats
        return c * (self.kappa + (rss + (n - 1 + sig2_inv_vb) * cs_m2 - (n - 1) * cs_b2
                                  + (n - 1) * self.m2_alpha.sum(axis=0)
                                  + self.m2_alpha.T @ zeta2_inv_vb - (n - 1) * self.cs_a2) / 2)

    # -- top of the iteration: R/locus_z_core.R:57-94 ------------------------------------------
    def begin_iteration(self, c=1.0):
        n, d, q = self.n, self.d, self.q
        cs_gam = self.stats[0]
        self.phi_vb = U.update_phi_z_vb_(self.phi, d, c)                              # :60
        xi_vb = c * (self.xi + self.xi_sum / 2)                                       # :61
        self.zeta2_inv_vb = self.phi_vb / xi_vb
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam, c)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2, c)
        self.sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = c * (self.eta + n / 2 + cs_gam / 2 + q / 2) - c + 1                  # update_eta_z_vb_
        kappa_vb = self._kappa_z(c, self.zeta2_inv_vb, self.sig2_inv_vb)
        self.tau_vb = eta_vb / kappa_vb
        # update_sig2_alpha_vb_ with tau (R/update_vb.R:32-49), q x d_loc
        self.sig2_alpha_vb = 1.0 / (c * np.outer(n - 1 + self.zeta2_inv_vb, self.tau_vb))
        self.sig2_beta_vb = U.update_sig2_beta_vb_(n, self.sig2_inv_vb, self.tau_vb, c)
        log_tau_vb = U.update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb, log_1_min_om_vb = self._log_om_terms(c)
        s2 = self.sig2_beta_vb
        return dict(
            lo=log_om_vb, l1=log_1_min_om_vb, c=float(c),
            scale=c * s2 * self.tau_vb, hinv=1.0 / (2.0 * s2),
            cst=-(log_tau_vb + log_sig2_inv_vb + np.log(s2)) / 2, s2k=s2,
            # alpha_vb[i, ] = c * sig2_alpha_vb[i, ] * (tau_vb * crossprod(...))  (:99-100)
            alpha_coef=c * self.sig2_alpha_vb * self.tau_vb[None, :])

    # -- after the sweep: :204-211 and elbo_z_ :299-342 ----------------------------------------
    def end_iteration(self, stats, rs_gam_local, ent_local, alpha_vb, c=1.0, with_elbo=True):
        n, d, q = self.n, self.d, self.q
        self.stats = np.asarray(stats, dtype=np.float64)
        self._last = (rs_gam_local, ent_local, alpha_vb)
        cs_gam, cs_m2, cs_b2, rss = self.stats
        alpha_vb = np.asarray(alpha_vb, dtype=np.float64).reshape(q, -1)
        self.m2_alpha = U.update_m2_alpha_(alpha_vb, self.sig2_alpha_vb)
        self.cs_a2 = (alpha_vb ** 2).sum(axis=0)
        tau, s2, si, zi = self.tau_vb, self.sig2_beta_vb, self.sig2_inv_vb, self.zeta2_inv_vb
        t_loc = float(tau @ cs_m2)
        xi_loc = self.m2_alpha @ tau
        loc = 0.0
        if with_elbo:
            eta_vb = self.eta + n / 2 + cs_gam / 2 + q / 2
            kappa_vb = self._kappa_z(1.0, zi, si)
            log_tau_vb = digamma(eta_vb) - np.log(kappa_vb)
            m2a_z = self.m2_alpha.T @ zi                                              # length d_loc
            loc = (E.e_y_(n, self.kappa, kappa_vb, log_tau_vb, cs_m2, si, tau)
                   + float(np.sum(tau * m2a_z / 2))                                   # R/elbo.R:195-196
                   + E.e_beta_gamma_local_(cs_gam, cs_m2, ent_local, log_tau_vb, s2, si, tau)
                   + E.e_tau_(self.eta, eta_vb, self.kappa, kappa_vb, log_tau_vb, tau)
                   # e_alpha_ (R/elbo.R:11-19) without the log zeta2_inv term (replicated)
                   + 0.5 * float(np.sum(-tau * m2a_z) + q * np.sum(log_tau_vb)
                                 + np.sum(np.log(self.sig2_alpha_vb)) + self.m2_alpha.size))
        sc = self._reduce_p(rs_gam_local, [t_loc, loc] + list(xi_loc), omega=with_elbo)
        self.sum_tau_m2 = float(sc[0])
        self.xi_sum = np.array(sc[2:2 + q])
        self._c_last = c
        if not with_elbo:
            return None
        s_bg, e_om = self._omega_terms()
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
        xi_vb = self.xi + self.xi_sum / 2                                             # :309
        log_zeta2_inv_vb = digamma(self.phi_vb) - np.log(xi_vb)
        return float(sc[1] + log_sig2_inv_vb * self.sum_gam / 2 + s_bg
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si)
                     + e_om + 0.5 * d * float(np.sum(log_zeta2_inv_vb))
                     + E.e_zeta2_inv_(log_zeta2_inv_vb, self.phi, self.phi_vb, self.xi, xi_vb, zi))


class ZEngine(IdentityEngine):
    """One GPU's shard of the identity-link model with covariates."""

    def __init__(self, px, Y_dev, Z, state, alpha_vb, host, device, cfg=None, dv=None):
        super().__init__(px, Y_dev, state, host, device, cfg=cfg, dv=dv)
        t, dv = self.torch, self.dv
        self.q = Z.shape[1]
        self.Z_host = np.asfortranarray(Z, dtype=np.float64)
        self.Zt = dv.to_device_colmajor(self.Z_host, device).t().contiguous()          # [q][n]
        self.alpha = self._up(np.ascontiguousarray(alpha_vb, dtype=np.float64).reshape(self.q, -1))
        cm = lambda: t.empty((self.d_loc, self.n), dtype=t.float64, device=device).t()  # noqa: E731
        self.M, self.Zmu = cm(), cm()

    def initialise(self):
        """R/locus_z_core.R:32-41."""
        t, dv = self.torch, self.dv
        n, p, d = self.n, self.p, self.d_loc
        cols = dv.state_cols(self.state, p, d, s2k=self._up(self.host.sig2_beta_vb))
        dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.state[2], d, 1, self.M, 1, n)
        self.Zmu.copy_(self._up((self.Z_host @ self.alpha.cpu().numpy()).T).t())
        self.R = dv.residual(self.Y, self.M, self.Zmu)
        rss = (self.R * self.R).sum(dim=0)
        rs_gam, _ = dv.gamma_rows(self.state[0], p, d)
        stats = t.cat([cols, rss.reshape(1, d)], dim=0).cpu().numpy()
        self.host.set_initial(stats, rs_gam)

    def iterate(self, shuffled_ind, c=1.0, with_elbo=True, alpha_order=None, before_readback=None):
        """`before_readback()`: host work of the caller once the kernels of this iteration are
        queued, before the host blocks on the read-back; if it returns the next sweep order, that
        sweep's X-dependent operands are packed on a side stream now (IdentityEngine._pack_ahead)."""
        dv, t = self.dv, self.torch
        n, d = self.n, self.d_loc
        prm = self.host.begin_iteration(c)
        pre, self._pre_ind = getattr(self, "_pre_ind", None), None
        if pre is not None and pre[0] is shuffled_ind:
            ind = pre[1]           # uploaded, and its operands packed, while the last sweep ran
        else:
            ind = shuffled_ind if t.is_tensor(shuffled_ind) else self._up(shuffled_ind)
        # the iteration's constants in one upload (seven small copies before)
        coef, lo, l1, scale, hinv, cst, s2k = self._up_many(
            prm["alpha_coef"], prm["lo"], prm["l1"], prm["scale"], prm["hinv"], prm["cst"], prm["s2k"])
        for i in (range(self.q) if alpha_order is None else alpha_order):            # :96-103
            dv.alpha_step(self.Y, None, self.M, self.Zmu, self.Zt[i], coef[i], self.alpha[i], n, d)
        dv.residual(self.Y, self.M, self.Zmu, out=self.R)
        self.sweeper.identity(ind, lo, l1, scale, hinv, cst, s2k, c, self.R, self.state, self.stats)
        dv.residual(self.Y, self.R, self.Zmu, out=self.M)                            # M = Y - R - Zmu
        rs_gam, ent = dv.gamma_rows(self.state[0], self.p, d,
                                    eps=self.host.ent_eps if with_elbo else None)
        tail = ent.sum().reshape(1) if with_elbo else t.zeros(1, dtype=t.float64,
                                                              device=self.device)
        h_dev = t.cat([self.stats.reshape(-1), tail, self.alpha.reshape(-1)])
        if before_readback is not None:
            self._pack_ahead(before_readback())   # the next sweep order, if the caller returns it
        h = h_dev.cpu().numpy()
        self.sweeper.status()
        stats = h[: 4 * d].reshape(4, d)
        return self.host.end_iteration(stats, rs_gam, float(h[4 * d]), h[4 * d + 1:], c, with_elbo)

    def held_out_elbo_z(self, Y_test_loc, X_test, Z_test):
        """elbo_z_ on held-out rows (R/cross_validate.R:331-343): mat_z_mu = Z_test alpha_vb."""
        Zmu = self._up((np.asarray(Z_test, dtype=np.float64) @ self.alpha.cpu().numpy()).T).t()
        return self.held_out_elbo(Y_test_loc, X_test, Zmu)

    def download(self):
        out = LazyFields(gam_vb=lambda: to_host_F(self.state[0]),
                         mu_beta_vb=lambda: to_host_F(self.state[1]),
                         beta_vb=lambda: to_host_F(self.state[2]),
                         mat_x_m1=lambda: to_host_F(self.M), mat_z_mu=lambda: to_host_F(self.Zmu),
                         alpha_vb=lambda: self.alpha.cpu().numpy().copy())
        return out


def locus_z_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb, sig2_beta_vb,
                  tau_vb, tol, maxit, anneal=None, verbose=False, batch="y", full_output=False,
                  debug=True, perm_fn=None, alpha_perm_fn=None, comm=None, device=None, trace=None,
                  held_out=None, _test_backend=None):
    """R/locus_z_core.R:8-290 on the GPU.  Y (n x d) centred, X (n x p) and Z (n x q) standardised.
    Arguments as locus_core_ plus the covariate block; `alpha_perm_fn(it)` supplies the order of
    the alpha loop (R: sample(1:q), :96), 0-based.  `held_out = (Y_test, X_test, Z_test)` adds
    `lb_held_out` (R/cross_validate.R:331-343)."""
    dv, device = resolve_backend(_test_backend, device)
    import torch
    if batch != "y":
        raise NotImplementedError("only the 'y' scheme (coreZLoop) is device-resident; "
                                  "coreZBatch is available as locus_amd.coreZBatch")
    comm = comm or Comm()
    Y = np.asarray(Y, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    k_lo, k_hi = comm.shard(d)
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731
    if alpha_perm_fn is None:
        rng_a = np.random.default_rng(1)
        alpha_perm_fn = lambda it: rng_a.permutation(q)  # noqa: E731

    hyper_loc = dict(list_hyper)
    for key in ("eta", "kappa"):
        hyper_loc[key] = np.broadcast_to(np.asarray(list_hyper[key], dtype=np.float64),
                                         (d,))[k_lo:k_hi]
    alpha_loc = np.asarray(alpha_vb, dtype=np.float64).reshape(q, d)[:, k_lo:k_hi]
    s2a_loc = np.broadcast_to(np.asarray(sig2_alpha_vb, dtype=np.float64), (q, d))[:, k_lo:k_hi]
    host = ZHost(n, p, q, d, hyper_loc, np.asarray(tau_vb)[k_lo:k_hi],
                 np.asarray(sig2_beta_vb)[k_lo:k_hi], alpha_loc, s2a_loc, comm)
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, k_lo:k_hi], device)
    state = torch.empty((3, p, k_hi - k_lo), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, slice(k_lo, k_hi), device)
    state[1] = columns_to_device(mu_beta_vb, slice(k_lo, k_hi), device)
    eng = ZEngine(px, Y_dev, Z, state, alpha_loc, host, device, dv=dv)
    eng.initialise()

    if anneal is None:
        annealing, c, ladder = False, 1.0, None
    else:
        annealing, ladder = True, U.get_annealing_ladder_(anneal)
        c = float(ladder[0])

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nxt = {}
    draw = lambda i: (np.ascontiguousarray(perm_fn(i), dtype=np.float64), alpha_perm_fn(i))  # noqa: E731
    while (not converged) and it < maxit:                                # :49
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        ind, a_order = nxt.pop(it) if it in nxt else draw(it)

        def draw_next(it=it):                 # while the GPU runs this iteration
            if it < maxit:
                nxt[it + 1] = draw(it + 1)
                return nxt[it + 1][0]         # packed ahead by the engine, beside the running sweep
        lb = eng.iterate(ind, c, with_elbo=not annealing, alpha_order=a_order,
                         before_readback=draw_next)
        if annealing:                                                    # :214-228
            c = float(ladder[it]) if it < len(ladder) else 1.0
            if np.isclose(c, 1.0, rtol=1.5e-8, atol=0):
                annealing = False
        else:
            lb_new = lb
            if trace is not None:
                trace.append((it, lb_new))
            if verbose and (it == 1 or it % 5 == 0):
                print(f"ELBO = {lb_new}\n")
            if debug and lb_new + EPS_SQRT < lb_old:                     # :238-239
                raise RuntimeError("ELBO not increasing monotonically. Exit. ")
            converged = abs(lb_new - lb_old) < tol
    if verbose and not converged:
        print("Maximal number of iterations reached before convergence. Exit.")

    loc = eng.download()
    out = dict(beta_vb=_gather(comm, loc["beta_vb"], d), gam_vb=_gather(comm, loc["gam_vb"], d),
               om_vb=host.om_vb, alpha_vb=_gather(comm, loc["alpha_vb"], d), converged=converged,
               it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old),
               annealing=False if anneal is None else anneal[0])
    if held_out is not None:
        Yt = np.asarray(held_out[0], dtype=np.float64)
        out["lb_held_out"] = eng.held_out_elbo_z(Yt[:, k_lo:k_hi], held_out[1], held_out[2])
    if full_output:                                                      # :267-274
        out.update(mu_beta_vb=_gather(comm, loc["mu_beta_vb"], d),
                   mat_x_m1=_gather(comm, loc["mat_x_m1"], d),
                   mat_z_mu=_gather(comm, loc["mat_z_mu"], d),
                   sig2_alpha_vb=_gather(comm, host.sig2_alpha_vb, d),
                   sig2_beta_vb=_gather_vec(comm, host.sig2_beta_vb, d),
                   tau_vb=_gather_vec(comm, host.tau_vb, d), sig2_inv_vb=host.sig2_inv_vb,
                   zeta2_inv_vb=host.zeta2_inv_vb, a_vb=host.a_vb, b_vb=host.b_vb,
                   sum_gam=host.sum_gam)
    return out
