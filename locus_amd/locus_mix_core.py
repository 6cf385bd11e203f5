"""Mixed identity / probit CAVI driver with fixed covariates, device-resident (mirrors
R/locus_mix_core.R, scheme batch = "y").

Some responses are continuous (identity link, precision tau_k), the others binary (probit link
through the latent W, tau_k = 1, log tau_k = 0); the intercept Z[:, 0] is not scaled.  The sweep is
coreZLoop on W (R/locus_mix_core.R:116-118), i.e. the identity-family kernel with per-response
constants, so nothing new runs on the device: the shard's responses are laid out
[continuous | binary] and the probit stages (`probit_w`, the X^2 contraction of e_y_probit_) are
applied to the binary column range.
"""
import numpy as np
from scipy.special import digamma

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F
from .locus_core import IdentityEngine, IdentityHost, _gather, _gather_vec, resolve_backend


class MixHost(IdentityHost):
    """Host arithmetic of locus_mix_core_ on reduced statistics; local responses are ordered
    [dc continuous | db binary].  eta, kappa: length dc."""

    def __init__(self, n, p, q, d_total, dc, hyper, tau_vb, sig2_beta_vb, alpha_vb, sig2_alpha_vb,
                 comm=None):
        super().__init__(n, p, d_total, hyper, tau_vb, sig2_beta_vb, comm)
        self.q, self.dc = int(q), int(dc)
        self.phi = np.broadcast_to(np.asarray(hyper["phi"], dtype=np.float64), (q,)).copy()
        self.xi = np.broadcast_to(np.asarray(hyper["xi"], dtype=np.float64), (q,)).copy()
        self.phi_vb = U.update_phi_z_vb_(self.phi, self.d)                            # :41
        self.alpha = np.array(alpha_vb, dtype=np.float64).reshape(q, -1)
        self.sig2_alpha_vb = np.array(sig2_alpha_vb, dtype=np.float64).reshape(q, -1)
        self.m2_alpha = U.update_m2_alpha_(self.alpha, self.sig2_alpha_vb)
        self.log_tau_vb = np.zeros(self.tau_vb.shape[0])
        self.den_add = np.zeros(q)
        self.den_add[0] = 1.0                                    # intercept = TRUE
        self.xi_sum = self.zeta2_inv_vb = None

    def set_initial(self, stats, rs_gam_local):
        self.stats = np.asarray(stats, dtype=np.float64)
        t_loc = float(self.tau_vb @ self.stats[1])
        sc = self._reduce_p(rs_gam_local, [t_loc] + list(self.m2_alpha @ self.tau_vb))
        self.sum_tau_m2 = float(sc[0])
        self.xi_sum = np.array(sc[1:1 + self.q])

    def _eta_kappa(self, zeta2_inv_vb, sig2_inv_vb):
        """update_eta_z_vb_ and update_kappa_z_vb_(intercept = TRUE) for the continuous responses
        (R/update_vb.R:226, 255-272) on reduced statistics."""
        n, q, dc = self.n, self.q, self.dc
        cs_gam, cs_m2, cs_b2, rss = (v[:dc] for v in self.stats)
        m2a, al = self.m2_alpha[:, :dc], self.alpha[:, :dc]
        eta_vb = self.eta + n / 2 + cs_gam / 2 + q / 2
        kappa_vb = (self.kappa + (rss + (n - 1 + sig2_inv_vb) * cs_m2 - (n - 1) * cs_b2
                                  + (n - 1) * m2a.sum(axis=0) + m2a.T @ zeta2_inv_vb
                                  - (n - 1) * (al ** 2).sum(axis=0)) / 2
                    + (m2a[0] - al[0] ** 2) / 2)
        return eta_vb, kappa_vb

    # -- top of the iteration: R/locus_mix_core.R:61-102 ---------------------------------------
    def begin_iteration(self):
        n, dc = self.n, self.dc
        xi_vb = self.xi + self.xi_sum / 2                                             # :62
        self.zeta2_inv_vb = self.phi_vb / xi_vb
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        self.sig2_inv_vb = lambda_vb / nu_vb
        eta_vb, kappa_vb = self._eta_kappa(self.zeta2_inv_vb, self.sig2_inv_vb)
        self.tau_vb[:dc] = eta_vb / kappa_vb                                          # :87
        self.sig2_alpha_vb = 1.0 / np.outer(n - 1 + self.zeta2_inv_vb + self.den_add, self.tau_vb)
        self.sig2_beta_vb = U.update_sig2_beta_vb_(n, self.sig2_inv_vb, self.tau_vb)
        self.log_tau_vb[:dc] = U.update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb, log_1_min_om_vb = self._log_om_terms(1.0)
        s2 = self.sig2_beta_vb
        return dict(lo=log_om_vb, l1=log_1_min_om_vb, scale=s2 * self.tau_vb, hinv=1.0 / (2.0 * s2),
                    cst=-(self.log_tau_vb + log_sig2_inv_vb + np.log(s2)) / 2, s2k=s2,
                    alpha_coef=self.sig2_alpha_vb * self.tau_vb[None, :])

    # -- after the sweep: :215-236 and elbo_mix_ :287-349 ---------------------------------------
    def end_iteration(self, stats, rs_gam_local, ent_local, alpha_vb, ey_bin, zz):
        """ey_bin: sum over the local binary responses of  y log Phi(U) + (1 - y) log(1 - Phi(U))
        - Q / 2  (device); zz = colSums(Z^2)."""
        n, d, q, dc = self.n, self.d, self.q, self.dc
        self.stats = np.asarray(stats, dtype=np.float64)
        self._last = (rs_gam_local, ent_local, alpha_vb)
        cs_gam, cs_m2, cs_b2, rss = self.stats
        self.alpha = np.asarray(alpha_vb, dtype=np.float64).reshape(q, -1)
        self.m2_alpha = U.update_m2_alpha_(self.alpha, self.sig2_alpha_vb)
        tau, s2, si, zi = self.tau_vb, self.sig2_beta_vb, self.sig2_inv_vb, self.zeta2_inv_vb
        eta_vb, kappa_vb = self._eta_kappa(zi, si)
        log_tau_vb = self.log_tau_vb.copy()
        log_tau_vb[:dc] = digamma(eta_vb) - np.log(kappa_vb)
        m2a_z = self.m2_alpha.T @ zi
        loc = (E.e_y_(n, self.kappa, kappa_vb, log_tau_vb[:dc], cs_m2[:dc], si, tau[:dc])
               + float(np.sum(tau[:dc] * m2a_z[:dc] / 2))
               # e_y_probit_(sweep = FALSE), R/elbo.R:221-237
               + ey_bin - float(np.sum(zz @ self.sig2_alpha_vb[:, dc:]) / 2)
               + E.e_beta_gamma_local_(cs_gam, cs_m2, ent_local, log_tau_vb, s2, si, tau)
               + E.e_tau_(self.eta, eta_vb, self.kappa, kappa_vb, log_tau_vb[:dc], tau[:dc])
               + 0.5 * float(np.sum(-tau * m2a_z) + q * np.sum(log_tau_vb)
                             + np.sum(np.log(self.sig2_alpha_vb)) + self.m2_alpha.size))
        sc = self._reduce_p(rs_gam_local, [float(tau @ cs_m2), loc] + list(self.m2_alpha @ tau),
                            omega=True)
        self.sum_tau_m2 = float(sc[0])
        self.xi_sum = np.array(sc[2:2 + q])
        s_bg, e_om = self._omega_terms()
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
        xi_vb = self.xi + self.xi_sum / 2
        log_zeta2_inv_vb = digamma(self.phi_vb) - np.log(xi_vb)
        return float(sc[1] + log_sig2_inv_vb * self.sum_gam / 2 + s_bg
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si)
                     + e_om + 0.5 * d * float(np.sum(log_zeta2_inv_vb))
                     + E.e_zeta2_inv_(log_zeta2_inv_vb, self.phi, self.phi_vb, self.xi, xi_vb, zi))


    def held_out_elbo_mix_(self, n_test, rss_test, ey_bin_test, zz_test):
        """elbo_mix_ (R/locus_mix_core.R:287-349) with the held-out rows' n, residual sums of
        squares (continuous responses), e_y_probit_ sum (binary ones) and colSums(Z_test^2);
        everything else as fitted (R/cross_validate.R:363-383)."""
        n_fit, st_fit = self.n, self.stats
        try:
            self.n = int(n_test)
            st = st_fit.copy()
            st[3] = rss_test
            return self.end_iteration(st, *self._last, ey_bin_test, zz_test)
        finally:
            self.n, self.stats = n_fit, st_fit


class MixEngine(IdentityEngine):
    """One GPU's shard: responses [0, dc) continuous, [dc, d_loc) binary (0/1)."""

    def __init__(self, px, Y_dev, Z, state, host, device, dv=None):
        super().__init__(px, Y_dev, state, host, device, dv=dv)
        t, dv = self.torch, self.dv
        self.q, self.dc = Z.shape[1], host.dc
        self.db = self.d_loc - self.dc
        self.Z_host = np.asfortranarray(Z, dtype=np.float64)
        self.zz = (self.Z_host ** 2).sum(axis=0)
        self.Zt = dv.to_device_colmajor(self.Z_host, device).t().contiguous()          # [q][n]
        self.alpha = self._up(host.alpha)
        cm = lambda k: t.empty((k, self.n), dtype=t.float64, device=device).t()        # noqa: E731
        self.M, self.Zmu = cm(self.d_loc), cm(self.d_loc)
        self.W = self.Y.clone(memory_format=t.preserve_format)       # continuous part stays Y
        self.Q = cm(max(self.db, 1))
        self.ey = t.zeros((max(self.db, 1),), dtype=t.float64, device=device)
        self.zs0 = t.zeros((self.n,), dtype=t.float64, device=device)

    def _bin(self, A):
        return A[:, self.dc:]

    def _update_W(self, with_ey):
        """W[, ind_bin] <- update_W_probit_ (R/locus_mix_core.R:35-37, :218-220); with_ey also the
        column sums of e_y_probit_ without its Z^2 sig2_alpha term."""
        if self.db == 0:
            return 0.0
        dv, t = self.dv, self.torch
        n, p, dc, db = self.n, self.p, self.dc, self.db
        if not with_ey:
            dv.probit_w(self._bin(self.Y), self._bin(self.M), self._bin(self.Zmu), None, None,
                        self._bin(self.W), None, n, db)
            return 0.0
        s2 = self._up(self.host.sig2_beta_vb[dc:])
        g, m, b = (self.state[k][:, dc:] for k in range(3))
        D = (g * (m * m + s2) - b * b).contiguous()                  # m2_beta - beta_vb^2, [p][db]
        dv.gemm(n, db, p, self.px.Xp, 1, self.px.ld, D, db, 1, self.Q, 1, n, sqA=True)
        dv.probit_w(self._bin(self.Y), self._bin(self.M), self._bin(self.Zmu), self.Q, self.zs0,
                    self._bin(self.W), self.ey, n, db)
        return float(self.ey[:db].sum().item())

    def initialise(self):
        """R/locus_mix_core.R:27-46."""
        t, dv = self.torch, self.dv
        n, p, d = self.n, self.p, self.d_loc
        cols = dv.state_cols(self.state, p, d, s2k=self._up(self.host.sig2_beta_vb))
        dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.state[2], d, 1, self.M, 1, n)
        self.Zmu.copy_(self._up((self.Z_host @ self.host.alpha).T).t())
        self._update_W(False)
        self.R = dv.residual(self.W, self.M, self.Zmu)
        rss = (self.R * self.R).sum(dim=0)
        rs_gam, _ = dv.gamma_rows(self.state[0], p, d)
        stats = t.cat([cols, rss.reshape(1, d)], dim=0).cpu().numpy()
        self.host.set_initial(stats, rs_gam)

    def iterate(self, shuffled_ind, alpha_order=None, before_readback=None):
        """`before_readback()`: host work of the caller once the kernels of this iteration are
        queued, before the host blocks on the read-back; if it returns the next sweep order, that
        sweep's X-dependent operands are packed on a side stream now (IdentityEngine._pack_ahead)."""
        dv, t = self.dv, self.torch
        n, d = self.n, self.d_loc
        prm = self.host.begin_iteration()
        pre, self._pre_ind = getattr(self, "_pre_ind", None), None
        if pre is not None and pre[0] is shuffled_ind:
            ind = pre[1]           # uploaded, and its operands packed, while the last sweep ran
        else:
            ind = shuffled_ind if t.is_tensor(shuffled_ind) else self._up(shuffled_ind)
        # the iteration's constants in one upload (seven small copies before)
        coef, lo, l1, scale, hinv, cst, s2k = self._up_many(
            prm["alpha_coef"], prm["lo"], prm["l1"], prm["scale"], prm["hinv"], prm["cst"], prm["s2k"])
        for i in (range(self.q) if alpha_order is None else alpha_order):            # :104-111
            dv.alpha_step(self.W, None, self.M, self.Zmu, self.Zt[i], coef[i], self.alpha[i], n, d)
        dv.residual(self.W, self.M, self.Zmu, out=self.R)
        self.sweeper.identity(ind, lo, l1, scale, hinv, cst, s2k, 1.0, self.R, self.state, self.stats)
        dv.residual(self.W, self.R, self.Zmu, out=self.M)
        rs_gam, ent = dv.gamma_rows(self.state[0], self.p, d, eps=self.host.ent_eps)
        ey_bin = self._update_W(True)
        h_dev = t.cat([self.stats.reshape(-1), ent.sum().reshape(1), self.alpha.reshape(-1)])
        if before_readback is not None:
            self._pack_ahead(before_readback())   # the next sweep order, if the caller returns it
        h = h_dev.cpu().numpy()
        self.sweeper.status()
        return self.host.end_iteration(h[: 4 * d].reshape(4, d), rs_gam, float(h[4 * d]),
                                       h[4 * d + 1:], ey_bin, self.zz)

    def held_out_elbo_mix(self, Y_test_loc, X_test, Z_test):
        """Y_test_loc in the local [continuous | binary] order (continuous columns centred)."""
        t, dv = self.torch, self.dv
        p, d, dc, db = self.p, self.d_loc, self.dc, self.db
        Zt = np.asarray(Z_test, dtype=np.float64)
        nt = Zt.shape[0]
        Xt = dv.to_device_colmajor(X_test, self.device)
        Yt = dv.to_device_colmajor(Y_test_loc, self.device)
        cm = lambda k: t.empty((k, nt), dtype=t.float64, device=self.device).t()  # noqa: E731
        M = cm(d)
        dv.gemm(nt, d, p, Xt, 1, nt, self.state[2], d, 1, M, 1, nt)
        Zmu = self._up((Zt @ self.alpha.cpu().numpy()).T).t()
        R = dv.residual(Yt, M, Zmu)
        rss = (R * R).sum(dim=0).cpu().numpy()
        ey = 0.0
        if db > 0:
            s2 = self._up(self.host.sig2_beta_vb[dc:])
            g, m, b = (self.state[k][:, dc:] for k in range(3))
            D = (g * (m * m + s2) - b * b).contiguous()
            Q, W = cm(db), cm(db)
            eyv = t.zeros((db,), dtype=t.float64, device=self.device)
            dv.gemm(nt, db, p, Xt, 1, nt, D, db, 1, Q, 1, nt, sqA=True)
            dv.probit_w(Yt[:, dc:], M[:, dc:], Zmu[:, dc:], Q,
                        t.zeros((nt,), dtype=t.float64, device=self.device), W, eyv, nt, db)
            ey = float(eyv.sum().item())
        return self.host.held_out_elbo_mix_(nt, rss, ey, (Zt ** 2).sum(axis=0))

    def download(self):
        out = LazyFields(gam_vb=lambda: to_host_F(self.state[0]),
                         mu_beta_vb=lambda: to_host_F(self.state[1]),
                         beta_vb=lambda: to_host_F(self.state[2]),
                         mat_x_m1=lambda: to_host_F(self.M), mat_z_mu=lambda: to_host_F(self.Zmu),
                         alpha_vb=lambda: self.alpha.cpu().numpy().copy())
        return out


def locus_mix_core_(Y, X, Z, ind_bin, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                    sig2_beta_vb, tau_vb, tol, maxit, verbose=False, batch="y", full_output=False,
                    debug=False, perm_fn=None, alpha_perm_fn=None, comm=None, device=None,
                    trace=None, held_out=None, _test_backend=None):
    """R/locus_mix_core.R:8-280 on the GPU.  `held_out = (Y_test, X_test, Z_test)` adds
    `lb_held_out` (R/cross_validate.R:363-383).  `ind_bin`: 0-based indices of the binary (0/1)
    responses of Y; the continuous ones are centred, X and Z[:, 1:] standardised, Z[:, 0] is the
    intercept.  eta, kappa of list_hyper have one entry per continuous response (in order);
    tau_vb has length d with 1 at the binary responses."""
    dv, device = resolve_backend(_test_backend, device)
    import torch
    if batch != "y":
        raise NotImplementedError("only the 'y' scheme is device-resident")
    comm = comm or Comm()
    Y = np.asarray(Y, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    n, d = Y.shape
    p, q = X.shape[1], Z.shape[1]
    is_bin = np.zeros(d, dtype=bool)
    is_bin[np.asarray(ind_bin, dtype=np.int64)] = True
    cont_rank = np.cumsum(~is_bin) - 1                      # position among the continuous ones
    k_lo, k_hi = comm.shard(d)
    loc = np.arange(k_lo, k_hi)
    order = np.concatenate([loc[~is_bin[loc]], loc[is_bin[loc]]])   # [continuous | binary]
    dc = int((~is_bin[loc]).sum())
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731
    if alpha_perm_fn is None:
        rng_a = np.random.default_rng(1)
        alpha_perm_fn = lambda it: rng_a.permutation(q)  # noqa: E731

    n_cont = int((~is_bin).sum())
    hyper_loc = dict(list_hyper)
    for key in ("eta", "kappa"):
        full = np.broadcast_to(np.asarray(list_hyper[key], dtype=np.float64), (n_cont,))
        hyper_loc[key] = full[cont_rank[order[:dc]]]
    take = lambda a: np.asarray(a, dtype=np.float64)[..., order]  # noqa: E731
    host = MixHost(n, p, q, d, dc, hyper_loc, take(tau_vb), take(sig2_beta_vb),
                   take(np.asarray(alpha_vb).reshape(q, d)),
                   take(np.broadcast_to(np.asarray(sig2_alpha_vb, dtype=np.float64), (q, d))), comm)
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, order], device)
    state = torch.empty((3, p, len(order)), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, order, device)
    state[1] = columns_to_device(mu_beta_vb, order, device)
    eng = MixEngine(px, Y_dev, Z, state, host, device, dv=dv)
    eng.initialise()

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nxt = {}
    draw = lambda i: (np.ascontiguousarray(perm_fn(i), dtype=np.float64), alpha_perm_fn(i))  # noqa: E731
    while (not converged) and it < maxit:                                # :53
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        ind, a_order = nxt.pop(it) if it in nxt else draw(it)

        def draw_next(it=it):                 # while the GPU runs this iteration
            if it < maxit:
                nxt[it + 1] = draw(it + 1)
                return nxt[it + 1][0]         # packed ahead by the engine, beside the running sweep
        lb_new = eng.iterate(ind, alpha_order=a_order, before_readback=draw_next)
        if trace is not None:
            trace.append((it, lb_new))
        if verbose and (it == 1 or it % 5 == 0):
            print(f"ELBO = {lb_new}\n")
        if debug and lb_new < lb_old:                                    # :234-235
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol
    if verbose and not converged:
        print("Maximal number of iterations reached before convergence. Exit.")

    back = np.argsort(order)                                # local [cont | bin] -> original order

    def gather(a):
        return _gather(comm, np.asfortranarray(np.asarray(a)[:, back]), d)

    loc_out = eng.download()
    out = dict(beta_vb=gather(loc_out["beta_vb"]), gam_vb=gather(loc_out["gam_vb"]),
               om_vb=host.om_vb, alpha_vb=gather(loc_out["alpha_vb"]), converged=converged, it=it,
               lb_opt=lb_new, diff_lb=abs(lb_new - lb_old))
    if held_out is not None:
        Yt = np.asarray(held_out[0], dtype=np.float64)
        out["lb_held_out"] = eng.held_out_elbo_mix(Yt[:, order], held_out[1], held_out[2])
    if full_output:                                                      # :264-270
        out.update(mu_beta_vb=gather(loc_out["mu_beta_vb"]), mat_x_m1=gather(loc_out["mat_x_m1"]),
                   mat_z_mu=gather(loc_out["mat_z_mu"]), sig2_alpha_vb=gather(host.sig2_alpha_vb),
                   sig2_beta_vb=_gather_vec(comm, host.sig2_beta_vb[back], d),
                   tau_vb=_gather_vec(comm, host.tau_vb[back], d), sig2_inv_vb=host.sig2_inv_vb,
                   zeta2_inv_vb=host.zeta2_inv_vb, a_vb=host.a_vb, b_vb=host.b_vb,
                   sum_gam=host.sum_gam, ind_bin=np.nonzero(is_bin)[0])
    return out
