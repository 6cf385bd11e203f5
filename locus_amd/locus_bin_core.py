"""Binary-response CAVI drivers with device-resident state:
  locus_logit_core_   mirrors R/locus_logistic_core.R:8-265 (Jaakkola-Jordan bound, chi / psi)
  locus_probit_core_  mirrors R/locus_probit_core.R:8-250   (latent W, inverse Mills ratio)

Resident per GPU (responses sharded, X and Z replicated): packed X, Y, mat_x_m1, mat_z_mu,
the residual, Q = X^2 (m2_beta - beta_vb^2), the [p][d] state planes and, for logit, the
[p][d] sig2_beta_vb / xwx planes and double-buffered chi_vb / 2 psi_vb.

X^2 contractions per iteration (fp64 matrix cores): the reference does five for logit
(update_chi_vb_ x2, update_sig2_beta_logit_vb_, e_y_logit_ x2); here two suffice:
  Q   = X^2 (m2_beta - beta_vb^2)   serves e_y_logit_ of iteration t AND chi_vb of iteration t+1
        (both are evaluated on the same end-of-iteration state, R/elbo.R:213-216 vs
        R/update_vb.R:165-166 with Z^2 m2_alpha - Z^2 alpha^2 = Z^2 sig2_alpha);
  xwx = crossprod(X^2, 2 psi_vb)    serves sig2_beta_vb AND the sweep's folded dot product.
Probit needs Q only (e_y_probit_, R/elbo.R:228).

Exchange: ONE all-reduce per iteration of [rowSums(gam_vb) (p) | sum(m2_beta) |
rowSums(m2_alpha) (q) | shard-local ELBO sums].
"""
import numpy as np
from scipy.special import digamma

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F
from .locus_core import EPS_MACH, HostBase, _gather, resolve_backend


class BinHost(HostBase):
    """Replicated vector arithmetic shared by the logit and probit drivers."""

    def __init__(self, n, p, d_total, q, hyper, comm=None):
        self.n, self.p, self.d, self.q = int(n), int(p), int(d_total), int(q)
        self.comm = comm or Comm()
        self.a = np.broadcast_to(np.asarray(hyper["a"], dtype=np.float64), (p,)).copy()
        self.b = np.broadcast_to(np.asarray(hyper["b"], dtype=np.float64), (p,)).copy()
        self.lam, self.nu = float(hyper["lambda"]), float(hyper["nu"])
        self.phi = np.atleast_1d(np.asarray(hyper["phi"], dtype=np.float64))
        self.xi = np.atleast_1d(np.asarray(hyper["xi"], dtype=np.float64))
        self.phi_vb = U.update_phi_z_vb_(self.phi, self.d)          # logistic_core.R:30
        self.sum_gam = self.sum_m2 = self.rs_m2_alpha = None
        self.sig2_inv_vb = self.zeta2_inv_vb = None
        # everything of length p lives where the all-reduced rowSums(gam_vb) lives (the GPU; CPU
        # tensors under the CPU test backend): five digamma and six lgamma passes over p per
        # iteration cost 3 ms in numpy at p = 20 000 with the GPU idle, against 0.1 ms there
        self.rs_gam_t = self.a_t = self.b_t = self.digam_sum_t = None

    def _on_device(self, dev):
        import torch
        if self.a_t is None or self.a_t.device != dev:
            self.a_t = torch.from_numpy(self.a).to(dev)
            self.b_t = torch.from_numpy(self.b).to(dev)
            self.digam_sum_t = torch.digamma(self.a_t + self.b_t + self.d)   # :34
            self.lbeta_ab_t = (torch.lgamma(self.a_t) + torch.lgamma(self.b_t)
                               - torch.lgamma(self.a_t + self.b_t)).sum()

    # length-p results for the caller, on the host (logistic_core.R:229-243)
    rs_gam = property(lambda self: self.rs_gam_t.cpu().numpy())
    a_vb = property(lambda self: U.update_a_vb(self.a, self.rs_gam))
    b_vb = property(lambda self: U.update_b_vb(self.b, self.d, self.rs_gam))
    om_vb = property(lambda self: self.a_vb / (self.a_vb + self.b_vb))

    def exchange(self, rs_gam_local, sum_m2_loc, rs_m2_alpha_loc, extra):
        """ONE all-reduce of [rowSums(gam) (p) | scalars]; rowSums stays on its device, the
        scalars and sum(rowSums) come back."""
        import torch
        dev = rs_gam_local.device
        self._on_device(dev)
        tail = np.concatenate([[sum_m2_loc], rs_m2_alpha_loc, np.atleast_1d(extra)])
        buf = torch.cat([rs_gam_local.reshape(-1), torch.as_tensor(tail, dtype=torch.float64).to(dev)])
        self.comm.all_reduce_sum_(buf)
        self.rs_gam_t = buf[: self.p]
        sc = torch.cat([buf[self.p:], self.rs_gam_t.sum().reshape(1)]).cpu().numpy()
        self.sum_gam = float(sc[-1])
        self.sum_m2 = float(sc[0])
        self.rs_m2_alpha = sc[1:1 + self.q].copy()
        return sc[1 + self.q:-1]

    def exchange_t(self, rs_gam_local, tail_t):
        """exchange() with the scalars as a device tensor [sum(m2_beta) | rowSums(m2_alpha) (q) |
        extra ...] and nothing brought back: returns the all-reduced tail, on the device."""
        import torch
        self._on_device(rs_gam_local.device)
        if self.comm.solo:
            self.rs_gam_t = rs_gam_local.reshape(-1)
            return tail_t
        buf = torch.cat([rs_gam_local.reshape(-1), tail_t.reshape(-1)])
        self.comm.all_reduce_sum_(buf)
        self.rs_gam_t = buf[: self.p]
        return buf[self.p:]

    def cov_t(self):
        """The covariate-wise terms of e_beta_gamma_bin_ (R/elbo.R:81-83) and e_omega_ (:114-118)
        as ONE device scalar, from the all-reduced rowSums(gam_vb)."""
        import torch
        d = self.d
        rs, a, b = self.rs_gam_t, self.a_t, self.b_t
        a_vb, b_vb = a + rs, b - rs + d                                   # update_a_vb / update_b_vb
        dg = torch.digamma(a_vb + b_vb)
        log_om_vb, log_1_min_om_vb = torch.digamma(a_vb) - dg, torch.digamma(b_vb) - dg
        return ((rs * log_om_vb).sum() + ((d - rs) * log_1_min_om_vb).sum()
                + ((a - a_vb) * log_om_vb + (b - b_vb) * log_1_min_om_vb + torch.lgamma(a_vb)
                   + torch.lgamma(b_vb) - torch.lgamma(a_vb + b_vb)).sum() - self.lbeta_ab_t)

    def begin_iteration(self):
        """logistic_core.R:55-77 / probit_core.R:51-74 (the parts common to both links)."""
        import torch
        xi_vb = U.update_xi_bin_vb_(self.xi, self.rs_m2_alpha)
        self.zeta2_inv_vb = self.phi_vb / xi_vb
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_bin_vb_(self.nu, self.sum_m2)
        self.sig2_inv_vb = lambda_vb / nu_vb
        rs = self.rs_gam_t
        return dict(log_sig2_inv_vb=float(U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)),
                    lo=torch.digamma(self.a_t + rs) - self.digam_sum_t,                 # R/update_vb.R:187
                    l1=torch.digamma(self.b_t - rs + self.d) - self.digam_sum_t)        # :190

    def finish_elbo(self, loc_sum, e_alpha_fn, cov=None):
        """Replicated part of elbo_logit_ / elbo_probit_ (logistic_core.R:276-301,
        probit_core.R:262-286); loc_sum = all-reduced shard-local sums; cov = cov_t() already on
        the host (it travels with the iteration's one read-back), else fetched here."""
        if cov is None:
            cov = float(self.cov_t())
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_bin_vb_(self.nu, self.sum_m2)
        xi_vb = U.update_xi_bin_vb_(self.xi, self.rs_m2_alpha)
        log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
        log_zeta2_inv_vb = digamma(self.phi_vb) - np.log(xi_vb)
        si = self.sig2_inv_vb
        glob = log_sig2_inv_vb * self.sum_gam / 2 - self.sum_m2 * si / 2 + cov
        return float(loc_sum + glob
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si)
                     + e_alpha_fn(log_zeta2_inv_vb)
                     + E.e_zeta2_inv_(log_zeta2_inv_vb, self.phi, self.phi_vb, self.xi, xi_vb,
                                      self.zeta2_inv_vb))


class BinEngine:
    """One GPU's shard of a binary-response model (link = 'logit' | 'probit')."""

    def __init__(self, link, px, Y_dev, Z, state, alpha_vb, sig2_alpha_vb, sig2_beta_vb, host,
                 device, dv=None):
        import torch
        if dv is None:
            from . import device as dv
        assert link in ("logit", "probit")
        self.link, self.torch, self.dv = link, torch, dv
        self.px, self.Y, self.state, self.host, self.device = px, Y_dev, state, host, device
        self.n, self.d_loc = Y_dev.shape
        self.p, self.q = px.p, Z.shape[1]
        n, p, d, q = self.n, self.p, self.d_loc, self.q
        assert tuple(state.shape) == (3, p, d)
        self.Z_host = np.asfortranarray(Z, dtype=np.float64)
        self.Z = dv.to_device_colmajor(self.Z_host, device)            # n x q, column j contiguous
        self.Zt = self.Z.t().contiguous()                               # [q][n]
        self.alpha = self._up(np.ascontiguousarray(alpha_vb, dtype=np.float64).reshape(q, d))
        f64 = torch.float64
        cm = lambda: torch.empty((d, n), dtype=f64, device=device).t()  # noqa: E731
        self.M, self.Zmu, self.R, self.Q = cm(), cm(), cm(), cm()
        self.D = torch.empty((p, d), dtype=f64, device=device)
        self.sweeper = dv.Sweeper(px, d, device)
        if link == "logit":
            self.s2 = sig2_beta_vb if torch.is_tensor(sig2_beta_vb) else self._up(sig2_beta_vb)
            assert tuple(self.s2.shape) == (p, d)
            self.xwx = torch.empty((p, d), dtype=f64, device=device)
            self.ls2 = torch.empty((p, d), dtype=f64, device=device)
            self.s2a = self._up(np.ascontiguousarray(sig2_alpha_vb, dtype=np.float64).reshape(q, d))
            self.chi = [cm(), cm()]
            self.W2 = [cm(), cm()]
            self.cur = 0
        else:
            self.s2c = float(sig2_beta_vb)
            self.s2a_vec = np.atleast_1d(np.asarray(sig2_alpha_vb, dtype=np.float64)).copy()
            self.W = cm()
        self.ey = torch.empty((d,), dtype=f64, device=device)

    def _up(self, a):
        if self.torch.is_tensor(a):
            return a
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    # -- shared pieces -----------------------------------------------------------------
    def _rows(self):
        dv = self.dv
        if self.link == "logit":
            return dv.bin_rows(self.state, self.p, self.d_loc, EPS_MACH, s2jk=self.s2)
        return dv.bin_rows(self.state, self.p, self.d_loc, EPS_MACH, s2c=self.s2c)

    def _contract_Q(self):
        """Q = X^2 (m2_beta - beta_vb^2): n x d on the matrix cores."""
        dv, n, p, d = self.dv, self.n, self.p, self.d_loc
        if self.link == "logit":
            dv.m2_minus_b2(self.state, p, d, self.D, s2jk=self.s2)
        else:
            dv.m2_minus_b2(self.state, p, d, self.D, s2c=self.s2c)
        dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.D, d, 1, self.Q, 1, n, sqA=True)

    def _m2_alpha_rows(self):
        """rowSums(m2_alpha) over the local responses (update_m2_alpha_, R/update_vb.R:18-29)
        and sum(log(sig2_alpha_vb)) for e_alpha_logit_."""
        a = self.alpha.cpu().numpy()
        if self.link == "logit":
            s2a = self.s2a.cpu().numpy()
            return (s2a + a ** 2).sum(axis=1), float(np.log(s2a).sum())
        return (a ** 2 + self.s2a_vec[:, None]).sum(axis=1), 0.0

    def initialise(self):
        """logistic_core.R:23-34 / probit_core.R:23-36."""
        t, dv = self.torch, self.dv
        n, p, d, q = self.n, self.p, self.d_loc, self.q
        if self.link == "logit":
            dv.state_cols(self.state, p, d, s2jk=self.s2)    # writes beta = gam * mu
        else:
            dv.state_cols(self.state, p, d, s2k=t.full((d,), self.s2c, dtype=t.float64,
                                                       device=self.device))
        dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.state[2], d, 1, self.M, 1, n)
        self.Zmu.copy_(self._up((self.Z_host @ self.alpha.cpu().numpy()).T).t())
        rows = self._rows()
        rs_m2a, _ = self._m2_alpha_rows()
        self.host.exchange(rows[0], float(rows[3].sum().item()), rs_m2a, [])
        if self.link == "logit":
            self._contract_Q()
            nxt = 1 - self.cur
            dv.logit_chi_psi(self.Q, self.M, self.Zmu, None, self.Z, self.s2a, None, None,
                             self.chi[nxt], self.W2[nxt], None, n, d, q)
        else:
            dv.probit_w(self.Y, self.M, self.Zmu, None, None, self.W, None, n, d)

    def iterate(self, shuffled_ind, alpha_order=None, before_readback=None):
        """One pass of the loop body (logistic_core.R:40-225 / probit_core.R:42-210).
        `before_readback()` runs on the host once every kernel of the iteration is queued and
        before the host blocks on the read-back (the caller draws the next sweep order there and
        may return it: see _pack_ahead)."""
        t, dv, host = self.torch, self.dv, self.host
        n, p, d, q = self.n, self.p, self.d_loc, self.q
        prm = host.begin_iteration()
        pre, self._pre_ind = getattr(self, "_pre_ind", None), None
        if pre is not None and pre[0] is shuffled_ind:
            ind = pre[1]           # uploaded, and its operands packed, while the last sweep ran
        else:
            ind = shuffled_ind if t.is_tensor(shuffled_ind) else self._up(shuffled_ind)
        lo, l1 = self._up(prm["lo"]), self._up(prm["l1"])
        order = range(q) if alpha_order is None else alpha_order
        if self.link == "logit":
            self.cur = 1 - self.cur                                   # chi_vb, psi_vb of this iteration
            W2 = self.W2[self.cur]
            dv.logit_sig2_alpha(self.Z, W2, self._up(host.zeta2_inv_vb), self.s2a, n, d, q)
            # xwx = crossprod(X^2, 2 psi) -> sig2_beta_vb = 1 / (xwx + sig2_inv_vb)
            dv.gemm(p, d, n, self.px.Xp, self.px.ld, 1, W2, 1, n, self.xwx, d, 1, sqA=True)
            dv.s2_from_xwx(self.xwx, host.sig2_inv_vb, self.s2)
            dv.half_log(self.s2, self.ls2)          # log(sig2_beta_vb) / 2 for the sweep
            for i in order:
                dv.alpha_step(self.Y, W2, self.M, self.Zmu, self.Zt[i], self.s2a[i], self.alpha[i],
                              n, d)
            dv.residual(self.Y, self.M, self.Zmu, W2, out=self.R, to_residual=True)
            self.sweeper.logit(ind, lo, l1, self.s2, self.xwx, W2, prm["log_sig2_inv_vb"] / 2,
                               self.R, self.state, ls2=self.ls2)
            dv.residual(self.Y, self.R, self.Zmu, W2, out=self.M, to_residual=False)
        else:
            self.s2a_vec = U.update_sig2_alpha_vb_(n, host.zeta2_inv_vb, intercept=True)  # :63
            self.s2c = float(U.update_sig2_beta_vb_(n, host.sig2_inv_vb))                 # :64
            for i in order:
                s2a_i = t.full((d,), float(self.s2a_vec[i]), dtype=t.float64, device=self.device)
                dv.alpha_step(self.W, None, self.M, self.Zmu, self.Zt[i], s2a_i, self.alpha[i], n, d)
            dv.residual(self.W, self.M, self.Zmu, out=self.R)
            full = lambda v: t.full((d,), float(v), dtype=t.float64, device=self.device)  # noqa: E731
            s2 = self.s2c                                             # src/coreLoop.cpp:152,160
            self.sweeper.identity(ind, lo, l1, full(s2), full(1.0 / (2.0 * s2)),
                                  full(-(prm["log_sig2_inv_vb"] + np.log(s2)) / 2), full(s2), 1.0,
                                  self.R, self.state)
            dv.residual(self.W, self.R, self.Zmu, out=self.M)
        rows = self._rows()
        self._contract_Q()
        if self.link == "logit":
            nxt = 1 - self.cur
            dv.logit_chi_psi(self.Q, self.M, self.Zmu, self.Y, self.Z, self.s2a, self.chi[self.cur],
                             self.W2[self.cur], self.chi[nxt], self.W2[nxt], self.ey, n, d, q)
        else:
            zs = self._up((self.Z_host ** 2) @ self.s2a_vec)
            dv.probit_w(self.Y, self.M, self.Zmu, self.Q, zs, self.W, self.ey, n, d)
        # Everything the host needs from this iteration in ONE read-back (round 3; it took seven
        # synchronisations, the GPU idle in between): the shard-local sums are formed, exchanged and
        # turned into the covariate-wise ELBO terms on the device.
        sums = rows[1:].sum(dim=1)                                   # ent, gls, sum_m2 (local)
        ey = self.ey.sum()
        if self.link == "logit":                                     # update_m2_alpha_, R/update_vb.R:18-29
            rs_m2a = (self.s2a + self.alpha ** 2).sum(dim=1)
            sum_log_s2a = t.log(self.s2a).sum().reshape(1)
        else:
            rs_m2a = (self.alpha ** 2).sum(dim=1) + self._up(self.s2a_vec) * d
            sum_log_s2a = t.zeros(1, dtype=t.float64, device=self.device)
        # shard-local sums of the ELBO: e_y_*, and of e_beta_gamma_bin_ the terms
        # gam (log sig2_beta + 1) / 2 and the entropy (R/elbo.R:84-85)
        loc = ey + (sums[1] + rows[0].sum()) / 2 - sums[0]
        tot = host.exchange_t(rows[0], t.cat([sums[2].reshape(1), rs_m2a, loc.reshape(1), sum_log_s2a]))
        h_dev = t.cat([tot, host.rs_gam_t.sum().reshape(1), host.cov_t().reshape(1), ey.reshape(1)])
        if before_readback is not None:
            self._pack_ahead(before_readback())
        h = h_dev.cpu().numpy()
        self.sweeper.status()
        host.sum_m2, host.rs_m2_alpha = float(h[0]), h[1:1 + q].copy()
        loc_sum, sum_log_s2a_all = float(h[1 + q]), float(h[2 + q])
        host.sum_gam, cov, self._ey_fit = float(h[3 + q]), float(h[4 + q]), float(h[5 + q])
        if self.link == "logit":
            fn = lambda lz: E.e_alpha_logit_(host.rs_m2_alpha, sum_log_s2a_all, q, host.d, lz,  # noqa: E731
                                             host.zeta2_inv_vb)
        else:
            fn = lambda lz: E.e_alpha_probit_(host.rs_m2_alpha, self.s2a_vec, q, host.d, lz,  # noqa: E731
                                              host.zeta2_inv_vb)
        return host.finish_elbo(loc_sum, fn, cov=cov)

    def _pack_ahead(self, next_ind):
        """Probit (the matrix-core identity sweep): the next sweep's X-dependent operands are
        packed on a side stream beside the running iteration (locus_core.IdentityEngine._pack_ahead)."""
        if next_ind is None or self.link != "probit" or not hasattr(self.sweeper, "prepack"):
            return
        dev_ind = self.sweeper.prepack(next_ind)
        if dev_ind is not None:
            self._pre_ind = (next_ind, dev_ind)

    def held_out_elbo(self, lb_fit, Y_test_loc, X_test, Z_test):
        """elbo_probit_ on held-out rows (R/cross_validate.R:345-361).  Of its six terms only
        e_y_probit_ (R/elbo.R:221-237) sees the data, so the held-out bound is the fitted one with
        that term re-evaluated on (Y_test, X_test, Z_test): X_test beta_vb and
        X_test^2 (m2_beta - beta_vb^2) on the matrix cores, then the probit_w kernel."""
        if self.link != "probit":
            raise NotImplementedError("held-out ELBO: probit only (R/prepare_locus.R:550-552)")
        t, dv = self.torch, self.dv
        p, d = self.p, self.d_loc
        Zt = np.asarray(Z_test, dtype=np.float64)
        nt = Zt.shape[0]
        Xt = dv.to_device_colmajor(X_test, self.device)
        Yt = dv.to_device_colmajor(Y_test_loc, self.device)
        cm = lambda: t.empty((d, nt), dtype=t.float64, device=self.device).t()  # noqa: E731
        M, Q, W = cm(), cm(), cm()
        dv.gemm(nt, d, p, Xt, 1, nt, self.state[2], d, 1, M, 1, nt)
        dv.gemm(nt, d, p, Xt, 1, nt, self.D, d, 1, Q, 1, nt, sqA=True)   # D of the last iteration
        Zmu = self._up((Zt @ self.alpha.cpu().numpy()).T).t()
        zs = self._up((Zt ** 2) @ self.s2a_vec)
        ey = t.empty((d,), dtype=t.float64, device=self.device)
        dv.probit_w(Yt, M, Zmu, Q, zs, W, ey, nt, d)
        delta = (ey.sum() - self._ey_fit).reshape(1)
        self.host.comm.all_reduce_sum_(delta)
        return lb_fit + float(delta.item())

    def download(self):
        out = LazyFields(gam_vb=lambda: to_host_F(self.state[0]),
                         mu_beta_vb=lambda: to_host_F(self.state[1]),
                         beta_vb=lambda: to_host_F(self.state[2]),
                         mat_x_m1=lambda: to_host_F(self.M), mat_z_mu=lambda: to_host_F(self.Zmu),
                         alpha_vb=lambda: self.alpha.cpu().numpy().copy())
        if self.link == "logit":
            out._makers.update(sig2_beta_vb=lambda: to_host_F(self.s2),
                               psi_vb=lambda: to_host_F(self.W2[self.cur] / 2),
                               chi_vb=lambda: to_host_F(self.chi[self.cur]))
        else:
            out["sig2_beta_vb"] = self.s2c
        return out


def _run(link, Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb, sig2_beta_vb,
         tol, maxit, verbose, batch, full_output, debug, perm_fn, alpha_perm_fn, comm, device,
         trace, _test_backend, held_out=None):
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
    k_lo, k_hi = comm.shard(d)
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731
    if alpha_perm_fn is None:     # R: sample(1:q) every iteration (R/locus_logistic_core.R:79,
        rng_a = np.random.default_rng(1)   # R/locus_probit_core.R:76); same stream on every rank
        alpha_perm_fn = lambda it: rng_a.permutation(q)  # noqa: E731
    host = BinHost(n, p, d, q, list_hyper, comm)
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, k_lo:k_hi], device)
    state = torch.empty((3, p, k_hi - k_lo), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, slice(k_lo, k_hi), device)
    state[1] = columns_to_device(mu_beta_vb, slice(k_lo, k_hi), device)
    alpha_loc = np.asarray(alpha_vb, dtype=np.float64).reshape(q, d)[:, k_lo:k_hi]
    if link == "logit":
        s2a_loc = np.asarray(sig2_alpha_vb, dtype=np.float64).reshape(q, d)[:, k_lo:k_hi]
        s2b_loc = columns_to_device(sig2_beta_vb, slice(k_lo, k_hi), device)
    else:
        s2a_loc, s2b_loc = sig2_alpha_vb, float(sig2_beta_vb)
    eng = BinEngine(link, px, Y_dev, Z, state, alpha_loc, s2a_loc, s2b_loc, host, device, dv=dv)
    eng.initialise()

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nxt = {}
    draw = lambda i: (np.ascontiguousarray(perm_fn(i), dtype=np.float64), alpha_perm_fn(i))  # noqa: E731
    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        ind, a_order = nxt.pop(it) if it in nxt else draw(it)

        def draw_next(it=it):                 # while the GPU runs this iteration (0.2 ms at p = 20 000)
            if it < maxit:
                nxt[it + 1] = draw(it + 1)
                return nxt[it + 1][0]
        lb_new = eng.iterate(ind, a_order, before_readback=draw_next)
        if trace is not None:
            trace.append((it, lb_new))
        if verbose and (it == 1 or it % 5 == 0):
            print(f"ELBO = {lb_new}\n")
        if debug and lb_new < lb_old:
            raise RuntimeError("ELBO not increasing monotonically. Exit. ")
        converged = abs(lb_new - lb_old) < tol
    if verbose and not converged:
        print("Maximal number of iterations reached before convergence. Exit.")

    loc = eng.download()
    g = lambda key: _gather(comm, loc[key], d)  # noqa: E731
    out = dict(beta_vb=g("beta_vb"), gam_vb=g("gam_vb"), om_vb=host.om_vb,
               alpha_vb=g("alpha_vb"), converged=converged, it=it, lb_opt=lb_new,
               diff_lb=abs(lb_new - lb_old))
    if held_out is not None:
        Yt = np.asarray(held_out[0], dtype=np.float64)
        out["lb_held_out"] = eng.held_out_elbo(lb_new, Yt[:, k_lo:k_hi], held_out[1], held_out[2])
    if full_output:
        out.update(mu_beta_vb=g("mu_beta_vb"), mat_x_m1=g("mat_x_m1"), mat_z_mu=g("mat_z_mu"),
                   sig2_inv_vb=host.sig2_inv_vb, zeta2_inv_vb=host.zeta2_inv_vb,
                   a_vb=host.a_vb, b_vb=host.b_vb, sum_gam=host.sum_gam)
        if link == "logit":
            out.update(sig2_beta_vb=g("sig2_beta_vb"), psi_vb=g("psi_vb"), chi_vb=g("chi_vb"))
        else:
            out.update(sig2_beta_vb=loc["sig2_beta_vb"])
    return out


def locus_logit_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                      sig2_beta_vb, tol, maxit, verbose=False, batch="y", full_output=False,
                      debug=False, perm_fn=None, alpha_perm_fn=None, comm=None, device=None,
                      trace=None, _test_backend=None):
    """R/locus_logistic_core.R:8-265 on the GPU.  Y has 1/2 subtracted, X standardised, Z holds
    the intercept / fixed covariates (n x q); alpha_vb, sig2_alpha_vb are q x d and
    sig2_beta_vb is p x d.  The reference's `chi_vb` argument is overwritten before its
    first use (:49) and is therefore not taken.  `alpha_perm_fn(it)` gives the order of the
    alpha loop (R: sample(1:q), :79; 0-based here), `perm_fn(it)` the SNP sweep order."""
    return _run("logit", Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                sig2_beta_vb, tol, maxit, verbose, batch, full_output, debug, perm_fn,
                alpha_perm_fn, comm, device, trace, _test_backend)


def locus_probit_core_(Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                       sig2_beta_vb, tol, maxit, verbose=False, batch="y", full_output=False,
                       debug=False, perm_fn=None, alpha_perm_fn=None, comm=None, device=None,
                       trace=None, held_out=None, _test_backend=None):
    """R/locus_probit_core.R:8-250 on the GPU.  Y is 0/1, Z[:, 0] is the intercept (:13);
    sig2_alpha_vb is a length-q vector and sig2_beta_vb a scalar.  `held_out = (Y_test, X_test,
    Z_test)` adds `lb_held_out` (R/cross_validate.R:345-361)."""
    return _run("probit", Y, X, Z, list_hyper, gam_vb, alpha_vb, mu_beta_vb, sig2_alpha_vb,
                sig2_beta_vb, tol, maxit, verbose, batch, full_output, debug, perm_fn,
                alpha_perm_fn, comm, device, trace, _test_backend, held_out)
