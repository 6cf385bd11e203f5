"""Identity-link CAVI driver with device-resident state (mirrors R/locus_core.R).

Three layers:
  IdentityHost    O(p + d) host arithmetic of one iteration on REDUCED statistics, plus
                  the packed all-reduce -- no GPU needed (exercised under gloo on CPU);
  IdentityEngine  the shard of the model that lives on one GPU: packed X, residual,
                  state planes; runs the sweep + reductions through liblocus_hip;
  locus_core_     numpy-facing wrapper with the reference's signature and return fields
                  (R/locus_core.R:8-10, :242-252).
"""
import glob
import os

import numpy as np
from scipy.special import digamma

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F

EPS_SQRT = np.finfo(np.float64).eps ** 0.5  # R/locus_core.R:31
EPS_MACH = np.finfo(np.float64).eps         # R/elbo.R:46


class HostBase:
    """What every driver's host side shares: the ONE packed all-reduce per exchange."""

    ent_eps = EPS_MACH  # eps inside the logs of the entropy term (R/elbo.R:46)

    def _all_reduce(self, rs_gam_local, scalars):
        """ONE all-reduce of [rowSums(gam) (p) | scalars]; returns (rs_gam, scalars) on host."""
        import torch
        tail = torch.as_tensor(np.asarray(scalars, dtype=np.float64).reshape(-1),
                               device=rs_gam_local.device)
        buf = torch.cat([rs_gam_local.reshape(-1), tail])
        self.comm.all_reduce_sum_(buf)
        h = buf.cpu().numpy()
        return h[: self.p].copy(), h[self.p:].copy()


class IdentityHost(HostBase):
    """Replicated / shard-local vector arithmetic of locus_core_ (R/locus_core.R:49-213).

    Shard-local (length d_loc): eta, kappa, tau_vb, sig2_beta_vb and the column statistics.
    Replicated (length p or scalar): a, b, lambda, nu, rs_gam, sum_gam, sum_tau_m2.
    """

    def __init__(self, n, p, d_total, hyper, tau_vb, sig2_beta_vb, comm=None):
        self.n, self.p, self.d = int(n), int(p), int(d_total)
        self.comm = comm or Comm()
        self.a = np.broadcast_to(np.asarray(hyper["a"], dtype=np.float64), (p,)).copy()
        self.b = np.broadcast_to(np.asarray(hyper["b"], dtype=np.float64), (p,)).copy()
        self.eta = np.asarray(hyper["eta"], dtype=np.float64)      # shard-local
        self.kappa = np.asarray(hyper["kappa"], dtype=np.float64)  # shard-local
        self.lam, self.nu = float(hyper["lambda"]), float(hyper["nu"])
        self.tau_vb = np.array(tau_vb, dtype=np.float64)
        self.sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
        self.sig2_inv_vb = None
        self.sum_gam = self.sum_tau_m2 = None
        self.stats = None
        # length-p quantities live where rowSums(gam) lives (the GPU; CPU tensors in the CPU
        # tests): the all-reduced buffer never comes back to the host, only scalars do
        self.rs_gam_t = self.a_t = self.b_t = None
        self._digam_sum, self._c_last = {}, 1.0

    def _reduce_p(self, rs_gam_local, scalars, omega=False):
        """ONE all-reduce of [rowSums(gam) (p) | scalars]; rowSums stays a tensor on its
        device, the scalars (+ sum(rs_gam)) come back to the host.  omega: the covariate-wise
        ELBO terms (_omega_terms) are formed right away and ride in the same transfer, so that an
        iteration costs one host synchronisation here instead of two."""
        import torch
        dev = rs_gam_local.device
        if self.a_t is None or self.a_t.device != dev:
            self.a_t = torch.from_numpy(self.a).to(dev)
            self.b_t = torch.from_numpy(self.b).to(dev)
            self._digam_sum = {}
        if self.comm.solo:                       # nothing to exchange: no staging buffer
            self.rs_gam_t = rs_gam_local.reshape(-1)
            parts = [self.rs_gam_t.sum().reshape(1)]
            if omega:
                parts.append(self._omega_terms_t())
            h = torch.cat(parts).cpu().numpy()
            self.sum_gam = float(h[0])
            self._omega_cache = (float(h[1]), float(h[2])) if omega else None
            return np.asarray(scalars, dtype=np.float64).reshape(-1)
        tail = torch.as_tensor(np.asarray(scalars, dtype=np.float64).reshape(-1), device=dev)
        buf = torch.cat([rs_gam_local.reshape(-1), tail])
        self.comm.all_reduce_sum_(buf)
        self.rs_gam_t = buf[: self.p]
        parts = [buf[self.p:], self.rs_gam_t.sum().reshape(1)]
        if omega:
            parts.append(self._omega_terms_t())
        h = torch.cat(parts).cpu().numpy()
        k = len(h) - (3 if omega else 1)
        self.sum_gam = float(h[k])
        self._omega_cache = (float(h[k + 1]), float(h[k + 2])) if omega else None
        return h[:k]

    # length-p results for the caller (R/locus_core.R:242-252), on the host
    @property
    def rs_gam(self):
        return self.rs_gam_t.cpu().numpy()

    @property
    def a_vb(self):
        return U.update_a_vb(self.a, self.rs_gam, self._c_last)

    @property
    def b_vb(self):
        return U.update_b_vb(self.b, self.d, self.rs_gam, self._c_last)

    @property
    def om_vb(self):
        a_vb, b_vb = self.a_vb, self.b_vb
        return a_vb / (a_vb + b_vb)

    def held_out_elbo_(self, n_test, rss_test):
        """The ELBO of the last iteration re-evaluated with the held-out rows' n and residual
        sums of squares, everything else (gam_vb, m2_beta, tau_vb, sig2_inv_vb, ...) as fitted:
        what elbo_(Y_test, ..., mat_x_m1 = X_test %*% beta_vb, ...) computes."""
        n_fit, st_fit = self.n, self.stats
        try:
            self.n = int(n_test)
            st = st_fit.copy()
            st[3] = rss_test
            return self._end_again(st)
        finally:
            self.n, self.stats = n_fit, st_fit

    def _end_again(self, stats):
        return self.end_iteration(stats, *self._last)

    def _omega_terms(self):
        """Covariate-wise terms of the ELBO at c = 1 (R/locus_core.R:270-275; R/elbo.R:43-54,
        114-118), evaluated where rowSums(gam) lives; two scalars come back:
        sum_j rs_gam log om + (d - rs_gam) log(1 - om)  and  e_omega_."""
        if getattr(self, "_omega_cache", None) is not None:     # came with the last reduction
            return self._omega_cache
        s_bg, e_om = self._omega_terms_t().cpu().numpy()
        return float(s_bg), float(e_om)

    def _omega_terms_t(self):
        import torch
        d = self.d
        rs, a, b = self.rs_gam_t, self.a_t, self.b_t
        a_vb = a + rs                                    # update_a_vb / update_b_vb with c = 1
        b_vb = b - rs + d
        dg = torch.digamma(a_vb + b_vb)
        log_om_vb = torch.digamma(a_vb) - dg
        log_1_min_om_vb = torch.digamma(b_vb) - dg
        lbeta = lambda x, y: torch.lgamma(x) + torch.lgamma(y) - torch.lgamma(x + y)  # noqa: E731
        s_bg = (rs * log_om_vb).sum() + ((d - rs) * log_1_min_om_vb).sum()
        e_om = ((a - a_vb) * log_om_vb + (b - b_vb) * log_1_min_om_vb - lbeta(a, b)
                + lbeta(a_vb, b_vb)).sum()
        return torch.stack([s_bg, e_om])

    def _log_om_terms(self, c):
        """log_om_vb, log_1_min_om_vb (R/update_vb.R:187,190) on the device of rowSums(gam)."""
        import torch
        d = self.d
        if c not in self._digam_sum:
            self._digam_sum = {c: torch.digamma(c * (self.a_t + self.b_t + d) - 2 * c + 2)}
        digam_sum = self._digam_sum[c]
        rs = self.rs_gam_t
        return (torch.digamma(c * (self.a_t + rs) - c + 1) - digam_sum,
                torch.digamma(c * (self.b_t - rs + d) - c + 1) - digam_sum)

    def set_initial(self, stats, rs_gam_local):
        """R/locus_core.R:36-42: statistics of the initial gam_vb / mu_beta_vb."""
        self.stats = np.asarray(stats, dtype=np.float64)
        t_loc = float(self.tau_vb @ self.stats[1])
        sc = self._reduce_p(rs_gam_local, [t_loc])
        self.sum_tau_m2 = float(sc[0])

    # -- top of the iteration: R/locus_core.R:57-84 ----------------------------
    def begin_iteration(self, c=1.0):
        n, d = self.n, self.d
        cs_gam, cs_m2, cs_b2, rss = self.stats
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam, c)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2, c)
        self.sig2_inv_vb = lambda_vb / nu_vb
        eta_vb = U.update_eta_vb_(n, self.eta, cs_gam, c)
        kappa_vb = U.update_kappa_vb_(n, self.kappa, rss, cs_m2, cs_b2, self.sig2_inv_vb, c)
        self.tau_vb = eta_vb / kappa_vb
        self.sig2_beta_vb = U.update_sig2_beta_vb_(n, self.sig2_inv_vb, self.tau_vb, c)
        log_tau_vb = U.update_log_tau_vb_(eta_vb, kappa_vb)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        log_om_vb, log_1_min_om_vb = self._log_om_terms(c)
        s2 = self.sig2_beta_vb
        return dict(
            lo=log_om_vb, l1=log_1_min_om_vb, log_sig2_inv_vb=float(log_sig2_inv_vb),
            log_tau_vb=log_tau_vb, tau_vb=self.tau_vb, sig2_beta_vb=s2, c=float(c),
            # per-response constants of the sweep kernel (src/coreLoop.cpp:34,42)
            scale=c * s2 * self.tau_vb, hinv=1.0 / (2.0 * s2),
            cst=-(log_tau_vb + log_sig2_inv_vb + np.log(s2)) / 2, s2k=s2)

    # -- after the sweep: R/locus_core.R:95,169-197 and elbo_ :261-292 ---------
    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True):
        n, d = self.n, self.d
        self.stats = np.asarray(stats, dtype=np.float64)
        self._last = (rs_gam_local, ent_local)
        cs_gam, cs_m2, cs_b2, rss = self.stats
        tau, s2, si = self.tau_vb, self.sig2_beta_vb, self.sig2_inv_vb
        t_loc = float(tau @ cs_m2)
        abc_loc = 0.0
        if with_elbo:
            eta_vb = U.update_eta_vb_(n, self.eta, cs_gam)
            kappa_vb = U.update_kappa_vb_(n, self.kappa, rss, cs_m2, cs_b2, si)
            log_tau_vb = digamma(eta_vb) - np.log(kappa_vb)
            abc_loc = (E.e_y_(n, self.kappa, kappa_vb, log_tau_vb, cs_m2, si, tau)
                       + E.e_beta_gamma_local_(cs_gam, cs_m2, ent_local, log_tau_vb, s2, si, tau)
                       + E.e_tau_(self.eta, eta_vb, self.kappa, kappa_vb, log_tau_vb, tau))
        import torch
        sc = self._reduce_p(rs_gam_local, [t_loc, abc_loc], omega=with_elbo)
        self.sum_tau_m2 = float(sc[0])
        self._c_last = c
        if not with_elbo:
            return None
        s_bg, e_om = self._omega_terms()
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        log_sig2_inv_vb = digamma(lambda_vb) - np.log(nu_vb)
        return float(sc[1] + log_sig2_inv_vb * self.sum_gam / 2 + s_bg
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si)
                     + e_om)


class IdentityTail(IdentityHost):
    """IdentityHost with every vector and scalar of the iteration held as a tensor on the device of
    rowSums(gam) (the GPU; CPU tensors under the CPU test backend): the O(d_loc) updates
    (R/locus_core.R:57-84), the packed all-reduce and the ELBO (:261-292, R/elbo.R) are queued
    behind the sweep without a host round trip.  An iteration costs no H2D copy and ONE
    device-to-host read, of the ELBO (none while annealing).  Same interface as IdentityHost;
    begin_iteration returns device tensors and end_iteration takes them."""

    device_resident = True

    def _t(self, a, dev):
        import torch
        return torch.from_numpy(np.array(a, dtype=np.float64, copy=True)).to(dev)

    def _to_device(self, dev):
        import math

        import torch
        if getattr(self, "_dev", None) == dev:
            return
        self._dev = dev
        self.a_t, self.b_t = self._t(self.a, dev), self._t(self.b, dev)
        self.eta_t = self._t(np.broadcast_to(self.eta, self.tau_vb_init.shape), dev)
        self.kappa_t = self._t(np.broadcast_to(self.kappa, self.tau_vb_init.shape), dev)
        self.tau_t = self._t(self.tau_vb_init, dev)
        self.s2_t = self._t(self.s2_init, dev)
        # constants of e_tau_ / e_sig2_inv_ / e_omega_ (R/elbo.R:114-143)
        self._e_tau_const = (self.eta_t * torch.log(self.kappa_t) - torch.lgamma(self.eta_t)).sum()
        self._e_si_const = self.lam * math.log(self.nu) - math.lgamma(self.lam)
        self._lbeta_ab = (torch.lgamma(self.a_t) + torch.lgamma(self.b_t)
                          - torch.lgamma(self.a_t + self.b_t)).sum()
        self._digam_sum = {}

    def __init__(self, n, p, d_total, hyper, tau_vb, sig2_beta_vb, comm=None):
        super().__init__(n, p, d_total, hyper, tau_vb, sig2_beta_vb, comm)
        self.tau_vb_init = np.array(tau_vb, dtype=np.float64).reshape(-1)
        self.s2_init = np.broadcast_to(np.asarray(sig2_beta_vb, dtype=np.float64),
                                       self.tau_vb_init.shape).copy()
        self.si_t = self.sum_gam_t = self.sum_tau_m2_t = self.stats_t = None

    # host views of the device-resident quantities (outputs, R/locus_core.R:242-252)
    tau_vb = property(lambda self: self.tau_t.cpu().numpy() if getattr(self, "tau_t", None)
                      is not None else self.tau_vb_init, lambda self, v: None)
    sig2_beta_vb = property(lambda self: self.s2_t.cpu().numpy() if getattr(self, "s2_t", None)
                            is not None else self.s2_init, lambda self, v: None)
    sig2_inv_vb = property(lambda self: None if self.si_t is None else float(self.si_t),
                           lambda self, v: None)
    sum_gam = property(lambda self: None if self.sum_gam_t is None else float(self.sum_gam_t),
                       lambda self, v: None)
    sum_tau_m2 = property(lambda self: None if self.sum_tau_m2_t is None
                          else float(self.sum_tau_m2_t), lambda self, v: None)

    def _reduce_t(self, rs_gam_local, scalars_t):
        """ONE all-reduce of [rowSums(gam) (p) | scalars]; everything stays on the device."""
        import torch
        if self.comm.solo:
            self.rs_gam_t = rs_gam_local.reshape(-1)
            out = scalars_t
        else:
            buf = torch.cat([rs_gam_local.reshape(-1), scalars_t.reshape(-1)])
            self.comm.all_reduce_sum_(buf)
            self.rs_gam_t, out = buf[: self.p], buf[self.p:]
        self.sum_gam_t = self.rs_gam_t.sum()
        return out

    def set_initial(self, stats, rs_gam_local):
        import torch
        self._to_device(rs_gam_local.device)
        self.stats_t = torch.as_tensor(stats, dtype=torch.float64).to(self._dev)
        sc = self._reduce_t(rs_gam_local, (self.tau_t @ self.stats_t[1]).reshape(1))
        self.sum_tau_m2_t = sc[0]

    def _lambda_nu(self, c=1.0):
        lambda_vb = c * (self.lam + self.sum_gam_t / 2) - c + 1          # R/update_vb.R:198
        nu_vb = c * (self.nu + self.sum_tau_m2_t / 2)                    # :204
        return lambda_vb, nu_vb

    def _eta_kappa(self, si, c=1.0):
        n = self.n
        cs_gam, cs_m2, cs_b2, rss = self.stats_t
        eta_vb = c * (self.eta_t + n / 2 + cs_gam / 2) - c + 1           # :220
        kappa_vb = c * (self.kappa_t + (rss + (n - 1 + si) * cs_m2 - (n - 1) * cs_b2) / 2)  # :229-237
        return eta_vb, kappa_vb

    def prepare_iteration(self, c=1.0):
        """The updates at the top of an iteration (R/locus_core.R:57-84) from the statistics of
        the last sweep, WITHOUT making them the host's current tau_vb / sig2_beta_vb /
        sig2_inv_vb: the engine queues this behind the ELBO of the previous iteration, before it
        knows whether another iteration follows, and commits it when one does."""
        import torch
        n = self.n
        lambda_vb, nu_vb = self._lambda_nu(c)
        si = lambda_vb / nu_vb
        eta_vb, kappa_vb = self._eta_kappa(si, c)
        tau = eta_vb / kappa_vb
        s2 = 1 / (c * (n - 1 + si) * tau)                                 # :93-104
        log_tau_vb = torch.digamma(eta_vb) - torch.log(kappa_vb)          # :275
        log_sig2_inv_vb = torch.digamma(lambda_vb) - torch.log(nu_vb)     # :213
        lo, l1 = self._log_om_terms(c)
        return dict(lo=lo, l1=l1, log_sig2_inv_vb=log_sig2_inv_vb, log_tau_vb=log_tau_vb,
                    tau_vb=tau, sig2_beta_vb=s2, sig2_inv_vb=si, c=float(c),
                    # per-response constants of the sweep kernel (src/coreLoop.cpp:34,42)
                    scale=c * s2 * tau, hinv=1.0 / (2.0 * s2),
                    cst=-(log_tau_vb + log_sig2_inv_vb + torch.log(s2)) / 2, s2k=s2)

    def commit_iteration(self, prm):
        self.tau_t, self.s2_t, self.si_t = prm["tau_vb"], prm["sig2_beta_vb"], prm["sig2_inv_vb"]
        return prm

    def begin_iteration(self, c=1.0):                                     # R/locus_core.R:57-84
        return self.commit_iteration(self.prepare_iteration(c))

    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True, sync=True):
        """sync=False returns the ELBO as a 0-dim device tensor (nothing waits for the GPU)."""
        import math

        import torch
        n, d = self.n, self.d
        self.stats_t = stats if torch.is_tensor(stats) else self._t(stats, self._dev)
        ent = ent_local if torch.is_tensor(ent_local) else torch.tensor(
            float(ent_local), dtype=torch.float64, device=self._dev)
        self._last = (rs_gam_local, ent)
        cs_gam, cs_m2, cs_b2, rss = self.stats_t
        tau, s2, si = self.tau_t, self.s2_t, self.si_t
        parts = [(tau @ cs_m2).reshape(1)]
        if with_elbo:                                                    # R/elbo.R, summed over k
            eta_vb, kappa_vb = self._eta_kappa(si)
            log_tau_vb = torch.digamma(eta_vb) - torch.log(kappa_vb)
            e_y = (-n / 2 * math.log(2 * math.pi) + n / 2 * log_tau_vb
                   - tau * (kappa_vb - cs_m2 * si / 2 - self.kappa_t)).sum()          # :189-200
            e_bg = (cs_gam * log_tau_vb / 2 - cs_m2 * tau * si / 2
                    + cs_gam * (torch.log(s2) + 1) / 2).sum() - ent.reshape(())       # :43-54
            e_tau = ((self.eta_t - eta_vb) * log_tau_vb - (self.kappa_t - kappa_vb) * tau
                     - eta_vb * torch.log(kappa_vb) + torch.lgamma(eta_vb)).sum() \
                + self._e_tau_const                                                  # :138-143
            parts.append((e_y + e_bg + e_tau).reshape(1))
        sc = self._reduce_t(rs_gam_local, torch.cat(parts))
        self.sum_tau_m2_t = sc[0]
        self._c_last = c
        if not with_elbo:
            return None
        s_bg, e_om = self._omega_terms_t()                               # :270-275
        lambda_vb, nu_vb = self._lambda_nu()
        log_sig2_inv_vb = torch.digamma(lambda_vb) - torch.log(nu_vb)
        e_si = ((self.lam - lambda_vb) * log_sig2_inv_vb - (self.nu - nu_vb) * si
                - lambda_vb * torch.log(nu_vb) + torch.lgamma(lambda_vb)) + self._e_si_const  # :125-130
        lb = sc[1] + log_sig2_inv_vb * self.sum_gam_t / 2 + s_bg + e_si + e_om
        return float(lb) if sync else lb                                 # the iteration's one D2H

    def held_out_elbo_(self, n_test, rss_test):
        n_fit, st_fit = self.n, self.stats_t
        try:
            self.n = int(n_test)
            st = st_fit.clone()
            st[3] = self._t(rss_test, self._dev)
            return self.end_iteration(st, *self._last)
        finally:
            self.n, self.stats_t = n_fit, st_fit

    def _omega_terms_t(self):
        import torch
        d = self.d
        rs, a, b = self.rs_gam_t, self.a_t, self.b_t
        a_vb = a + rs                                    # update_a_vb / update_b_vb with c = 1
        b_vb = b - rs + d
        dg = torch.digamma(a_vb + b_vb)
        log_om_vb = torch.digamma(a_vb) - dg
        log_1_min_om_vb = torch.digamma(b_vb) - dg
        s_bg = (rs * log_om_vb).sum() + ((d - rs) * log_1_min_om_vb).sum()
        e_om = ((a - a_vb) * log_om_vb + (b - b_vb) * log_1_min_om_vb + torch.lgamma(a_vb)
                + torch.lgamma(b_vb) - torch.lgamma(a_vb + b_vb)).sum() - self._lbeta_ab
        return s_bg, e_om


class FusedTail(IdentityTail):
    """IdentityTail with the steady-state tail in three HIP kernels (csrc/tail.hip,
    `locus_hip_identity_tail_dev`) instead of ~150 tensor kernels: phase 0 behind the sweep, the
    all-reduce, phase 1 -- ELBO of the iteration that ended AND the constants of the next one, into
    one of two buffer sets that alternate, so that what the running sweep reads is never
    overwritten.  The first iteration's set-up (set_initial / the first prepare_iteration) and any
    prepare_iteration the engine did not announce use the tensor path of the base class."""

    wants_ent_vector = True

    def __init__(self, *args, dv=None, **kw):
        super().__init__(*args, **kw)
        self._dvl = dv
        self._sets = None

    def _alloc(self):
        import torch
        dev, f64 = self._dev, torch.float64
        d_loc = self.tau_t.numel()
        mk = lambda *shape: torch.zeros(shape, dtype=f64, device=dev)  # noqa: E731
        self._sets = [dict(vec=mk(5, d_loc), lp=mk(2, self.p), scal=mk(8)) for _ in range(2)]
        self._buf = mk(self.p + 2)
        self._part = mk(192)
        self._turn = 0
        self._next = None

    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True, sync=True,
                      c_next=None):
        import torch
        from . import _lib
        if self._sets is None:
            self._alloc()
        dvl = self._dvl
        self.stats_t = stats
        self._last = (rs_gam_local, ent_local)
        st = self._sets[self._turn]
        self._turn ^= 1
        world1 = self.comm.solo
        a = _lib.TailArgs()
        a.n, a.p, a.d_loc, a.d_total = self.n, self.p, self.tau_t.numel(), self.d
        a.a, a.b = self.a_t.data_ptr(), self.b_t.data_ptr()
        a.eta, a.kappa = self.eta_t.data_ptr(), self.kappa_t.data_ptr()
        a.lam, a.nu = self.lam, self.nu
        a.stats = stats.data_ptr()
        tau, s2, si = self.tau_t.contiguous(), self.s2_t.contiguous(), self.si_t
        a.tau, a.s2, a.si_in = tau.data_ptr(), s2.data_ptr(), si.data_ptr()
        a.ent = ent_local.data_ptr() if (with_elbo and ent_local is not None) else None
        a.shard2 = self._buf[self.p:].data_ptr()
        a.scal = st["scal"].data_ptr()
        a.with_elbo = 1 if with_elbo else 0
        a.c_next = float(c_next if c_next is not None else 1.0)
        a.part = self._part.data_ptr()
        dvl.identity_tail(a, 0)
        if world1:
            self.rs_gam_t = rs_gam_local.reshape(-1)
        else:
            self._buf[: self.p].copy_(rs_gam_local.reshape(-1))
            self.comm.all_reduce_sum_(self._buf)
            self.rs_gam_t = self._buf[: self.p]
        a.rs_gam = self.rs_gam_t.data_ptr()
        if c_next is not None:
            a.lo, a.l1 = st["lp"][0].data_ptr(), st["lp"][1].data_ptr()
            v = st["vec"]
            a.tau_next, a.s2_next, a.scale, a.hinv, a.cst = (v[i].data_ptr() for i in range(5))
        dvl.identity_tail(a, 1)
        self._keep = (tau, s2, si, stats, ent_local, rs_gam_local)  # alive until the kernels ran
        scal = st["scal"]
        self.sum_gam_t, self.sum_tau_m2_t = scal[1], scal[2]
        self._c_last = c
        self._next = None
        if c_next is not None:
            v = st["vec"]
            self._next = dict(lo=st["lp"][0], l1=st["lp"][1], log_sig2_inv_vb=scal[5], tau_vb=v[0],
                              sig2_beta_vb=v[1], sig2_inv_vb=scal[4], c=float(c_next), scale=v[2],
                              hinv=v[3], cst=v[4], s2k=v[1])
        if not with_elbo:
            return None
        return float(scal[0]) if sync else scal[0]

    def prepare_iteration(self, c=1.0):
        nxt, self._next = getattr(self, "_next", None), None
        if nxt is not None and nxt["c"] == float(c):
            return nxt
        return super().prepare_iteration(c)

    def held_out_elbo_(self, n_test, rss_test):
        n_fit, st_fit = self.n, self.stats_t
        # the fitted sig2_inv_vb may live in the buffer set this call is about to write
        self.si_t = self.si_t.clone()
        try:
            self.n = int(n_test)
            st = st_fit.clone()
            st[3] = self._t(rss_test, self._dev)
            return self.end_iteration(st, *self._last)
        finally:
            self.n, self.stats_t = n_fit, st_fit


class IdentityEngine:
    """One GPU's shard of the identity-link model."""

    def held_out_elbo(self, Y_test_loc, X_test, Zmu_test=None):
        """elbo_ with the fitted variational parameters on held-out rows (R/cross_validate.R:
        313-322): mat_x_m1 = X_test beta_vb on the matrix cores, the residual sum of squares per
        response, then the host's ELBO with n = nrow(Y_test)."""
        t, dv = self.torch, self.dv
        nt, p, d = Y_test_loc.shape[0], self.p, self.d_loc
        Xt = dv.to_device_colmajor(X_test, self.device)
        Yt = dv.to_device_colmajor(Y_test_loc, self.device)
        M = t.empty((d, nt), dtype=t.float64, device=self.device).t()
        dv.gemm(nt, d, p, Xt, 1, nt, self.state[2], d, 1, M, 1, nt)
        R = dv.residual(Yt, M, Zmu_test)
        return self.host.held_out_elbo_(nt, (R * R).sum(dim=0).cpu().numpy())

    def __init__(self, px, Y_dev, state, host, device, cfg=None, dv=None):
        import torch
        if dv is None:
            from . import device as dv
        self.torch, self.dv = torch, dv
        self.px, self.Y, self.state, self.host, self.device = px, Y_dev, state, host, device
        self.n, self.d_loc = Y_dev.shape
        self.p = px.p
        assert tuple(state.shape) == (3, self.p, self.d_loc)
        self.sweeper = dv.Sweeper(px, self.d_loc, device, cfg=cfg)
        self.stats = torch.empty((4, self.d_loc), dtype=torch.float64, device=device)
        self.R = None

    def _up(self, a):
        if self.torch.is_tensor(a):
            return a
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def _up_many(self, *arrays):
        """Several host arrays in ONE host-to-device copy: returns device views of their shapes (the
        per-iteration constants of the drivers whose tail runs on the host)."""
        host = [i for i, a in enumerate(arrays) if not self.torch.is_tensor(a)]   # tensors pass through
        flat = [np.ascontiguousarray(arrays[i], dtype=np.float64).reshape(-1) for i in host]
        out = list(arrays)
        if host:
            buf = self.torch.from_numpy(np.concatenate(flat)).to(self.device)
            o = 0
            for i, f in zip(host, flat):
                out[i] = buf[o:o + f.size].reshape(np.shape(arrays[i]))
                o += f.size
        return out

    def initialise(self):
        """beta = gam * mu, mat_x_m1 = X beta (fp64 MFMA GEMM), residual, statistics
        (R/locus_core.R:36-42)."""
        t, dv = self.torch, self.dv
        n, p, d = self.n, self.p, self.d_loc
        s2k = self._up(self.host.sig2_beta_vb)
        cols = dv.state_cols(self.state, p, d, s2k=s2k)  # also writes the beta plane
        M = t.empty((d, n), dtype=t.float64, device=self.device).t()
        # M (n x d, col-major) = X (n x p) beta (p x d); X is read from its packed copy
        dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.state[2], d, 1, M, 1, n)
        self.R = dv.residual(self.Y, M)
        del M
        rss = (self.R * self.R).sum(dim=0)
        rs_gam, _ = dv.gamma_rows(self.state[0], p, d)
        stats = t.cat([cols, rss.reshape(1, d)], dim=0)
        if not getattr(self.host, "device_resident", False):
            stats = stats.cpu().numpy()
        self.host.set_initial(stats, rs_gam)

    def iterate(self, shuffled_ind, c=1.0, with_elbo=True, after_launch=None, c_next=None):
        """One pass of the loop body R/locus_core.R:49-211; returns the ELBO (or None).
        `after_launch()` runs on the host once the kernels of this iteration are queued and before
        the host waits for their statistics (e.g. drawing the next sweep order).  `c_next`
        (device-resident host side only): annealing factor of the iteration that would follow;
        its top-of-iteration updates are then queued before the host waits for this ELBO."""
        dv = self.dv
        resident = getattr(self.host, "device_resident", False)
        nxt = getattr(self, "_prm_next", None)
        self._prm_next = None
        if resident and nxt is not None and nxt["c"] == float(c):
            prm = self.host.commit_iteration(nxt)
        else:
            prm = self.host.begin_iteration(c)
        pre, self._pre_ind = getattr(self, "_pre_ind", None), None
        if pre is not None and pre[0] is shuffled_ind:
            ind = pre[1]           # uploaded, and its operands packed, while the last sweep ran
        else:
            ind = shuffled_ind if self.torch.is_tensor(shuffled_ind) else self._up(shuffled_ind)
        self.sweeper.identity(ind, self._up(prm["lo"]), self._up(prm["l1"]),
                              self._up(prm["scale"]), self._up(prm["hinv"]), self._up(prm["cst"]),
                              self._up(prm["s2k"]), c, self.R, self.state, self.stats)
        rs_gam, ent = dv.gamma_rows(self.state[0], self.p, self.d_loc,
                                    eps=self.host.ent_eps if with_elbo else None)
        tail = ent.sum().reshape(1) if with_elbo else self.torch.zeros(1, dtype=self.torch.float64,
                                                                      device=self.device)
        if resident:
            # everything after the sweep is queued on the device; the ELBO is the one read-back
            if getattr(self.host, "wants_ent_vector", False):
                lb = self.host.end_iteration(self.stats, rs_gam, ent if with_elbo else None, c,
                                             with_elbo, sync=False, c_next=c_next)
            else:
                lb = self.host.end_iteration(self.stats, rs_gam, tail, c, with_elbo, sync=False)
            if c_next is not None:
                self._prm_next = self.host.prepare_iteration(c_next)
            if after_launch is not None:
                self._pack_ahead(after_launch())
            self.sweeper.status()
            return None if lb is None else float(lb)
        packed = self.torch.cat([self.stats.reshape(-1), tail])
        if after_launch is not None:
            self._pack_ahead(after_launch())
        h = packed.cpu().numpy()
        self.sweeper.status()
        stats = h[:-1].reshape(4, self.d_loc)
        return self.host.end_iteration(stats, rs_gam, float(h[-1]), c, with_elbo)

    def _pack_ahead(self, next_ind):
        """`after_launch` may return the NEXT iteration's sweep order: whatever of that sweep
        depends only on X and the order (permutation check, block-packed operands) is then queued
        on a side stream now, beside the running sweep (Sweeper.prepack).  The caller passes the
        same object to the next iterate()."""
        if next_ind is None or not hasattr(self.sweeper, "prepack"):
            return
        dev_ind = self.sweeper.prepack(next_ind)
        if dev_ind is not None:
            self._pre_ind = (next_ind, dev_ind)

    def iterate_scheme(self, scheme, c=1.0, with_elbo=True, c_next=None, k_order=None,
                       j_orders=None):
        """One iteration of the reference's internal schemes (R/locus_core.R:97-161; "used only
        internally, convergence not ensured"):
          "x-y"  coreBatch: every (j, k) at once from the residual of the last iteration -- two
                 GEMMs on the matrix cores around the element-wise update (src/coreBatch.cpp:19-43);
          "x"    the same update response by response (:97-117); the responses do not interact
                 inside an iteration, so the result is that of "x-y" whatever the order;
          "0"    response by response with log_om_vb refreshed from rowSums(gam_vb) in between
                 (:135-161): d sweeps of one response each, in the order k_order, response k over
                 the covariates in the order j_orders(k).  Sequential over the responses: one GPU."""
        dv, t, host = self.dv, self.torch, self.host
        n, p, d = self.n, self.p, self.d_loc
        resident = getattr(host, "device_resident", False)
        nxt = getattr(self, "_prm_next", None)
        self._prm_next = None
        if resident and nxt is not None and nxt["c"] == float(c):
            prm = host.commit_iteration(nxt)
        else:
            prm = host.begin_iteration(c)
        up = self._up
        if scheme in ("x", "x-y"):
            # "x" anneals (R/locus_core.R:104-111: c scales mu_beta_vb and the argument of
            # log_one_plus_exp_); "x-y" = coreBatch has no annealing factor and locus_core_
            # refuses that combination up front (:123-124)
            if scheme == "x-y" and float(c) != 1.0:
                raise ValueError("Annealing not implemented for this scheme. Exit.")
            log_si = float(prm["log_sig2_inv_vb"])
            # T = X'(Y - mat_x_m1) into the mu plane, then mu, gam, beta element-wise
            dv.gemm(p, d, n, self.px.Xp, self.px.ld, 1, self.R, 1, n, self.state[1], d, 1)
            cstk = -up(prm["cst"]) - log_si / 2              # log_tau / 2 + log(sig2_beta) / 2
            dv.batch_update(self.state, up(prm["lo"]), up(prm["l1"]), up(prm["scale"]),
                            up(prm["hinv"]), cstk, log_si, n - 1, 0, p, d, cfac=float(c))
            M = t.empty((d, n), dtype=t.float64, device=self.device).t()
            dv.gemm(n, d, p, self.px.Xp, 1, self.px.ld, self.state[2], d, 1, M, 1, n)
            self.R = dv.residual(self.Y, M)
            self.stats[:3] = dv.state_cols(self.state, p, d, s2k=up(prm["s2k"]))
            self.stats[3] = (self.R * self.R).sum(dim=0)
        elif scheme == "0":
            if host.comm.world != 1:
                raise NotImplementedError('scheme "0" is sequential over the responses: one rank')
            if getattr(self, "_sw1", None) is None:
                self._sw1 = dv.Sweeper(self.px, 1, self.device)
                self._stats1 = t.empty((4, 1), dtype=t.float64, device=self.device)
            scale, hinv, cst, s2k = (up(prm[key]) for key in ("scale", "hinv", "cst", "s2k"))
            for k in k_order:
                k = int(k)
                lo, l1 = host._log_om_terms(c)              # from the CURRENT rowSums(gam_vb), :139
                st1 = self.state[:, :, k].contiguous()      # [3][p] = [3][p][1]
                g_old = st1[0].clone()
                self._sw1.identity(up(j_orders(k)), lo, l1, scale[k:k + 1], hinv[k:k + 1],
                                   cst[k:k + 1], s2k[k:k + 1], c, self.R[:, k], st1, self._stats1)
                self._sw1.status()
                self.state[:, :, k] = st1
                self.stats[:, k] = self._stats1[:, 0]
                host.rs_gam_t = host.rs_gam_t + (st1[0] - g_old)      # :157
        else:
            raise ValueError("Batch scheme not defined. Exit.")
        rs_gam, ent = dv.gamma_rows(self.state[0], p, d, eps=host.ent_eps if with_elbo else None)
        if resident and getattr(host, "wants_ent_vector", False):
            lb = host.end_iteration(self.stats, rs_gam, ent if with_elbo else None, c, with_elbo,
                                    c_next=c_next)
        elif resident:
            tail = ent.sum().reshape(1) if with_elbo else t.zeros(1, dtype=t.float64,
                                                                  device=self.device)
            lb = host.end_iteration(self.stats, rs_gam, tail, c, with_elbo)
        else:
            lb = host.end_iteration(self.stats.cpu().numpy(), rs_gam,
                                    float(ent.sum()) if with_elbo else 0.0, c, with_elbo)
        if resident and c_next is not None:
            self._prm_next = host.prepare_iteration(c_next)
        return lb

    def download(self):
        """gam_vb, mu_beta_vb, beta_vb (p x d_loc) and mat_x_m1 (n x d_loc) as numpy, R layout;
        each is fetched when first asked for."""
        return LazyFields(gam_vb=lambda: to_host_F(self.state[0]),
                          mu_beta_vb=lambda: to_host_F(self.state[1]),
                          beta_vb=lambda: to_host_F(self.state[2]),
                          mat_x_m1=lambda: to_host_F(self.dv.residual(self.Y, self.R)))  # M = Y - R


def resolve_backend(test_backend, device):
    """(device layer, torch device).  The product always gets the HIP layer (and fails loudly
    without a GPU); a test may inject an emulation of locus_amd.device on CPU tensors."""
    import torch
    if test_backend is not None:
        return test_backend, torch.device("cpu")
    from . import device as dv
    dv.require_gpu()
    return dv, device or torch.device("cuda", torch.cuda.current_device())


def checkpoint_(it, checkpoint_path, gam_vb_fn, converged, lb_new, lb_old, om_vb_fn, rate=100):
    """R/utils.R:316-336 (npz instead of RData): every `rate` iterations, keep the last two.
    gam_vb_fn / om_vb_fn are called only when a file is written (both download from the GPU)."""
    if checkpoint_path is not None and it % rate == 0:
        np.savez(os.path.join(checkpoint_path, f"tmp_output_it_{it}.npz"), gam_vb=gam_vb_fn(),
                 converged=converged, it=it, lb_new=lb_new, diff_lb=abs(lb_new - lb_old),
                 om_vb=om_vb_fn())
        old = os.path.join(checkpoint_path, f"tmp_output_it_{it - 2 * rate}.npz")
        if os.path.exists(old):
            os.remove(old)


def checkpoint_clean_up_(checkpoint_path):  # R/utils.R:339-352
    if checkpoint_path is not None:
        for f in glob.glob(os.path.join(checkpoint_path, "tmp_output_it_*")):
            os.remove(f)


def locus_core_(Y, X, list_hyper, gam_vb, mu_beta_vb, sig2_beta_vb, tau_vb, tol, maxit,
                anneal=None, verbose=False, batch="y", full_output=False, debug=True,
                checkpoint_path=None, perm_fn=None, comm=None, device=None, trace=None,
                held_out=None, k_perm_fn=None, perm0_fn=None, _test_backend=None):
    """R/locus_core.R:8-255 on the GPU.  Y (n x d) must be centred and X (n x p) standardised.

    `held_out = (Y_test, X_test)` adds `lb_held_out`, the ELBO of the fit on those rows (the
    cross-validation objective, R/cross_validate.R:313-322).

    `perm_fn(it)` supplies the sweep order of iteration `it` (float64, 0-based); by default
    numpy draws it (R: sample(0:(p-1)), :88).  With a DistComm every rank passes the FULL
    arrays and keeps only its response shard on its GPU; outputs are gathered on every rank.

    `_test_backend` is a test hook, never set by the product: tests/cpu_device.py emulates the
    device layer on CPU tensors so that the host arithmetic and the d-sharding can run under
    gloo without a GPU.  Without it the HIP library is required (no fallback).
    """
    dv, device = resolve_backend(_test_backend, device)
    import torch
    if batch not in ("y", "x", "x-y", "0"):
        raise ValueError("Batch scheme not defined. Exit.")                 # :159-161
    if batch == "x-y" and anneal is not None:
        raise ValueError("Annealing not implemented for this scheme. Exit.")  # :123-124
    comm = comm or Comm()
    Y = np.asarray(Y, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    n, d = Y.shape
    p = X.shape[1]
    k_lo, k_hi = comm.shard(d)
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731
    if batch in ("x", "0"):
        # the internal schemes of R/locus_core.R:97-161: sample(1:d) per iteration (:102,137) and,
        # for "0", sample(1:p) per response (:142)
        if k_perm_fn is None:
            rng_k = np.random.default_rng(1)
            k_perm_fn = lambda it: rng_k.permutation(d)  # noqa: E731
        if perm0_fn is None:
            rng_0 = np.random.default_rng(2)
            perm0_fn = lambda it, k: rng_0.permutation(p).astype(np.float64)  # noqa: E731

    hyper_loc = dict(list_hyper)
    hyper_loc["eta"] = np.broadcast_to(np.asarray(list_hyper["eta"], dtype=np.float64),
                                       (d,))[k_lo:k_hi]
    hyper_loc["kappa"] = np.broadcast_to(np.asarray(list_hyper["kappa"], dtype=np.float64),
                                         (d,))[k_lo:k_hi]
    host_cls = FusedTail if hasattr(dv, "identity_tail") else IdentityTail  # CPU test backend: tensors
    host = host_cls(n, p, d, hyper_loc, np.asarray(tau_vb)[k_lo:k_hi],
                    np.asarray(sig2_beta_vb)[k_lo:k_hi], comm, **(
                        dict(dv=dv) if host_cls is FusedTail else {}))
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, k_lo:k_hi], device)
    state = torch.empty((3, p, k_hi - k_lo), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, slice(k_lo, k_hi), device)
    state[1] = columns_to_device(mu_beta_vb, slice(k_lo, k_hi), device)
    eng = IdentityEngine(px, Y_dev, state, host, device, dv=dv)
    eng.initialise()

    if anneal is None:
        annealing, c, ladder = False, 1.0, None
    else:
        annealing, ladder = True, U.get_annealing_ladder_(anneal)
        c = float(ladder[0])

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nxt = {}
    while (not converged) and it < maxit:                                # :49
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        ind = nxt.pop(it) if it in nxt else np.ascontiguousarray(perm_fn(it), dtype=np.float64)

        def draw_next(it=it):                                            # overlaps the sweep
            if it < maxit:
                nxt[it + 1] = np.ascontiguousarray(perm_fn(it + 1), dtype=np.float64)
                return nxt[it + 1] if batch == "y" else None             # packed ahead by the engine
        c_next = (float(ladder[it]) if it < len(ladder) else 1.0) if annealing else 1.0
        if batch == "y":
            lb = eng.iterate(ind, c, with_elbo=not annealing, after_launch=draw_next,
                             c_next=c_next if it < maxit else None)
        else:
            lb = eng.iterate_scheme(
                batch, c, with_elbo=not annealing, c_next=c_next if it < maxit else None,
                k_order=k_perm_fn(it) if batch in ("x", "0") else None,
                j_orders=(lambda k, it=it: np.ascontiguousarray(perm0_fn(it, k), dtype=np.float64)))
        if annealing:                                                    # :177-190
            c = c_next
            if np.isclose(c, 1.0, rtol=1.5e-8, atol=0):
                annealing = False
        else:
            lb_new = lb
            if trace is not None:
                trace.append((it, lb_new))
            if verbose and (it == 1 or it % 5 == 0):
                print(f"ELBO = {lb_new}\n")
            if debug and lb_new + EPS_SQRT < lb_old:                     # :202-203
                raise RuntimeError("ELBO not increasing monotonically. Exit. ")
            converged = abs(lb_new - lb_old) < tol                       # :205
            checkpoint_(it, checkpoint_path, lambda: _gather(comm, eng.download()["gam_vb"], d),
                        converged, lb_new, lb_old, lambda: host.om_vb)
    checkpoint_clean_up_(checkpoint_path)
    if verbose and not converged:
        print("Maximal number of iterations reached before convergence. Exit.")

    loc = eng.download()
    out = dict(beta_vb=_gather(comm, loc["beta_vb"], d), gam_vb=_gather(comm, loc["gam_vb"], d),
               om_vb=host.om_vb, converged=converged, it=it, lb_opt=lb_new,
               diff_lb=abs(lb_new - lb_old), annealing=False if anneal is None else anneal[0])
    if held_out is not None:
        Yt = np.asarray(held_out[0], dtype=np.float64)
        out["lb_held_out"] = eng.held_out_elbo(Yt[:, k_lo:k_hi], held_out[1])
    if full_output:                                                      # :242-247
        out.update(mu_beta_vb=_gather(comm, loc["mu_beta_vb"], d),
                   mat_x_m1=_gather(comm, loc["mat_x_m1"], d),
                   sig2_beta_vb=_gather_vec(comm, host.sig2_beta_vb, d),
                   tau_vb=_gather_vec(comm, host.tau_vb, d), sig2_inv_vb=host.sig2_inv_vb,
                   a_vb=host.a_vb, b_vb=host.b_vb, sum_gam=host.sum_gam)
    return out


def _gather(comm, local, d):
    """Concatenate the response shards (columns) of every rank, on every rank: ONE all-gather of
    the shards themselves (rank r contributes rows x d_r doubles; nothing is zero-padded or
    summed)."""
    if comm.solo:
        return local
    return comm.all_gather_columns(np.asarray(local, dtype=np.float64), d)


def _gather_vec(comm, local, d):
    if comm.solo:
        return local
    return _gather(comm, np.asarray(local, dtype=np.float64).reshape(1, -1), d).reshape(-1)
