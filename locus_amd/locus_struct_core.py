"""Structured-prior CAVI driver with device-resident state (mirrors R/locus_struct_core.R).

The sweep is coreStructLoop (src/coreStructLoop.cpp:18-52) = the identity-family kernel with
c = 1 and per-covariate terms log Phi(theta_j), log(1 - Phi(theta_j)).  Everything that
involves theta (W, the block matvecs, e_theta_) needs only rowSums(gam_vb), so it is
replicated host arithmetic on the all-reduced vector.
"""
import numpy as np

from . import elbo as E
from . import update_vb as U
from .comm import Comm
from .hostio import LazyFields, columns_to_device, to_host_F
from .locus_core import HostBase, IdentityEngine, _gather, _gather_vec, resolve_backend
from .struct import LOG_SQRT_2PI, StructBlocks

EPS_STRUCT = np.finfo(np.float64).eps ** 0.75  # R/elbo.R:94


class StructHost(HostBase):
    """Replicated / shard-local vector arithmetic of locus_struct_core_
    (R/locus_struct_core.R:51-145).  Same interface as IdentityHost.

    Everything of length p -- theta_vb, log Phi(theta_vb), rowSums(W), the per-block products with
    S0_inv and sig2_theta_vb (80 MB of matrices at C5, read once per iteration) -- lives on the
    device that holds the all-reduced rowSums(gam_vb); only scalars come back to the host."""

    ent_eps = EPS_STRUCT

    def __init__(self, n, p, d_total, hyper, tau_vb, sig2_beta_vb, blocks, comm=None):
        self.n, self.p, self.d = int(n), int(p), int(d_total)
        self.comm = comm or Comm()
        self.eta = np.asarray(hyper["eta"], dtype=np.float64)      # shard-local
        self.kappa = np.asarray(hyper["kappa"], dtype=np.float64)  # shard-local
        self.lam, self.nu = float(hyper["lambda"]), float(hyper["nu"])
        self.m0 = np.asarray(hyper["m0"], dtype=np.float64)
        self.blocks = blocks
        self.tau_vb = np.array(tau_vb, dtype=np.float64)
        self.sig2_beta_vb = np.array(sig2_beta_vb, dtype=np.float64)
        self.sig2_inv_vb = None
        self.sum_gam = self.sum_tau_m2 = None
        self.stats = None
        self.rs_gam_t = self.theta_t = self._lp = None

    # -- length-p state on the device ---------------------------------------------------------
    def _on_device(self, dev):
        import torch
        if self.theta_t is not None and self.theta_t.device == dev:
            return
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.m0_t = up(self.m0)
        self.theta_t = self.m0_t.clone()                           # :25
        self._lp = None
        self.groups_t = [(up(pos.astype(np.int64)), up(S), up(V)) for pos, S, V in self.blocks.groups]
        self.S_m0_t = torch.empty_like(self.m0_t)                  # S0_inv_bl m0_bl, constant
        for pos, S, _ in self.groups_t:
            self.S_m0_t[pos] = torch.bmm(S, self.m0_t[pos].unsqueeze(-1)).squeeze(-1)

    @property
    def theta_vb(self):
        return self.theta_t.cpu().numpy()

    @property
    def rs_gam(self):
        return self.rs_gam_t.cpu().numpy()

    def _reduce(self, rs_gam_local, scalars):
        """ONE all-reduce of [rowSums(gam) (p) | scalars]; rowSums stays on its device."""
        import torch
        self._on_device(rs_gam_local.device)
        tail = torch.as_tensor(np.asarray(scalars, dtype=np.float64).reshape(-1),
                               device=rs_gam_local.device)
        buf = torch.cat([rs_gam_local.reshape(-1), tail])
        self.comm.all_reduce_sum_(buf)
        self.rs_gam_t = buf[: self.p]
        return buf[self.p:]

    def _log_phi(self):
        """log Phi(theta_vb), log(1 - Phi(theta_vb)): needed by the sweep, by W and by the ELBO;
        evaluated once per value of theta_vb."""
        import torch
        if self._lp is None:
            self._lp = (torch.special.log_ndtr(self.theta_t), torch.special.log_ndtr(-self.theta_t))
        return self._lp

    def set_initial(self, stats, rs_gam_local):
        import torch
        self.stats = np.asarray(stats, dtype=np.float64)
        sc = self._reduce(rs_gam_local, [float(self.tau_vb @ self.stats[1])])
        h = torch.cat([sc, self.rs_gam_t.sum().reshape(1)]).cpu().numpy()
        self.sum_tau_m2, self.sum_gam = float(h[0]), float(h[1])

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
        lp, l1p = self._log_phi()
        return dict(lo=lp, l1=l1p, c=1.0,
                    scale=s2 * self.tau_vb, hinv=1.0 / (2.0 * s2),
                    cst=-(log_tau_vb + log_sig2_inv_vb + np.log(s2)) / 2, s2k=s2)

    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True):
        import torch
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
        sc = self._reduce(rs_gam_local, [float(tau @ cs_m2), loc])
        rs, th = self.rs_gam_t, self.theta_t
        # W and theta (:127-129): only rowSums(W) enters, W_jk = gam_jk (imr1_j - imr0_j) +
        # theta_j + imr0_j with the inverse Mills ratios of R/utils.R:201-220
        lp, l1p = self._log_phi()
        dens = -th * th / 2 - LOG_SQRT_2PI
        imr0 = -torch.exp(dens - l1p)
        imr0 = torch.where(imr0 > -th, -th, imr0)
        imr1 = torch.exp(dens - lp)
        imr1 = torch.where(imr1 < -th, -th, imr1)
        x = (imr1 - imr0) * rs + d * (th + imr0) + self.S_m0_t
        # theta_bl = sig2_theta_bl (rowSums(W_bl) + S0_inv_bl m0_bl)  (R/update_vb.R:315-319)
        theta = torch.empty_like(th)
        for pos, _, V in self.groups_t:
            theta[pos] = torch.bmm(V, x[pos].unsqueeze(-1)).squeeze(-1)
        self.theta_t, self._lp = theta, None
        lp, l1p = self._log_phi()                                         # of the NEW theta_vb
        # replicated part of elbo_struct_, with the NEW theta_vb (:132-135; R/elbo.R:98-104,
        # 166-180)
        dlt = theta - self.m0_t
        quad = torch.zeros((), dtype=torch.float64, device=th.device)
        for pos, S, _ in self.groups_t:
            db = dlt[pos]
            quad = quad + (db * torch.bmm(S, db.unsqueeze(-1)).squeeze(-1)).sum()
        h = torch.cat([sc, torch.stack([rs.sum(), (rs * lp).sum() + ((d - rs) * l1p).sum(),
                                        quad])]).cpu().numpy()
        self.sum_tau_m2, self.sum_gam = float(h[0]), float(h[2])
        lambda_vb = U.update_lambda_vb_(self.lam, self.sum_gam)
        nu_vb = U.update_nu_vb_(self.nu, self.sum_tau_m2)
        log_sig2_inv_vb = U.update_log_sig2_inv_vb_(lambda_vb, nu_vb)
        glob = (log_sig2_inv_vb * self.sum_gam / 2 + float(h[3])
                - d * self.blocks.sum_diag_sig2_theta / 2)
        return float(h[1] + glob + self.blocks.e_theta_const - float(h[4]) / 2
                     + E.e_sig2_inv_(self.lam, lambda_vb, log_sig2_inv_vb, self.nu, nu_vb, si))


class StructTail(StructHost):
    """StructHost whose per-response arithmetic, scalars and ELBO run in csrc/tail.hip
    (`locus_hip_identity_tail_dev`, phases 0 and 2) behind the sweep: the response-wise sums of
    the ELBO and the constants of the next sweep never visit the host, theta_vb and its terms stay
    tensor operations (R/update_vb.R:315-319, R/elbo.R:98-104,166-180), and an iteration ends with
    ONE read-back, of the ELBO.  The first iteration's set-up uses the numpy path of the base class."""

    device_resident = True
    wants_ent_vector = True

    def __init__(self, *args, dv=None, **kw):
        super().__init__(*args, **kw)
        self._dvl, self._sets, self._next = dv, None, None
        self.tau_t = self.s2_t = self.si_t = None

    def _host(name, tname):   # host views of the device-resident quantities (outputs)
        def get(self):
            tt = getattr(self, tname, None)
            if tt is None:
                return self.__dict__.get("_" + name)
            return float(tt) if tt.dim() == 0 else tt.cpu().numpy()

        def put(self, v):
            self.__dict__["_" + name] = v
        return property(get, put)

    tau_vb = _host("tau_vb", "tau_t")
    sig2_beta_vb = _host("sig2_beta_vb", "s2_t")
    sig2_inv_vb = _host("sig2_inv_vb", "si_t")
    del _host

    def set_initial(self, stats, rs_gam_local):
        import torch
        st = stats.cpu().numpy() if torch.is_tensor(stats) else stats
        super().set_initial(st, rs_gam_local)

    def begin_iteration(self, c=1.0):
        """First iteration (and any iteration the engine did not announce): the numpy path of the
        base class, its results moved to the device."""
        import torch
        if self._next is not None and self._next["c"] == float(c):
            return self.commit_iteration(self.prepare_iteration(c))
        self.tau_t = self.s2_t = self.si_t = None         # the properties fall back to numpy
        prm = super().begin_iteration(c)
        dev = self.theta_t.device
        up = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True)).to(dev)  # noqa: E731
        out = dict(lo=prm["lo"], l1=prm["l1"], c=1.0, scale=up(prm["scale"]), hinv=up(prm["hinv"]),
                   cst=up(prm["cst"]), s2k=up(prm["s2k"]), tau_vb=up(self.tau_vb),
                   sig2_beta_vb=up(self.sig2_beta_vb),
                   sig2_inv_vb=torch.tensor(float(self.sig2_inv_vb), dtype=torch.float64, device=dev))
        return self.commit_iteration(out)

    def commit_iteration(self, prm):
        self.tau_t, self.s2_t, self.si_t = prm["tau_vb"], prm["sig2_beta_vb"], prm["sig2_inv_vb"]
        return prm

    def prepare_iteration(self, c=1.0):
        nxt, self._next = self._next, None
        assert nxt is not None and nxt["c"] == float(c), "prepare_iteration without end_iteration"
        return nxt

    def end_iteration(self, stats, rs_gam_local, ent_local, c=1.0, with_elbo=True, sync=True,
                      c_next=None):
        import torch
        from . import _lib
        dev, f64 = rs_gam_local.device, torch.float64
        self._on_device(dev)
        d_loc, d = stats.shape[1], self.d
        if self._sets is None:
            mk = lambda *shape: torch.zeros(shape, dtype=f64, device=dev)  # noqa: E731
            self._sets = [dict(vec=mk(5, d_loc), scal=mk(8)) for _ in range(2)]
            self._buf, self._part, self._turn = mk(self.p + 2), mk(192), 0
            up = lambda a: torch.from_numpy(np.array(np.broadcast_to(a, (d_loc,)), dtype=np.float64,  # noqa: E731
                                                     copy=True)).to(dev)
            self.eta_t, self.kappa_t = up(self.eta), up(self.kappa)
        st = self._sets[self._turn]
        self._turn ^= 1
        a = _lib.TailArgs()
        a.n, a.p, a.d_loc, a.d_total = self.n, self.p, d_loc, d
        a.eta, a.kappa, a.lam, a.nu = self.eta_t.data_ptr(), self.kappa_t.data_ptr(), self.lam, self.nu
        a.stats = stats.data_ptr()
        tau, s2, si = self.tau_t.contiguous(), self.s2_t.contiguous(), self.si_t
        a.tau, a.s2, a.si_in = tau.data_ptr(), s2.data_ptr(), si.data_ptr()
        a.ent = ent_local.data_ptr() if ent_local is not None else None
        a.shard2, a.scal = self._buf[self.p:].data_ptr(), st["scal"].data_ptr()
        a.with_elbo, a.c_next, a.part = 1, 1.0, self._part.data_ptr()
        self._dvl.identity_tail(a, 0)                     # shard-local sums of elbo_struct_ (:201-221)
        if self.comm.solo:
            self.rs_gam_t = rs_gam_local.reshape(-1)
        else:
            self._buf[: self.p].copy_(rs_gam_local.reshape(-1))
            self.comm.all_reduce_sum_(self._buf)
            self.rs_gam_t = self._buf[: self.p]
        rs, th = self.rs_gam_t, self.theta_t
        # W and theta (:127-129), as in StructHost.end_iteration
        lp, l1p = self._log_phi()
        dens = -th * th / 2 - LOG_SQRT_2PI
        imr0 = -torch.exp(dens - l1p)
        imr0 = torch.where(imr0 > -th, -th, imr0)
        imr1 = torch.exp(dens - lp)
        imr1 = torch.where(imr1 < -th, -th, imr1)
        x = (imr1 - imr0) * rs + d * (th + imr0) + self.S_m0_t
        theta = torch.empty_like(th)
        for pos, _, V in self.groups_t:
            theta[pos] = torch.bmm(V, x[pos].unsqueeze(-1)).squeeze(-1)
        self.theta_t, self._lp = theta, None
        lp, l1p = self._log_phi()                         # of the NEW theta_vb
        dlt = theta - self.m0_t
        quad = torch.zeros((), dtype=f64, device=dev)
        for pos, S, _ in self.groups_t:
            db = dlt[pos]
            quad = quad + (db * torch.bmm(S, db.unsqueeze(-1)).squeeze(-1)).sum()
        # {sum(gam_vb), covariate terms of e_beta_gamma_ (with log Phi for log omega), e_theta_}
        self._part[:3] = torch.stack([
            rs.sum(), (rs * lp).sum() + ((d - rs) * l1p).sum(),
            self.blocks.e_theta_const - quad / 2 - d * self.blocks.sum_diag_sig2_theta / 2])
        if c_next is not None:
            v = st["vec"]
            a.tau_next, a.s2_next, a.scale, a.hinv, a.cst = (v[i].data_ptr() for i in range(5))
        self._dvl.identity_tail(a, 2)                     # scalars, ELBO, next sweep's constants
        self._keep = (tau, s2, si, stats, ent_local, rs_gam_local)
        scal = st["scal"]
        self._next = None
        if c_next is not None:
            v = st["vec"]
            self._next = dict(lo=lp, l1=l1p, c=1.0, scale=v[2], hinv=v[3], cst=v[4], s2k=v[1],
                              tau_vb=v[0], sig2_beta_vb=v[1], sig2_inv_vb=scal[4])
        self._sum_t = (scal[1], scal[2])
        return float(scal[0]) if sync else scal[0]

    sum_gam = property(lambda self: float(self._sum_t[0]) if getattr(self, "_sum_t", None)
                       else self.__dict__.get("_sum_gam"),
                       lambda self, v: self.__dict__.__setitem__("_sum_gam", v))
    sum_tau_m2 = property(lambda self: float(self._sum_t[1]) if getattr(self, "_sum_t", None)
                          else self.__dict__.get("_sum_tau_m2"),
                          lambda self, v: self.__dict__.__setitem__("_sum_tau_m2", v))


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

    # :30 -- on the GPU when there is one (the CPU test backend keeps the numpy path)
    blocks = StructBlocks(d, list_struct["vec_fac_st"], list_hyper["s02"], X,
                          device=device if device.type == "cuda" else None)
    hyper_loc = dict(list_hyper)
    for key in ("eta", "kappa"):
        hyper_loc[key] = np.broadcast_to(np.asarray(list_hyper[key], dtype=np.float64),
                                         (d,))[k_lo:k_hi]
    fused = hasattr(dv, "identity_tail")          # the CPU test backend keeps the numpy host side
    host = (StructTail if fused else StructHost)(
        n, p, d, hyper_loc, np.asarray(tau_vb)[k_lo:k_hi], np.asarray(sig2_beta_vb)[k_lo:k_hi],
        blocks, comm, **(dict(dv=dv) if fused else {}))
    px = dv.PackedX(dv.to_device_colmajor(X, device))
    Y_dev = dv.to_device_colmajor(Y[:, k_lo:k_hi], device)
    state = torch.empty((3, p, k_hi - k_lo), dtype=torch.float64, device=device)
    state[0] = columns_to_device(gam_vb, slice(k_lo, k_hi), device)
    state[1] = columns_to_device(mu_beta_vb, slice(k_lo, k_hi), device)
    eng = IdentityEngine(px, Y_dev, state, host, device, dv=dv)
    eng.initialise()

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nxt = {}
    while (not converged) and it < maxit:                                # :51
        lb_old = lb_new
        it += 1
        if verbose and (it == 1 or it % 5 == 0):
            print(f"Iteration {it}... ")
        ind = nxt.pop(it) if it in nxt else np.ascontiguousarray(perm_fn(it), dtype=np.float64)

        def draw_next(it=it):                                            # overlaps the sweep
            if it < maxit:
                nxt[it + 1] = np.ascontiguousarray(perm_fn(it + 1), dtype=np.float64)
                return nxt[it + 1]                                       # packed ahead by the engine
        lb_new = eng.iterate(ind, 1.0, after_launch=draw_next,
                             **(dict(c_next=1.0 if it < maxit else None) if fused else {}))
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
