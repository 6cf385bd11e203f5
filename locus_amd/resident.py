"""The resident-context entry points of the C ABI (include/locus_hip.h, section 3) from Python,
and locus_core_ written against them exactly as an R maintainer would write it in R: every
length-p / length-d update on the host (R/update_vb.R, R/elbo.R), one ctx_sweep per iteration.

The drivers of this package keep their vectors on the GPU as well (locus_core.py); this module is
the reference-side view of the boundary -- host arrays only, no torch -- and its parity test is the
end-to-end check of that part of the ABI."""
import ctypes

import numpy as np

from . import _lib
from . import update_vb as U
from .locus_core import EPS_MACH, EPS_SQRT, IdentityHost


def _f(a):
    a = np.asarray(a, dtype=np.float64)
    return a if a.flags.f_contiguous else np.asfortranarray(a)


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class ResidentContext:
    """locus_hip_ctx_*: X, Y and the p x d state live on the GPU until close()."""

    def __init__(self, X, Y):
        X, Y = _f(X), _f(Y)
        self.n, self.p = X.shape
        self.d = Y.shape[1]
        if Y.shape[0] != self.n:
            raise ValueError("X and Y row counts differ")
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().locus_hip_ctx_create(_p(X), _p(Y), self.n, self.p, self.d,
                                                   ctypes.byref(h)))
        self._h = h

    def set_state(self, gam_vb, mu_beta_vb, mat_z_mu=None):
        g, m = _f(gam_vb), _f(mu_beta_vb)
        if g.shape != (self.p, self.d) or m.shape != (self.p, self.d):
            raise ValueError("gam_vb and mu_beta_vb must be p x d")
        z = None if mat_z_mu is None else _f(mat_z_mu)
        _lib.check(_lib.lib().locus_hip_ctx_set_state(self._h, _p(g), _p(m), _p(z)))

    def set_response(self, Y_or_W=None, mat_z_mu=None):
        """New latent W and / or Z alpha_vb between iterations (coreZLoop / coreProbitLoop
        drivers); the resident gam_vb / mu_beta_vb / mat_x_m1 are kept."""
        y = None if Y_or_W is None else _f(Y_or_W)
        z = None if mat_z_mu is None else _f(mat_z_mu)
        for a in (y, z):
            if a is not None and a.shape != (self.n, self.d):
                raise ValueError("Y_or_W and mat_z_mu must be n x d")
        _lib.check(_lib.lib().locus_hip_ctx_set_response(self._h, _p(y), _p(z)))

    def _outs(self, ent):
        return (np.empty(self.p), np.empty((4, self.d)), ctypes.c_double(0.0) if ent else None)

    def stats(self, sig2_beta_vb, ent_eps=None):
        rs, cs, e = self._outs(ent_eps is not None)
        s2 = np.ascontiguousarray(np.broadcast_to(sig2_beta_vb, (self.d,)), dtype=np.float64)
        _lib.check(_lib.lib().locus_hip_ctx_stats(self._h, _p(s2), float(ent_eps or 0.0), _p(rs),
                                                  _p(cs), ctypes.byref(e) if e is not None else None))
        return rs, cs, (e.value if e is not None else None)

    def sweep(self, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb, sig2_beta_vb, tau_vb,
              shuffled_ind, c=1.0, ent_eps=None):
        """coreLoop (R/RcppExports.R:16-18 argument meaning) on the resident state; returns
        rowSums(gam_vb), the [4][d] column sums and the entropy sum (None without ent_eps)."""
        v = lambda a, k: np.ascontiguousarray(np.broadcast_to(a, (k,)), dtype=np.float64)  # noqa: E731
        rs, cs, e = self._outs(ent_eps is not None)
        args = [v(log_om_vb, self.p), v(log_1_min_om_vb, self.p), v(log_tau_vb, self.d),
                v(sig2_beta_vb, self.d), v(tau_vb, self.d), v(shuffled_ind, self.p)]
        _lib.check(_lib.lib().locus_hip_ctx_sweep(
            self._h, _p(args[0]), _p(args[1]), float(log_sig2_inv_vb), _p(args[2]), _p(args[3]),
            _p(args[4]), _p(args[5]), float(c), float(ent_eps or 0.0), _p(rs), _p(cs),
            ctypes.byref(e) if e is not None else None))
        return rs, cs, (e.value if e is not None else None)

    def get_state(self, names=("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1")):
        shp = dict(gam_vb=(self.p, self.d), mu_beta_vb=(self.p, self.d), beta_vb=(self.p, self.d),
                   mat_x_m1=(self.n, self.d))
        out = {k: np.empty(shp[k], order="F") for k in names}
        _lib.check(_lib.lib().locus_hip_ctx_get_state(
            self._h, *[_p(out.get(k)) for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1")]))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:     # also safe when __init__ raised
            _lib.lib().locus_hip_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def locus_core_ctx_(Y, X, list_hyper, gam_vb, mu_beta_vb, sig2_beta_vb, tau_vb, tol, maxit,
                    anneal=None, perm_fn=None, full_output=False, trace=None):
    """R/locus_core.R:8-255 with the native calls replaced by the resident context: the shape of
    the change an R maintainer makes."""
    import torch
    Y, X = _f(Y), _f(X)
    n, d = Y.shape
    p = X.shape[1]
    if perm_fn is None:
        rng = np.random.default_rng(0)
        perm_fn = lambda it: rng.permutation(p).astype(np.float64)  # noqa: E731
    host = IdentityHost(n, p, d, list_hyper, tau_vb, sig2_beta_vb)
    with ResidentContext(X, Y) as ctx:
        ctx.set_state(gam_vb, mu_beta_vb)                                  # :36-42
        rs, cs, _ = ctx.stats(host.sig2_beta_vb)
        host.set_initial(cs, torch.from_numpy(rs))
        annealing, c, ladder = False, 1.0, None
        if anneal is not None:
            annealing, ladder = True, U.get_annealing_ladder_(anneal)
            c = float(ladder[0])
        converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
        while (not converged) and it < maxit:                              # :49
            lb_old = lb_new
            it += 1
            prm = host.begin_iteration(c)                                  # :57-84
            rs, cs, ent = ctx.sweep(prm["lo"].numpy(), prm["l1"].numpy(), prm["log_sig2_inv_vb"],
                                    prm["log_tau_vb"], prm["sig2_beta_vb"], prm["tau_vb"],
                                    perm_fn(it), c, None if annealing else EPS_MACH)   # :88-95
            lb = host.end_iteration(cs, torch.from_numpy(rs), ent or 0.0, c, not annealing)
            if annealing:                                                  # :177-190
                c = float(ladder[it]) if it < len(ladder) else 1.0
                if np.isclose(c, 1.0, rtol=1.5e-8, atol=0):
                    annealing = False
            else:
                lb_new = lb
                if trace is not None:
                    trace.append((it, lb_new))
                if lb_new + EPS_SQRT < lb_old:                             # :202-203
                    raise RuntimeError("ELBO not increasing monotonically. Exit. ")
                converged = abs(lb_new - lb_old) < tol                     # :205
        st = ctx.get_state(("gam_vb", "beta_vb") + (("mu_beta_vb", "mat_x_m1") if full_output else ()))
    out = dict(beta_vb=st["beta_vb"], gam_vb=st["gam_vb"], om_vb=host.om_vb, converged=converged,
               it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old))
    if full_output:
        out.update(mu_beta_vb=st["mu_beta_vb"], mat_x_m1=st["mat_x_m1"],
                   sig2_beta_vb=host.sig2_beta_vb, tau_vb=host.tau_vb, sig2_inv_vb=host.sig2_inv_vb)
    return out
