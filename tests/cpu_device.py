"""TEST INFRASTRUCTURE: an emulation of locus_amd.device (the HIP device layer) on CPU torch
tensors, built on numpy and the C oracle.  It lets the drivers' host arithmetic, their
orchestration and the response sharding run without a GPU (world_size 1 and 2 under gloo),
through the `_test_backend` hook of the driver functions.  Never imported by the product.

Same function names, argument order and in-place semantics as locus_amd/device.py; the sweeps
call the oracle's coreLoop / coreLogitLoop on the resident residual (Y := R, mat_x_m1 := 0).
"""
import numpy as np
import torch
from scipy.special import log_ndtr

from oracle import c_oracle

F64 = torch.float64


def to_device_colmajor(a, device):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a))
    return torch.from_numpy(np.ascontiguousarray(a.T)).t()


def _view(t, shape, strides):
    return torch.as_strided(t, shape, strides, storage_offset=t.storage_offset()).numpy()


class PackedX:
    def __init__(self, X_dev):
        self.n, self.p = X_dev.shape
        self.ld = (self.n + 127) // 128 * 128
        self.Xp = torch.zeros((self.p, self.ld), dtype=F64)
        self.Xp[:, : self.n] = X_dev.t()
        self.xtx = (self.Xp * self.Xp).sum(dim=1)
        self.X = np.asfortranarray(X_dev.numpy())


def _state_views(state):
    """gam, mu, beta as writable numpy views of the [3][p][d] planes (C order)."""
    return state[0].numpy(), state[1].numpy(), state[2].numpy()


class Sweeper:
    def __init__(self, px, d, device, cfg=None):
        self.px, self.d = px, int(d)
        self.bad = False

    def _check(self, ind):
        v = ind.numpy()
        p = self.px.p
        self.bad = not (np.all(v >= 0) and np.all(v < p) and np.unique(v.astype(np.int64)).size == v.size)
        return not self.bad

    def identity(self, shuffled_ind, lo, l1, scale, hinv, cst, s2k, cfac, R, state, stats=None,
                 nsteps=0):
        if not self._check(shuffled_ind):
            return
        n, d = R.shape
        gam, mu, beta = (np.asfortranarray(a) for a in _state_views(state))
        s2 = s2k.numpy()
        tau = scale.numpy() / (cfac * s2)
        log_tau = -2.0 * cst.numpy() - np.log(s2)          # log_sig2_inv := 0
        Rn = np.asfortranarray(R.numpy())
        M = np.zeros((n, d), order="F")
        c_oracle.coreLoop(self.px.X, Rn, gam, lo.numpy(), l1.numpy(), 0.0, log_tau, beta, M, mu, s2,
                          tau, shuffled_ind.numpy(), float(cfac))
        self._store(R, Rn - M, state, gam, mu, beta)
        if stats is not None:
            st = stats.numpy()
            st[0] = gam.sum(axis=0)
            st[1] = (gam * (mu ** 2 + s2[None, :])).sum(axis=0)
            st[2] = (beta ** 2).sum(axis=0)
            st[3] = (R.numpy() ** 2).sum(axis=0)

    def logit(self, shuffled_ind, lo, l1, s2, xwx, W2, half_lsi, R, state, stats=None, nsteps=0,
              ls2=None):
        if not self._check(shuffled_ind):
            return
        n, d = R.shape
        gam, mu, beta = (np.asfortranarray(a) for a in _state_views(state))
        Rn = np.asfortranarray(R.numpy())
        M = np.zeros((n, d), order="F")
        Z0 = np.zeros((n, d), order="F")
        psi = np.asfortranarray(W2.numpy() / 2)
        c_oracle.coreLogitLoop(self.px.X, Rn, gam, lo.numpy(), l1.numpy(), 2.0 * float(half_lsi),
                               beta, M, Z0, mu, psi, np.asfortranarray(s2.numpy()),
                               shuffled_ind.numpy())
        self._store(R, Rn - W2.numpy() * M, state, gam, mu, beta)

    @staticmethod
    def _store(R, Rnew, state, gam, mu, beta):
        R.numpy()[...] = Rnew
        for q, a in enumerate((gam, mu, beta)):
            state[q].numpy()[...] = a

    def status(self):
        if self.bad:
            from locus_amd._lib import LocusHipError
            raise LocusHipError(-3, "shuffled_ind is not a permutation")


def residual(Y, A, Zmu=None, W2=None, out=None, to_residual=True):
    n, d = Y.shape
    if out is None:
        out = torch.empty((d, n), dtype=F64).t()
    y, a = Y.numpy(), A.numpy()
    z = Zmu.numpy() if Zmu is not None else 0.0
    if W2 is not None:
        w = W2.numpy()
        res = (y - w * (a + z)) if to_residual else ((y - a) / w - z)
    else:
        res = y - a - z
    out.numpy()[...] = res
    return out


def gamma_rows(gam_plane, p, d, eps=None):
    g = gam_plane.numpy()
    rs = torch.from_numpy(g.sum(axis=1))
    ent = None
    if eps is not None:
        ent = torch.from_numpy((g * np.log(g + eps) + (1 - g) * np.log(1 - g + eps)).sum(axis=1))
    return rs, ent


def state_cols(state, p, d, s2k=None, s2jk=None, nchunks=64):
    gam, mu, beta = _state_views(state)
    beta[...] = gam * mu
    s2 = s2jk.numpy() if s2jk is not None else s2k.numpy()[None, :]
    return torch.from_numpy(np.stack([gam.sum(axis=0), (gam * (mu ** 2 + s2)).sum(axis=0),
                                      (beta ** 2).sum(axis=0)]))


def gemm(m, n, k, A, sa_m, sa_k, B, sb_k, sb_n, C, sc_m, sc_n, alpha=1.0, beta=0.0, sqA=False,
         sqB=False):
    a = _view(A, (m, k), (sa_m, sa_k))
    b = _view(B, (k, n), (sb_k, sb_n))
    c = _view(C, (m, n), (sc_m, sc_n))
    if sqA:
        a = a ** 2
    if sqB:
        b = b ** 2
    c[...] = alpha * (a @ b) + (beta * c if beta != 0.0 else 0.0)
    return C


def m2_minus_b2(state, p, d, D, s2jk=None, s2c=0.0):
    gam, mu, _ = _state_views(state)
    s2 = s2jk.numpy() if s2jk is not None else s2c
    D.numpy()[...] = (mu ** 2 + s2) * gam - (gam * mu) ** 2
    return D


def s2_from_xwx(xwx, sig2_inv, s2):
    s2.numpy()[...] = 1.0 / (xwx.numpy() + sig2_inv)
    return s2


def half_log(src, out):
    out.numpy()[...] = 0.5 * np.log(src.numpy())
    return out


def logit_chi_psi(Q, M, Zmu, Y, Z, s2a, chi_old, W2_old, chi_new, W2_new, ey, n, d, q):
    m, zm = M.numpy(), Zmu.numpy()
    c2 = Q.numpy() + m ** 2 + (Z.numpy() ** 2) @ s2a.numpy() + zm ** 2 + 2 * m * zm
    if ey is not None:
        co, po = chi_old.numpy(), W2_old.numpy() / 2
        ey.numpy()[...] = (-np.log1p(np.exp(-co)) + Y.numpy() * (m + zm) - co / 2
                           - po * (c2 - co ** 2)).sum(axis=0)
    c = np.sqrt(c2)
    chi_new.numpy()[...] = c
    W2_new.numpy()[...] = 2 * ((1 / (1 + np.exp(-c)) - 0.5) / (2 * c))


def logit_sig2_alpha(Z, W2, zeta2_inv, s2a, n, d, q):
    s2a.numpy()[...] = 1.0 / ((Z.numpy() ** 2).T @ W2.numpy() + zeta2_inv.numpy()[:, None])


def alpha_step(Yw, W2, M, Zmu, Zi, s2a_i, alpha_i, n, d):
    z = Zi.numpy()
    zm = Zmu.numpy() - np.outer(z, alpha_i.numpy())
    if W2 is not None:
        t = Yw.numpy() - W2.numpy() * (zm + M.numpy())
    else:
        t = Yw.numpy() - zm - M.numpy()
    a_new = s2a_i.numpy() * (t.T @ z)
    alpha_i.numpy()[...] = a_new
    Zmu.numpy()[...] = zm + np.outer(z, a_new)


def probit_w(Y, M, Zmu, Q, zs, W, ey, n, d):
    u = M.numpy() + Zmu.numpy()
    y = Y.numpy()
    lp, lq = log_ndtr(u), log_ndtr(-u)
    dl = -u ** 2 / 2 - 0.5 * np.log(2 * np.pi)
    m1 = np.exp(dl - lp)
    m1 = np.where(m1 < -u, -u, m1)
    m0 = -np.exp(dl - lq)
    m0 = np.where(m0 > -u, -u, m0)
    W.numpy()[...] = u + np.where(y == 1, m1, m0)
    if Q is not None:
        ey.numpy()[...] = (y * lp + (1 - y) * lq - Q.numpy() / 2
                           - zs.numpy()[:, None] / 2).sum(axis=0)


def bin_rows(state, p, d, eps, s2jk=None, s2c=1.0):
    gam, mu, _ = _state_views(state)
    s2 = s2jk.numpy() if s2jk is not None else np.full((1, 1), s2c)
    out = np.stack([gam.sum(axis=1),
                    (gam * np.log(gam + eps) + (1 - gam) * np.log(1 - gam + eps)).sum(axis=1),
                    (gam * np.log(s2)).sum(axis=1),
                    (gam * (mu ** 2 + s2)).sum(axis=1)])
    return torch.from_numpy(out)
