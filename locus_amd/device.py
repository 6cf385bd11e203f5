"""Device-resident side of the drivers: torch tensors own the HBM, liblocus_hip does the
work (section 2 of include/locus_hip.h).  torch is used for allocation, streams and
torch.distributed only.

Layouts:
  Xp     [p][ld]      packed, zero padded columns of X (replicated on every GPU)
  R      n x d_loc    column-major residual  Y - mat_x_m1 (- mat_z_mu)   | logit: Y - 2 psi (M + Zmu)
  state  [3][p][d_loc] planes gam_vb, mu_beta_vb, beta_vb, response-contiguous
  stats  [4][d_loc]   per-response sums produced by the sweep epilogue
"""
import ctypes

import numpy as np
import torch

from . import _lib

F64 = torch.float64


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise RuntimeError("locus_amd needs a HIP device (no CPU fallback)")


def colmajor(t):
    """n x d tensor whose storage is column-major (R layout): a transposed view of [d][n]."""
    return t.t().contiguous().t()


def to_device_colmajor(a, device):
    """numpy (any order) -> device tensor with column-major storage, shape kept."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    t = torch.from_numpy(np.ascontiguousarray(a.T)).to(device)  # [d][n] row-major
    return t.t()  # n x d view, column-major storage


def is_colmajor(t):
    return t.dim() == 2 and t.stride(0) == 1 and t.stride(1) == t.shape[0]


class PackedX:
    """X packed once per data set: Xp [p][ld] and xtx[j] = X[,j]'X[,j]."""

    def __init__(self, X_dev):
        assert is_colmajor(X_dev) and X_dev.dtype == F64
        self.n, self.p = X_dev.shape
        self.ld = int(_lib.lib().locus_hip_packed_ld(self.n))
        self.Xp = torch.empty((self.p, self.ld), dtype=F64, device=X_dev.device)
        self.xtx = torch.empty((self.p,), dtype=F64, device=X_dev.device)
        _lib.check(_lib.lib().locus_hip_pack_x_dev(_p(X_dev), self.n, self.p, _p(self.Xp),
                                                   _p(self.xtx), _stream()))


class Sweeper:
    """Runs CAVI sweeps on device-resident state (wraps locus_hip_sweep_dev)."""

    def __init__(self, px, d, device, cfg=None):
        self.px = px
        self.n, self.p, self.d = px.n, px.p, int(d)
        nbytes = int(_lib.lib().locus_hip_sweep_workspace_bytes(self.n, self.p))
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.cfg = cfg
        self._keep = None
        self.device = device
        self._pre = None          # pack-ahead state, allocated by the first prepack()

    # ---- operands of the NEXT sweep packed while the current one runs ------------------------
    def prepack(self, shuffled_ind_next):
        """Queue, on a side stream, the part of the next sweep that depends only on X and the
        sweep order (locus_hip_sweep_prepack_dev) into the workspace the running sweep does NOT
        use.  `shuffled_ind_next`: the order as a host array (uploaded on the side stream, so
        that nothing waits for the running sweep) or a device tensor.  identity() given the
        returned device tensor picks the packed operands up.  Returns None when the shape is not
        swept by the matrix-core kernels (nothing queued)."""
        if self.cfg is not None and int(self.cfg[0]) != -1:
            return None
        pre = self._pre
        if pre is None:
            nb = int(_lib.lib().locus_hip_sweep_prepack_workspace_bytes(self.n, self.p, self.d))
            if nb <= 0:
                self._pre = False
                return None
            pre = self._pre = dict(
                ws=[torch.empty((nb,), dtype=torch.uint8, device=self.device) for _ in range(2)],
                side=torch.cuda.Stream(device=self.device), turn=0, used=[None, None], ready=None)
        if pre is False:
            return None
        main = torch.cuda.current_stream(self.device)
        slot = pre["turn"]
        pre["turn"] ^= 1
        side = pre["side"]
        if pre["used"][slot] is not None:          # the sweep that last read this workspace
            side.wait_event(pre["used"][slot])
        with torch.cuda.stream(side):
            ind = shuffled_ind_next
            if not torch.is_tensor(ind):
                ind = torch.from_numpy(np.ascontiguousarray(ind, dtype=np.float64)).to(self.device)
            a = _lib.SweepArgs()
            a.n, a.p, a.d = self.n, self.p, self.d
            a.Xp, a.xtx = self.px.Xp.data_ptr(), self.px.xtx.data_ptr()
            a.shuffled_ind = ind.data_ptr()
            a.workspace, a.workspace_bytes = pre["ws"][slot].data_ptr(), pre["ws"][slot].numel()
            if self.cfg is not None:
                for i in range(4):
                    a.cfg[i] = int(self.cfg[i])
            rc = _lib.lib().locus_hip_sweep_prepack_dev(ctypes.byref(a), side.cuda_stream)
            if rc == -2:                           # LOCUS_HIP_EUNSUPPORTED: sweep as usual
                self._pre = False
                return None
            _lib.check(rc)
            done = torch.cuda.Event()
            done.record(side)
        ind.record_stream(main)
        pre["ready"] = (ind, slot, done)
        return ind

    def _args(self, shuffled_ind, lo, l1, R, state, stats, nsteps=0):
        a = _lib.SweepArgs()
        a.n, a.p, a.d = self.n, self.p, self.d
        a.Xp, a.xtx = self.px.Xp.data_ptr(), self.px.xtx.data_ptr()
        a.shuffled_ind, a.lo, a.l1 = shuffled_ind.data_ptr(), lo.data_ptr(), l1.data_ptr()
        a.R, a.state = R.data_ptr(), state.data_ptr()
        a.stats = stats.data_ptr() if stats is not None else None
        a.workspace, a.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
        a.nsteps = nsteps
        if self.cfg is not None:
            for i in range(4):
                a.cfg[i] = int(self.cfg[i])
        return a

    def identity(self, shuffled_ind, lo, l1, scale, hinv, cst, s2k, cfac, R, state, stats=None,
                 nsteps=0):
        """coreLoop / coreZLoop / coreStructLoop / coreProbitLoop on the resident residual."""
        a = self._args(shuffled_ind, lo, l1, R, state, stats, nsteps)
        a.scale, a.hinv, a.cst, a.s2k = (scale.data_ptr(), hinv.data_ptr(), cst.data_ptr(),
                                         s2k.data_ptr())
        a.cfac = float(cfac)
        self._keep = (shuffled_ind, lo, l1, scale, hinv, cst, s2k)
        pre = self._pre if isinstance(self._pre, dict) else None
        self._status_ws = self.workspace
        if pre is not None and pre["ready"] is not None:
            ind, slot, done = pre["ready"]
            pre["ready"] = None
            if ind is shuffled_ind and nsteps == 0:      # the order these operands were packed for
                main = torch.cuda.current_stream(self.device)
                main.wait_event(done)
                a.workspace, a.workspace_bytes = pre["ws"][slot].data_ptr(), pre["ws"][slot].numel()
                a.prepacked = 1
                _lib.check(_lib.lib().locus_hip_sweep_dev(ctypes.byref(a), 0, _stream()))
                used = torch.cuda.Event()
                used.record(main)
                pre["used"][slot] = used
                self._status_ws = pre["ws"][slot]
                return
        _lib.check(_lib.lib().locus_hip_sweep_dev(ctypes.byref(a), 0, _stream()))

    def logit(self, shuffled_ind, lo, l1, s2, xwx, W2, half_lsi, R, state, stats=None, nsteps=0,
              ls2=None):
        """coreLogitLoop on the resident psi-weighted residual; ls2 = log(s2) / 2 (half_log)
        spares the sweep one logarithm per (covariate, response) pair."""
        a = self._args(shuffled_ind, lo, l1, R, state, stats, nsteps)
        a.s2, a.xwx, a.W2 = s2.data_ptr(), xwx.data_ptr(), W2.data_ptr()
        a.ls2 = ls2.data_ptr() if ls2 is not None else None
        a.half_lsi = float(half_lsi)
        self._keep = (shuffled_ind, lo, l1, s2, xwx, W2, ls2)
        self._status_ws = self.workspace
        _lib.check(_lib.lib().locus_hip_sweep_dev(ctypes.byref(a), 1, _stream()))

    def status(self):
        """Synchronise and raise if the last sweep saw an invalid shuffled_ind."""
        ws = getattr(self, "_status_ws", None)
        _lib.check(_lib.lib().locus_hip_sweep_status_dev(_p(ws if ws is not None else self.workspace),
                                                         self.p, _stream()))


def residual(Y, A, Zmu=None, W2=None, out=None, to_residual=True):
    n, d = Y.shape
    if out is None:
        out = torch.empty((d, n), dtype=F64, device=Y.device).t()
    _lib.check(_lib.lib().locus_hip_residual_dev(_p(Y), _p(A), _p(Zmu), _p(W2), _p(out), n, d,
                                                 1 if to_residual else 0, _stream()))
    return out


def transpose(inp, rows, cols, out=None):
    """Row-major [rows][cols] -> [cols][rows]."""
    if out is None:
        out = torch.empty((cols, rows), dtype=F64, device=inp.device)
    _lib.check(_lib.lib().locus_hip_transpose_dev(_p(inp), _p(out), rows, cols, _stream()))
    return out


def gamma_rows(gam_plane, p, d, eps=None):
    rs = torch.empty((p,), dtype=F64, device=gam_plane.device)
    ent = torch.empty((p,), dtype=F64, device=gam_plane.device) if eps is not None else None
    _lib.check(_lib.lib().locus_hip_gamma_rows_dev(_p(gam_plane), _p(rs), _p(ent), p, d,
                                                   float(eps if eps is not None else 0.0),
                                                   _stream()))
    return rs, ent


def identity_tail(args, phase):
    """locus_hip_identity_tail_dev on a filled _lib.TailArgs (locus_core.FusedTail)."""
    _lib.check(_lib.lib().locus_hip_identity_tail_dev(ctypes.byref(args), int(phase), _stream()))


def state_cols(state, p, d, s2k=None, s2jk=None, nchunks=64):
    nchunks = max(1, min(int(nchunks), p))
    part = torch.empty((nchunks, 3, d), dtype=F64, device=state.device)
    _lib.check(_lib.lib().locus_hip_state_cols_dev(_p(state), _p(s2k), _p(s2jk), _p(part), p, d,
                                                   nchunks, _stream()))
    return part.sum(dim=0)  # [3][d]


def gemm(m, n, k, A, sa_m, sa_k, B, sb_k, sb_n, C, sc_m, sc_n, alpha=1.0, beta=0.0, sqA=False,
         sqB=False):
    _lib.check(_lib.lib().locus_hip_gemm_dev(m, n, k, float(alpha), _p(A), sa_m, sa_k, int(sqA),
                                             _p(B), sb_k, sb_n, int(sqB), float(beta), _p(C), sc_m,
                                             sc_n, _stream()))
    return C


def batch_update(state, lo, l1, fac, hinv, cstk, log_sig2_inv, nm1, probit, p, d, cfac=1.0):
    _lib.check(_lib.lib().locus_hip_batch_update_dev(_p(state), _p(lo), _p(l1), _p(fac), _p(hinv),
                                                     _p(cstk), float(log_sig2_inv), float(nm1),
                                                     float(cfac), int(probit), p, d, _stream()))


# ---- binary-response drivers (link.hip) ---------------------------------------------------
def m2_minus_b2(state, p, d, D, s2jk=None, s2c=0.0):
    _lib.check(_lib.lib().locus_hip_m2_minus_b2_dev(_p(state), _p(s2jk), float(s2c), _p(D), p, d,
                                                    _stream()))
    return D


def s2_from_xwx(xwx, sig2_inv, s2):
    _lib.check(_lib.lib().locus_hip_s2_from_xwx_dev(_p(xwx), float(sig2_inv), _p(s2), xwx.numel(),
                                                    _stream()))
    return s2


def half_log(src, out):
    _lib.check(_lib.lib().locus_hip_half_log_dev(_p(src), _p(out), src.numel(), _stream()))
    return out


def logit_chi_psi(Q, M, Zmu, Y, Z, s2a, chi_old, W2_old, chi_new, W2_new, ey, n, d, q):
    _lib.check(_lib.lib().locus_hip_logit_chi_psi_dev(_p(Q), _p(M), _p(Zmu), _p(Y), _p(Z), _p(s2a),
                                                      _p(chi_old), _p(W2_old), _p(chi_new),
                                                      _p(W2_new), _p(ey), n, d, q, _stream()))


def logit_sig2_alpha(Z, W2, zeta2_inv, s2a, n, d, q):
    _lib.check(_lib.lib().locus_hip_logit_sig2_alpha_dev(_p(Z), _p(W2), _p(zeta2_inv), _p(s2a), n, d,
                                                         q, _stream()))


def alpha_step(Yw, W2, M, Zmu, Zi, s2a_i, alpha_i, n, d):
    _lib.check(_lib.lib().locus_hip_alpha_step_dev(_p(Yw), _p(W2), _p(M), _p(Zmu), _p(Zi),
                                                   _p(s2a_i), _p(alpha_i), n, d, _stream()))


def probit_w(Y, M, Zmu, Q, zs, W, ey, n, d):
    _lib.check(_lib.lib().locus_hip_probit_w_dev(_p(Y), _p(M), _p(Zmu), _p(Q), _p(zs), _p(W), _p(ey),
                                                 n, d, _stream()))


def bin_rows(state, p, d, eps, s2jk=None, s2c=1.0):
    """rs_gam, ent, gls, m2 as one [4][p] tensor."""
    out = torch.empty((4, p), dtype=F64, device=state.device)
    _lib.check(_lib.lib().locus_hip_bin_rows_dev(_p(state), _p(s2jk), float(s2c), _p(out[0]),
                                                 _p(out[1]), _p(out[2]), _p(out[3]), p, d,
                                                 float(eps), _stream()))
    return out
