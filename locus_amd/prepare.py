"""Data preparation, default hyper-parameters and default initial values
(mirrors the parts of R/prepare_locus.R and R/set_hyper_init.R the hot path depends on).

R's RNG is replaced by numpy's Generator: for a given seed the draws differ from R's,
their distributions are the reference's.
"""
import numpy as np
from scipy.stats import norm


def scale_rm_(X):
    """scale(X) (centre, sd with n - 1), then rm_constant_ and rm_collinear_ (R/utils.R:244-311):
    returns the kept columns and the mask of the removed ones."""
    X = np.asarray(X, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        Xs = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
    bool_cst = np.isnan(Xs.sum(axis=0))                       # rm_constant_
    Xs = Xs[:, ~bool_cst]
    bool_coll = _duplicated_columns(Xs)                       # rm_collinear_: keep first copies
    Xs = Xs[:, ~bool_coll]
    bool_rmvd = bool_cst.copy()
    bool_rmvd[~bool_cst] = bool_coll
    return Xs, bool_rmvd


def _duplicated_columns(Xs):
    """duplicated(Xs, MARGIN = 2) (R/utils.R:283): mask of the columns equal to an earlier one.
    One exact 64-bit hash per column (wrapping integer dot product of the bit patterns with fixed
    odd multipliers), then an exact comparison inside each group of equal hashes -- O(n p) instead
    of a lexicographic sort of the columns."""
    n, p = Xs.shape
    bits = np.ascontiguousarray(Xs + 0.0).view(np.uint64)     # + 0.0: -0.0 and 0.0 compare equal
    mult = np.random.default_rng(0x10C05).integers(1, 2 ** 63, size=n, dtype=np.uint64) * 2 + 1
    with np.errstate(over="ignore"):
        h = mult @ bits
    order = np.argsort(h, kind="stable")
    hs = h[order]
    dup = np.zeros(p, dtype=bool)
    start = np.nonzero(np.r_[True, hs[1:] != hs[:-1]])[0]
    stop = np.r_[start[1:], p]
    for a, b in zip(start[stop - start > 1], stop[stop - start > 1]):
        kept = []
        for j in order[a:b]:                                  # ascending column index
            if any(np.array_equal(Xs[:, j], Xs[:, k]) for k in kept):
                dup[j] = True
            else:
                kept.append(j)
    return dup


def prepare_data_(Y, X, link="identity", ind_bin=None):
    """R/prepare_locus.R:109-141: scale(X) (centre, sd with n-1), drop constant and duplicated
    columns, centre Y (identity) / subtract 1/2 (logit).  Returns (Y, X, bool_rmvd_x)."""
    Y = np.asarray(Y, dtype=np.float64)
    Xs, bool_rmvd_x = scale_rm_(X)
    if Xs.shape[1] < 1:
        raise ValueError("There must be at least 1 non-constant candidate predictor stored in X.")
    if link == "identity":
        Y = Y - Y.mean(axis=0)
    elif link == "mix":                                      # R/prepare_locus.R:133-135
        cont = np.setdiff1d(np.arange(Y.shape[1]), ind_bin)
        Y = Y.copy()
        Y[:, cont] = Y[:, cont] - Y[:, cont].mean(axis=0)
    elif link == "logit":
        Y = Y - 0.5
    return np.asfortranarray(Y), np.asfortranarray(Xs), bool_rmvd_x


def set_blocks(p, pos_bl):
    """R/prepare_locus.R:701-761: partition of the p candidate predictors into consecutive blocks;
    pos_bl = first index (1-based, as in R) of every block.  (n_cpus has no counterpart: blocks
    run one after the other on a GPU, or one block per rank, see locus_amd.locus.)"""
    pos_bl = np.asarray(pos_bl, dtype=np.int64)
    if (pos_bl < 1).any() or (pos_bl > p).any():
        raise ValueError("The positions provided in pos_bl must range between 1 and p.")
    if len(np.unique(pos_bl)) != len(pos_bl) or (np.diff(pos_bl) <= 0).any():
        raise ValueError("The positions provided in pos_bl must be unique and increasing.")
    vec_fac_bl = np.cumsum(np.isin(np.arange(1, p + 1), pos_bl))
    return dict(p_blocks=int(p), n_bl=len(np.unique(vec_fac_bl)), vec_fac_bl=vec_fac_bl)


def convert_p0_av_(p0_av, p, n_bl=None):
    """R/prepare_locus.R:155-214 (scalar or length-p p0_av); n_bl: number of blocks of a
    block-wise run, for which the sparsity level is adapted (:205-212)."""
    p_star = _convert_p0_av(p0_av, p)
    if n_bl is not None:
        p_star = np.minimum(np.asarray(p_star, dtype=np.float64) * n_bl, 0.975 * p)
        if p_star.ndim == 0:
            p_star = float(p_star)
    return p_star


def _convert_p0_av(p0_av, p):
    eps = np.finfo(np.float64).eps ** 0.5
    p0_av = np.atleast_1d(np.asarray(p0_av, dtype=np.float64))
    if p0_av.size == 1:
        v = float(p0_av[0])
        if v / p < eps:
            raise ValueError("invalid p0_av: the prior sparsity level p0_av / p must be > 0")
        if v / p > 0.95:
            raise ValueError("invalid p0_av: induces a non-sparse formulation")
        return v
    if p0_av.size != p:
        raise ValueError("p0_av must have length 1 or p")
    if (p0_av < eps).any() or (p0_av > 1 - eps).any():
        raise ValueError("All entries of p0_av must lie strictly between 0 and 1")
    return p0_av * p


def auto_set_hyper_(Y, p, p_star, q=None, link="identity", struct=False, ind_bin=None):
    """R/set_hyper_init.R:321-419 (no groups).  link = "mix": eta, kappa for the continuous
    responses only (:342-344), ind_bin 0-based."""
    d = Y.shape[1]
    if link == "mix":
        Y = Y[:, np.setdiff1d(np.arange(d), ind_bin)]
    h = dict(d_hyper=d, p_hyper=p, q_hyper=q, link_hyper=link)
    h["lambda"] = 1e-2
    h["nu"] = 1.0
    if link in ("identity", "mix"):
        eta = 1.0 / np.median(Y.var(axis=0, ddof=1))
        if not np.isfinite(eta):
            eta = 1e3
        h["eta"] = np.full(Y.shape[1], eta)
        h["kappa"] = np.ones(Y.shape[1])
    else:
        h["eta"] = h["kappa"] = None
    a = np.ones(p)
    b = np.broadcast_to(d * (p - np.asarray(p_star, dtype=np.float64)) / p_star, (p,)).copy()
    if (a <= 0).any() or (b <= 0).any():
        raise ValueError("hyper-parameters a and b must be positive")
    if struct:
        h["s02"] = 0.1
        h["m0"] = -norm.ppf(b / (a + b)) * np.sqrt(1 + h["s02"])
        h["a"] = h["b"] = None
    else:
        h["a"], h["b"] = a, b
        h["m0"] = h["s02"] = None
    if q is not None:
        h["phi"] = np.ones(q)
        h["xi"] = np.ones(q)
    else:
        h["phi"] = h["xi"] = None
    return h


def auto_set_init_(Y, p, p_star, q=None, user_seed=None, link="identity", ind_bin=None):
    """R/set_hyper_init.R:750-874 (no groups); link = "mix": tau_vb = 1 at the binary responses
    (:783-799), ind_bin 0-based."""
    d = Y.shape[1]
    rng = np.random.default_rng(user_seed)
    shape2 = d * (p - np.asarray(p_star, dtype=np.float64)) / p_star
    if np.ndim(shape2) == 0:
        gam_vb = rng.beta(1.0, float(shape2), size=(p, d))
    else:
        gam_vb = rng.beta(1.0, np.asarray(shape2)[:, None], size=(p, d))
    mu_beta_vb = rng.normal(size=(p, d))
    sig2_inv_vb = 1e-2
    init = dict(d_init=d, p_init=p, q_init=q, link_init=link, gam_vb=np.asfortranarray(gam_vb),
                mu_beta_vb=np.asfortranarray(mu_beta_vb), sig2_inv_vb=sig2_inv_vb)
    if link in ("identity", "mix"):
        cont = np.arange(d) if link == "identity" else np.setdiff1d(np.arange(d), ind_bin)
        tau = 1.0 / np.median(Y[:, cont].var(axis=0, ddof=1))
        if not np.isfinite(tau):
            tau = 1e3
        tau_vb = np.ones(d)
        tau_vb[cont] = tau
        init["tau_vb"] = tau_vb
        # 1 / rgamma(d, shape = 2, rate = 1 / (sig2_inv_vb * tau_vb))
        init["sig2_beta_vb"] = 1.0 / rng.gamma(2.0, sig2_inv_vb * tau_vb, size=d)
    elif link == "logit":
        init["tau_vb"] = None
        init["sig2_beta_vb"] = np.asfortranarray(1.0 / rng.gamma(2.0, sig2_inv_vb, size=(p, d)))
    else:  # probit
        init["tau_vb"] = None
        init["sig2_beta_vb"] = float(1.0 / rng.gamma(2.0, sig2_inv_vb))
    if q is not None:                                        # R/set_hyper_init.R:823-848
        init["alpha_vb"] = rng.normal(size=(q, d))
        if link in ("identity", "mix"):
            zeta2_inv_vb = rng.gamma(1.0, 1.0, size=q)
            init["sig2_alpha_vb"] = 1.0 / rng.gamma(2.0, np.outer(zeta2_inv_vb, init["tau_vb"]))
        elif link == "logit":
            init["sig2_alpha_vb"] = 1.0 / rng.gamma(2.0, rng.gamma(1.0, 1.0, size=(q, d)))
        else:
            init["sig2_alpha_vb"] = 1.0 / rng.gamma(2.0, rng.gamma(1.0, 1.0, size=q))
    else:
        init["alpha_vb"] = init["sig2_alpha_vb"] = None
    return init


class DrawnMatrix:
    """A p x d matrix of independent draws that is never materialised on the host: every rank
    draws the columns of its own response shard directly on its GPU (at C3 the two p x d initial
    matrices are 4 GB each; numpy needs ~15 s to draw them and PCIe ~1 s to move them, the CAVI
    iterations to convergence take 0.5 s).

    The columns come in chunks of CHUNK responses, each from its own generator seeded with
    (seed, kind, chunk), so the values do not depend on how the responses are sharded.
    kind: "beta1"  Beta(1, param) by inversion, param a scalar or one value per row
                   (gam_vb, R/set_hyper_init.R:765);
          "normal" N(0, 1) (mu_beta_vb, :780);
          "inv_gamma2" 1 / Gamma(shape 2, scale param) (logit sig2_beta_vb, :811).
    np.asarray(m) materialises the matrix (column-major) for host-side callers.
    torch's CPU and GPU generators give different streams, so a matrix belongs to ONE device
    type: `device_type` ("cuda" / "cpu"), or, if None, the type of the first draw; asking for
    columns on the other type raises instead of silently returning different numbers."""
    CHUNK = 256
    KINDS = ("beta1", "normal", "inv_gamma2")

    def __init__(self, kind, p, d, seed=None, param=None, device_type=None):
        assert kind in self.KINDS
        self.kind, self.shape, self.param = kind, (int(p), int(d)), param
        self.device_type = device_type
        ss = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(seed)
        self.seed = int(ss.generate_state(1, dtype=np.uint64)[0])
        self.ndim, self.dtype = 2, np.dtype(np.float64)

    def _chunk(self, c, device):
        import torch
        p, d = self.shape
        w = min(self.CHUNK, d - c * self.CHUNK)
        key = np.random.SeedSequence([self.seed, self.KINDS.index(self.kind), c])
        g = torch.Generator(device=device)
        g.manual_seed(int(key.generate_state(1, dtype=np.uint64)[0] >> np.uint64(1)))
        kw = dict(generator=g, dtype=torch.float64, device=device)
        if self.kind == "normal":
            return torch.randn((p, w), **kw)
        lu = torch.log1p(-torch.rand((p, w), **kw))           # log of a uniform on (0, 1]
        if self.kind == "beta1":                                # 1 - U^(1/b)
            b = torch.as_tensor(self.param, dtype=torch.float64, device=device).reshape(-1, 1)
            return -torch.expm1(lu / b)
        lu += torch.log1p(-torch.rand((p, w), **kw))          # Gamma(2) = -log(U1 U2)
        return 1.0 / (-float(self.param) * lu).clamp_min(1e-300)

    def columns_on(self, cols, device):
        """[p][len(cols)] row-major float64 tensor on `device`; cols: slice or index array."""
        import torch
        device = torch.device(device)
        if self.device_type is None:
            self.device_type = device.type
        elif device.type != self.device_type:
            raise ValueError(f"this DrawnMatrix is drawn on '{self.device_type}' devices, "
                             f"not on '{device.type}'")
        p, d = self.shape
        cols = np.arange(d)[cols]
        out = torch.empty((p, len(cols)), dtype=torch.float64, device=device)
        which = cols // self.CHUNK
        for c in np.unique(which):
            pos = np.nonzero(which == c)[0]
            blk = self._chunk(int(c), device)
            src = cols[pos] - c * self.CHUNK
            if len(pos) == blk.shape[1] and pos[-1] - pos[0] == len(pos) - 1 \
                    and np.array_equal(src, np.arange(len(pos))):
                out[:, pos[0]:pos[-1] + 1] = blk
            else:
                out[:, torch.as_tensor(pos, device=device)] = blk[:, torch.as_tensor(src, device=device)]
        return out

    def __array__(self, dtype=None, copy=None):
        import torch
        kind = self.device_type or ("cuda" if torch.cuda.is_available() else "cpu")
        dev = torch.device(kind)
        from .hostio import to_host_F
        return to_host_F(self.columns_on(slice(None), dev))


def auto_set_init_drawn_(Y, p, p_star, q=None, user_seed=None, link="identity", ind_bin=None,
                         device_type=None):
    """auto_set_init_ with the p x d matrices (gam_vb, mu_beta_vb, logit sig2_beta_vb) left as
    DrawnMatrix recipes for the drivers to draw on their GPUs; everything else as auto_set_init_."""
    d = Y.shape[1]
    # the small fields (tau_vb, sig2_beta_vb of length d, alpha_vb, ...) from the host routine,
    # called with a single stand-in covariate
    init = auto_set_init_(Y, 1, 0.5, q=q, user_seed=user_seed, link=link, ind_bin=ind_bin)
    ss = np.random.SeedSequence(user_seed).spawn(3)
    shape2 = d * (p - np.asarray(p_star, dtype=np.float64)) / p_star
    init["p_init"] = p
    init["gam_vb"] = DrawnMatrix("beta1", p, d, ss[0], shape2, device_type)
    init["mu_beta_vb"] = DrawnMatrix("normal", p, d, ss[1], device_type=device_type)
    if link == "logit":
        init["sig2_beta_vb"] = DrawnMatrix("inv_gamma2", p, d, ss[2], init["sig2_inv_vb"],
                                           device_type)
    return init


def add_intercept_(n, d, list_hyper, list_init, link, Z=None):
    """R/locus.R:379-424: intercept column with a vague prior for logit/probit, in front of the
    user's covariates if any."""
    one = np.ones((n, 1))
    if Z is None:
        list_hyper["phi"] = np.array([1e-3])
        list_hyper["xi"] = np.array([1e-3])
        list_init["alpha_vb"] = np.zeros((1, d))
        list_init["sig2_alpha_vb"] = np.array([1.0]) if link == "probit" else np.ones((1, d))
        return np.asfortranarray(one)
    list_hyper["phi"] = np.concatenate([[1e-3], np.atleast_1d(list_hyper["phi"])])
    list_hyper["xi"] = np.concatenate([[1e-3], np.atleast_1d(list_hyper["xi"])])
    list_init["alpha_vb"] = np.vstack([np.zeros((1, d)), list_init["alpha_vb"]])
    if link == "probit":
        list_init["sig2_alpha_vb"] = np.concatenate([[1.0], np.atleast_1d(list_init["sig2_alpha_vb"])])
    else:
        list_init["sig2_alpha_vb"] = np.vstack([np.ones((1, d)), list_init["sig2_alpha_vb"]])
    return np.asfortranarray(np.hstack([one, Z]))
