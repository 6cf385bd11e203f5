"""Cross-validation of p0_av on the variational lower bound (R/cross_validate.R): for every fold
and every value of the grid the model is fitted on the training rows by the device-resident
drivers and the ELBO is evaluated on the held-out rows; the value with the best mean wins.

All the links the reference's procedure covers: "identity" with or without fixed covariates Z
(R/cross_validate.R:303-343), "probit" (:345-361) and "mix" (:363-383); logit is refused by the
reference itself (R/prepare_locus.R:550-552).
The reference spreads the folds over CPU cores (parallel::mclapply, :393); here fold k runs on
rank k mod world when a communicator of several ranks is given (each rank fits whole folds on its
own GPU), else the folds run one after the other.
"""
import numpy as np

from .prepare import add_intercept_, auto_set_hyper_, auto_set_init_


def create_grid_(p, size_p0_av_grid):
    """R/cross_validate.R:153-173."""
    if p < 75:  # a different treatment to avoid a single element in the grid
        g = np.round(np.linspace(max(p // 4, 1), max(-(-p // 3), 2), size_p0_av_grid))
    else:
        g = np.linspace(max(min(1000.0, p / 4), 1.0), max(min(1500.0, p / 2), 1.0),
                        size_p0_av_grid)
        g = np.maximum(np.round(g / 25) * 25, 1.0)
    return np.unique(g)


def set_cv(n, p, n_folds, size_p0_av_grid, n_cpus=1, tol_cv=0.1, maxit_cv=1000, verbose=True):
    """R/cross_validate.R:89-150.  `n_cpus` is kept for signature parity; the folds are spread
    over the ranks of the communicator given to locus(), not over CPU cores."""
    for name, v in (("n_folds", n_folds), ("size_p0_av_grid", size_p0_av_grid),
                    ("maxit_cv", maxit_cv), ("n_cpus", n_cpus)):
        if int(v) != v or v < 1:
            raise ValueError(f"{name} must be a natural number.")
    if not 2 <= n_folds <= n:
        raise ValueError("n_folds must be a natural number greater than 2 and smaller than the "
                         "number of samples.")
    if size_p0_av_grid < 2:
        raise ValueError("size_p0_av_grid must be at greater 1 to allow for comparisons.")
    if size_p0_av_grid > 10:
        raise ValueError("size_p0_av_grid is large and may induce expensive computations. "
                         "Choose size_p0_av_grid in {2, 3, ..., 10}.")
    if not tol_cv > np.finfo(np.float64).eps:
        raise ValueError("tol_cv must be positive.")
    grid = create_grid_(p, size_p0_av_grid)
    if size_p0_av_grid > len(grid) and verbose:
        print(f"Cross-validation p0_av_grid reduced to {len(grid)} elements as p is small.")
    return dict(n_cv=int(n), p_cv=int(p), n_folds=int(n_folds), p0_av_grid=grid,
                size_p0_av_grid=len(grid), tol_cv=float(tol_cv), n_cpus=min(int(n_cpus), n_folds),
                maxit_cv=int(maxit_cv))


def prepare_cv_(list_cv, n, p, bool_rmvd_x, p0_av, link, list_hyper, list_init, verbose):
    """R/prepare_locus.R:541-590."""
    if not isinstance(list_cv, dict) or "p0_av_grid" not in list_cv:
        raise ValueError("The provided list_cv must be the result of set_cv (or None to skip the "
                         "cross-validation step).")
    if link == "logit":
        raise ValueError("Cross-validation not implemented for logistic regression. Please, set "
                         "list_cv to None or use probit regression.")
    if p0_av is not None or list_hyper is not None or list_init is not None:
        raise ValueError("p0_av, list_hyper and list_init must all be None if list_cv is "
                         "provided (cross-validation).")
    if list_cv["n_cv"] != n:
        raise ValueError("The number of samples n provided to set_cv is not consistent with "
                         "those of the data.")
    if list_cv["p_cv"] != len(bool_rmvd_x):
        raise ValueError("The number of candidate predictors p provided to set_cv is not "
                         "consistent with X.")
    list_cv = dict(list_cv)
    if np.any(list_cv["p0_av_grid"] > p):  # p reduced by the removal of constant predictors
        list_cv["p0_av_grid"] = create_grid_(p, list_cv["size_p0_av_grid"])
        list_cv["size_p0_av_grid"] = len(list_cv["p0_av_grid"])
        if verbose:
            print("The cross-validation grid has been readjusted to account for the removal of "
                  f"constant candidate predictors. Grid used: {list_cv['p0_av_grid']}.")
    return list_cv


def _scale_train(A_tr, A_test):
    """scale(A_tr); columns that are constant on the training rows leave both parts
    (R/cross_validate.R:200-209, 229-237)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        S = (A_tr - A_tr.mean(axis=0)) / A_tr.std(axis=0, ddof=1)
    keep = ~np.isnan(S.sum(axis=0))
    return np.asfortranarray(S[:, keep]), np.asfortranarray(A_test[:, keep])


def cross_validate_(Y, X, Z, link, ind_bin, list_cv, user_seed, verbose, comm=None, device=None,
                    _test_backend=None, perm_fn=None, alpha_perm_fn=None):
    """R/cross_validate.R:177-418.  Returns the selected p0_av; the matrix of held-out lower
    bounds (folds x grid) is left in list_cv["lb_mat"]."""
    if link not in ("identity", "probit", "mix"):
        raise ValueError("cross-validation: link must be 'identity', 'probit' or 'mix'")
    from .locus_bin_core import locus_probit_core_
    from .locus_mix_core import locus_mix_core_
    from .locus_core import locus_core_
    from .locus_z_core import locus_z_core_
    Y, X = np.asarray(Y, dtype=np.float64), np.asarray(X, dtype=np.float64)
    n, d = Y.shape
    grid = np.asarray(list_cv["p0_av_grid"], dtype=np.float64)
    n_folds = list_cv["n_folds"]
    folds = np.arange(n) % n_folds                               # rep_len(1:n_folds, n), 0-based
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    kw = dict(tol=list_cv["tol_cv"], maxit=list_cv["maxit_cv"], verbose=False, full_output=False,
              comm=None, device=device, _test_backend=_test_backend)
    if perm_fn is not None:
        kw["perm_fn"] = perm_fn
    lb_mat = np.zeros((n_folds, len(grid)))
    for k in range(n_folds):
        if k % world != rank:
            continue
        if verbose:
            print(f"Evaluating fold k = {k + 1}... \n-------------------------")
        cur = folds == k
        X_tr, X_test = _scale_train(X[~cur], X[cur])
        p = X_tr.shape[1]
        if link == "identity":                                  # :211-222
            Y_tr = np.asfortranarray(Y[~cur] - Y[~cur].mean(axis=0))
            Y_test = np.asfortranarray(Y[cur] - Y[cur].mean(axis=0))
        elif link == "mix":
            cont = np.setdiff1d(np.arange(d), ind_bin)
            Y_tr, Y_test = np.array(Y[~cur], order="F"), np.array(Y[cur], order="F")
            Y_tr[:, cont] -= Y_tr[:, cont].mean(axis=0)
            Y_test[:, cont] -= Y_test[:, cont].mean(axis=0)
        else:
            Y_tr, Y_test = np.asfortranarray(Y[~cur]), np.asfortranarray(Y[cur])
        q = Z_tr = Z_test = None
        if Z is not None:
            Z_tr, Z_test = _scale_train(np.asarray(Z, dtype=np.float64)[~cur],
                                        np.asarray(Z, dtype=np.float64)[cur])
            q = Z_tr.shape[1]
        for g, pg in enumerate(grid):
            if verbose:
                print(f"Evaluating p0_av = {pg:g}... ")
            hy = auto_set_hyper_(Y_tr, p, pg, q=q, link=link, ind_bin=ind_bin)
            li = auto_set_init_(Y_tr, p, pg, q=q, user_seed=user_seed, link=link, ind_bin=ind_bin)
            if link != "identity":                               # intercept, :258-300
                Zi_tr = add_intercept_(X_tr.shape[0], d, hy, li, link, Z_tr)
                one = np.ones((X_test.shape[0], 1))
                Zi_test = one if Z_test is None else np.hstack([one, Z_test])
            if link == "mix":                                    # :363-383
                vb = locus_mix_core_(Y_tr, X_tr, Zi_tr, ind_bin, hy, li["gam_vb"], li["alpha_vb"],
                                     li["mu_beta_vb"], li["sig2_alpha_vb"], li["sig2_beta_vb"],
                                     li["tau_vb"], held_out=(Y_test, X_test, Zi_test),
                                     alpha_perm_fn=alpha_perm_fn, **kw)
            elif link == "probit":                               # :345-361
                vb = locus_probit_core_(Y_tr, X_tr, Zi_tr, hy, li["gam_vb"], li["alpha_vb"],
                                        li["mu_beta_vb"], li["sig2_alpha_vb"], li["sig2_beta_vb"],
                                        held_out=(Y_test, X_test, Zi_test),
                                        alpha_perm_fn=alpha_perm_fn, **kw)
            elif q is None:
                vb = locus_core_(Y_tr, X_tr, hy, li["gam_vb"], li["mu_beta_vb"], li["sig2_beta_vb"],
                                 li["tau_vb"], held_out=(Y_test, X_test), **kw)
            else:
                vb = locus_z_core_(Y_tr, X_tr, Z_tr, hy, li["gam_vb"], li["alpha_vb"],
                                   li["mu_beta_vb"], li["sig2_alpha_vb"], li["sig2_beta_vb"],
                                   li["tau_vb"], held_out=(Y_test, X_test, Z_test),
                                   alpha_perm_fn=alpha_perm_fn, **kw)
            lb_mat[k, g] = vb["lb_held_out"]
            if verbose:
                print(f"Lower bound on test set, fold {k + 1}, p0_av {pg:g}: {lb_mat[k, g]}. \n"
                      "-------------------------")
    if world > 1:
        import torch
        t = torch.from_numpy(lb_mat)
        _reduce(comm, t)
        lb_mat = t.numpy()
    list_cv["lb_mat"] = lb_mat
    p0_av_opt = float(grid[int(np.argmax(lb_mat.mean(axis=0)))])
    if verbose:
        print("Lower bounds on test sets for each fold and each grid element: ")
        print(lb_mat)
        print(f"===== ...end of cross-validation with selected p0_av = {p0_av_opt:g} ===== ")
    return p0_av_opt


def _reduce(comm, t):
    """Sum a small host tensor over the ranks (rows of other ranks are zero)."""
    import torch.distributed as dist
    if dist.get_backend(comm.group) == "nccl":
        g = t.cuda()
        comm.all_reduce_sum_(g)
        t.copy_(g.cpu())
    else:
        comm.all_reduce_sum_(t)
