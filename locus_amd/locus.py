"""locus(Y, X, p0_av, ...): the user-level entry of the reference (R/locus.R:240-640), for the
paths rebuilt here.  Prepares the data (standardisation, constant / duplicated columns), sets the
default hyper-parameters and initial values when none are given, dispatches on link / Z /
list_struct to the device-resident drivers and returns the reference's `vb` fields.

`list_cv` (set_cv): p0_av is chosen by cross-validation on the held-out ELBO
(locus_amd.cross_validate, R/cross_validate.R) for the identity, probit and mix links (logit is refused, as in R/prepare_locus.R:550-552).

Not rebuilt (out of scope): group selection (list_groups); asking for it raises
NotImplementedError rather than silently doing something else.
"""
import numpy as np

from .prepare import (add_intercept_, auto_set_hyper_, auto_set_init_, auto_set_init_drawn_,
                      convert_p0_av_, prepare_data_, scale_rm_)


def locus(Y, X, p0_av, Z=None, link="identity", ind_bin=None, list_hyper=None, list_init=None,
          list_cv=None, list_blocks=None, list_groups=None, list_struct=None, user_seed=None,
          tol=0.1, maxit=1000, anneal=None, full_output=False, verbose=True,
          checkpoint_path=None, comm=None, device=None, _test_backend=None):
    """Arguments and return fields as R/locus.R:240-245, :605-640 (`comm`: response sharding over
    the GPUs of a node, see locus_amd.comm; every rank passes the full arrays)."""
    if list_groups is not None:
        raise NotImplementedError("list_groups is outside the rebuilt path (SURVEY.md §8)")
    if link not in ("identity", "logit", "probit", "mix"):
        raise ValueError("link must be 'identity', 'logit', 'probit' or 'mix'")
    if (link == "mix") != (ind_bin is not None):             # R/prepare_locus.R:769-785
        raise ValueError("ind_bin (0-based indices of the binary responses) must be given for "
                         "link = 'mix' and only then")
    if ind_bin is not None:
        ind_bin = np.unique(np.asarray(ind_bin, dtype=np.int64))
    # check_annealing_ (R/utils.R:47-68): annealing only for the identity link without structure
    if anneal is not None and (link != "identity" or list_struct is not None):
        raise ValueError("Annealing has been implemented only for the identity link, "
                         "without structured sparsity")
    if list_blocks is not None and list_struct is not None:      # prepare_blocks_ (:611-613)
        raise ValueError("list_blocks and list_struct can't be both non-NULL.")
    if list_struct is not None and (link != "identity" or Z is not None):
        raise ValueError("list_struct requires link = 'identity' and Z = NULL")  # set_hyper:211

    if verbose:
        print("== Preparing the data ... ")
    Y, X, bool_rmvd_x = prepare_data_(Y, X, link, ind_bin)
    n, d = Y.shape
    p = X.shape[1]
    q = bool_rmvd_z = None
    if Z is not None:                                            # R/prepare_locus.R:69-101
        Z = np.asarray(Z, dtype=np.float64)
        if Z.shape[0] != n:
            raise ValueError("Z must have the same number of samples as Y and X.")
        Z, bool_rmvd_z = scale_rm_(Z)
        Z = np.asfortranarray(Z)
        q = Z.shape[1]
    if list_struct is not None:                                  # prepare_struct_: drop removed SNPs
        list_struct = dict(list_struct)
        list_struct["vec_fac_st"] = np.asarray(list_struct["vec_fac_st"])[~bool_rmvd_x]

    vec_fac_bl = n_bl = None
    if list_blocks is not None:                                  # prepare_blocks_ (:594-631)
        if list_blocks["p_blocks"] != len(bool_rmvd_x):
            raise ValueError("The number of candidate predictors given to set_blocks is not "
                             "consistent with X.")
        vec_fac_bl = np.asarray(list_blocks["vec_fac_bl"])[~bool_rmvd_x]
        n_bl = len(np.unique(vec_fac_bl))
    p_star = None
    if list_cv is not None and list_blocks is None and list_struct is None:  # R/locus.R:280-292
        from .cross_validate import cross_validate_, prepare_cv_
        if verbose:
            print("===== Cross-validation... =====")
        list_cv = prepare_cv_(list_cv, n, p, bool_rmvd_x, p0_av, link, list_hyper, list_init,
                              verbose)
        p_star = cross_validate_(Y, X, Z, link, ind_bin, list_cv, user_seed, verbose, comm=comm,
                                 device=device, _test_backend=_test_backend)
    elif list_hyper is None or list_init is None:
        p_star = convert_p0_av_(p0_av, len(bool_rmvd_x), n_bl)
        if np.ndim(p_star) > 0:
            p_star = p_star[~bool_rmvd_x]
    if list_hyper is None:
        list_hyper = auto_set_hyper_(Y, p, p_star, q=q, link=link, struct=list_struct is not None,
                                     ind_bin=ind_bin)
    else:
        list_hyper = dict(list_hyper)
    if list_init is None:
        # the p x d initial matrices are drawn by each rank on its GPU (prepare.DrawnMatrix);
        # the block-wise path slices them by rows on the host, so it keeps the host draw
        kw_i = dict(q=q, user_seed=user_seed, link=link, ind_bin=ind_bin)
        if vec_fac_bl is not None:
            list_init = auto_set_init_(Y, p, p_star, **kw_i)
        else:
            list_init = auto_set_init_drawn_(Y, p, p_star, **kw_i,
                                             device_type="cpu" if _test_backend else "cuda")
    else:
        list_init = dict(list_init)
    if link != "identity":
        Z = add_intercept_(n, d, list_hyper, list_init, link, Z)

    if vec_fac_bl is not None:
        vb = _run_blocks(Y, X, Z, link, ind_bin, list_hyper, list_init, vec_fac_bl, tol, maxit,
                         anneal, full_output, comm, device, _test_backend)
    else:
        vb = _run_one(Y, X, Z, link, ind_bin, list_hyper, list_init, list_struct, anneal,
                      checkpoint_path, dict(tol=tol, maxit=maxit, verbose=verbose,
                                            full_output=full_output, comm=comm, device=device,
                                            _test_backend=_test_backend))
    vb["p_star"] = p_star
    if list_cv is not None and "lb_mat" in list_cv:
        vb["cv_lb_mat"] = list_cv["lb_mat"]                      # held-out ELBO, folds x grid
    vb["rmvd_x"] = np.nonzero(bool_rmvd_x)[0]                    # R/locus.R:620-626
    vb["rmvd_z"] = None if bool_rmvd_z is None else np.nonzero(bool_rmvd_z)[0]
    return vb


def _run_one(Y, X, Z, link, ind_bin, list_hyper, list_init, list_struct, anneal, checkpoint_path,
             kw):
    """Dispatch of R/locus.R:447-503 on link / Z / list_struct."""
    li = list_init
    if link == "identity" and list_struct is not None:
        from .locus_struct_core import locus_struct_core_
        vb = locus_struct_core_(Y, X, list_hyper, li["gam_vb"], li["mu_beta_vb"],
                                li["sig2_beta_vb"], li["tau_vb"], list_struct, **kw)
    elif link == "identity" and Z is None:
        from .locus_core import locus_core_
        vb = locus_core_(Y, X, list_hyper, li["gam_vb"], li["mu_beta_vb"], li["sig2_beta_vb"],
                         li["tau_vb"], anneal=anneal, checkpoint_path=checkpoint_path, **kw)
    elif link == "identity":
        from .locus_z_core import locus_z_core_
        vb = locus_z_core_(Y, X, Z, list_hyper, li["gam_vb"], li["alpha_vb"], li["mu_beta_vb"],
                           li["sig2_alpha_vb"], li["sig2_beta_vb"], li["tau_vb"], anneal=anneal,
                           **kw)
    elif link == "mix":
        from .locus_mix_core import locus_mix_core_
        vb = locus_mix_core_(Y, X, Z, ind_bin, list_hyper, li["gam_vb"], li["alpha_vb"],
                             li["mu_beta_vb"], li["sig2_alpha_vb"], li["sig2_beta_vb"],
                             li["tau_vb"], **kw)
    else:
        from . import locus_bin_core
        fn = getattr(locus_bin_core, f"locus_{link}_core_")
        vb = fn(Y, X, Z, list_hyper, li["gam_vb"], li["alpha_vb"], li["mu_beta_vb"],
                li["sig2_alpha_vb"], li["sig2_beta_vb"], **kw)
    return vb


def _run_blocks(Y, X, Z, link, ind_bin, list_hyper, list_init, vec_fac_bl, tol, maxit, anneal,
                full_output, comm, device, _test_backend):
    """Block-wise inference (R/locus.R:505-603): the candidate predictors are split into blocks
    that are fitted independently, each with all the responses.  The reference fans the blocks out
    over CPU cores (parallel::mclapply, :590); here they run one after the other on the GPU, or --
    with a communicator of several ranks -- block k on rank k mod world, each rank fitting its
    blocks on its own GPU without response sharding, results exchanged at the end."""
    labels = np.unique(vec_fac_bl)
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    mine = {}
    for k, lab in enumerate(labels):
        if k % world != rank:
            continue
        pos = np.nonzero(vec_fac_bl == lab)[0]
        hy = dict(list_hyper)                                    # :509-514
        hy["a"], hy["b"] = np.asarray(list_hyper["a"])[pos], np.asarray(list_hyper["b"])[pos]
        li = dict(list_init)                                     # :516-523
        li["gam_vb"] = np.asarray(list_init["gam_vb"])[pos]
        li["mu_beta_vb"] = np.asarray(list_init["mu_beta_vb"])[pos]
        if link == "logit":
            li["sig2_beta_vb"] = np.asarray(list_init["sig2_beta_vb"])[pos]
        kw = dict(tol=tol, maxit=maxit, verbose=False, full_output=full_output, comm=None,
                  device=device, _test_backend=_test_backend)
        vb = _run_one(Y, np.asfortranarray(X[:, pos]), Z, link, ind_bin, hy, li, None, anneal, None,
                      kw)
        mine[k] = {key: vb[key] for key in ("converged", "it", "lb_opt", "om_vb", "gam_vb")}
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=comm.group)
        mine = {k: v for part in parts for k, v in part.items()}
    order = range(len(labels))
    return dict(converged=np.array([mine[k]["converged"] for k in order]),       # :592-601
                it=np.array([mine[k]["it"] for k in order]),
                lb_opt=np.array([mine[k]["lb_opt"] for k in order]),
                om_vb=np.concatenate([mine[k]["om_vb"] for k in order]),
                gam_vb=np.asfortranarray(np.vstack([mine[k]["gam_vb"] for k in order])))
