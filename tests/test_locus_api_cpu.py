"""The user-level locus() entry (R/locus.R:240-640) on the CPU emulation of the device layer:
data preparation (constant / duplicated columns), defaults, dispatch on link / Z / list_struct,
refusal of the options outside the rebuilt path."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
for pth in (os.path.dirname(HERE), HERE):
    if pth not in sys.path:
        sys.path.insert(0, pth)

N, P, D, P0 = 80, 60, 6, 5


def _data(link="identity"):
    from locus_amd.simulate import simulate
    sim = simulate(N, P, D, P0, seed=5, link=link)
    X = np.array(sim["X_raw"] if "X_raw" in sim else sim["X"], dtype=np.float64)
    X = np.hstack([X, X[:, [3]], np.full((N, 1), 2.0)])       # one duplicated, one constant column
    return sim, X


@pytest.mark.parametrize("mode", ["identity", "z", "struct", "logit", "probit", "mix"])
def test_locus_dispatch_and_preparation(oracle, mode):
    import cpu_device
    from locus_amd.locus import locus
    from locus_amd.struct import set_struct
    link = mode if mode in ("logit", "probit") else "identity"
    sim, X = _data(link)
    kw = dict(link=link, user_seed=3, tol=1e-30, maxit=3, verbose=False, full_output=True,
              _test_backend=cpu_device)
    if mode == "z":
        kw["Z"] = np.random.default_rng(1).normal(size=(N, 2))
    if mode == "mix":
        Yb = sim["Y"].copy()
        Yb[:, [1, 4]] = (Yb[:, [1, 4]] > 0).astype(np.float64)
        sim = dict(sim, Y=Yb)
        kw.update(link="mix", ind_bin=[1, 4])
    if mode == "struct":
        kw["list_struct"] = set_struct(N, X.shape[1], np.arange(1, X.shape[1] + 1, 20))
    vb = locus(sim["Y"], X, P0, **kw)
    assert list(vb["rmvd_x"]) == [P, P + 1]                   # the duplicate and the constant
    assert vb["gam_vb"].shape == (P, D) and vb["it"] == 3
    assert np.isfinite(vb["lb_opt"]) and np.all((vb["gam_vb"] >= 0) & (vb["gam_vb"] <= 1))
    if mode == "z":
        assert vb["alpha_vb"].shape == (2, D)
    if link != "identity" or mode == "mix":
        assert vb["alpha_vb"].shape == (1, D)                 # the added intercept


def test_locus_refuses_what_is_out_of_scope(oracle):
    import cpu_device
    from locus_amd.locus import locus
    sim, X = _data()
    for kw in (dict(list_groups={}),):
        with pytest.raises(NotImplementedError):
            locus(sim["Y"], X, P0, _test_backend=cpu_device, verbose=False, **kw)
    with pytest.raises(ValueError):
        locus(sim["Y"], X, P0, ind_bin=[1], _test_backend=cpu_device, verbose=False)
    with pytest.raises(ValueError):
        locus(sim["Y"], X, P0, link="logit", anneal=(1, 2.0, 3), _test_backend=cpu_device,
              verbose=False)


def _blocks_run(comm=None):
    import cpu_device
    from locus_amd.locus import locus
    from locus_amd.prepare import set_blocks
    sim, X = _data()
    lb = set_blocks(X.shape[1], [1, 21, 45])                  # 3 blocks over the raw columns
    return sim, X, locus(sim["Y"], X, P0, list_blocks=lb, user_seed=3, tol=1e-30, maxit=3,
                         verbose=False, comm=comm, _test_backend=cpu_device)


def test_locus_block_wise_equals_per_block_fits(oracle):
    """list_blocks (R/locus.R:505-603): every block of predictors is an independent fit with all
    the responses; p_star is scaled by the number of blocks (R/prepare_locus.R:205-212)."""
    import cpu_device
    from locus_amd.locus_core import locus_core_
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    sim, X, vb = _blocks_run()
    assert vb["gam_vb"].shape == (P, D) and vb["it"].shape == (3,) and vb["om_vb"].shape == (P,)
    assert vb["p_star"] == P0 * 3
    Yc, Xs, rm = prepare_data_(sim["Y"], X)
    hy = auto_set_hyper_(Yc, P, P0 * 3)
    ini = auto_set_init_(Yc, P, P0 * 3, user_seed=3)
    fac = np.cumsum(np.isin(np.arange(1, X.shape[1] + 1), [1, 21, 45]))[~rm]
    for k, lab in enumerate(np.unique(fac)):
        pos = np.nonzero(fac == lab)[0]
        h = dict(hy, a=hy["a"][pos], b=hy["b"][pos])
        one = locus_core_(Yc, np.asfortranarray(Xs[:, pos]), h, ini["gam_vb"][pos],
                          ini["mu_beta_vb"][pos], ini["sig2_beta_vb"], ini["tau_vb"], tol=1e-30,
                          maxit=3, _test_backend=cpu_device)
        np.testing.assert_array_equal(vb["gam_vb"][pos], one["gam_vb"])
        assert vb["lb_opt"][k] == one["lb_opt"]


@pytest.mark.parametrize("with_z", [False, True])
def test_cross_validation_matches_restatement(oracle, with_z):
    """cross_validate_ (R/cross_validate.R:177-418): held-out lower bounds of every fold and grid
    value against the full-matrix restatement; then locus(list_cv = ...) end to end."""
    import cpu_device
    from oracle import ref_driver
    from locus_amd.cross_validate import create_grid_, cross_validate_, set_cv
    from locus_amd.locus import locus
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    np.testing.assert_array_equal(create_grid_(60, 3), [15, 18, 20])      # R: round(seq(15, 20, l = 3))
    np.testing.assert_array_equal(create_grid_(2000, 4), [500, 675, 825, 1000])
    np.testing.assert_array_equal(create_grid_(100, 5), [25, 50])         # base 25, duplicates dropped
    sim, X = _data()
    Yc, Xs, _ = prepare_data_(sim["Y"], X)
    Z = np.random.default_rng(3).normal(size=(N, 2)) if with_z else None
    cv = set_cv(N, X.shape[1], n_folds=3, size_p0_av_grid=3, maxit_cv=3, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([3.0, 8.0, 20.0]), 1e-30   # exactly 3 iterations
    perms = {}

    def perm_fn(it):
        p = 60
        return perms.setdefault(it, np.random.default_rng(70 + it).permutation(p).astype(np.float64))

    def defaults(Y_tr, p, pg, q):
        return (auto_set_hyper_(Y_tr, p, pg, q=q), auto_set_init_(Y_tr, p, pg, q=q, user_seed=5))
    ref_opt, ref_lb = ref_driver.cross_validate_(Yc, Xs, Z, cv, defaults, perm_fn)
    got_opt = cross_validate_(Yc, Xs, Z, "identity", None, cv, 5, False, _test_backend=cpu_device,
                              perm_fn=perm_fn, alpha_perm_fn=lambda it: np.arange(2))
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt and cv["lb_mat"].shape == (3, 3)
    assert len(set(np.round(ref_lb.mean(axis=0), 3))) == 3                 # the grid matters

    cv = set_cv(N, X.shape[1], n_folds=2, size_p0_av_grid=2, tol_cv=1.0, maxit_cv=5, verbose=False)
    vb = locus(sim["Y"], X, None, Z=Z, list_cv=cv, user_seed=5, tol=1e-30, maxit=2, verbose=False,
               _test_backend=cpu_device)
    assert vb["p_star"] in cv["p0_av_grid"] and vb["cv_lb_mat"].shape == (2, len(cv["p0_av_grid"]))
    with pytest.raises(ValueError):                                        # p0_av must be None
        locus(sim["Y"], X, 5, list_cv=cv, verbose=False, _test_backend=cpu_device)


@pytest.mark.parametrize("with_z", [False, True])
def test_cross_validation_probit_matches_restatement(oracle, with_z):
    """The probit branch of cross_validate_ (R/cross_validate.R:258-300, 345-361)."""
    import cpu_device
    from oracle import ref_driver
    from locus_amd.cross_validate import cross_validate_, set_cv
    from locus_amd.locus import locus
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    sim, X = _data("probit")
    Yb, Xs, _ = prepare_data_(sim["Y"], X, link="probit")
    Z = np.random.default_rng(3).normal(size=(N, 2)) if with_z else None
    q = 2 if with_z else None
    cv = set_cv(N, X.shape[1], n_folds=3, size_p0_av_grid=3, maxit_cv=3, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([3.0, 8.0, 20.0]), 1e-30
    perm_fn = lambda it: np.random.default_rng(70 + it).permutation(60).astype(np.float64)  # noqa: E731
    aperm = lambda it: np.arange(1 + (q or 0))  # noqa: E731

    def defaults(Y_tr, p, pg, qq):
        return (auto_set_hyper_(Y_tr, p, pg, q=qq, link="probit"),
                auto_set_init_(Y_tr, p, pg, q=qq, user_seed=5, link="probit"))
    ref_opt, ref_lb = ref_driver.cross_validate_(Yb, Xs, Z, cv, defaults, perm_fn, link="probit",
                                                 alpha_perm_fn=aperm)
    got_opt = cross_validate_(Yb, Xs, Z, "probit", None, cv, 5, False, _test_backend=cpu_device,
                              perm_fn=perm_fn, alpha_perm_fn=aperm)
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt
    cv = set_cv(N, X.shape[1], n_folds=2, size_p0_av_grid=2, tol_cv=1.0, maxit_cv=4, verbose=False)
    vb = locus(sim["Y"], X, None, Z=Z, link="probit", list_cv=cv, user_seed=5, maxit=2,
               verbose=False, _test_backend=cpu_device)
    assert vb["p_star"] in cv["p0_av_grid"]
    with pytest.raises(ValueError):                                        # R/prepare_locus.R:550
        locus(sim["Y"] - 0.0, X, None, link="logit", list_cv=cv, verbose=False,
              _test_backend=cpu_device)


@pytest.mark.parametrize("with_z", [False, True])
def test_cross_validation_mix_matches_restatement(oracle, with_z):
    """The mix branch of cross_validate_ (R/cross_validate.R:217-222, 258-300, 363-383)."""
    import cpu_device
    from oracle import ref_driver
    from locus_amd.cross_validate import cross_validate_, set_cv
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    sim, X = _data()
    ind_bin = np.array([1, 4])
    Yb = sim["Y"].copy()
    Yb[:, ind_bin] = (Yb[:, ind_bin] > 0).astype(np.float64)
    Ym, Xs, _ = prepare_data_(Yb, X, link="mix", ind_bin=ind_bin)
    Z = np.random.default_rng(3).normal(size=(N, 2)) if with_z else None
    q = 2 if with_z else None
    cv = set_cv(N, X.shape[1], n_folds=3, size_p0_av_grid=3, maxit_cv=3, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([3.0, 8.0, 20.0]), 1e-30
    perm_fn = lambda it: np.random.default_rng(70 + it).permutation(60).astype(np.float64)  # noqa: E731
    aperm = lambda it: np.arange(1 + (q or 0))  # noqa: E731

    def defaults(Y_tr, p, pg, qq):
        return (auto_set_hyper_(Y_tr, p, pg, q=qq, link="mix", ind_bin=ind_bin),
                auto_set_init_(Y_tr, p, pg, q=qq, user_seed=5, link="mix", ind_bin=ind_bin))
    ref_opt, ref_lb = ref_driver.cross_validate_(Ym, Xs, Z, cv, defaults, perm_fn, link="mix",
                                                 alpha_perm_fn=aperm, ind_bin=ind_bin)
    got_opt = cross_validate_(Ym, Xs, Z, "mix", ind_bin, cv, 5, False, _test_backend=cpu_device,
                              perm_fn=perm_fn, alpha_perm_fn=aperm)
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt


def test_duplicated_columns_match_lexicographic_unique():
    """prepare._duplicated_columns (hash + exact check) against numpy's sort-based unique, the
    semantics of R's duplicated(X, MARGIN = 2): every copy after the first is flagged."""
    from locus_amd.prepare import _duplicated_columns, scale_rm_
    rng = np.random.default_rng(8)
    X = rng.binomial(2, 0.3, size=(40, 900)).astype(np.float64)     # short columns: many repeats
    X[:, 700] = X[:, 2]
    X[:, 30] = 0.0
    X[:, 31] = -0.0                                                 # equal to column 30
    dup = _duplicated_columns(X)
    _, first = np.unique(X + 0.0, axis=1, return_index=True)
    ref = np.ones(X.shape[1], dtype=bool)
    ref[first] = False
    np.testing.assert_array_equal(dup, ref)
    assert dup[700] and dup[31] and not dup[2] and not dup[30]
    Xs, rm = scale_rm_(X)
    assert rm[30] and rm[31] and rm[700] and Xs.shape[1] == X.shape[1] - rm.sum()


def test_drawn_initial_values():
    """prepare.DrawnMatrix: the default p x d initial matrices are drawn per response chunk, so
    a rank's shard does not depend on the sharding; distributions as R/set_hyper_init.R:765,780,811."""
    import torch
    from locus_amd.prepare import DrawnMatrix, auto_set_init_drawn_
    cpu = torch.device("cpu")
    m = DrawnMatrix("beta1", 400, 600, 5, 50.0, "cpu")
    full = np.asarray(m)
    assert full.shape == (400, 600) and full.flags.f_contiguous
    assert abs(full.mean() - 1 / 51) < 2e-4 and full.min() >= 0 and full.max() < 1
    np.testing.assert_array_equal(m.columns_on(slice(100, 333), cpu).numpy(), full[:, 100:333])
    idx = np.array([5, 599, 256, 3, 511])
    np.testing.assert_array_equal(m.columns_on(idx, cpu).numpy(), full[:, idx])
    np.testing.assert_array_equal(np.asarray(DrawnMatrix("beta1", 400, 600, 5, 50.0, "cpu")), full)
    assert not np.array_equal(np.asarray(DrawnMatrix("beta1", 400, 600, 6, 50.0, "cpu")), full)
    rows = np.asarray(DrawnMatrix("beta1", 3, 4000, 1, np.array([1.0, 9.0, 99.0]), "cpu"))
    np.testing.assert_allclose(rows.mean(axis=1), [0.5, 0.1, 0.01], rtol=0.08)
    z = np.asarray(DrawnMatrix("normal", 300, 400, 2, device_type="cpu"))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    g = np.asarray(DrawnMatrix("inv_gamma2", 300, 400, 2, 1e-2, "cpu"))
    assert abs(np.median(1 / g) / (1e-2 * 1.67835) - 1) < 0.02      # median of Gamma(2, scale)
    Y = np.random.default_rng(0).normal(size=(50, 7))
    ini = auto_set_init_drawn_(Y, 30, 3, q=2, user_seed=4, link="logit")
    assert all(isinstance(ini[k], DrawnMatrix) for k in ("gam_vb", "mu_beta_vb", "sig2_beta_vb"))
    assert ini["alpha_vb"].shape == (2, 7) and ini["sig2_alpha_vb"].shape == (2, 7)
    with pytest.raises(ValueError):                                 # one device type per matrix
        DrawnMatrix("normal", 4, 4, 1, device_type="cuda").columns_on(slice(None), cpu)
    ini = auto_set_init_drawn_(Y, 30, 3, user_seed=4)
    assert ini["sig2_beta_vb"].shape == (7,) and ini["tau_vb"].shape == (7,)


def _default_init_worker(rank, world, port, q):
    """locus() with its default (drawn) initial values: responses sharded over two ranks give the
    single-rank result (the draw does not depend on the sharding; sums differ in association)."""
    try:
        import torch
        import torch.distributed as dist
        import cpu_device
        from locus_amd.locus import locus
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locus_amd.comm import DistComm
        for link in ("identity", "logit"):
            sim, X = _data(link)
            kw = dict(link=link, user_seed=11, tol=1e-30, maxit=3, verbose=False,
                      full_output=True, _test_backend=cpu_device)
            two = locus(sim["Y"], X, P0, comm=DistComm(), **kw)
            one = locus(sim["Y"], X, P0, **kw)
            for key in ("gam_vb", "mu_beta_vb", "beta_vb"):
                np.testing.assert_allclose(two[key], one[key], rtol=1e-9, atol=1e-300)
            assert abs(two["lb_opt"] - one["lb_opt"]) <= 1e-8 * abs(one["lb_opt"])
        dist.destroy_process_group()
        q.put((rank, None))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


def _cv_worker(rank, world, port, q):
    """Cross-validation with the folds spread over two ranks gives the single-rank matrix."""
    try:
        import torch
        import torch.distributed as dist
        import cpu_device
        from locus_amd.cross_validate import cross_validate_, set_cv
        from locus_amd.prepare import prepare_data_
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locus_amd.comm import DistComm
        sim, X = _data()
        Yc, Xs, _ = prepare_data_(sim["Y"], X)
        perm_fn = lambda it: np.random.default_rng(70 + it).permutation(60).astype(np.float64)  # noqa: E731
        res = []
        for comm in (DistComm(), None):
            cv = set_cv(N, X.shape[1], n_folds=3, size_p0_av_grid=2, tol_cv=1.0, maxit_cv=3,
                        verbose=False)
            opt = cross_validate_(Yc, Xs, None, "identity", None, cv, 5, False, comm=comm,
                                  _test_backend=cpu_device, perm_fn=perm_fn)
            res.append((opt, cv["lb_mat"]))
        assert res[0][0] == res[1][0]
        np.testing.assert_array_equal(res[0][1], res[1][1])
        dist.destroy_process_group()
        q.put((rank, None))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


def _blocks_worker(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locus_amd.comm import DistComm
        _, _, vb = _blocks_run(DistComm())
        _, _, ref = _blocks_run(None)
        np.testing.assert_array_equal(vb["gam_vb"], ref["gam_vb"])
        np.testing.assert_array_equal(vb["lb_opt"], ref["lb_opt"])
        dist.destroy_process_group()
        q.put((rank, None))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("worker", ["blocks", "default_init", "cv"])
def test_locus_over_ranks_world2_gloo(oracle, worker):
    """blocks: with several ranks block k runs on rank k mod world, every rank ends with all
    blocks.  default_init: see _default_init_worker."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    target = dict(blocks=_blocks_worker, default_init=_default_init_worker, cv=_cv_worker)[worker]
    procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    for rank, err in res:
        assert err is None, f"rank {rank}: {err}"
