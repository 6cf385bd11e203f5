"""GPU parity of the device-resident identity driver (locus_amd.locus_core.locus_core_)
against the numpy restatement of R/locus_core.R (oracle.ref_driver.locus_core_), on the
BASELINE config C1 (n=200, p=500, d=20, p0_av=15) and identical sweep orders."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n=200, p=500, d=20, p0=15, seed=123):
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    sim = simulate(n, p, d, p0, seed=seed)
    hyper = auto_set_hyper_(sim["Y"], p, p0)
    init = auto_set_init_(sim["Y"], p, p0, user_seed=seed)
    perms = np.random.default_rng(1)
    table = {}

    def perm_fn(it):
        if it not in table:
            table[it] = perms.permutation(p).astype(np.float64)
        return table[it]
    return sim, hyper, init, perm_fn


@pytest.mark.parametrize("maxit", [1, 5])
def test_identity_fixed_iterations(oracle, hip, maxit):
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup()
    kw = dict(tol=1e-30, maxit=maxit, perm_fn=perm_fn, full_output=True)
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], **kw)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], **kw)
    assert got["it"] == ref["it"] == maxit
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(got["om_vb"], ref["om_vb"], rtol=1e-9)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])  # north_star: ELBO 1e-8
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_identity_multi_launch_statistics(oracle, hip, monkeypatch):
    """The fused column statistics across the launches of the planned matrix-core sweep: with
    one pretended CU the 20 responses go to a T = 2 launch (all 20 <= 32)... so pretend half a
    tile less: LOCUS_HIP_MFMA_CUS=1 and d = 50 -> 32 (T = 2) + 16 (T = 1) + 2 (T = 1)."""
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    monkeypatch.setenv("LOCUS_HIP_MFMA_CUS", "1")
    sim, hyper, init, perm_fn = _setup(n=120, p=90, d=50, p0=8, seed=7)
    kw = dict(tol=1e-30, maxit=3, perm_fn=perm_fn, full_output=True)
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], **kw)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], **kw)
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])


@pytest.mark.parametrize("anneal", [None, (1, 2.0, 3)])
def test_z_covariates_driver(oracle, hip, anneal):
    """locus_z_core_ (R/locus_z_core.R): identity link with q = 3 fixed covariates, against the
    numpy restatement; 5 iterations, identical sweep and alpha orders."""
    from oracle import ref_driver
    from locus_amd.locus_z_core import locus_z_core_
    sim, hyper, init, perm_fn = _setup(n=150, p=300, d=37, p0=10, seed=11)
    n, d, q = 150, 37, 3
    rng = np.random.default_rng(2)
    Z = rng.normal(size=(n, q))
    Z = np.asfortranarray((Z - Z.mean(axis=0)) / Z.std(axis=0, ddof=1))
    hyper["phi"], hyper["xi"] = np.ones(q), np.array([1.0, 0.5, 2.0])
    alpha0 = rng.normal(scale=0.2, size=(q, d))
    s2a0 = rng.gamma(2.0, 0.3, size=(q, d))
    aperms = [np.random.default_rng(40 + i).permutation(q) for i in range(8)]
    kw = dict(tol=1e-30, maxit=5, perm_fn=perm_fn, alpha_perm_fn=lambda it: aperms[it],
              anneal=anneal)
    args = (sim["Y"], sim["X"], Z, hyper, init["gam_vb"], alpha0, init["mu_beta_vb"], s2a0,
            init["sig2_beta_vb"], init["tau_vb"])
    ref = ref_driver.locus_z_core_(*args, **kw)
    got = locus_z_core_(*args, full_output=True, **kw)
    assert got["it"] == ref["it"] == 5
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "alpha_vb", "mat_z_mu",
              "sig2_alpha_vb", "tau_vb"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_mix_driver(oracle, hip):
    """locus_mix_core_ (R/locus_mix_core.R): continuous and binary responses interleaved,
    intercept + 2 covariates, against the numpy restatement; 4 iterations."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_drivers_cpu import _mix_problem
    from oracle import ref_driver
    from locus_amd.locus_mix_core import locus_mix_core_
    sim, hyper, init, extra, perm_fn, aperm_fn = _mix_problem(n=150, p=300, d=37, p0=10, seed=5)
    args = (sim["Y"], sim["X"], extra["Z"], extra["ind_bin"], hyper, init["gam_vb"],
            init["alpha_vb"], init["mu_beta_vb"], init["sig2_alpha_vb"], init["sig2_beta_vb"],
            init["tau_vb"])
    kw = dict(tol=1e-30, maxit=4, perm_fn=perm_fn, alpha_perm_fn=aperm_fn)
    ref = ref_driver.locus_mix_core_(*args, **kw)
    got = locus_mix_core_(*args, full_output=True, **kw)
    assert got["it"] == ref["it"] == 4
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "alpha_vb", "mat_z_mu",
              "sig2_alpha_vb", "tau_vb"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-11, err_msg=k)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_identity_to_convergence_recovers_actives(oracle, hip):
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup()
    tr_ref, tr_got = [], []
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], tol=0.1, maxit=1000,
                                 perm_fn=perm_fn, trace=tr_ref)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], tol=0.1, maxit=1000, perm_fn=perm_fn,
                      trace=tr_got)
    assert got["converged"] and ref["converged"] and got["it"] == ref["it"]
    lb_r = np.array([x[1] for x in tr_ref]); lb_g = np.array([x[1] for x in tr_got])
    np.testing.assert_allclose(lb_g, lb_r, rtol=1e-8)
    assert np.all(np.diff(lb_g) > -1.49e-8)  # ELBO monotone (R/locus_core.R:202-203)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)
    # the simulated active SNPs are found (each is associated with >= 1 response)
    found = (got["gam_vb"] > 0.5).any(axis=1)
    assert found[sim["ind_p0"]].mean() >= 0.8
    assert found.sum() <= 2 * len(sim["ind_p0"])


def test_identity_annealing(oracle, hip):
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup(n=100, p=80, d=6, p0=5)
    kw = dict(tol=1e-30, maxit=7, perm_fn=perm_fn, anneal=(1, 2.0, 4), full_output=True)
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], **kw)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], **kw)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], rtol=1e-9, atol=1e-12)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])


def test_identity_scheme_x_anneals(oracle, hip):
    """batch = "x" with an annealing ladder (R/locus_core.R:104-111 carry c; only "x-y" refuses
    annealing, :123-124): the element-wise kernel of coreBatch with the factor c."""
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup(n=90, p=40, d=6, p0=5, seed=3)
    kt = {}

    def k_perm_fn(it):
        if it not in kt:
            kt[it] = np.random.default_rng(100 + it).permutation(6)
        return kt[it]
    kw = dict(tol=1e-30, maxit=6, perm_fn=perm_fn, anneal=(1, 2.0, 4), full_output=True, batch="x",
              debug=False, k_perm_fn=k_perm_fn)
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], **kw)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], **kw)
    for key in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[key], ref[key], rtol=1e-9, atol=1e-12, err_msg=key)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    with pytest.raises(ValueError):
        locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                    init["sig2_beta_vb"], init["tau_vb"], **dict(kw, batch="x-y"))


# ---- structured model (R/locus_struct_core.R) ---------------------------------------------
def _setup_struct(n=200, p=500, d=20, p0=15, bl=50, seed=5):
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    from locus_amd.struct import set_struct
    sim = simulate(n, p, d, p0, seed=seed)
    hyper = auto_set_hyper_(sim["Y"], p, p0, struct=True)
    init = auto_set_init_(sim["Y"], p, p0, user_seed=seed)
    ls = set_struct(n, p, np.arange(1, p + 1, bl))
    perms = [np.random.default_rng(100 + i).permutation(p).astype(np.float64) for i in range(64)]
    return sim, hyper, init, ls, (lambda it: perms[it])


@pytest.mark.parametrize("maxit", [1, 4])
def test_struct_fixed_iterations(oracle, hip, maxit):
    from oracle import ref_driver
    from locus_amd.locus_struct_core import locus_struct_core_
    sim, hyper, init, ls, perm_fn = _setup_struct()
    ref = ref_driver.locus_struct_core_(sim["Y"], sim["X"], hyper, init["gam_vb"],
                                        init["mu_beta_vb"], init["sig2_beta_vb"], init["tau_vb"],
                                        ls["vec_fac_st"], tol=1e-30, maxit=maxit, perm_fn=perm_fn)
    got = locus_struct_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                             init["sig2_beta_vb"], init["tau_vb"], ls, tol=1e-30, maxit=maxit,
                             perm_fn=perm_fn, full_output=True)
    assert got["it"] == ref["it"] == maxit
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(got["theta_vb"], ref["theta_vb"], rtol=1e-9, atol=1e-12)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_struct_to_convergence(oracle, hip):
    from oracle import ref_driver
    from locus_amd.locus_struct_core import locus_struct_core_
    sim, hyper, init, ls, perm_fn = _setup_struct(n=150, p=300, d=12, p0=10, bl=30)
    tr_ref, tr_got = [], []
    ref = ref_driver.locus_struct_core_(sim["Y"], sim["X"], hyper, init["gam_vb"],
                                        init["mu_beta_vb"], init["sig2_beta_vb"], init["tau_vb"],
                                        ls["vec_fac_st"], tol=0.1, maxit=60, perm_fn=perm_fn,
                                        trace=tr_ref)
    got = locus_struct_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                             init["sig2_beta_vb"], init["tau_vb"], ls, tol=0.1, maxit=60,
                             perm_fn=perm_fn, trace=tr_got)
    assert got["it"] == ref["it"] and got["converged"] == ref["converged"]
    np.testing.assert_allclose([x[1] for x in tr_got], [x[1] for x in tr_ref], rtol=1e-8)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


# ---- binary responses (R/locus_logistic_core.R, R/locus_probit_core.R) ----------------------
def _setup_bin(link, n=200, p=300, d=12, p0=10, q=1, seed=9):
    from locus_amd.prepare import add_intercept_, auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    sim = simulate(n, p, d, p0, seed=seed, link=link)
    hyper = auto_set_hyper_(sim["Y"], p, p0, q=1, link=link)
    init = auto_set_init_(sim["Y"], p, p0, q=1, user_seed=seed, link=link)
    Z = add_intercept_(n, d, hyper, init, link)
    if q > 1:  # fixed covariates next to the intercept (R/locus.R:401-425)
        rng = np.random.default_rng(seed + 1)
        Zc = rng.normal(size=(n, q - 1))
        Zc = (Zc - Zc.mean(axis=0)) / Zc.std(axis=0, ddof=1)
        Z = np.asfortranarray(np.hstack([Z, Zc]))
        hyper["phi"] = np.concatenate([hyper["phi"], np.ones(q - 1)])
        hyper["xi"] = np.concatenate([hyper["xi"], np.ones(q - 1)])
        init["alpha_vb"] = rng.normal(scale=0.1, size=(q, d))
        init["sig2_alpha_vb"] = (np.full(q, 0.5) if link == "probit"
                                 else rng.gamma(2.0, 0.2, size=(q, d)))
    perms = [np.random.default_rng(200 + i).permutation(p).astype(np.float64) for i in range(64)]
    aperms = [np.random.default_rng(300 + i).permutation(q) for i in range(64)]
    return sim, hyper, init, Z, (lambda it: perms[it]), (lambda it: aperms[it])


@pytest.mark.parametrize("link", ["logit", "probit"])
@pytest.mark.parametrize("q,maxit", [(1, 1), (1, 4), (3, 3)])
def test_binary_fixed_iterations(oracle, hip, link, q, maxit):
    from oracle import ref_driver
    from locus_amd import locus_bin_core
    sim, hyper, init, Z, perm_fn, aperm_fn = _setup_bin(link, q=q)
    ref_fn = getattr(ref_driver, f"locus_{link}_core_")
    got_fn = getattr(locus_bin_core, f"locus_{link}_core_")
    args = (sim["Y"], sim["X"], Z, hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
            init["sig2_alpha_vb"], init["sig2_beta_vb"])
    ref = ref_fn(*args, tol=1e-30, maxit=maxit, perm_fn=perm_fn, alpha_perm_fn=aperm_fn)
    got = got_fn(*args, tol=1e-30, maxit=maxit, perm_fn=perm_fn, alpha_perm_fn=aperm_fn,
                 full_output=True)
    assert got["it"] == ref["it"] == maxit
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "mat_z_mu", "alpha_vb"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(got["sig2_beta_vb"], ref["sig2_beta_vb"], rtol=1e-10)
    if link == "logit":
        np.testing.assert_allclose(got["psi_vb"], ref["psi_vb"], rtol=1e-9)
    np.testing.assert_allclose(got["om_vb"], ref["om_vb"], rtol=1e-9)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


@pytest.mark.parametrize("link", ["logit", "probit"])
def test_binary_to_convergence(oracle, hip, link):
    from oracle import ref_driver
    from locus_amd import locus_bin_core
    sim, hyper, init, Z, perm_fn, aperm_fn = _setup_bin(link, n=150, p=200, d=8, p0=8)
    args = (sim["Y"], sim["X"], Z, hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
            init["sig2_alpha_vb"], init["sig2_beta_vb"])
    tr_ref, tr_got = [], []
    ref = getattr(ref_driver, f"locus_{link}_core_")(*args, tol=0.1, maxit=60, perm_fn=perm_fn,
                                                     trace=tr_ref)
    got = getattr(locus_bin_core, f"locus_{link}_core_")(*args, tol=0.1, maxit=60,
                                                         perm_fn=perm_fn, trace=tr_got)
    assert got["it"] == ref["it"] and got["converged"] == ref["converged"]
    lb_r = np.array([x[1] for x in tr_ref]); lb_g = np.array([x[1] for x in tr_got])
    np.testing.assert_allclose(lb_g, lb_r, rtol=1e-8)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_locus_entry_point_on_gpu(hip):
    """locus(Y, X, p0_av) (R/locus.R:240) end to end on the GPU: defaults, convergence at the
    reference's tol = 0.1, the simulated active SNPs are selected."""
    from locus_amd import locus
    from locus_amd.simulate import simulate
    sim = simulate(200, 500, 20, 15, seed=123)
    vb = locus(sim["Y"], sim["X"], p0_av=15, user_seed=123, verbose=False)
    assert vb["converged"] and vb["it"] < 200
    found = (vb["gam_vb"] > 0.5).any(axis=1)
    assert found[sim["ind_p0"]].mean() >= 0.8 and found.sum() <= 2 * len(sim["ind_p0"])
    assert len(vb["rmvd_x"]) == 0 and vb["om_vb"].shape == (500,)


def test_default_initial_values_drawn_on_the_gpu(hip, oracle):
    """locus() draws its default p x d initial matrices on the GPU (prepare.DrawnMatrix): a shard's
    values do not depend on the sharding, the same seed gives the same fit, and the fit from those
    values equals the numpy restatement of the driver started from the same (downloaded) values."""
    import torch
    from oracle.ref_driver import locus_core_ as ref_core
    from locus_amd import locus
    from locus_amd.prepare import (auto_set_hyper_, auto_set_init_drawn_, prepare_data_)
    from locus_amd.locus_core import locus_core_
    from locus_amd.simulate import simulate
    sim = simulate(150, 300, 700, 10, seed=5)
    Yc, Xs, _ = prepare_data_(sim["Y"], sim["X"])
    ini = auto_set_init_drawn_(Yc, 300, 10, user_seed=9)
    dev = torch.device("cuda", 0)
    full = np.asarray(ini["gam_vb"])
    np.testing.assert_array_equal(ini["gam_vb"].columns_on(slice(200, 613), dev).cpu().numpy(),
                                  full[:, 200:613])
    a = locus(sim["Y"], sim["X"], p0_av=10, user_seed=9, tol=1e-30, maxit=3, verbose=False)
    b = locus(sim["Y"], sim["X"], p0_av=10, user_seed=9, tol=1e-30, maxit=3, verbose=False)
    np.testing.assert_array_equal(a["gam_vb"], b["gam_vb"])
    hy = auto_set_hyper_(Yc, 300, 10)
    perms = [np.random.default_rng(40 + i).permutation(300).astype(np.float64) for i in range(3)]
    kw = dict(tol=1e-30, maxit=3, perm_fn=lambda it: perms[it - 1])
    got = locus_core_(Yc, Xs, hy, ini["gam_vb"], ini["mu_beta_vb"], ini["sig2_beta_vb"],
                      ini["tau_vb"], **kw)
    ref = ref_core(Yc, Xs, hy, full.copy(order="F"), np.asarray(ini["mu_beta_vb"]).copy(order="F"),
                   ini["sig2_beta_vb"], ini["tau_vb"], **kw)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], rtol=1e-9, atol=1e-300)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])


def test_drivers_with_more_than_2048_samples(oracle, hip):
    """n above the register-resident kernels (streaming sweep kernel) through whole drivers:
    identity and logit, two iterations against the numpy restatements."""
    from oracle import ref_driver
    from locus_amd import locus_bin_core
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup(n=2300, p=40, d=6, p0=4, seed=21)
    kw = dict(tol=1e-30, maxit=2, perm_fn=perm_fn, full_output=True)
    args = (sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"], init["sig2_beta_vb"],
            init["tau_vb"])
    ref, got = ref_driver.locus_core_(*args, **kw), locus_core_(*args, **kw)
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])

    sim, hyper, init, Z, perm_fn, aperm_fn = _setup_bin("logit", n=2100, p=30, d=5, p0=4)
    args = (sim["Y"], sim["X"], Z, hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
            init["sig2_alpha_vb"], init["sig2_beta_vb"])
    kw = dict(tol=1e-30, maxit=2, perm_fn=perm_fn, alpha_perm_fn=aperm_fn)
    ref = ref_driver.locus_logit_core_(*args, **kw)
    got = locus_bin_core.locus_logit_core_(*args, full_output=True, **kw)
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "alpha_vb"):
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])


@pytest.mark.parametrize("with_z", [False, True])
def test_cross_validation_on_gpu(oracle, hip, with_z):
    """cross_validate_ (R/cross_validate.R:177-418) on the device: the held-out lower bounds of
    every fold and grid value against the full-matrix restatement, then locus(list_cv = ...)."""
    from oracle import ref_driver
    from locus_amd import locus
    from locus_amd.cross_validate import cross_validate_, set_cv
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    from locus_amd.simulate import simulate
    n, p, d = 150, 120, 12
    sim = simulate(n, p, d, 8, seed=31)
    Yc, Xs, _ = prepare_data_(sim["Y"], sim["X"])
    Z = np.random.default_rng(3).normal(size=(n, 2)) if with_z else None
    cv = set_cv(n, p, n_folds=3, size_p0_av_grid=3, maxit_cv=4, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([4.0, 10.0, 30.0]), 1e-30
    perms = {}

    def perm_fn(it):
        return perms.setdefault(it, np.random.default_rng(70 + it).permutation(p).astype(np.float64))

    def defaults(Y_tr, pp, pg, q):
        return (auto_set_hyper_(Y_tr, pp, pg, q=q), auto_set_init_(Y_tr, pp, pg, q=q, user_seed=5))
    ref_opt, ref_lb = ref_driver.cross_validate_(Yc, Xs, Z, cv, defaults, perm_fn)
    got_opt = cross_validate_(Yc, Xs, Z, "identity", None, cv, 5, False, perm_fn=perm_fn,
                              alpha_perm_fn=lambda it: np.arange(2))
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt
    cv = set_cv(n, p, n_folds=2, size_p0_av_grid=2, tol_cv=1.0, maxit_cv=5, verbose=False)
    vb = locus(sim["Y"], sim["X"], None, Z=Z, list_cv=cv, user_seed=5, maxit=3, verbose=False)
    assert vb["p_star"] in cv["p0_av_grid"] and vb["cv_lb_mat"].shape == (2, len(cv["p0_av_grid"]))


# ---- committed fixtures (tests/golden/, written by tests/golden/make_golden.py) ----------------
@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit", "logit", "batch",
                                     "zbatch", "probitbatch"])
def test_hip_routines_against_golden_vectors(hip, variant):
    """The eight C-ABI routines on the committed input / output vectors."""
    import os
    from helpers import ATOL, INOUT, RTOL, call_variant
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", f"sweep_{variant}.npz"))
    inp = {k[3:]: (np.asfortranarray(f[k]) if f[k].ndim == 2 else
                   (f[k] if f[k].ndim == 1 else float(f[k])))
           for k in f.files if k.startswith("in_")}
    out = call_variant(hip, variant, inp)
    for k in INOUT:
        np.testing.assert_allclose(out[k], f["out_" + k], rtol=RTOL, atol=ATOL, err_msg=k)


def test_hip_drivers_against_golden_results(hip):
    """Every driver (identity, annealed, struct, logit, probit, z, annealed z, mix) on the GPU
    against the committed results of the restated R drivers."""
    import test_drivers_cpu as T
    for kind in T.KINDS:
        T.check_against_golden(kind, T._run(kind, "hip"), rtol=1e-9)


def test_identity_c1_elbo_trace_against_golden(hip):
    """BASELINE config C1 to convergence at tol = 0.1: the committed ELBO trace, iteration count
    and selected (covariate, response) pairs of the restated driver."""
    import os
    from locus_amd.locus_core import locus_core_
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "driver_identity_c1.npz"))
    sim = simulate(200, 500, 20, 15, seed=123)
    hy = auto_set_hyper_(sim["Y"], 500, 15)
    ini = auto_set_init_(sim["Y"], 500, 15, user_seed=123)
    rng = np.random.default_rng(1)
    tr = []
    r = locus_core_(sim["Y"], sim["X"], hy, ini["gam_vb"], ini["mu_beta_vb"], ini["sig2_beta_vb"],
                    ini["tau_vb"], tol=0.1, maxit=100,
                    perm_fn=lambda it: rng.permutation(500).astype(np.float64), trace=tr)
    assert r["it"] == int(f["it"]) and r["converged"]
    np.testing.assert_allclose([x[1] for x in tr], f["elbo_trace"], rtol=1e-8)
    np.testing.assert_array_equal(np.argwhere(r["gam_vb"] > 0.5), f["selected"])


def test_cross_validation_probit_on_gpu(oracle, hip):
    """The probit branch of cross_validate_ (R/cross_validate.R:258-300, 345-361) on the device:
    the held-out bound re-evaluates e_y_probit_ with two MFMA contractions over the test rows."""
    from oracle import ref_driver
    from locus_amd.cross_validate import cross_validate_, set_cv
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    from locus_amd.simulate import simulate
    n, p, d = 150, 100, 9
    sim = simulate(n, p, d, 8, seed=33, link="probit")
    Yb, Xs, _ = prepare_data_(sim["Y"], sim["X"], link="probit")
    Z = np.random.default_rng(3).normal(size=(n, 2))
    cv = set_cv(n, p, n_folds=3, size_p0_av_grid=3, maxit_cv=4, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([4.0, 10.0, 30.0]), 1e-30
    perm_fn = lambda it: np.random.default_rng(70 + it).permutation(p).astype(np.float64)  # noqa: E731
    aperm = lambda it: np.arange(3)  # noqa: E731

    def defaults(Y_tr, pp, pg, q):
        return (auto_set_hyper_(Y_tr, pp, pg, q=q, link="probit"),
                auto_set_init_(Y_tr, pp, pg, q=q, user_seed=5, link="probit"))
    ref_opt, ref_lb = ref_driver.cross_validate_(Yb, Xs, Z, cv, defaults, perm_fn, link="probit",
                                                 alpha_perm_fn=aperm)
    got_opt = cross_validate_(Yb, Xs, Z, "probit", None, cv, 5, False, perm_fn=perm_fn,
                              alpha_perm_fn=aperm)
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt


def test_cross_validation_mix_on_gpu(oracle, hip):
    """The mix branch of cross_validate_ (R/cross_validate.R:363-383) on the device."""
    from oracle import ref_driver
    from locus_amd import locus
    from locus_amd.cross_validate import cross_validate_, set_cv
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_, prepare_data_
    from locus_amd.simulate import simulate
    n, p, d = 150, 100, 9
    sim = simulate(n, p, d, 8, seed=34)
    ind_bin = np.array([0, 3, 7])
    Yb = sim["Y"].copy()
    Yb[:, ind_bin] = (Yb[:, ind_bin] > 0).astype(np.float64)
    Ym, Xs, _ = prepare_data_(Yb, sim["X"], link="mix", ind_bin=ind_bin)
    Z = np.random.default_rng(3).normal(size=(n, 2))
    cv = set_cv(n, p, n_folds=3, size_p0_av_grid=3, maxit_cv=4, verbose=False)
    cv["p0_av_grid"], cv["tol_cv"] = np.array([4.0, 10.0, 30.0]), 1e-30
    perm_fn = lambda it: np.random.default_rng(70 + it).permutation(p).astype(np.float64)  # noqa: E731
    aperm = lambda it: np.arange(3)  # noqa: E731

    def defaults(Y_tr, pp, pg, q):
        return (auto_set_hyper_(Y_tr, pp, pg, q=q, link="mix", ind_bin=ind_bin),
                auto_set_init_(Y_tr, pp, pg, q=q, user_seed=5, link="mix", ind_bin=ind_bin))
    ref_opt, ref_lb = ref_driver.cross_validate_(Ym, Xs, Z, cv, defaults, perm_fn, link="mix",
                                                 alpha_perm_fn=aperm, ind_bin=ind_bin)
    got_opt = cross_validate_(Ym, Xs, Z, "mix", ind_bin, cv, 5, False, perm_fn=perm_fn,
                              alpha_perm_fn=aperm)
    np.testing.assert_allclose(cv["lb_mat"], ref_lb, rtol=1e-8)
    assert got_opt == ref_opt
    cv = set_cv(n, p, n_folds=2, size_p0_av_grid=2, tol_cv=1.0, maxit_cv=4, verbose=False)
    vb = locus(Yb, sim["X"], None, Z=Z, link="mix", ind_bin=ind_bin, list_cv=cv, user_seed=5,
               maxit=2, verbose=False)
    assert vb["p_star"] in cv["p0_av_grid"]


@pytest.mark.parametrize("mode", ["identity", "z", "struct", "logit", "probit", "mix", "blocks"])
def test_locus_every_mode_on_gpu(hip, mode):
    """locus() end to end on the device for every link / option, with its default hyper-parameters
    and (device-drawn) initial values: shapes, ranges, the removed columns and a finite ELBO."""
    from locus_amd import locus, set_blocks, set_struct
    from locus_amd.simulate import simulate
    n, p, d = 120, 90, 8
    link = mode if mode in ("logit", "probit") else "identity"
    sim = simulate(n, p, d, 6, seed=12, link=link)
    X = np.array(sim["X_raw"] if "X_raw" in sim else sim["X"], dtype=np.float64)
    X = np.hstack([X, X[:, [3]], np.full((n, 1), 2.0)])       # one duplicated, one constant column
    Y = sim["Y"].copy()
    kw = dict(link=link, user_seed=3, maxit=4, verbose=False, full_output=True)
    if mode == "z":
        kw["Z"] = np.random.default_rng(1).normal(size=(n, 2))
    if mode == "mix":
        Y[:, [1, 4]] = (Y[:, [1, 4]] > 0).astype(np.float64)
        kw.update(link="mix", ind_bin=[1, 4])
    if mode == "struct":
        kw["list_struct"] = set_struct(n, X.shape[1], np.arange(1, X.shape[1] + 1, 30))
    if mode == "blocks":
        kw["list_blocks"] = set_blocks(X.shape[1], [1, 31, 61])
        kw.pop("full_output")
    vb = locus(Y, X, 6, **kw)
    assert list(vb["rmvd_x"]) == [p, p + 1]
    assert vb["gam_vb"].shape == (p, d) and np.all((vb["gam_vb"] >= 0) & (vb["gam_vb"] <= 1))
    assert np.all(np.isfinite(np.atleast_1d(vb["lb_opt"])))


def test_fused_tail_matches_tensor_tail(hip):
    """csrc/tail.hip (three kernels per iteration) against the same arithmetic as torch tensor
    operations (IdentityTail), on made-up reduced statistics: annealed iterations without the
    ELBO, then plain ones with it; ELBO, the scalars and every per-response / per-covariate
    constant of the next sweep to 1e-12."""
    import torch
    from locus_amd import device as dv
    from locus_amd.locus_core import FusedTail, IdentityTail
    rng = np.random.default_rng(5)
    n, p, d = 50, 300, 7
    dev = torch.device("cuda:0")
    hyper = dict(a=rng.uniform(0.5, 3.0, p), b=rng.uniform(5.0, 400.0, p),
                 eta=rng.uniform(0.5, 2.0, d), kappa=rng.uniform(0.5, 2.0, d), nu=1.3)
    hyper["lambda"] = 0.7
    tau0, s20 = rng.uniform(0.5, 2.0, d), rng.uniform(1e-3, 1e-2, d)
    A = IdentityTail(n, p, d, hyper, tau0, s20)
    B = FusedTail(n, p, d, hyper, tau0, s20, dv=dv)

    def stats_rs():
        cs_gam = rng.uniform(0.0, 20.0, d)
        cs_b2 = rng.uniform(0.0, 3.0, d)
        st = np.stack([cs_gam, cs_b2 + rng.uniform(0.01, 1.0, d), cs_b2, rng.uniform(10.0, 80.0, d)])
        return (torch.from_numpy(st).to(dev), torch.from_numpy(rng.uniform(0.0, d, p)).to(dev),
                torch.from_numpy(rng.uniform(0.0, 0.05, p)).to(dev))

    st, rs, _ = stats_rs()
    A.set_initial(st.clone(), rs.clone())
    B.set_initial(st.clone(), rs.clone())
    ladder = [0.4, 0.7, 1.0, 1.0, 1.0]
    A.begin_iteration(ladder[0])
    B.begin_iteration(ladder[0])
    for i, c in enumerate(ladder[:-1]):
        c_next, with_elbo = ladder[i + 1], c == 1.0
        st, rs, ent = stats_rs()
        lbA = A.end_iteration(st.clone(), rs.clone(), ent.sum().reshape(1), c, with_elbo)
        pa = A.begin_iteration(c_next)
        lbB = B.end_iteration(st.clone(), rs.clone(), ent if with_elbo else None, c, with_elbo,
                              c_next=c_next)
        pb = B.begin_iteration(c_next)
        if with_elbo:
            assert abs(lbA - lbB) <= 1e-12 * abs(lbA), (i, lbA, lbB)
        else:
            assert lbA is None and lbB is None
        for key in ("lo", "l1", "tau_vb", "sig2_beta_vb", "scale", "hinv", "cst", "s2k",
                    "sig2_inv_vb", "log_sig2_inv_vb"):
            np.testing.assert_allclose(pb[key].cpu().numpy(), pa[key].cpu().numpy(), rtol=1e-12,
                                       atol=1e-14, err_msg=f"iteration {i}: {key}")
        assert abs(A.sum_gam - B.sum_gam) <= 1e-12 * abs(A.sum_gam)
        assert abs(A.sig2_inv_vb - B.sig2_inv_vb) <= 1e-12 * abs(A.sig2_inv_vb)
    rss_test = rng.uniform(5.0, 40.0, d)
    si = B.sig2_inv_vb
    lbA, lbB = A.held_out_elbo_(31, rss_test), B.held_out_elbo_(31, rss_test)
    assert abs(lbA - lbB) <= 1e-12 * abs(lbA)
    assert B.sig2_inv_vb == si  # the fitted value survives the held-out evaluation


def _shard_worker(rank, world, port, q):
    """One of two ranks that share the GPU (gloo: RCCL refuses two ranks on one device)."""
    try:
        import os
        import sys
        import torch
        import torch.distributed as dist
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_driver
        from locus_amd.comm import DistComm
        from locus_amd.locus_core import locus_core_
        sim, hyper, init, perm_fn = _setup(n=150, p=120, d=37, p0=9, seed=11)
        kw = dict(tol=1e-30, maxit=4, perm_fn=perm_fn, full_output=True)
        got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                          init["sig2_beta_vb"], init["tau_vb"], comm=DistComm(),
                          device=torch.device("cuda:0"), **kw)
        if rank == 0:
            ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"],
                                         init["mu_beta_vb"], init["sig2_beta_vb"], init["tau_vb"], **kw)
            for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "tau_vb", "sig2_beta_vb"):
                np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
            assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
            assert abs(got["sig2_inv_vb"] - ref["sig2_inv_vb"]) <= 1e-10 * abs(ref["sig2_inv_vb"])
        t = torch.from_numpy(np.concatenate([got["gam_vb"].ravel(), [got["lb_opt"]]]))
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks disagree"
        dist.destroy_process_group()
        q.put((rank, None))
    except BaseException as e:  # noqa: BLE001 -- report to the parent, which fails the test
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))


def test_identity_two_ranks_share_the_gpu(oracle, hip):
    """The d-sharded driver with the fused iteration tail on real device buffers: two ranks (gloo)
    on one GPU, 19 + 18 responses, against the single-process oracle driver."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    for rank, err in res:
        assert err is None, f"rank {rank}: {err}"


def test_bench_two_ranks_share_the_gpu(hip):
    """bench.py's N > 1 path end to end (torch.distributed.run, two ranks over gloo on ONE GPU) at a
    size where every rank has a long sweep queued when it reaches the all-reduce: d_loc = 5000
    responses per rank, p = 2000.  This stalled inside gloo until DistComm drained the stream
    first; the watchdog turns a stall into a failure with stacks instead of a hang."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LOCUS_BENCH_BACKEND="gloo", LOCUS_BENCH_WATCHDOG="90")
    env.pop("LOCUS_HIP_SWEEP_CFG", None)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
         "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--p", "2000"],
        cwd=root, env=env, capture_output=True, text=True, timeout=200)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["d_per_gpu"] == 5000
    assert rec["config"]["elbo_monotone"]


def _rccl_worker(port, q):
    """ONE rank on the GPU with backend "nccl" (= RCCL): the group is created before any other GPU
    call of this process, DistComm is forced through its collectives, so the RCCL branches of
    comm.py (all-reduce enqueued on the stream without a drain, device-side all-gather) and the
    packed all-reduce of FusedTail run as they do with N > 1 ranks."""
    try:
        import os
        import sys
        import torch
        import torch.distributed as dist
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        device = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        torch.cuda.set_device(device)
        from oracle import ref_driver
        from locus_amd.comm import DistComm
        from locus_amd.locus_core import locus_core_
        comm = DistComm(force_collectives=True)
        assert dist.get_backend() == "nccl" and not comm.solo and not comm._sync_first
        calls = {"ar": 0, "ag": 0}
        ar, ag = dist.all_reduce, dist.all_gather_into_tensor

        def count_ar(t, *a, **k):
            assert t.is_cuda
            calls["ar"] += 1
            return ar(t, *a, **k)

        def count_ag(o, i, *a, **k):
            assert o.is_cuda and i.is_cuda
            calls["ag"] += 1
            return ag(o, i, *a, **k)
        dist.all_reduce, dist.all_gather_into_tensor = count_ar, count_ag
        sim, hyper, init, perm_fn = _setup(n=150, p=120, d=37, p0=9, seed=11)
        kw = dict(tol=1e-30, maxit=3, perm_fn=perm_fn, full_output=True)
        got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                          init["sig2_beta_vb"], init["tau_vb"], comm=comm, device=device, **kw)
        assert calls["ar"] >= 3 and calls["ag"] >= 4, calls
        ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"],
                                     init["mu_beta_vb"], init["sig2_beta_vb"], init["tau_vb"], **kw)
        for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1", "tau_vb", "sig2_beta_vb"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)
        assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
        assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)
        # all_gather_columns on its own, ragged in nothing (one rank) but through RCCL
        loc = np.asfortranarray(np.random.default_rng(0).normal(size=(11, 7)))
        np.testing.assert_array_equal(comm.all_gather_columns(loc, 7), loc)
        dist.destroy_process_group()
        q.put((0, None))
    except BaseException as e:  # noqa: BLE001 -- report to the parent, which fails the test
        import traceback
        q.put((0, traceback.format_exc() + repr(e)))


def test_identity_one_rank_over_rccl(oracle, hip):
    """The RCCL code path on the one GPU: a child process whose first GPU call is
    init_process_group("nccl", world_size=1, device_id=cuda:0) runs 3 iterations of locus_core_
    through DistComm's collectives, against the oracle driver."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(port, q))
    pr.start()
    try:
        rank, err = q.get(timeout=240)
    finally:
        pr.join(timeout=60)
        if pr.is_alive():
            pr.kill()
    assert err is None, err


def test_bench_one_rank_over_rccl(hip):
    """bench.py through torch.distributed.run with ONE rank and LOCUS_BENCH_FORCE_DIST=1: the
    "nccl" process group with device_id, DistComm and the packed all-reduce of every step."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LOCUS_BENCH_FORCE_DIST="1", LOCUS_BENCH_WATCHDOG="120")
    env.pop("LOCUS_HIP_SWEEP_CFG", None)
    env.pop("LOCUS_BENCH_BACKEND", None)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
         "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--p", "2000",
         "--to-tol", "0"],
        cwd=root, env=env, capture_output=True, text=True, timeout=220)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["config"]["collectives"] == "nccl (forced, 1 rank)"
    assert rec["config"]["elbo_monotone"]


@pytest.mark.parametrize("batch", ["x-y", "x", "0"])
def test_identity_internal_batch_schemes(oracle, hip, batch):
    """The reference's internal schemes of locus_core_ (R/locus_core.R:97-161; "convergence not
    ensured"): coreBatch for every (j, k) at once, the same response by response, and the plain
    double loop with log_om_vb refreshed after every response."""
    from oracle import ref_driver
    from locus_amd.locus_core import locus_core_
    sim, hyper, init, perm_fn = _setup(n=90, p=40, d=6, p0=5, seed=3)
    p, d = 40, 6
    kt, jt = {}, {}

    def k_perm_fn(it):
        if it not in kt:
            kt[it] = np.random.default_rng(100 + it).permutation(d)
        return kt[it]

    def perm0_fn(it, k):
        if (it, k) not in jt:
            jt[(it, k)] = np.random.default_rng(1000 * it + k).permutation(p).astype(np.float64)
        return jt[(it, k)]
    kw = dict(tol=1e-30, maxit=3, perm_fn=perm_fn, full_output=True, batch=batch, debug=False,
              k_perm_fn=k_perm_fn, perm0_fn=perm0_fn)
    ref = ref_driver.locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"], **kw)
    got = locus_core_(sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"],
                      init["sig2_beta_vb"], init["tau_vb"], **kw)
    for key in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[key], ref[key], rtol=1e-9, atol=1e-12, err_msg=key)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
