"""CPU checks of the oracle itself (no GPU): the C restatement against a second, independent
restatement taken from the scalar R formulas, against the reference's own run-time
invariants, and against the committed golden fixtures.

The reference ships no known-answer vectors for this path (tests/testthat/test_dimensions.R
asserts shapes only) and cannot be run here, so these are the pins the oracle has --
"parity unpinned" with respect to the reference running."""
import os

import numpy as np
import pytest

from helpers import INOUT, call_variant, inputs_for

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_log_one_plus_exp(oracle):
    x = np.array([-800.0, -40.0, -1.0, -1e-9, 0.0, 1e-9, 1.0, 40.0, 800.0])
    got = oracle.log_one_plus_exp(x)
    # the reference's formula (src/utils.cpp:3-12) loses log1p precision for very negative x
    # (log(1 + 4e-18) == 0); that is part of its definition, so compare on its own terms
    m = np.where(x < 0, 0.0, x)
    np.testing.assert_allclose(got, np.log(np.exp(x - m) + np.exp(-m)) + m, rtol=4e-16, atol=0)
    np.testing.assert_allclose(got, np.logaddexp(0.0, x), rtol=1e-15, atol=2.3e-16)


@pytest.mark.parametrize("shape", [(50, 30, 4), (200, 120, 9)])
@pytest.mark.parametrize("c", [1.0, 0.4])
def test_c_oracle_vs_scalar_R_restatement(oracle, shape, c):
    """src/coreLoop.cpp:34-50 (C oracle) vs R/locus_core.R:144-155 (numpy, other association)."""
    from oracle import ref_driver
    n, p, d = shape
    inp = inputs_for("identity", n, p, d, seed=4, c=c)
    ref = call_variant(oracle, "identity", inp)
    a = {k: (v.copy(order="F") if isinstance(v, np.ndarray) and v.ndim == 2 else v)
         for k, v in inp.items()}
    ref_driver.sweep_identity_scalar_R(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"],
                                       a["log_1_min_om_vb"], a["log_sig2_inv_vb"], a["log_tau_vb"],
                                       a["beta_vb"], a["mat_x_m1"], a["mu_beta_vb"],
                                       a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], c)
    for k in INOUT:
        np.testing.assert_allclose(ref[k], a[k], rtol=1e-10, atol=1e-13, err_msg=k)


@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit", "logit", "batch",
                                     "zbatch", "probitbatch"])
def test_invariants(oracle, variant):
    """mat_x_m1 == X %*% beta_vb after the call (R/update_vb.R:114), 0 <= gam <= 1,
    beta = gam * mu (src/coreLoop.cpp:48)."""
    inp = inputs_for(variant, 120, 45, 6, seed=9)
    out = call_variant(oracle, variant, inp)
    np.testing.assert_allclose(out["mat_x_m1"], inp["X"] @ out["beta_vb"], atol=1e-11)
    assert (out["gam_vb"] >= 0).all() and (out["gam_vb"] <= 1).all()
    np.testing.assert_array_equal(out["beta_vb"], out["mu_beta_vb"] * out["gam_vb"])


def test_add_back_identity(oracle):
    """X_j'(Y - (M - X_j b_j)) == X_j'(Y - M) + (X_j'X_j) b_j: the identity the reference uses in
    src/coreBatch.cpp:32-33 and the HIP kernel uses in every step -- checked on the oracle by
    running coreBatch against one explicit batched evaluation."""
    inp = inputs_for("batch", 150, 40, 5, seed=2)
    out = call_variant(oracle, "batch", inp)
    X, Y, M, beta = inp["X"], inp["Y"], inp["mat_x_m1"], inp["beta_vb"]
    n = X.shape[0]
    mu = (X.T @ (Y - M) + (n - 1) * beta) * (inp["sig2_beta_vb"] * inp["tau_vb"])[None, :]
    np.testing.assert_allclose(out["mu_beta_vb"], mu, rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit", "logit", "batch",
                                     "zbatch", "probitbatch"])
def test_golden_fixture(oracle, variant):
    """Committed vectors (tests/golden/make_golden.py): guards the oracle against drift."""
    f = np.load(os.path.join(GOLDEN, f"sweep_{variant}.npz"))
    inp = {k[3:]: (np.asfortranarray(f[k]) if f[k].ndim == 2 else
                   (f[k] if f[k].ndim == 1 else float(f[k])))
           for k in f.files if k.startswith("in_")}
    out = call_variant(oracle, variant, inp)
    for k in INOUT:
        np.testing.assert_allclose(out[k], f["out_" + k], rtol=1e-13, atol=1e-15, err_msg=k)


def test_oracle_driver_elbo_monotone_and_recovery(oracle):
    """R/locus_core.R:202-203 (ELBO must not decrease) on the C1-shaped simulation, and the
    simulated active SNPs are recovered."""
    from oracle import ref_driver
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    sim = simulate(200, 500, 20, 15, seed=123)
    hy = auto_set_hyper_(sim["Y"], 500, 15)
    ini = auto_set_init_(sim["Y"], 500, 15, user_seed=123)
    rng = np.random.default_rng(1)
    tr = []
    r = ref_driver.locus_core_(sim["Y"], sim["X"], hy, ini["gam_vb"], ini["mu_beta_vb"],
                               ini["sig2_beta_vb"], ini["tau_vb"], tol=0.1, maxit=100,
                               perm_fn=lambda it: rng.permutation(500).astype(np.float64),
                               trace=tr, debug=True)
    assert r["converged"]
    lbs = np.array([x[1] for x in tr])
    assert np.all(np.diff(lbs) > -1.49e-8)
    found = (r["gam_vb"] > 0.5).any(axis=1)
    assert found[sim["ind_p0"]].mean() >= 0.8 and found.sum() <= 30
    f = np.load(os.path.join(GOLDEN, "driver_identity_c1.npz"))
    np.testing.assert_allclose(lbs, f["elbo_trace"], rtol=1e-10)
