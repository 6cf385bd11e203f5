"""Shared helpers of the parity tests: call one native routine through either the
oracle (oracle.c_oracle) or the product (locus_amd.core) -- both expose the reference's
R wrapper names and argument order (R/RcppExports.R:4-34)."""
import numpy as np

# Tolerances of BASELINE.json's north_star: gam_vb / mu_beta_vb within 1e-9 relative.
# atol covers entries whose dot product cancels to ~0 (|mu| << typical |mu| ~ 1e-2..1):
# there the reference's own Eigen summation order would differ by the same amount.
RTOL = 1e-9
ATOL = 1e-13

INOUT = ("gam_vb", "beta_vb", "mat_x_m1", "mu_beta_vb")


def call_variant(mod, variant, inp):
    """Run `variant` on copies of the in/out matrices; returns dict of the four outputs."""
    a = {k: (np.array(v, order="F", copy=True) if isinstance(v, np.ndarray) and v.ndim == 2 else
             (np.array(v, copy=True) if isinstance(v, np.ndarray) else v)) for k, v in inp.items()}
    if variant == "identity":
        mod.coreLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                     a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                     a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], a["c"])
    elif variant == "z":
        mod.coreZLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                      a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                      a["mat_z_mu"], a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"],
                      a["shuffled_ind"], a["c"])
    elif variant == "struct":
        mod.coreStructLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                           a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                           a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"])
    elif variant == "probit":
        mod.coreProbitLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                           a["log_sig2_inv_vb"], a["beta_vb"], a["mat_x_m1"], a["mat_z_mu"],
                           a["mu_beta_vb"], a["sig2_beta_vb"], a["shuffled_ind"])
    elif variant == "logit":
        mod.coreLogitLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                          a["log_sig2_inv_vb"], a["beta_vb"], a["mat_x_m1"], a["mat_z_mu"],
                          a["mu_beta_vb"], a["psi_vb"], a["sig2_beta_vb"], a["shuffled_ind"])
    elif variant == "batch":
        mod.coreBatch(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                      a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                      a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"])
    elif variant == "zbatch":
        mod.coreZBatch(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                       a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                       a["mat_z_mu"], a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"])
    elif variant == "probitbatch":
        mod.coreProbitBatch(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                            a["log_sig2_inv_vb"], a["beta_vb"], a["mat_x_m1"], a["mat_z_mu"],
                            a["mu_beta_vb"], a["sig2_beta_vb"])
    else:
        raise ValueError(variant)
    return {k: a[k] for k in INOUT}


def inputs_for(variant, n, p, d, seed=0, c=1.0):
    from locus_amd.simulate import sweep_inputs
    base = {"batch": "identity", "zbatch": "z", "probitbatch": "probit"}.get(variant, variant)
    return sweep_inputs(n, p, d, seed=seed, variant=base, c=c)


def assert_parity(got, ref, rtol=RTOL, atol=ATOL, what=""):
    for k in INOUT:
        g, r = got[k], ref[k]
        assert g.shape == r.shape
        err = np.abs(g - r)
        tol = atol + rtol * np.maximum(np.abs(g), np.abs(r))
        bad = err > tol
        assert not bad.any(), (
            f"{what} {k}: {int(bad.sum())} / {bad.size} entries out of tolerance; "
            f"max abs err {err.max():.3e}, max rel err "
            f"{(err / np.maximum(np.abs(r), 1e-300)).max():.3e}")
    # selected sets must be bit-identical (north_star); report the margin for context
    sel_g, sel_r = got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5
    assert np.array_equal(sel_g, sel_r), (
        f"{what}: selected sets differ; min |gam - 0.5| = {np.abs(ref['gam_vb'] - 0.5).min():.3e}")
