"""The resident-context part of the C ABI (include/locus_hip.h, section 3): host pointers in and
out, state kept on the GPU between calls -- against the oracle routines and the restated driver."""
import numpy as np
import pytest

from helpers import ATOL, RTOL, assert_parity, call_variant, inputs_for

pytestmark = pytest.mark.gpu


def _stats_ref(out, inp, zmu=None):
    gam, mu, beta = out["gam_vb"], out["mu_beta_vb"], out["beta_vb"]
    R = inp["Y"] - out["mat_x_m1"] - (0.0 if zmu is None else zmu)
    s2 = np.broadcast_to(inp["sig2_beta_vb"], (gam.shape[1],))
    return gam.sum(axis=1), np.stack([gam.sum(axis=0), (gam * (mu ** 2 + s2)).sum(axis=0),
                                      (beta ** 2).sum(axis=0), (R ** 2).sum(axis=0)])


@pytest.mark.parametrize("variant,shape", [("identity", (200, 90, 21)), ("z", (150, 40, 7)),
                                           ("struct", (64, 33, 1)), ("identity", (2300, 24, 5))])
def test_ctx_sweep_matches_oracle_routine(oracle, hip, variant, shape):
    from locus_amd.resident import ResidentContext
    n, p, d = shape
    inp = inputs_for(variant, n, p, d, seed=n + d, c=0.7 if variant == "identity" else 1.0)
    ref = call_variant(oracle, variant, inp)
    zmu = inp.get("mat_z_mu") if variant == "z" else None
    eps = np.finfo(np.float64).eps
    with ResidentContext(inp["X"], inp["Y"]) as ctx:
        ctx.set_state(inp["gam_vb"], inp["mu_beta_vb"], zmu)
        st0 = ctx.get_state()
        np.testing.assert_allclose(st0["mat_x_m1"], inp["mat_x_m1"], rtol=1e-10, atol=1e-11)
        rs0, cs0, e0 = ctx.stats(inp["sig2_beta_vb"], eps)
        rs_r, cs_r = _stats_ref(dict(inp), inp, zmu)
        np.testing.assert_allclose(rs0, rs_r, rtol=1e-12)
        np.testing.assert_allclose(cs0, cs_r, rtol=1e-10)
        rs, cs, ent = ctx.sweep(inp["log_om_vb"], inp["log_1_min_om_vb"], inp["log_sig2_inv_vb"],
                                inp["log_tau_vb"], inp["sig2_beta_vb"], inp["tau_vb"],
                                inp["shuffled_ind"], inp.get("c", 1.0), eps)
        got = ctx.get_state()
    assert_parity(got, ref, what=f"ctx {variant} {shape}")
    rs_r, cs_r = _stats_ref(ref, inp, zmu)
    np.testing.assert_allclose(rs, rs_r, rtol=1e-9)
    np.testing.assert_allclose(cs, cs_r, rtol=1e-9)
    g = ref["gam_vb"]
    np.testing.assert_allclose(ent, np.sum(g * np.log(g + eps) + (1 - g) * np.log(1 - g + eps)),
                               rtol=1e-9)
    assert e0 is not None and np.isfinite(e0)


def test_ctx_sweep_as_core_probit_loop(oracle, hip):
    """coreProbitLoop (src/coreLoop.cpp:139-171) through the context, as the header describes:
    W in place of Y, tau_vb = 1, log_tau_vb = 0, the scalar sig2_beta_vb repeated."""
    from locus_amd.resident import ResidentContext
    n, p, d = 130, 45, 6
    inp = inputs_for("probit", n, p, d, seed=9)
    ref = call_variant(oracle, "probit", inp)
    with ResidentContext(inp["X"], inp["Y"]) as ctx:              # inp["Y"] is W here
        ctx.set_state(inp["gam_vb"], inp["mu_beta_vb"], inp["mat_z_mu"])
        ctx.sweep(inp["log_om_vb"], inp["log_1_min_om_vb"], inp["log_sig2_inv_vb"], np.zeros(d),
                  np.full(d, float(inp["sig2_beta_vb"])), np.ones(d), inp["shuffled_ind"])
        got = ctx.get_state()
    assert_parity(got, ref, what="ctx probit")


@pytest.mark.parametrize("variant", ["z", "probit"])
def test_ctx_several_iterations_with_changing_response(oracle, hip, variant):
    """coreZLoop / coreProbitLoop through the context for several iterations, with mat_z_mu (and
    for probit the latent W) changing between them as in R/locus_z_core.R:100-111 and
    R/locus_probit_core.R:76-91: locus_hip_ctx_set_response must keep the resident gam_vb /
    mu_beta_vb / mat_x_m1 (ctx_set_state would reset them to the host's stale copies)."""
    from locus_amd.resident import ResidentContext
    n, p, d = 140, 50, 9
    inp = inputs_for(variant, n, p, d, seed=31)
    rng = np.random.default_rng(5)
    cur = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    with ResidentContext(inp["X"], inp["Y"]) as ctx:
        ctx.set_state(inp["gam_vb"], inp["mu_beta_vb"], inp["mat_z_mu"])
        for it in range(4):
            if it:       # what the alpha loop / update_W_probit_ do between sweeps
                cur["mat_z_mu"] = np.asfortranarray(
                    cur["mat_z_mu"] + rng.normal(scale=0.05, size=(n, d)))
                if variant == "probit":
                    cur["Y"] = np.asfortranarray(cur["Y"] + rng.normal(scale=0.1, size=(n, d)))
                    ctx.set_response(cur["Y"], cur["mat_z_mu"])
                else:
                    ctx.set_response(None, cur["mat_z_mu"])
            cur["shuffled_ind"] = rng.permutation(p).astype(np.float64)
            ref = call_variant(oracle, variant, cur)
            if variant == "probit":
                ctx.sweep(cur["log_om_vb"], cur["log_1_min_om_vb"], cur["log_sig2_inv_vb"],
                          np.zeros(d), np.full(d, float(cur["sig2_beta_vb"])), np.ones(d),
                          cur["shuffled_ind"])
            else:
                ctx.sweep(cur["log_om_vb"], cur["log_1_min_om_vb"], cur["log_sig2_inv_vb"],
                          cur["log_tau_vb"], cur["sig2_beta_vb"], cur["tau_vb"],
                          cur["shuffled_ind"], cur.get("c", 1.0))
            got = ctx.get_state()
            assert_parity(got, ref, rtol=1e-9, atol=1e-12, what=f"ctx {variant} iteration {it + 1}")
            cur.update(ref)           # the oracle continues from its own output, the context from its
    assert not np.array_equal(got["gam_vb"], inp["gam_vb"])


def test_ctx_rejects_misuse(hip):
    from locus_amd._lib import LocusHipError
    from locus_amd.resident import ResidentContext
    inp = inputs_for("identity", 100, 10, 3, seed=1)
    args = (inp["log_om_vb"], inp["log_1_min_om_vb"], inp["log_sig2_inv_vb"], inp["log_tau_vb"],
            inp["sig2_beta_vb"], inp["tau_vb"])
    with ResidentContext(inp["X"], inp["Y"]) as ctx:
        with pytest.raises(LocusHipError):                       # no state yet
            ctx.sweep(*args, inp["shuffled_ind"])
        ctx.set_state(inp["gam_vb"], inp["mu_beta_vb"])
        bad = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 8], dtype=np.float64)
        with pytest.raises(LocusHipError) as ei:
            ctx.sweep(*args, bad)
        assert ei.value.code == -3
        np.testing.assert_array_equal(ctx.get_state(("gam_vb",))["gam_vb"], inp["gam_vb"])
        ctx.sweep(*args, inp["shuffled_ind"])                    # still usable afterwards
        with pytest.raises(LocusHipError):                       # nothing to replace
            ctx.set_response(None, None)
        with pytest.raises(ValueError):
            ctx.set_state(inp["gam_vb"][:, :2], inp["mu_beta_vb"][:, :2])


@pytest.mark.parametrize("anneal", [None, (1, 2.0, 3)])
def test_locus_core_through_the_resident_context(oracle, hip, anneal):
    """R/locus_core.R's loop with ctx_sweep in place of coreLoop (the change INTEGRATION.md §2
    describes) against the restated driver: C1, five iterations."""
    from oracle import ref_driver
    from locus_amd.resident import locus_core_ctx_
    from test_driver_gpu import _setup
    sim, hyper, init, perm_fn = _setup()
    kw = dict(tol=1e-30, maxit=5, perm_fn=perm_fn, anneal=anneal, full_output=True)
    args = (sim["Y"], sim["X"], hyper, init["gam_vb"], init["mu_beta_vb"], init["sig2_beta_vb"],
            init["tau_vb"])
    ref, got = ref_driver.locus_core_(*args, **kw), locus_core_ctx_(*args, **kw)
    assert got["it"] == ref["it"] == 5
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"):
        np.testing.assert_allclose(got[k], ref[k], rtol=RTOL, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(got["om_vb"], ref["om_vb"], rtol=1e-9)
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"])
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


def test_c_abi_from_a_plain_c_program(hip):
    """tests/c/abi_smoke.c: the stateless coreLoop entry point and the resident context called
    from C (gcc, no Python in the data path), cross-checked against each other."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = os.path.join(here, "c", "abi_smoke")
    if not os.path.exists(exe):                       # normally built by __graft_entry__.build()
        subprocess.check_call(["gcc", "-O1", "-I", os.path.join(root, "include"),
                               os.path.join(here, "c", "abi_smoke.c"), "-o", exe,
                               "-L", os.path.join(root, "locus_amd"), "-llocus_hip", "-lm",
                               "-Wl,-rpath," + os.path.join(root, "locus_amd")])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "locus_amd") + ":" +
               os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([exe], capture_output=True, text=True, timeout=100, env=env)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr
