"""CPU tests of the drivers' host side: the O(p + d) arithmetic of the four CAVI drivers, their
orchestration and the response (d) sharding with its one all-reduce per iteration.

The HIP device layer is replaced by tests/cpu_device.py (numpy + C oracle on CPU tensors) via
the drivers' `_test_backend` hook; results are compared with the full-matrix numpy restatement
of the R drivers (oracle/ref_driver.py).  world_size 2 runs under gloo in two processes with
an uneven split of the responses (d = 7 -> 3 + 4).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for pth in (ROOT, HERE):
    if pth not in sys.path:
        sys.path.insert(0, pth)

N, P, D, P0 = 60, 80, 7, 6
MAXIT = 3
KINDS = ["identity", "anneal", "struct", "logit", "probit", "z", "z-anneal", "mix"]


def _problem(kind):
    from locus_amd.prepare import add_intercept_, auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    from locus_amd.struct import set_struct
    link = kind if kind in ("logit", "probit") else "identity"
    sim = simulate(N, P, D, P0, seed=17, link=link)
    hyper = auto_set_hyper_(sim["Y"], P, P0, q=1 if link != "identity" else None, link=link,
                            struct=(kind == "struct"))
    init = auto_set_init_(sim["Y"], P, P0, user_seed=17, link=link)
    extra = {}
    if link != "identity":
        Z = add_intercept_(N, D, hyper, init, link)
        rng = np.random.default_rng(3)
        Zc = rng.normal(size=(N, 1))
        Zc = (Zc - Zc.mean(axis=0)) / Zc.std(axis=0, ddof=1)
        extra["Z"] = np.asfortranarray(np.hstack([Z, Zc]))
        hyper["phi"] = np.array([1e-3, 1.0])
        hyper["xi"] = np.array([1e-3, 1.0])
        init["alpha_vb"] = rng.normal(scale=0.1, size=(2, D))
        init["sig2_alpha_vb"] = (np.array([1.0, 0.5]) if link == "probit"
                                 else rng.gamma(2.0, 0.2, size=(2, D)))
    if kind == "mix":
        return _mix_problem()
    if kind in ("z", "z-anneal"):
        rng = np.random.default_rng(4)
        Zc = rng.normal(size=(N, 3))
        extra["Z"] = np.asfortranarray((Zc - Zc.mean(axis=0)) / Zc.std(axis=0, ddof=1))
        hyper["phi"] = np.array([1.0, 1.0, 2.0])
        hyper["xi"] = np.array([1.0, 0.5, 1.0])
        init["alpha_vb"] = rng.normal(scale=0.2, size=(3, D))
        init["sig2_alpha_vb"] = rng.gamma(2.0, 0.3, size=(3, D))
        perms = [np.random.default_rng(50 + i).permutation(P).astype(np.float64) for i in range(16)]
        aperms = [np.random.default_rng(90 + i).permutation(3) for i in range(16)]
        return sim, hyper, init, extra, (lambda it: perms[it]), (lambda it: aperms[it])
    if kind == "struct":
        extra["ls"] = set_struct(N, P, np.arange(1, P + 1, 20))
    perms = [np.random.default_rng(50 + i).permutation(P).astype(np.float64) for i in range(16)]
    aperms = [np.random.default_rng(70 + i).permutation(2) for i in range(16)]
    return sim, hyper, init, extra, (lambda it: perms[it]), (lambda it: aperms[it])


def _mix_problem(n=N, p=P, d=D, p0=P0, seed=17):
    """Continuous (centred) and binary (0/1) responses interleaved; Z = intercept + 2 covariates."""
    from locus_amd.prepare import auto_set_hyper_, auto_set_init_
    from locus_amd.simulate import simulate
    sim = simulate(n, p, d, p0, seed=seed)
    rng = np.random.default_rng(seed + 1)
    ind_bin = np.sort(rng.choice(d, size=max(1, d // 2 - 1), replace=False))
    cont = np.setdiff1d(np.arange(d), ind_bin)
    Y = sim["Y"].copy()
    Y[:, ind_bin] = (Y[:, ind_bin] > 0).astype(np.float64)
    sim = dict(sim, Y=np.asfortranarray(Y))
    hyper = auto_set_hyper_(Y[:, cont], p, p0)
    b_full = d * (p - p0) / p0
    hyper["b"] = np.full(p, b_full)
    hyper["phi"], hyper["xi"] = np.array([1e-3, 1.0, 1.0]), np.array([1e-3, 1.0, 0.5])
    init = auto_set_init_(Y, p, p0, user_seed=seed)
    tau = np.ones(d)
    tau[cont] = 1.0 / np.median(Y[:, cont].var(axis=0, ddof=1))
    init["tau_vb"] = tau
    init["sig2_beta_vb"] = 1.0 / rng.gamma(2.0, 1e-2 * tau)
    Zc = rng.normal(size=(n, 2))
    Zc = (Zc - Zc.mean(axis=0)) / Zc.std(axis=0, ddof=1)
    extra = dict(Z=np.asfortranarray(np.hstack([np.ones((n, 1)), Zc])), ind_bin=ind_bin)
    init["alpha_vb"] = rng.normal(scale=0.1, size=(3, d))
    init["sig2_alpha_vb"] = rng.gamma(2.0, 0.3, size=(3, d))
    perms = [np.random.default_rng(50 + i).permutation(p).astype(np.float64) for i in range(16)]
    aperms = [np.random.default_rng(90 + i).permutation(3) for i in range(16)]
    return sim, hyper, init, extra, (lambda it: perms[it]), (lambda it: aperms[it])


def _run(kind, which, comm=None):
    """which = 'ref' (oracle/ref_driver.py, full matrices), 'got' (product host side on the
    CPU emulation of the device layer) or 'hip' (the product on the GPU, tests/test_driver_gpu.py)."""
    import cpu_device
    backend = None if which == "hip" else cpu_device
    sim, hyper, init, extra, perm_fn, aperm_fn = _problem(kind)
    Y, X = sim["Y"], sim["X"]
    common = dict(tol=1e-30, maxit=MAXIT, perm_fn=perm_fn)
    if kind == "mix":
        args = (Y, X, extra["Z"], extra["ind_bin"], hyper, init["gam_vb"], init["alpha_vb"],
                init["mu_beta_vb"], init["sig2_alpha_vb"], init["sig2_beta_vb"], init["tau_vb"])
        if which == "ref":
            from oracle import ref_driver as R
            return R.locus_mix_core_(*args, alpha_perm_fn=aperm_fn, **common)
        from locus_amd.locus_mix_core import locus_mix_core_
        return locus_mix_core_(*args, alpha_perm_fn=aperm_fn, comm=comm, full_output=True,
                               _test_backend=backend, **common)
    if kind in ("z", "z-anneal"):
        an = (1, 2.0, 2) if kind == "z-anneal" else None
        args = (Y, X, extra["Z"], hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
                init["sig2_alpha_vb"], init["sig2_beta_vb"], init["tau_vb"])
        if which == "ref":
            from oracle import ref_driver as R
            return R.locus_z_core_(*args, anneal=an, alpha_perm_fn=aperm_fn, **common)
        from locus_amd.locus_z_core import locus_z_core_
        return locus_z_core_(*args, anneal=an, alpha_perm_fn=aperm_fn, comm=comm,
                             full_output=True, _test_backend=backend, **common)
    if which == "ref":
        from oracle import ref_driver as R
        if kind in ("identity", "anneal"):
            return R.locus_core_(Y, X, hyper, init["gam_vb"], init["mu_beta_vb"],
                                 init["sig2_beta_vb"], init["tau_vb"],
                                 anneal=(1, 2.0, 2) if kind == "anneal" else None, **common)
        if kind == "struct":
            return R.locus_struct_core_(Y, X, hyper, init["gam_vb"], init["mu_beta_vb"],
                                        init["sig2_beta_vb"], init["tau_vb"],
                                        extra["ls"]["vec_fac_st"], **common)
        fn = getattr(R, f"locus_{kind}_core_")
        return fn(Y, X, extra["Z"], hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
                  init["sig2_alpha_vb"], init["sig2_beta_vb"], alpha_perm_fn=aperm_fn, **common)
    kw = dict(common, comm=comm, full_output=True, _test_backend=backend)
    if kind in ("identity", "anneal"):
        from locus_amd.locus_core import locus_core_
        return locus_core_(Y, X, hyper, init["gam_vb"], init["mu_beta_vb"], init["sig2_beta_vb"],
                           init["tau_vb"], anneal=(1, 2.0, 2) if kind == "anneal" else None, **kw)
    if kind == "struct":
        from locus_amd.locus_struct_core import locus_struct_core_
        return locus_struct_core_(Y, X, hyper, init["gam_vb"], init["mu_beta_vb"],
                                  init["sig2_beta_vb"], init["tau_vb"], extra["ls"], **kw)
    from locus_amd import locus_bin_core
    fn = getattr(locus_bin_core, f"locus_{kind}_core_")
    return fn(Y, X, extra["Z"], hyper, init["gam_vb"], init["alpha_vb"], init["mu_beta_vb"],
              init["sig2_alpha_vb"], init["sig2_beta_vb"], alpha_perm_fn=aperm_fn, **kw)


def _compare(kind, got, ref):
    assert got["it"] == ref["it"] == MAXIT
    keys = ["gam_vb", "mu_beta_vb", "beta_vb", "mat_x_m1"]
    if kind in ("logit", "probit", "z", "z-anneal", "mix"):
        keys += ["alpha_vb", "mat_z_mu"]
    for k in keys:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=f"{kind} {k}")
    for k in ("om_vb", "theta_vb"):
        if k in ref and ref[k] is not None:
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=f"{kind} {k}")
    assert abs(got["lb_opt"] - ref["lb_opt"]) <= 1e-8 * abs(ref["lb_opt"]), kind
    assert np.array_equal(got["gam_vb"] > 0.5, ref["gam_vb"] > 0.5)


@pytest.mark.parametrize("kind", KINDS)
def test_host_side_single_rank(oracle, kind):
    _compare(kind, _run(kind, "got"), _run(kind, "ref"))


GOLDEN_KEYS = ("gam_vb", "mu_beta_vb", "beta_vb", "lb_opt")


def check_against_golden(kind, res, rtol):
    """Committed results of the restated drivers (tests/golden/drivers_small.npz, written by
    tests/golden/make_golden.py) for the problems of _problem()."""
    f = np.load(os.path.join(HERE, "golden", "drivers_small.npz"))
    key = kind.replace("-", "_")
    for k in GOLDEN_KEYS[:3]:
        np.testing.assert_allclose(res[k], f[f"{key}__{k}"], rtol=rtol, atol=1e-12,
                                   err_msg=f"{kind} {k}")
    assert abs(res["lb_opt"] - float(f[f"{key}__lb_opt"])) <= max(rtol, 1e-8) * abs(res["lb_opt"])
    assert np.array_equal(res["gam_vb"] > 0.5, f[f"{key}__gam_vb"] > 0.5)


@pytest.mark.parametrize("kind", KINDS)
def test_restated_drivers_against_golden(oracle, kind):
    """Guards oracle/ref_driver.py against drift."""
    check_against_golden(kind, _run(kind, "ref"), rtol=1e-11)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locus_amd.comm import DistComm
        comm = DistComm()
        got = _run(kind, "got", comm=comm)
        ref = _run(kind, "ref") if rank == 0 else None
        if rank == 0:
            _compare(kind, got, ref)
        # every rank must hold the same gathered result and the same ELBO, bit for bit
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


@pytest.mark.parametrize("kind", KINDS)
def test_response_sharding_world2_gloo(oracle, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    for rank, err in res:
        assert err is None, f"rank {rank}: {err}"


def test_shard_bounds_cover_all_responses():
    from locus_amd.comm import shard_bounds
    for d in (1, 7, 20, 1000, 10000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(d, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == d
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
