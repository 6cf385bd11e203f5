"""locus_hip_sweep_static_dev -- the entry point that allocates nothing (include/locus_hip.h):
same sweep as locus_hip_sweep_dev, scratch in a fixed array of the library's code object instead
of a caller-owned workspace.  Through the C ABI (ctypes), on two calls in a row, the second with a
smaller input (so that the scratch still holds the first call's step records, visited flags and
operand copies): results must be bit-identical to locus_hip_sweep_dev's on the same inputs -- it
is the same kernels on the same data, only the scratch address differs.  Collected last
(test_zz_*): a failure here must not hide the other GPU tests behind `pytest -x`."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(n, p, d, logit, seed, dev):
    import torch
    from locus_amd import device as dv
    rng = np.random.default_rng(seed)
    X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float64)
    X[0, X.std(axis=0) == 0] += 1.0
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    px = dv.PackedX(dv.to_device_colmajor(X, dev))
    prob = dict(n=n, p=p, d=d, px=px, logit=logit,
                gam=rng.random((p, d)) ** 4, mu=rng.normal(scale=0.3, size=(p, d)),
                R0=rng.normal(size=(d, n)),
                lo=up(np.full(p, -3.0) + rng.normal(scale=0.1, size=p)), l1=up(np.full(p, -0.05)),
                order=up(rng.permutation(p).astype(np.float64)))
    if logit:
        psi2 = 2 * (0.05 + 0.2 * rng.random((d, n)))                    # W2 = 2 psi, [d][n] storage
        W2 = up(psi2).t()
        s2 = up(1.0 / ((X ** 2).T @ psi2.T + 0.7))                     # [p][d]
        xwx = torch.empty((p, d), dtype=torch.float64, device=dev)
        dv.gemm(p, d, n, px.Xp, px.ld, 1, W2, 1, n, xwx, d, 1, sqA=True)
        prob.update(W2=W2, s2=s2, xwx=xwx, half_lsi=float(np.log(0.7) / 2))
    else:
        tau = rng.gamma(2.0, 1.0, size=d)
        s2 = 1.0 / ((n - 1 + 0.7) * tau)
        prob.update(cons=[up(s2 * tau), up(1 / (2 * s2)),
                          up(-(np.log(tau) + np.log(0.7) + np.log(s2)) / 2), up(s2)])
    return prob


def _run(prob, static, dev, order=None):
    """One sweep through locus_hip_sweep_dev (static = False: torch-owned workspace) or
    locus_hip_sweep_static_dev; returns (state, R, stats) as numpy."""
    import torch
    from locus_amd import _lib
    from locus_amd import device as dv
    n, p, d = prob["n"], prob["p"], prob["d"]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    state = torch.empty((3, p, d), dtype=torch.float64, device=dev)
    state[0], state[1] = up(prob["gam"]), up(prob["mu"])
    state[2] = state[0] * state[1]
    R = up(prob["R0"]).t()
    stats = torch.zeros((4, d), dtype=torch.float64, device=dev)
    sw = dv.Sweeper(prob["px"], d, dev)
    ind = prob["order"] if order is None else order
    a = sw._args(ind, prob["lo"], prob["l1"], R, state, stats)
    if prob["logit"]:
        a.s2, a.xwx, a.W2 = prob["s2"].data_ptr(), prob["xwx"].data_ptr(), prob["W2"].data_ptr()
        a.half_lsi = prob["half_lsi"]
    else:
        a.scale, a.hinv, a.cst, a.s2k = (t.data_ptr() for t in prob["cons"])
        a.cfac = 1.0
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.lib()
    if static:
        a.workspace, a.workspace_bytes = None, 0          # must be ignored
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        rc = lib.locus_hip_sweep_static_dev(ctypes.byref(a), int(prob["logit"]), stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before, "the call changed free device memory"
        _lib.check(rc)
        _lib.check(lib.locus_hip_sweep_status_dev(None, p, stream))
    else:
        _lib.check(lib.locus_hip_sweep_dev(ctypes.byref(a), int(prob["logit"]), stream))
        sw.status()
    torch.cuda.synchronize()
    return state.cpu().numpy(), R.cpu().numpy(), stats.cpu().numpy()


@pytest.mark.parametrize("logit", [False, True])
def test_static_workspace_sweep_equals_sweep_dev_twice_in_a_row(hip, logit, monkeypatch):
    import torch
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    dev = torch.device("cuda:0")
    big = _problem(300, 400, 70, logit, seed=3, dev=dev)       # then a smaller one: fewer rows,
    small = _problem(130, 90, 21, logit, seed=4, dev=dev)      # covariates and responses
    before = torch.cuda.memory_allocated(dev)
    # set-up, once per device: the first use looks up the array's device address (the HIP runtime
    # materialises the code object's 64 MiB array then); a status query is the documented warm-up
    from locus_amd import _lib
    _lib.lib().locus_hip_sweep_status_dev(None, 1, None)
    for prob in (big, small):
        ref = _run(prob, False, dev)
        got = _run(prob, True, dev)
        for name, g, r in zip(("state", "R", "stats"), got, ref):
            assert np.array_equal(g, r), (
                f"{name} differs between sweep_static_dev and sweep_dev (p = {prob['p']}): "
                f"max abs diff {np.abs(g - r).max():.3e}")
        assert not np.array_equal(got[0][0], prob["gam"])        # the sweep did run
    assert torch.cuda.memory_allocated(dev) == before


def test_static_workspace_limits_and_errors(hip):
    import torch
    from locus_amd import _lib
    from locus_amd._lib import LocusHipError
    dev = torch.device("cuda:0")
    prob = _problem(100, 40, 5, False, seed=1, dev=dev)
    # a repeated index is reported through the static workspace's own error word, state untouched
    bad = prob["order"].clone()
    bad[3] = bad[7]
    with pytest.raises(LocusHipError) as ei:
        _run(prob, True, dev, order=bad)
    assert ei.value.code == -3
    ok = _run(prob, True, dev)                                   # and the next call is clean again
    assert np.array_equal(ok[0], _run(prob, False, dev)[0])
    # beyond the array: a clear error, nothing launched
    a = _lib.SweepArgs()
    a.n, a.p, a.d = 1000, 50000, 16
    rc = _lib.lib().locus_hip_sweep_static_dev(ctypes.byref(a), 0, None)
    assert rc == -2
    msg = _lib.lib().locus_hip_last_error().decode()
    assert "static workspace holds" in msg and "locus_hip_sweep_dev" in msg
