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
    for kw in (dict(list_cv={}), dict(list_blocks={}), dict(list_groups={})):
        with pytest.raises(NotImplementedError):
            locus(sim["Y"], X, P0, _test_backend=cpu_device, verbose=False, **kw)
    with pytest.raises(ValueError):
        locus(sim["Y"], X, P0, ind_bin=[1], _test_backend=cpu_device, verbose=False)
    with pytest.raises(ValueError):
        locus(sim["Y"], X, P0, link="logit", anneal=(1, 2.0, 3), _test_backend=cpu_device,
              verbose=False)
