"""GPU parity of the five sweep routines (coreLoop, coreZLoop, coreStructLoop,
coreProbitLoop, coreLogitLoop) against the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest

from helpers import assert_parity, call_variant, inputs_for

pytestmark = pytest.mark.gpu

VARIANTS = ["identity", "z", "struct", "probit", "logit"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(200, 500, 20), (100, 75, 20), (64, 33, 1), (257, 40, 7),
                                   (1000, 64, 41)])
def test_sweep_matches_oracle(oracle, hip, variant, shape):
    n, p, d = shape
    inp = inputs_for(variant, n, p, d, seed=n + p + d)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} {shape}")


# explicit kernel geometries NPL,C,S,G : every reduction / staging path
GEOMS = ["4,1,1,1", "4,2,1,1", "4,4,1,2", "4,5,1,1", "4,8,1,1", "4,10,1,2", "4,16,1,1",
         "2,4,2,1", "2,5,2,2", "2,16,2,1", "8,5,1,2", "8,10,1,2", "16,5,1,4", "16,4,1,4",
         "2,2,4,1", "2,10,4,1", "4,4,4,1", "8,2,2,2"]


@pytest.mark.parametrize("geom", GEOMS)
def test_sweep_geometries_identity(oracle, hip, geom, monkeypatch):
    npl, c, s, g = map(int, geom.split(","))
    n = min(64 * npl * s, 64 * npl * s - 7)  # ragged rows inside the padded column
    p, d = 37, 2 * c * g + 3                  # ragged responses: partial last workgroup
    inp = inputs_for("identity", n, p, d, seed=11)
    ref = call_variant(oracle, "identity", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what=f"identity geom {geom}")


@pytest.mark.parametrize("geom", ["4,1,1,1", "4,4,1,1", "4,5,1,2", "2,8,2,1", "8,4,1,2", "8,5,1,2",
                                  "16,2,1,4", "2,2,4,1"])
def test_sweep_geometries_logit(oracle, hip, geom, monkeypatch):
    npl, c, s, g = map(int, geom.split(","))
    n = 64 * npl * s - 5
    p, d = 29, 2 * c * g + 1
    inp = inputs_for("logit", n, p, d, seed=5)
    ref = call_variant(oracle, "logit", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, "logit", inp)
    assert_parity(got, ref, what=f"logit geom {geom}")


# logit look-ahead kernel, explicit geometries -3,CR,RED,NPL (CR responses per workgroup, six waves
# of 64 NPL rows; RED = 1: the vector waves finish the reduction of the partial sums, 2: the chain
# wave does): ragged rows and responses, sweeps shorter and longer than the prefetch rings
LA_GEOMS = [f"-3,{cr},{red},{npl}" for npl in (1, 2, 3, 4, 5, 6, 8) for cr in (1, 2, 4)
            for red in (1, 2) if not (npl == 8 and cr == 4)] + ["-3,0,0,0"]


@pytest.mark.parametrize("geom", LA_GEOMS)
def test_logit_lookahead_geometries(oracle, hip, geom, monkeypatch):
    _, cr, _, npl = map(int, geom.split(","))
    n = 6 * 64 * npl - 5 if npl else 300
    p, d = 29, 2 * max(cr, 1) + 1
    inp = inputs_for("logit", n, p, d, seed=5)
    ref = call_variant(oracle, "logit", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, "logit", inp)
    assert_parity(got, ref, what=f"logit look-ahead geom {geom}")


def _la_random_cases(count, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        n = int(rng.integers(20, 3073))
        cr = int(rng.choice([1, 2, 4]))
        if 64 * 6 * 6 < n:
            cr = min(cr, 2)                      # CR = 4 is built up to NPL = 6
        cases.append((n, int(rng.integers(1, 120)), int(rng.integers(1, 4 * cr + 2)), cr,
                      int(rng.integers(1, 3))))
    return cases


@pytest.mark.parametrize("case", _la_random_cases(24, seed=2024), ids=lambda c: "n%d_p%d_d%d_cr%d_red%d" % c)
def test_logit_lookahead_random_shapes(oracle, hip, case, monkeypatch):
    """Seeded random shapes up to the kernel's limit of 3072 rows, NPL chosen by the library, both
    ways of finishing the reduction, sweeps of 1 .. 119 steps (shorter and longer than every ring)."""
    n, p, d, cr, red = case
    inp = inputs_for("logit", n, p, d, seed=n + 7 * p)
    ref = call_variant(oracle, "logit", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", f"-3,{cr},{red},0")
    got = call_variant(hip, "logit", inp)
    assert_parity(got, ref, what=f"logit look-ahead {case}")


@pytest.mark.parametrize("geom", ["-3,2,3,6", "-3,2,-1,2", "-3,3,0,2", "-3,4,0,8", "-3,1,0,1", "-3,1,2,1",
                                  "-3,1,0,7"])
def test_logit_lookahead_bad_geometry_is_refused(hip, geom, monkeypatch):
    """A look-ahead geometry that does not exist (reduction mode other than 0 / 1 / 2, 3 responses
    per workgroup, 4 with NPL = 8, no NPL = 7) or does not cover n (6 * 64 rows for n = 600) is an
    error before any launch: the caller's matrices are untouched."""
    from locus_amd._lib import LocusHipError
    inp = inputs_for("logit", 600, 17, 5, seed=2)
    before = {k: np.array(inp[k], copy=True) for k in ("gam_vb", "beta_vb", "mat_x_m1", "mu_beta_vb")}
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    with pytest.raises(LocusHipError) as ei:
        call_variant(hip, "logit", inp)
    assert ei.value.code in (-1, -2), ei.value          # EINVAL / EUNSUPPORTED
    assert "look-ahead" in str(ei.value)
    for k, v in before.items():
        assert np.array_equal(v, inp[k]), k


@pytest.mark.parametrize("shape", [(100, 1, 3), (100, 2, 1), (64, 7, 9), (450, 8, 2), (449, 9, 5),
                                   (1000, 203, 41)])
def test_logit_lookahead_short_and_long_sweeps(oracle, hip, shape, monkeypatch):
    n, p, d = shape
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-3,0,0,0")
    inp = inputs_for("logit", n, p, d, seed=n + p)
    ref = call_variant(oracle, "logit", inp)
    got = call_variant(hip, "logit", inp)
    assert_parity(got, ref, what=f"logit look-ahead {shape}")


def test_annealing_factor(oracle, hip):
    inp = inputs_for("identity", 200, 60, 9, seed=3, c=0.37)
    ref = call_variant(oracle, "identity", inp)
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what="identity c=0.37")


def test_bad_permutation_rejected(hip):
    from locus_amd._lib import LocusHipError
    inp = inputs_for("identity", 100, 10, 3, seed=1)
    inp["shuffled_ind"] = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 8], dtype=np.float64)
    before = inp["gam_vb"].copy()
    with pytest.raises(LocusHipError) as ei:
        call_variant(hip, "identity", inp)
    assert ei.value.code == -3
    assert np.array_equal(before, inp["gam_vb"])


@pytest.mark.parametrize("t,cs,drop", [(1, 2, 1), (1, 3, 2), (1, 2, 0), (2, 3, 2), (2, 2, 0)])
def test_cluster_timeout_is_a_named_failure(oracle, hip, t, cs, drop, monkeypatch):
    """A row-split cluster whose partner workgroup never runs (LOCUS_HIP_TEST_DROP_MEMBER makes
    that member of every cluster return at entry, as if it had not been dispatched): the other
    members give up after their bounded wait, the call fails with LOCUS_HIP_ETIMEOUT and its own
    text -- not with the bad-permutation code --, the stateless entry point leaves the caller's
    matrices alone, and the next, normal sweep is correct.  The reference's sweep cannot fail
    half-way (src/coreLoop.cpp:36-52)."""
    from locus_amd._lib import LocusHipError
    inp = inputs_for("identity", 300, 53, 39, seed=4)
    before = {k: np.array(inp[k], copy=True) for k in ("gam_vb", "beta_vb", "mat_x_m1", "mu_beta_vb")}
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", f"-1,{t},{cs},0")
    monkeypatch.setenv("LOCUS_HIP_TEST_DROP_MEMBER", str(drop))
    with pytest.raises(LocusHipError) as ei:
        call_variant(hip, "identity", inp)
    assert ei.value.code == -4, ei.value
    assert "timed out" in str(ei.value) and "shuffled_ind" not in str(ei.value)
    for k, v in before.items():
        assert np.array_equal(v, inp[k]), k
    monkeypatch.delenv("LOCUS_HIP_TEST_DROP_MEMBER")
    ref = call_variant(oracle, "identity", inp)
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what=f"identity after a cluster time-out, T = {t}, CS = {cs}")


def test_cluster_timeout_resident_state(hip, monkeypatch):
    """The same failure through the device-resident API: Sweeper.status() raises ETIMEOUT, the
    residual and the statistics are those from before the sweep (the documented contract), and
    once the residual is rebuilt from the state a normal sweep equals a sweep from that state."""
    import torch
    from locus_amd import device as dv
    from locus_amd._lib import LocusHipError
    n, p, d = 300, 70, 37
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    X = rng.normal(size=(n, p))
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    Xd = dv.to_device_colmajor(X, dev)
    px = dv.PackedX(Xd)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    Y = up(rng.normal(size=(d, n))).t()
    lo, l1 = up(np.full(p, -3.0) + rng.normal(scale=0.1, size=p)), up(np.full(p, -0.05))
    tau = rng.gamma(2.0, 1.0, size=d)
    s2 = 1.0 / ((n - 1 + 0.7) * tau)
    cons = [up(s2 * tau), up(1 / (2 * s2)), up(-(np.log(tau) + np.log(0.7) + np.log(s2)) / 2), up(s2)]
    perm = up(rng.permutation(p).astype(np.float64))
    state = torch.empty((3, p, d), dtype=torch.float64, device=dev)
    state[0], state[1] = up(rng.random((p, d)) ** 4), up(rng.normal(scale=0.3, size=(p, d)))
    state[2] = state[0] * state[1]

    def residual_of(st):
        M = torch.empty((d, n), dtype=torch.float64, device=dev).t()
        dv.gemm(n, d, p, px.Xp, 1, px.ld, st[2], d, 1, M, 1, n)
        return dv.residual(Y, M)
    R = residual_of(state)
    R0 = R.clone()
    stats = torch.full((4, d), -7.0, dtype=torch.float64, device=dev)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-1,1,2,0")
    monkeypatch.setenv("LOCUS_HIP_TEST_DROP_MEMBER", "1")
    sw = dv.Sweeper(px, d, dev)
    sw.identity(perm, lo, l1, *cons, 1.0, R, state, stats)
    with pytest.raises(LocusHipError) as ei:
        sw.status()
    assert ei.value.code == -4
    assert torch.equal(R, R0) and bool((stats == -7.0).all())
    monkeypatch.delenv("LOCUS_HIP_TEST_DROP_MEMBER")
    # recover as documented: residual from the (partly updated) state, then sweep normally
    st_a = state.clone()
    st_a[2] = st_a[0] * st_a[1]
    st_b = st_a.clone()
    Ra, Rb = residual_of(st_a), residual_of(st_b)
    sw.identity(perm, lo, l1, *cons, 1.0, Ra, st_a, stats)
    sw.status()
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-1,1,1,0")      # one workgroup per tile: no clusters
    sw1 = dv.Sweeper(px, d, dev)
    sw1.identity(perm, lo, l1, *cons, 1.0, Rb, st_b, None)
    sw1.status()
    np.testing.assert_allclose(st_a.cpu().numpy(), st_b.cpu().numpy(), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(Ra.cpu().numpy(), Rb.cpu().numpy(), rtol=1e-9, atol=1e-12)


def test_two_sweeps_in_a_row_deterministic(hip):
    inp = inputs_for("identity", 200, 120, 20, seed=8)
    a = call_variant(hip, "identity", inp)
    b = call_variant(hip, "identity", inp)
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{k} differs run to run"


# matrix-core kernel, explicit geometries -1,T,S,KW4 (T tiles of 16 responses per workgroup,
# S waves, KW4 16-row tiles per wave): ragged rows / responses / last block of the sweep
MFMA_GEOMS = ([f"-1,1,1,{kt}" for kt in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 21)]
              + [f"-1,1,2,{kt}" for kt in range(1, 12)]          # rows cut over 2 workgroups
              + [f"-1,1,3,{kt}" for kt in range(1, 8)] + ["-1,1,3,0"]  # ... over 3
              + [f"-1,2,8,{kt}" for kt in range(1, 12)] + ["-1,1,1,0", "-1,1,2,0", "-1,1,0,0", "-1,2,8,0"]
              # two tiles (32 responses) per cluster of 2, 3 workgroups
              # (16-row tiles per wave up to what covers the kernel's n <= 2016)
              + [f"-1,2,{cs},{kt}" for cs, top in ((2, 8), (3, 7)) for kt in range(1, top + 1)]
              + ["-1,2,2,0", "-1,2,3,0"]
              # the odd deal: waves 3..5 of every workgroup take one 16-row tile fewer (G = 100 + KT)
              + [f"-1,1,{cs},{100 + kt}" for cs, top in ((1, 12), (2, 11), (3, 7)) for kt in range(2, top + 1)]
              + [f"-1,2,{cs},{100 + kt}" for cs, top in ((2, 8), (3, 7)) for kt in range(2, top + 1)])


@pytest.mark.parametrize("geom", MFMA_GEOMS)
def test_mfma_sweep_geometries_identity(oracle, hip, geom, monkeypatch):
    _, t, cs, kt = map(int, geom.split(","))
    if t == 2:
        waves = 6 * cs if cs in (2, 3) else 7      # clusters of cs workgroups / one workgroup
    else:
        waves = 6 * cs if cs in (2, 3) else 6      # matrix waves over the rows
    tiles = waves * kt
    if kt >= 100:                                  # odd deal: half of the waves hold kt - 1 tiles
        kt -= 100
        tiles = waves * kt - waves // 2
    n = min(16 * tiles - 5, 2016) if kt else 300            # 16-row tiles, ragged; kernel limit
    p, d = 53, 16 * t + 16 * t + 7    # 3 blocks + 5 steps; partial last workgroup / tile
    inp = inputs_for("identity", n, p, d, seed=21)
    ref = call_variant(oracle, "identity", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what=f"identity mfma geom {geom}")


EXTREME_CFGS = [("identity", "0,0,0,0"), ("identity", "-1,2,8,0"), ("identity", "-1,1,1,0"),
                ("identity", "-1,1,2,0"), ("identity", "-1,1,3,0"), ("identity", "-1,2,3,0"),
                ("identity", "4,1,1,1"),
                ("logit", "0,0,0,0"), ("logit", "-3,2,0,0"), ("logit", "4,1,1,1")]


@pytest.mark.parametrize("variant,geom", EXTREME_CFGS)
def test_extreme_log_odds(oracle, hip, variant, geom, monkeypatch):
    """Prior log-odds far outside the range of exp: gam must saturate at 0 / 1 exactly as the
    reference's logOnePlusExp does (src/utils.cpp:3-12), through every kernel's own sigmoid
    (Horner and Estrin exponentials, clamp at -800, Newton division)."""
    n, p, d = 300, 53, 39
    inp = inputs_for(variant, n, p, d, seed=77)
    inp["log_om_vb"][[3, 17]] = [-2000.0, -745.5]          # gam -> 0
    inp["log_1_min_om_vb"][[5, 40]] = [-1.0e4, -700.0]     # gam -> 1
    inp["log_om_vb"][8], inp["log_1_min_om_vb"][8] = -1.0e-300, -690.0
    ref = call_variant(oracle, variant, inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} {geom} extreme log-odds")
    g = got["gam_vb"]
    assert (g[3] < 1e-300).all() and (g[5] == 1.0).all() and np.isfinite(g).all()


@pytest.mark.parametrize("variant,geom", [c for c in EXTREME_CFGS if c[1] != "4,1,1,1"])
def test_nan_log_odds_propagate(oracle, hip, variant, geom, monkeypatch):
    """A NaN prior log-odds of one covariate: the reference's std::exp propagates it into gam, beta
    and, through the residual, into everything visited later.  The kernels clamp their exponent
    with a select (not fmax), so the same entries are NaN -- none is silently turned into 0 or 1."""
    n, p, d = 200, 40, 21
    inp = inputs_for(variant, n, p, d, seed=5)
    j = int(inp["shuffled_ind"][25])        # visited 26th: 25 clean updates before it
    inp["log_om_vb"][j] = np.nan
    ref = call_variant(oracle, variant, inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    got = call_variant(hip, variant, inp)
    for k in ref:
        assert np.array_equal(np.isnan(np.asarray(got[k])), np.isnan(np.asarray(ref[k]))), f"{k}: NaN pattern differs"
        ok = ~np.isnan(np.asarray(ref[k]))
        np.testing.assert_allclose(np.asarray(got[k])[ok], np.asarray(ref[k])[ok], rtol=1e-9, atol=1e-11)
    assert np.isnan(ref["gam_vb"][j]).all() and not np.isnan(ref["gam_vb"]).all()


@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit"])
@pytest.mark.parametrize("shape", [(200, 500, 20), (1000, 64, 41), (100, 16, 3), (64, 7, 1),
                                   (333, 1, 5), (500, 33, 70)])
def test_mfma_sweep_variants(oracle, hip, variant, shape, monkeypatch):
    n, p, d = shape
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", f"-1,{1 if d <= 16 else 2},{2 if n >= 300 else 1},0")
    inp = inputs_for(variant, n, p, d, seed=n + p + d, c=0.6 if variant == "identity" else 1.0)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} {shape} mfma")


@pytest.mark.parametrize("n,d", [(769, 1270), (1000, 1280), (1000, 1281), (1153, 700), (2016, 1000),
                                 (384, 900), (385, 900), (768, 1300), (1000, 4100)])
def test_default_dispatch_cluster_boundaries(oracle, hip, n, d, monkeypatch):
    """Default dispatch around the limits of the cluster-size choice (one workgroup per tile up
    to n = 384; three per tile for n > 768 while 3 x tiles fit on the CUs, else two, else one)
    and of the instantiated row tilings, with ragged last tiles; d = 4100 adds the planned split
    into a T = 2 launch and a clustered remainder.  Short sweep (p = 40) against the oracle."""
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    inp = inputs_for("identity", n, 40, d, seed=n + d)
    ref = call_variant(oracle, "identity", inp)
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what=f"identity default dispatch n={n} d={d}")


@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit"])
@pytest.mark.parametrize("shape", [(1000, 64, 41), (500, 33, 70), (130, 21, 3)])
def test_mfma_three_member_clusters_variants(oracle, hip, variant, shape, monkeypatch):
    """Rows cut over three workgroups (the default for n > 768 and <= 80 tiles) for every member of
    the identity family, ragged rows and tiles included."""
    n, p, d = shape
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-1,1,3,0")
    inp = inputs_for(variant, n, p, d, seed=3 * n + p + d, c=0.6 if variant == "identity" else 1.0)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} {shape} three-member clusters")


@pytest.mark.parametrize("cs", [2, 3])
@pytest.mark.parametrize("variant", ["identity", "z", "struct", "probit"])
@pytest.mark.parametrize("shape", [(1000, 64, 71), (500, 33, 70), (130, 21, 3), (333, 100, 33)])
def test_mfma_two_tile_clusters_variants(oracle, hip, variant, shape, cs, monkeypatch):
    """Two tiles (32 responses) per cluster of 2, 3 workgroups for every member of the identity
    family: ragged rows, a ragged last cluster (one tile, or part of one), sweeps of 2 to 7 blocks."""
    n, p, d = shape
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", f"-1,2,{cs},0")
    inp = inputs_for(variant, n, p, d, seed=5 * n + p + d + cs, c=0.6 if variant == "identity" else 1.0)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} {shape} two tiles per cluster of {cs}")


@pytest.mark.parametrize("variant", ["identity", "struct"])
def test_mfma_planned_multi_launch(oracle, hip, variant, monkeypatch):
    """The default (planned) mode splits the responses over several launches when they exceed
    what the CUs hold at once: T = 2 launches of 32 responses per CU, then a T = 1 remainder.
    Forced at small d by pretending the device has 2 CUs: d = 150 -> 64 + 64 (T = 2) + 22 (T = 1)."""
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    monkeypatch.setenv("LOCUS_HIP_MFMA_CUS", "2")
    inp = inputs_for(variant, 300, 70, 150, seed=5)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} planned multi-launch")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(200, 90, 7), (3, 5, 2), (257, 33, 1), (513, 20, 4)])
def test_streaming_kernel_forced(oracle, hip, variant, shape, monkeypatch):
    """stream_sweep_kernel (the fallback for n beyond the register-resident kernels) on shapes the
    other kernels also cover: every variant, ragged rows, n < one wave."""
    n, p, d = shape
    inp = inputs_for(variant, n, p, d, seed=n + d)
    ref = call_variant(oracle, variant, inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-2,0,0,0")
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} streaming {shape}")


@pytest.mark.parametrize("variant,n", [("identity", 2049), ("struct", 3000), ("z", 2500),
                                       ("probit", 4096), ("logit", 2100), ("logit", 4090),
                                       ("probit", 4100), ("logit", 8000), ("identity", 16001)])
def test_more_than_2048_samples(oracle, hip, variant, n, monkeypatch):
    """No limit on n.  2048 < n <= 4096: the wide instantiations of the vector kernel (NPL = 16,
    four row slices, four staging pieces per thread).  Above: the streaming kernel -- residual
    column in LDS up to 120 KB (n <= 15 359; logit, which also keeps the 2 psi column:
    n <= 7679), in global memory beyond."""
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    inp = inputs_for(variant, n, 24, 2 * (n % 7) + 3, seed=n)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, what=f"{variant} n={n}")


@pytest.mark.parametrize("geom", ["16,1,4,2", "16,2,4,2", "16,4,4,2"])
def test_wide_vector_geometries(oracle, hip, geom, monkeypatch):
    """The wide (2048 < n <= 4096) instantiations, every C, ragged rows and responses."""
    npl, c, s, g = map(int, geom.split(","))
    inp = inputs_for("identity", 4096 - 37, 21, 2 * c * g + 3, seed=c)
    ref = call_variant(oracle, "identity", inp)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", geom)
    monkeypatch.setenv("LOCUS_HIP_SWEEP_VECTOR", "1")
    got = call_variant(hip, "identity", inp)
    assert_parity(got, ref, what=f"identity wide geom {geom}")


@pytest.mark.parametrize("shape,cus,only", [((300, 200, 70), None, None), ((300, 70, 150), "2", None),
                                             ((130, 40, 5), None, None),
                                             # several launches of ONE geometry share a packed region:
                                             # one tile per CU on 8 CUs (128 + 22), and clusters of
                                             # two on 24 CUs (128 + 128 + 44) -- what 2 GPUs do at C3
                                             ((300, 60, 150), "8", None), ((300, 60, 300), "24", "1,2")])
def test_operands_packed_ahead(hip, shape, cus, only, monkeypatch):
    """locus_hip_sweep_prepack_dev + locus_hip_sweep_dev(prepacked = 1), as Sweeper.prepack queues
    them on a side stream while another sweep runs: three sweeps in a row with the next one's
    operands packed ahead are bit-identical to three plain sweeps, also when the plan has several
    launch geometries side by side in the workspace (2 pretended CUs: d = 150 -> 64 + 64 + 22) and
    when the order finally swept is not the one that was packed for (the sweep packs itself)."""
    import torch
    from locus_amd import device as dv
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    if cus:
        monkeypatch.setenv("LOCUS_HIP_MFMA_CUS", cus)
    if only:
        monkeypatch.setenv("LOCUS_HIP_MFMA_PLAN", only)
    n, p, d = shape
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    X = rng.normal(size=(n, p))
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    px = dv.PackedX(dv.to_device_colmajor(X, dev))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    gam, mu = rng.random((p, d)) ** 4, rng.normal(scale=0.3, size=(p, d))
    R0 = rng.normal(size=(d, n))
    los = [up(np.full(p, -3.0) + rng.normal(scale=0.1, size=p)) for _ in range(3)]
    l1 = up(np.full(p, -0.05))
    tau = rng.gamma(2.0, 1.0, size=d)
    s2 = 1.0 / ((n - 1 + 0.7) * tau)
    cons = [up(s2 * tau), up(1 / (2 * s2)), up(-(np.log(tau) + np.log(0.7) + np.log(s2)) / 2), up(s2)]
    perms = [rng.permutation(p).astype(np.float64) for _ in range(4)]

    def run(ahead):
        state = torch.empty((3, p, d), dtype=torch.float64, device=dev)
        state[0], state[1] = up(gam), up(mu)
        state[2] = state[0] * state[1]
        R = up(R0).t()
        stats = torch.empty((4, d), dtype=torch.float64, device=dev)
        sw = dv.Sweeper(px, d, dev)
        nxt = None
        for i in range(3):
            ind = nxt if nxt is not None else up(perms[i])
            sw.identity(ind, los[i], l1, *cons, 1.0, R, state, stats)
            nxt = None
            if ahead and i < 2:
                # i = 0: pack for the order swept next; i = 1: pack for ANOTHER order (perms[3])
                got = sw.prepack(perms[i + 1] if i == 0 else perms[3])
                assert got is not None
                nxt = got if i == 0 else None
            sw.status()
        return state.cpu().numpy(), R.cpu().numpy(), stats.cpu().numpy()

    plain, ahead = run(False), run(True)
    for a, b in zip(plain, ahead):
        assert np.array_equal(a, b)
    assert not np.array_equal(plain[0][0], gam)


@pytest.mark.parametrize("vector", [False, True, "stream"])
def test_partial_sweeps_compose(hip, vector, monkeypatch):
    """nsteps < p (locus_hip_sweep_args.nsteps): visiting the first k covariates of the order and
    then the remaining p - k equals one full sweep, on the resident state (both kernels)."""
    import torch
    from locus_amd import device as dv
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    if vector == "stream":
        monkeypatch.setenv("LOCUS_HIP_SWEEP_CFG", "-2,0,0,0")
    elif vector:
        monkeypatch.setenv("LOCUS_HIP_SWEEP_VECTOR", "1")
    n, p, d, k = 150, 77, 21, 30
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    X = rng.normal(size=(n, p))
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    px = dv.PackedX(dv.to_device_colmajor(X, dev))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    gam, mu = rng.random((p, d)) ** 4, rng.normal(scale=0.3, size=(p, d))
    R0 = rng.normal(size=(d, n))
    lo, l1 = up(np.full(p, -3.0) + rng.normal(scale=0.1, size=p)), up(np.full(p, -0.05))
    tau = rng.gamma(2.0, 1.0, size=d)
    s2 = 1.0 / ((n - 1 + 0.7) * tau)
    cons = [up(s2 * tau), up(1 / (2 * s2)), up(-(np.log(tau) + np.log(0.7) + np.log(s2)) / 2), up(s2)]
    perm = rng.permutation(p).astype(np.float64)

    def run(parts):
        state = torch.empty((3, p, d), dtype=torch.float64, device=dev)
        state[0], state[1] = up(gam), up(mu)
        state[2] = state[0] * state[1]
        R = up(R0).t()
        sw = dv.Sweeper(px, d, dev)
        for order, nsteps in parts:
            sw.identity(up(order), lo, l1, *cons, 1.0, R, state, None, nsteps)
            sw.status()
        return state.cpu().numpy(), R.cpu().numpy()

    full_s, full_r = run([(perm, 0)])
    part_s, part_r = run([(perm, k), (np.roll(perm, -k), p - k)])
    np.testing.assert_allclose(part_s, full_s, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(part_r, full_r, rtol=1e-10, atol=1e-12)
    assert not np.array_equal(full_s[0], gam)
