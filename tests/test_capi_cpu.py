"""CPU checks of the boundary: liblocus_hip.so loads, exports every symbol the header
declares, and the argument validation that runs before any GPU work behaves."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "locus_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(locus_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from locus_amd import _lib
    h = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(h, nm), f"{nm} declared in include/locus_hip.h but not exported"
        assert nm in _lib.SIGNATURES, f"{nm} has no ctypes signature in locus_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert h.locus_hip_version() >= 100
    assert h.locus_hip_packed_ld(1000) == 1024 and h.locus_hip_packed_ld(128) == 128


def test_sweep_args_struct_matches_header():
    from locus_amd import _lib
    # 3 int64 + 5 ptr + 4 ptr + double + 3 ptr + double + 3 ptr + ptr + 2 int64 + 4 int32 + ptr
    # + 2 int32 (prepacked, pad_)
    assert ctypes.sizeof(_lib.SweepArgs) == 8 * (3 + 5 + 4 + 1 + 3 + 1 + 3 + 1 + 2) + 16 + 8 + 8


def test_tail_args_struct_matches_header():
    from locus_amd import _lib
    # 4 int64 + 4 ptr + 2 double + 5 ptr + 3 ptr + 2 int32 + double + 8 ptr
    assert ctypes.sizeof(_lib.TailArgs) == 8 * (4 + 4 + 2 + 5 + 3) + 8 + 8 + 8 * 8


@pytest.mark.parametrize("n,d", [(200, 20), (1000, 1000), (1000, 1250), (1000, 10000),
                                 (2000, 500), (1000, 2000), (64, 1), (2048, 8)])
@pytest.mark.parametrize("logit", [0, 1])
def test_geometry_chooser(n, d, logit, monkeypatch):
    from locus_amd import _lib
    cfg = (ctypes.c_int32 * 4)()
    _lib.check(_lib.lib().locus_hip_sweep_config(n, d, logit, cfg))
    if not logit and n <= 2016:
        # identity family: the matrix-core sweep, planned from d at launch time
        assert list(cfg) == [-1, 0, 0, 0]
        monkeypatch.setenv("LOCUS_HIP_SWEEP_VECTOR", "1")
        _lib.check(_lib.lib().locus_hip_sweep_config(n, d, logit, cfg))
    if logit and d <= 1024:
        # logit family, at most one workgroup of 4 responses per CU (256 CUs assumed without a
        # GPU): the look-ahead kernel, responses per workgroup and rows per lane set at launch
        assert list(cfg) == [-3, 0, 0, 0]
        monkeypatch.setenv("LOCUS_HIP_SWEEP_VECTOR", "1")
        _lib.check(_lib.lib().locus_hip_sweep_config(n, d, logit, cfg))
    npl, c, s, g = list(cfg)
    assert npl in (2, 4, 8, 16) and 64 * npl * s >= n and 64 * s * g <= 1024
    assert 2 * npl * c * (2 if logit else 1) <= 170  # residual fits the 256-VGPR budget
    assert g >= (npl + 3) // 4                        # <= 2 staging pieces per thread


@pytest.mark.parametrize("n,logit", [(2049, 0), (2049, 1), (4096, 1), (4097, 0), (100000, 0)])
def test_geometry_chooser_beyond_2048_rows(n, logit, monkeypatch):
    """2048 < n <= 4096: the wide vector instantiations (NPL = 16, S = 4, G = 2: 512 threads,
    four staging pieces each); above: the streaming kernel, NPL = -2.  (Logit with few responses
    and n <= 3072: the look-ahead kernel first.)"""
    from locus_amd import _lib
    cfg = (ctypes.c_int32 * 4)()
    _lib.check(_lib.lib().locus_hip_sweep_config(n, 10, logit, cfg))
    if logit and n <= 3072:
        assert list(cfg) == [-3, 0, 0, 0]
        monkeypatch.setenv("LOCUS_HIP_SWEEP_VECTOR", "1")
        _lib.check(_lib.lib().locus_hip_sweep_config(n, 10, logit, cfg))
    if n <= 4096:
        assert list(cfg)[0] == 16 and list(cfg)[2:] == [4, 2] and cfg[1] <= (1 if logit else 4)
    else:
        assert list(cfg) == [-2, 0, 0, 0]


def test_rejects_bad_arguments_before_touching_the_gpu():
    import locus_amd
    from locus_amd._lib import LocusHipError
    from helpers import call_variant, inputs_for
    inp = inputs_for("identity", 50, 8, 2, seed=0)
    inp["shuffled_ind"] = np.array([0, 1, 2, 3, 4, 5, 6, 9], dtype=np.float64)  # out of range
    with pytest.raises(LocusHipError) as ei:
        call_variant(locus_amd.core, "identity", inp)
    assert ei.value.code == -3
    inp = inputs_for("identity", 50, 8, 2, seed=0)
    with pytest.raises(ValueError):
        locus_amd.coreLoop(inp["X"], inp["Y"], inp["gam_vb"][:, :1], inp["log_om_vb"],
                           inp["log_1_min_om_vb"], 0.0, inp["log_tau_vb"], inp["beta_vb"],
                           inp["mat_x_m1"], inp["mu_beta_vb"], inp["sig2_beta_vb"], inp["tau_vb"],
                           inp["shuffled_ind"])


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, not fall back to a CPU path."""
    import locus_amd
    from locus_amd import _lib
    from locus_amd._lib import LocusHipError
    from helpers import call_variant, inputs_for
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(LocusHipError):
        call_variant(locus_amd.core, "identity", inputs_for("identity", 50, 8, 2, seed=0))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "locus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liblocus_oracle" not in txt, f


def test_workspace_bytes_cover_the_matrix_core_streams():
    """locus_hip_sweep_workspace_bytes(n, p) must hold the block-packed operand streams of the
    matrix-core sweep: (p / 16 + 2) iterations x padded rows x 32 doubles (both operand forms),
    for the larger of the T = 1 (6 waves) and T = 2 (7 waves) row paddings."""
    from locus_amd import _lib
    f = _lib.lib().locus_hip_sweep_workspace_bytes
    for n, p in [(200, 500), (1000, 50000), (1008, 16), (2016, 1000), (1, 1)]:
        nit = (p + 15) // 16 + 2
        tiles = (n + 15) // 16
        rows = 16 * max(6 * -(-tiles // 6), 7 * -(-tiles // 7) if n <= 1232 else 0)
        assert f(n, p) >= 8 * nit * rows * 32, (n, p)
    assert f(1000, 50000) > f(1000, 25000) > f(1000, 16)
    # n beyond the matrix-core kernel: only the step records remain
    assert f(4000, 50000) < 64 * 50000


def _top_level_args(text, start):
    """Number of top-level comma-separated arguments of the call whose '(' is at text[start]."""
    depth, n_args, i, seen = 0, 1, start, False
    while True:
        ch = text[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return n_args if seen else 0
        elif ch == "," and depth == 1:
            n_args += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1


@pytest.mark.parametrize("n", [100, 200, 384, 500, 769, 1000, 1153, 2016])
@pytest.mark.parametrize("cus", ["1", "2", "37", "256", "304"])
def test_launch_plan_covers_every_response_once(n, cus, monkeypatch):
    """The launch plan of the matrix-core sweep is a host-side dynamic
    programme: whatever d and CU count, its launches must tile [0, d) without gap or overlap, stay
    within their capacity, use a row tiling that covers n, and share a packed-operand region only
    between launches of one geometry.  (No GPU needed: without one the library assumes 256 CUs.)"""
    import ctypes as C
    from locus_amd import _lib
    monkeypatch.delenv("LOCUS_HIP_SWEEP_CFG", raising=False)
    monkeypatch.delenv("LOCUS_HIP_MFMA_PLAN", raising=False)
    monkeypatch.delenv("LOCUS_HIP_MFMA_CS", raising=False)
    monkeypatch.setenv("LOCUS_HIP_MFMA_CUS", cus)
    lib = _lib.lib()
    rows = (C.c_int32 * (8 * 4096))()
    ds = sorted(set(list(range(1, 70)) + [127, 128, 129, 1000, 1250, 1280, 1281, 2000, 2500, 2560,
                                          2561, 4096, 5000, 8192, 8193, 10000, 16385, 40000]))
    for d in ds:
        k = lib.locus_hip_sweep_plan(n, d, rows, 4096)
        assert 1 <= k <= 4096, (n, d, cus, k)
        pos, geom_of_region = 0, {}
        for r in range(k):
            T, CS, KT, ODD, k0, k1, region, cap = rows[8 * r: 8 * r + 8]
            assert k0 == pos and k0 < k1 <= d, (n, d, cus, r, k0, k1, pos)
            assert k1 - k0 <= cap, (n, d, cus, r, k1 - k0, cap)
            if CS == 0:
                assert T == 2 and ODD == 0 and 7 * 16 * KT >= n
            else:
                assert (T, CS) in ((1, 1), (1, 2), (1, 3), (2, 2), (2, 3)) and ODD in (0, 1)
                assert CS * 3 * (2 * KT - ODD) * 16 >= n and KT - ODD >= 1
            assert geom_of_region.setdefault(region, (T, CS, KT, ODD)) == (T, CS, KT, ODD)
            pos = k1
        assert pos == d, (n, d, cus, pos)
        assert sorted(geom_of_region) == list(range(len(geom_of_region)))


def test_launch_plan_prices_the_chain(monkeypatch):
    """Few rows per response: the 32-response kernel would be bound by its chain wave (4.85 us per
    block at n = 200), one tile per CU is not -- the plan must know (round 3: n = 200, d = 1000 ran
    in 6.3 ms on the 32-response kernel, 2.9 ms with one tile per CU).  At n = 1000 the 32-response
    kernel is matrix-bound and stays the choice for d >= 5121."""
    import ctypes as C
    from locus_amd import _lib
    for k in ("LOCUS_HIP_SWEEP_CFG", "LOCUS_HIP_MFMA_PLAN", "LOCUS_HIP_MFMA_CS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCUS_HIP_MFMA_CUS", "256")
    lib = _lib.lib()
    rows = (C.c_int32 * (8 * 64))()

    def plan(n, d):
        k = lib.locus_hip_sweep_plan(n, d, rows, 64)
        return [tuple(rows[8 * r: 8 * r + 8]) for r in range(k)]

    assert all(r[1] >= 1 for r in plan(200, 1000))          # clusters (CS >= 1), not CS = 0
    assert all(r[1] >= 1 for r in plan(400, 1000))
    big = plan(1000, 10000)
    assert big[0][:2] == (2, 0) and big[0][5] == 8192       # the 32-response kernel on 256 CUs
    assert big[-1][1] >= 1 and big[-1][5] == 10000          # the remainder in clusters
    assert [r[:2] for r in plan(1000, 5000)] == [(2, 3), (2, 3)]   # two launches of 80 three-CU clusters
    assert [r[:2] for r in plan(1000, 1250)] == [(1, 3)]


def test_rcpp_glue_calls_match_the_header():
    """integration/rcpp/*.cpp cannot be compiled here (no R / Rcpp in the image): at least every
    locus_hip_* call in it must name a declared entry point and pass as many arguments as the
    header declares, and all eight .Call routines of the reference (src/RcppExports.cpp:191-201)
    must be there."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "locus_hip.h")).read(),
                 flags=re.S)
    hdr = re.sub(r"\(\s*void\s*\)", "()", hdr)
    arity = {m.group(1): _top_level_args(hdr, m.end() - 1)
             for m in re.finditer(r"\b(locus_hip_[a-z0-9_]+)\s*\(", hdr)}
    glue_dir = os.path.join(ROOT, "integration", "rcpp")
    exported, called = set(), set()
    for f in sorted(os.listdir(glue_dir)):
        if not f.endswith((".cpp", ".h")):
            continue
        src = re.sub(r"//[^\n]*", "", open(os.path.join(glue_dir, f)).read())
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        exported |= set(re.findall(r"^(?:void|SEXP|Rcpp::List)\s+(\w+)\(", src, flags=re.M))
        for m in re.finditer(r"\b(locus_hip_[a-z0-9_]+)\s*\(", src):
            name = m.group(1)
            if name in ("locus_hip_check",):
                continue
            assert name in arity, f"{f}: {name} is not declared in include/locus_hip.h"
            assert _top_level_args(src, m.end() - 1) == arity[name], f"{f}: arity of {name}"
            called.add(name)
    for r_name in ("coreBatch", "coreZBatch", "coreProbitBatch", "coreLoop", "coreZLoop",
                   "coreLogitLoop", "coreProbitLoop", "coreStructLoop"):
        assert r_name in exported, r_name
    stateless = {n for n in arity if n.startswith("locus_hip_core_")}
    assert stateless <= called and len(stateless) == 8
    assert {n for n in arity if n.startswith("locus_hip_ctx_")} <= called
