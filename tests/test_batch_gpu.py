"""GPU parity of coreBatch / coreZBatch / coreProbitBatch (fp64 MFMA GEMMs + element-wise
kernel) and of the GEMM / helper kernels of the device API, against the oracle / numpy."""
import numpy as np
import pytest

from helpers import assert_parity, call_variant, inputs_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["batch", "zbatch", "probitbatch"])
@pytest.mark.parametrize("shape", [(200, 500, 20), (100, 75, 20), (65, 130, 3), (300, 64, 64)])
def test_batch_matches_oracle(oracle, hip, variant, shape):
    n, p, d = shape
    inp = inputs_for(variant, n, p, d, seed=n + d)
    ref = call_variant(oracle, variant, inp)
    got = call_variant(hip, variant, inp)
    assert_parity(got, ref, atol=1e-12, what=f"{variant} {shape}")


# (130, 70, 9001) and (64, 64, 2049): few output tiles and a deep contraction -> the split-K path
# (8 and 2 slices of k, ragged last slice, partial products summed by the reduce kernel)
@pytest.mark.parametrize("m,n,k", [(64, 64, 16), (70, 33, 19), (1, 1, 1), (200, 130, 257),
                                   (130, 70, 9001), (64, 64, 2049)])
@pytest.mark.parametrize("layout", ["nn", "tn", "nt", "tt"])
def test_gemm_f64(hip, m, n, k, layout):
    import torch
    from locus_amd import device as dv
    rng = np.random.default_rng(m + n + k)
    A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n)); C0 = rng.normal(size=(m, n))
    dev = torch.device("cuda:0")
    # row-major storage of A (or its transpose) picked by `layout`
    if layout[0] == "n":
        At, sa_m, sa_k = torch.from_numpy(A.copy()).to(dev), k, 1
    else:
        At, sa_m, sa_k = torch.from_numpy(A.T.copy()).to(dev), 1, m
    if layout[1] == "n":
        Bt, sb_k, sb_n = torch.from_numpy(B.copy()).to(dev), n, 1
    else:
        Bt, sb_k, sb_n = torch.from_numpy(B.T.copy()).to(dev), 1, k
    Ct = torch.from_numpy(C0.copy()).to(dev)
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct, n, 1, alpha=0.5, beta=-2.0)
    np.testing.assert_allclose(Ct.cpu().numpy(), 0.5 * A @ B - 2.0 * C0, rtol=1e-12, atol=1e-12 * k ** 0.5)
    Ct2 = torch.empty((m, n), dtype=torch.float64, device=dev)
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct2, n, 1, sqA=True)
    np.testing.assert_allclose(Ct2.cpu().numpy(), (A * A) @ B, rtol=1e-12, atol=1e-12 * k ** 0.5)
    # the run-time squaring flags (B alone, both operands) are a separate instantiation
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct2, n, 1, sqB=True)
    np.testing.assert_allclose(Ct2.cpu().numpy(), A @ (B * B), rtol=1e-12, atol=1e-12 * k ** 0.5)
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct2, n, 1, sqA=True, sqB=True)
    np.testing.assert_allclose(Ct2.cpu().numpy(), (A * A) @ (B * B), rtol=1e-12, atol=1e-12 * k ** 0.5)


@pytest.mark.parametrize("layout", ["nn", "tn", "nt", "tt"])
@pytest.mark.parametrize("forced", [False, True])
def test_gemm_f64_wide_tiles(hip, layout, forced, monkeypatch):
    """The 64 x 128 tile: chosen by itself for many output tiles (4000 x 1279: 63 x 10 tiles on 256 CUs;
    ragged in m, n and k), and forced on a small ragged shape (LOCUS_HIP_GEMM_TW=w)."""
    import torch
    from locus_amd import device as dv
    if forced:
        monkeypatch.setenv("LOCUS_HIP_GEMM_TW", "w")
        m, n, k = 70, 133, 37
    else:
        monkeypatch.delenv("LOCUS_HIP_GEMM_TW", raising=False)
        m, n, k = 4000, 1279, 75
    rng = np.random.default_rng(m + n + k)
    A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n)); C0 = rng.normal(size=(m, n))
    dev = torch.device("cuda:0")
    if layout[0] == "n":
        At, sa_m, sa_k = torch.from_numpy(A.copy()).to(dev), k, 1
    else:
        At, sa_m, sa_k = torch.from_numpy(A.T.copy()).to(dev), 1, m
    if layout[1] == "n":
        Bt, sb_k, sb_n = torch.from_numpy(B.copy()).to(dev), n, 1
    else:
        Bt, sb_k, sb_n = torch.from_numpy(B.T.copy()).to(dev), 1, k
    Ct = torch.from_numpy(C0.copy()).to(dev)
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct, n, 1, alpha=0.5, beta=-2.0)
    np.testing.assert_allclose(Ct.cpu().numpy(), 0.5 * A @ B - 2.0 * C0, rtol=1e-12, atol=1e-12 * k ** 0.5)
    dv.gemm(m, n, k, At, sa_m, sa_k, Bt, sb_k, sb_n, Ct, n, 1, sqA=True, sqB=True)
    np.testing.assert_allclose(Ct.cpu().numpy(), (A * A) @ (B * B), rtol=1e-12, atol=1e-12 * k ** 0.5)


@pytest.mark.parametrize("m,n,k", [(70, 33, 19), (200, 130, 257)])
def test_gemm_f64_negative_stride(hip, m, n, k):
    """A negative stride takes the instantiation with 64-bit lane offsets (the default one
    addresses a tile as scalar base + unsigned 32-bit offset): rows of A walked backwards."""
    import torch
    from locus_amd import device as dv
    rng = np.random.default_rng(m * n + k)
    A = rng.normal(size=(m, k)); B = rng.normal(size=(k, n))
    dev = torch.device("cuda:0")
    At = torch.from_numpy(A.copy()).to(dev)
    Bt = torch.from_numpy(B.copy()).to(dev)
    Ct = torch.empty((m, n), dtype=torch.float64, device=dev)
    dv.gemm(m, n, k, At[m - 1], -k, 1, Bt, n, 1, Ct, n, 1)
    np.testing.assert_allclose(Ct.cpu().numpy(), A[::-1] @ B, rtol=1e-12, atol=1e-12 * k ** 0.5)
    dv.gemm(m, n, k, At[m - 1], -k, 1, Bt, n, 1, Ct, n, 1, sqA=True)
    np.testing.assert_allclose(Ct.cpu().numpy(), (A * A)[::-1] @ B, rtol=1e-12, atol=1e-12 * k ** 0.5)


def test_pack_transpose_rows_helpers(hip):
    import torch
    from locus_amd import device as dv
    rng = np.random.default_rng(0)
    n, p, d = 130, 37, 11
    X = np.asfortranarray(rng.normal(size=(n, p)))
    dev = torch.device("cuda:0")
    px = dv.PackedX(dv.to_device_colmajor(X, dev))
    Xp = px.Xp.cpu().numpy()
    assert px.ld == 256
    np.testing.assert_array_equal(Xp[:, :n], X.T)
    assert (Xp[:, n:] == 0).all()
    np.testing.assert_allclose(px.xtx.cpu().numpy(), (X * X).sum(axis=0), rtol=1e-14)
    G = rng.random((p, d))
    G[0, :4] = [0.0, 1.0, 1e-300, 1.0 - 2.0 ** -53]   # ends of the range of gam_vb
    G[1] = 10.0 ** -rng.uniform(0, 20, size=d)         # tiny inclusion probabilities
    t = dv.transpose(torch.from_numpy(G.T.copy()).to(dev), d, p)     # [d][p] -> [p][d]
    np.testing.assert_array_equal(t.cpu().numpy(), G)
    rs, ent = dv.gamma_rows(t, p, d, eps=2.2e-16)
    np.testing.assert_allclose(rs.cpu().numpy(), G.sum(axis=1), rtol=1e-14)
    ref_ent = (G * np.log(G + 2.2e-16) + (1 - G) * np.log(1 - G + 2.2e-16)).sum(axis=1)
    np.testing.assert_allclose(ent.cpu().numpy(), ref_ent, rtol=1e-12)
