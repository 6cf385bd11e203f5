"""One-time set-up of the structured model (R/update_vb.R:327-374): nearPD regularisation of the
per-block correlation matrices, their inverses and log-determinants.  Every path of the product
(numpy, batched torch on CPU, batched torch on the GPU under -m gpu) is compared with the
ORACLE's restatement (oracle/ref_driver.py: near_pd_corr, update_sig2_theta_vb_, e_theta_), never
with another path of the product."""
import numpy as np
import pytest


def _stacks():
    rng = np.random.default_rng(4)
    out = []
    for n in (60, 12, 25):                       # well conditioned, rank-deficient, borderline
        X = rng.binomial(2, 0.3, size=(n, 5 * 20)).astype(np.float64) + 0.01 * rng.normal(size=(n, 100))
        Xc = X - X.mean(axis=0)
        C = np.stack([np.corrcoef(Xc[:, 20 * k:20 * (k + 1)], rowvar=False) for k in range(5)])
        out.append(C)
    ind = np.stack([np.eye(20) + 0.9 * (np.ones((20, 20)) - np.eye(20)) * ((-1.0) ** np.add.outer(np.arange(20), np.arange(20)))] * 2)
    ind[1, 0, 1] = ind[1, 1, 0] = 0.99           # not a valid correlation matrix as a whole
    out.append(ind)
    return out


def test_near_pd_numpy_against_oracle_restatement(oracle):
    from oracle import ref_driver
    from locus_amd.struct import near_pd_corr
    for C in _stacks():
        got = near_pd_corr(C)
        for k in range(C.shape[0]):
            np.testing.assert_allclose(got[k], ref_driver.near_pd_corr(C[k]), rtol=1e-9, atol=1e-11)
            assert np.linalg.eigvalsh(got[k]).min() > 0 and np.allclose(np.diag(got[k]), 1.0)


def _check_torch(device):
    import torch
    from oracle import ref_driver
    from locus_amd.struct import near_pd_corr_t
    for C in _stacks():
        got = near_pd_corr_t(torch.from_numpy(C).to(device)).cpu().numpy()
        for k in range(C.shape[0]):
            np.testing.assert_allclose(got[k], ref_driver.near_pd_corr(C[k]), rtol=1e-7, atol=1e-9)
            assert np.linalg.eigvalsh(got[k]).min() > 0 and np.allclose(np.diag(got[k]), 1.0)


def test_near_pd_torch_cpu(oracle):
    import torch
    _check_torch(torch.device("cpu"))


@pytest.mark.gpu
def test_near_pd_torch_gpu(oracle, hip):
    import torch
    _check_torch(torch.device("cuda", 0))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [150, 14])
def test_struct_blocks_on_gpu_match_oracle(oracle, hip, n):
    """StructBlocks(device = cuda) against the oracle's update_sig2_theta_vb_ / e_theta_
    (R/update_vb.R:343-371, R/elbo.R:151-182): S0_inv, sig2_theta_vb, log-determinants of every
    block, including rank-deficient blocks (n smaller than the block) and a ragged last block."""
    import torch
    _check_blocks(n, torch.device("cuda", 0))


@pytest.mark.parametrize("n", [150, 14])
def test_struct_blocks_numpy_match_oracle(oracle, n):
    _check_blocks(n, None)


def _check_blocks(n, device):
    from oracle import ref_driver
    from locus_amd.struct import StructBlocks, set_struct
    rng = np.random.default_rng(n)
    p, d, s02 = 230, 30, 0.1
    X = rng.binomial(2, 0.3, size=(n, p)).astype(np.float64) + 0.05 * rng.normal(size=(n, p))
    X = np.asfortranarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    ls = set_struct(n, p, np.arange(1, p + 1, 40))
    fac = np.asarray(ls["vec_fac_st"])
    r_S0, r_V, r_ld, bl_ids = ref_driver.update_sig2_theta_vb_(d, p, fac, s02, X)
    kw = {} if device is None else dict(device=device)
    got = StructBlocks(d, ls["vec_fac_st"], s02, X, **kw)
    assert len(got.S0_inv) == len(r_S0)
    for a, b in zip(got.S0_inv, r_S0):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6 * np.abs(b).max())
    for a, b in zip(got.sig2_theta, r_V):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9 * np.abs(b).max())
    np.testing.assert_allclose(got.sum_log_det, r_ld, rtol=1e-8)
    m0, th = rng.normal(size=p), rng.normal(size=p)
    ref_e = ref_driver.e_theta_(m0, th, r_S0, r_V, fac, bl_ids, r_ld)
    assert abs(got.e_theta_(m0, th) - ref_e) <= 1e-7 * abs(ref_e)
