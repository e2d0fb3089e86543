"""fp32 MFMA GEMM (avsi_gemm_f32) against numpy float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import avsi_amd
    from avsi_amd import ops
    return ops


def _pad_cols(a, ld):
    out = np.zeros((a.shape[0], ld), dtype=np.float32)
    out[:, :a.shape[1]] = a
    return out


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 257, 264), (1000, 2048, 512), (64, 100, 8), (257, 40, 1024)])
def test_gemm_matches_numpy(ops, ta, tb, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + 2 * ta + tb)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    a_store = _pad_cols(A.T.copy(), -(-M // 4) * 4) if ta else A
    b_store = B.T.copy() if tb else _pad_cols(B, -(-N // 4) * 4)
    a_t, b_t = torch.from_numpy(a_store).cuda(), torch.from_numpy(b_store).cuda()
    got = ops.gemm(a_t, b_t, trans_a=ta, trans_b=tb, m=M, n=N, k=K, bias=torch.from_numpy(bias).cuda())
    assert got.shape == (M, N)
    err = np.abs(got.cpu().numpy() - ref).max()
    assert err < 2e-6 * K * 4, err          # fp32 fma chain: ~1e-7 * sum|a b|


def test_alpha_beta_accumulate(ops):
    rng = np.random.default_rng(0)
    A = rng.normal(size=(130, 64)).astype(np.float32)
    B = rng.normal(size=(64, 132)).astype(np.float32)
    C0 = rng.normal(size=(130, 132)).astype(np.float32)
    c = torch.from_numpy(C0.copy()).cuda()
    ops.gemm(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), out=c, alpha=0.5, beta=2.0)
    ref = 0.5 * (A.astype(np.float64) @ B) + 2.0 * C0
    assert np.abs(c.cpu().numpy() - ref).max() < 1e-4


def test_row_scale_and_time_major_row_map(ops):
    """prediction epilogue: rows are (t, b) time-major with batch pitch Bp -> [B, T, N]."""
    T, B, Bp, K, N = 7, 5, 32, 64, 257
    rng = np.random.default_rng(1)
    X = rng.normal(size=(T, Bp, K)).astype(np.float32)
    W = _pad_cols(rng.normal(size=(K, N)).astype(np.float32), 260)
    bias = rng.normal(size=N).astype(np.float32)
    scale = (rng.uniform(size=(T, Bp)) > 0.3).astype(np.float32)
    out = torch.full((B, T, N), 7.0, device='cuda')
    ops.gemm(torch.from_numpy(X).cuda().view(T * Bp, K), torch.from_numpy(W).cuda(), out=out.view(B * T, N), n=N,
             bias=torch.from_numpy(bias).cuda(), row_scale=torch.from_numpy(scale).cuda().view(-1),
             row_map=(Bp, T, B))
    ref = (X.astype(np.float64) @ W[:, :N] + bias) * scale[:, :, None]
    np.testing.assert_allclose(out.cpu().numpy(), ref[:, :B].transpose(1, 0, 2), atol=1e-4)


def test_rejects_unaligned(ops):
    import avsi_amd
    a = torch.zeros(8, 6, device='cuda')
    b = torch.zeros(6, 8, device='cuda')
    with pytest.raises(avsi_amd._lib.AvsiError):
        ops.gemm(a, b)                       # K = 6 is not a multiple of 4
