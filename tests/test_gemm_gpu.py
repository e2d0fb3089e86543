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


@pytest.mark.parametrize("T,B,Bp,padded", [(250, 270, 288, False), (9, 7500, 7520, False), (250, 270, 288, True)])
def test_projection_with_the_last_bin_folded_into_the_wide_tile(ops, T, B, Bp, padded):
    """The 257-bin projection the way the model issues it from 65536 rows on (models.py:117-123): ONE launch on the
    128 x 256 tile with the general epilogue (sequence mask, time-major -> batch-major rows, row pitch 257) whose workgroups
    take the 257th bin as a dot product on the VALU over the rows they have staged (round 6) -- against numpy, and against
    the round-3 form: N = 256 on the wide tile (bit for bit: the same MFMA chains) + the last bin on a 32-wide tile on column
    views of W, bias and the output (another order of the 512-long sum: to rounding).  `padded`: with the zero padding of
    the two 250-unit halves promised (k_zero), i.e. through the short special tiles, as the model calls it."""
    K, N = 512, 257
    rng = np.random.default_rng(B)
    X = rng.normal(size=(T, Bp, K)).astype(np.float32)
    W = _pad_cols((rng.normal(size=(K, N)) * 0.05).astype(np.float32), 260)
    kz = None
    if padded:          # (the promise is about the ROWS OF W; X keeps values there, as the training pass's column of ones does)
        W[250:256] = 0
        W[506:512] = 0
        kz = ((250, 256), (506, 512))
    bias = rng.normal(size=260).astype(np.float32)
    scale = (rng.uniform(size=(T, Bp)) > 0.3).astype(np.float32)
    x, w, b = torch.from_numpy(X).cuda().view(T * Bp, K), torch.from_numpy(W).cuda(), torch.from_numpy(bias).cuda()
    rs = torch.from_numpy(scale).cuda().view(-1)
    one = torch.full((B * T, N), 7.0, device='cuda')
    two = torch.full((B * T, N), 9.0, device='cuda')
    ops.gemm(x, w, out=one, n=N, bias=b, row_scale=rs, row_map=(Bp, T, B), k_zero=kz)
    ops.gemm(x, w, out=two, n=256, bias=b, row_scale=rs, row_map=(Bp, T, B), k_zero=kz)
    assert float(two[:, 256].min()) == 9.0          # the 256-bin product leaves the last column alone
    ops.gemm(x, w[:, 256:N], out=two[:, 256:], n=1, bias=b[256:N], row_scale=rs, row_map=(Bp, T, B), k_zero=kz)
    assert torch.equal(one[:, :256], two[:, :256])
    assert float((one[:, 256] - two[:, 256]).abs().max()) < 2e-5
    assert float(one[:, 256].abs().max()) > 0.5     # (the folded column was written: not the 7.0 fill, not zeros)
    ref = (X.astype(np.float64) @ W[:, :N].astype(np.float64) + bias[:N]) * scale[:, :, None]
    np.testing.assert_allclose(one.view(B, T, N).cpu().numpy(), ref[:, :B].transpose(1, 0, 2), atol=2e-4)
    np.testing.assert_allclose(two.view(B, T, N).cpu().numpy(), ref[:, :B].transpose(1, 0, 2), atol=2e-4)


@pytest.mark.parametrize("K,ranges", [(272, ((257, 272),)), (512, ((250, 256), (506, 512)))])
def test_persistent_wide_tiles_equal_one_workgroup_per_tile(ops, K, ranges):
    """From 2048 wide (128 x 256) tiles on, A . B runs on RESIDENT workgroups that walk the tiles and request the next
    tile's first k-tiles before they store the current one (round 6).  The MFMA chains are the same, so the product over
    262,221 rows (16,392 tiles: 32 per workgroup, a ragged last row block) equals, bit for bit, the same rows computed
    in pieces of 8192 rows (512 tiles each: one workgroup per tile), and numpy to rounding."""
    M, N = 262144 + 77, 2048
    g = torch.Generator(device='cuda')
    g.manual_seed(K)
    a = torch.randn(M, K, device='cuda', generator=g)
    b = torch.randn(K, N, device='cuda', generator=g) * 0.05
    for lo, hi in ranges:
        b[lo:hi] = 0
    bias = torch.randn(N, device='cuda', generator=g)
    full = torch.full((M, N), 7.0, device='cuda')
    ops.gemm(a, b, out=full, bias=bias, k_zero=ranges)
    piece = torch.empty(8192, N, device='cuda')
    for lo in range(0, M, 8192):
        n = min(8192, M - lo)
        ops.gemm(a[lo:lo + n], b, out=piece[:n], bias=bias, k_zero=ranges)
        assert torch.equal(full[lo:lo + n], piece[:n]), lo
    rows = torch.tensor([0, 127, 128, 65535, 131072, M - 78, M - 1], device='cuda')
    ref = a[rows].double() @ b.double() + bias.double()
    assert float((full[rows].double() - ref).abs().max()) < 2e-6 * K * 4


def test_persistent_projection_with_the_folded_last_bin(ops):
    """The 257-bin projection at a size where the wide tile is persistent (2375 row blocks): the folded 257th bin's column of
    W is staged once per WORKGROUP there, and its accumulator restarts with every tile -- against numpy on sampled rows."""
    T, B, Bp, K, N = 250, 1200, 1216, 512, 257
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    x = torch.randn(T * Bp, K, device='cuda', generator=g)
    w = torch.zeros(K, 260, device='cuda')
    w[:, :N] = torch.randn(K, N, device='cuda', generator=g) * 0.05
    w[250:256] = 0
    w[506:512] = 0
    bias = torch.randn(260, device='cuda', generator=g)
    rs = (torch.rand(T * Bp, device='cuda', generator=g) > 0.3).float()
    out = torch.full((B * T, N), 7.0, device='cuda')
    ops.gemm(x, w, out=out, n=N, bias=bias, row_scale=rs, row_map=(Bp, T, B), k_zero=((250, 256), (506, 512)))
    ts = torch.tensor([0, 1, 100, 249], device='cuda')
    bs = torch.tensor([0, 31, 32, 640, 1199], device='cuda')
    rows = (ts[:, None] * Bp + bs[None, :]).reshape(-1)
    ref = (x[rows].double() @ w[:, :N].double() + bias[:N].double()) * rs[rows, None].double()
    got = out.view(B, T, N)[bs[None, :].expand(4, 5).reshape(-1), ts[:, None].expand(4, 5).reshape(-1)]
    assert float((got.double() - ref).abs().max()) < 2e-4
    assert float(out.min()) > -50 and float(out.max()) < 50 and not bool((out == 7.0).all(dim=1).any())


@pytest.mark.parametrize("N", [1, 5, 32])
def test_at_most_32_columns_take_the_narrow_tile(ops, N):
    M, K = 1000, 272
    rng = np.random.default_rng(N)
    A = rng.normal(size=(M, K)).astype(np.float32)
    Bm = _pad_cols(rng.normal(size=(K, N)).astype(np.float32), 36)
    bias = rng.normal(size=N).astype(np.float32)
    got = ops.gemm(torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda(), n=N, bias=torch.from_numpy(bias).cuda())
    ref = A.astype(np.float64) @ Bm[:, :N].astype(np.float64) + bias
    assert got.shape == (M, N)
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6 * K * 4


@pytest.mark.parametrize("M,N,K,ranges", [(8192, 2048, 512, ((250, 256), (506, 512))), (8192, 2048, 272, ((257, 272),)),
                                          (300, 257, 512, ((250, 256), (506, 512))), (8192, 2048, 400, ((393, 400),)),
                                          (4096, 256, 512, ((250, 256), (506, 512))), (640, 2048, 64, ((3, 61),)),
                                          (640, 128, 48, ((0, 16), (40, 48)))])
def test_zero_padding_inside_the_reduction_is_skipped_not_changed(ops, M, N, K, ranges):
    """avsi_gemm_epilogue::k_zero: rows of B the caller knows to be zero (the 250 -> 256 padding of a BLSTM layer's two
    input halves, models.py:95-115 has no padding at all) -- the 16-deep tiles skip the MFMA steps that would multiply
    padding only.  A keeps NON-zero values there (the training pass puts a column of ones into the padding), so a
    skipped step that was needed, or a needed one that was skipped, shows: equal to the plain product bit for bit."""
    rng = np.random.default_rng(K + N)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    for lo, hi in ranges:
        B[lo:hi] = 0.0
    bias = rng.normal(size=N).astype(np.float32)
    a, b, bi = torch.from_numpy(A).cuda(), torch.from_numpy(_pad_cols(B, (N + 3) // 4 * 4)).cuda(), torch.from_numpy(bias).cuda()
    plain = ops.gemm(a, b, n=N, bias=bi)
    skipped = ops.gemm(a, b, n=N, bias=bi, k_zero=ranges)
    assert torch.equal(plain, skipped)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    assert np.abs(skipped.cpu().numpy() - ref).max() < 2e-6 * K * 4
    # the promise is the caller's: with a non-zero row inside a range the skipped product differs (the steps ARE skipped)
    if (M, N, K) == (8192, 2048, 272):
        b2 = b.clone()
        b2[271] = 1.0
        assert not torch.equal(ops.gemm(a, b2, n=N, bias=bi), ops.gemm(a, b2, n=N, bias=bi, k_zero=ranges))


def test_k_zero_ranges_are_checked(ops):
    import avsi_amd
    a, b = torch.zeros(256, 64, device='cuda'), torch.zeros(64, 128, device='cuda')
    for bad in (((10, 5),), ((0, 65),), ((-1, 4),)):
        with pytest.raises(avsi_amd._lib.AvsiError):
            ops.gemm(a, b, k_zero=bad)


def test_rejects_unaligned(ops):
    import avsi_amd
    a = torch.zeros(8, 6, device='cuda')
    b = torch.zeros(6, 8, device='cuda')
    with pytest.raises(avsi_amd._lib.AvsiError):
        ops.gemm(a, b)                       # K = 6 is not a multiple of 4


def test_randomised_shapes_and_options(ops):
    """80 seeded random cases over the dispatch space: DMA / register-staged kernels, 64-wide tail tiles,
    column-group tile order, transposes, bias, alpha / beta, split-K."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(80):
        ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
        M = int(rng.choice([1, 31, 128, 129, 700, 2500]))
        N = int(rng.choice([1, 33, 64, 65, 200, 257, 300, 1100]))
        K = int(rng.choice([8, 16, 48, 264, 272, 512, 1040]))
        A = rng.normal(size=(M, K)).astype(np.float32)
        B = rng.normal(size=(K, N)).astype(np.float32)
        a_store = _pad_cols(A.T.copy(), -(-M // 4) * 4) if ta else A
        b_store = B.T.copy() if tb else _pad_cols(B, -(-N // 4) * 4)
        a_t, b_t = torch.from_numpy(a_store).cuda(), torch.from_numpy(b_store).cuda()
        ref = A.astype(np.float64) @ B.astype(np.float64)
        mode = case % 3
        if mode == 0:       # plain + bias
            bias = rng.normal(size=N).astype(np.float32)
            got = ops.gemm(a_t, b_t, trans_a=ta, trans_b=tb, m=M, n=N, k=K, bias=torch.from_numpy(bias).cuda())
            ref = ref + bias
        elif mode == 1:     # alpha / beta accumulate
            C0 = rng.normal(size=(M, N)).astype(np.float32)
            got = torch.from_numpy(C0.copy()).cuda()
            ops.gemm(a_t, b_t, out=got, trans_a=ta, trans_b=tb, m=M, n=N, k=K, alpha=-0.75, beta=1.0)
            ref = -0.75 * ref + C0
        else:               # split-K (contiguous [M, N] output)
            got = torch.full((M, N), 7.0, device='cuda')
            ops.gemm_splitk(a_t, b_t, got, trans_a=ta, trans_b=tb, m=M, n=N, k=K, splits=int(rng.choice([1, 2, 5])))
        err = np.abs(got.cpu().numpy()[:, :N] - ref).max() / (1e-6 * K * 4 + 1e-5)
        worst = max(worst, err)
        assert err < 2.0, (case, ta, tb, M, N, K, mode, err)
    assert worst > 0


@pytest.mark.parametrize("M,K", [(8192, 272), (8192, 512), (16384, 272), (16384, 512), (8192 + 40, 512)])
def test_wide_tile_layer_gemm_matches_numpy(ops, M, K):
    """The kernel the benchmark times: gemm_dma_kernel<false, false, 16, 3, false, 256> (128 x 256 tiles, short
    epilogue) is selected for A.B with N % 256 == 0, N >= 1024 and m_blocks * (N / 256) >= 512, i.e. from 8192 rows
    at N = 2048 -- the layer input projections x . Wx + b of the stacked BLSTM (reference models.py:95-115).
    M = 8232 adds a partial last row tile (general epilogue of the same kernel)."""
    N = 2048
    rng = np.random.default_rng(M + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    got = ops.gemm(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), bias=torch.from_numpy(bias).cuda())
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    assert got.shape == (M, N)
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6 * K * 4
    # accumulate form (the speaker-embedding variants: beta = 1 onto a broadcast bias) takes the general epilogue
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    acc = torch.from_numpy(C0.copy()).cuda()
    ops.gemm(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), out=acc, beta=1.0)
    assert np.abs(acc.cpu().numpy() - (ref - bias + C0)).max() < 2e-6 * K * 4


@pytest.mark.parametrize("Mk,rows,splits", [(512, 16384, 16), (272, 32768, 32), (400, 20000, 16)])
def test_wide_tile_split_k_weight_gradient_matches_numpy(ops, Mk, rows, splits):
    """dWx = X^T . dZ of training (the gradient of models.py:95-115): A^T . B, N = 2048, reduction over all T * Bp
    rows cut into slabs; m_blocks * 8 * splits >= 512 selects the 128 x 256 tile of the DMA kernel."""
    rng = np.random.default_rng(rows + Mk)
    X = rng.normal(size=(rows, Mk)).astype(np.float32)
    dZ = rng.normal(size=(rows, 2048)).astype(np.float32)
    out = torch.full((Mk, 2048), 7.0, device='cuda')
    ops.gemm_splitk(torch.from_numpy(X).cuda(), torch.from_numpy(dZ).cuda(), out, trans_a=True, m=Mk, n=2048, k=rows,
                    splits=splits)
    ref = X.astype(np.float64).T @ dZ.astype(np.float64)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * rows * 4 * 0.05 + 1e-3   # random-sign sums grow ~sqrt(K)
