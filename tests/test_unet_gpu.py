"""U-Net inpainter (BASELINE configs[4]) on the GPU against the torch-CPU float64 oracle: building-block
kernels, forward, losses, every gradient, Adam steps and waveform output."""
import numpy as np
import pytest
import torch

from oracle import frontend as OF
from oracle import unet as OU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import avsi_amd
    from avsi_amd import models, ops
    return models, ops


def _cfg(N):
    return dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
                starter_learning_rate=1e-3, learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=2,
                l2=0.0)


def _inputs(B, N, seed):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = N // 128
    masks = np.ones((B, T, 128), dtype=np.float32)
    for b in range(B):
        s = rng.integers(4, T - 20)
        masks[b, s:s + 12] = 0
    st = OF.get_stft(wav, window_size=16, step_size=8, n_fft=256)[:, :, :128]
    spec = OF.get_spectrogram(st, log=True)
    mean, std = OF.feature_stats(list(spec))
    return wav, masks, mean.astype(np.float32), std.astype(np.float32), T, spec


def test_im2col_col2im_adjoint_and_reference(mods):
    """im2col vs an explicit numpy gather (incl. the fused 2x up-sampling + concat); col2im is its adjoint."""
    models, ops = mods
    rng = np.random.default_rng(0)
    B, H, W, C0, C1, k = 2, 8, 6, 3, 5, 3
    a = rng.normal(size=(B, H, W, C0)).astype(np.float32)
    c = rng.normal(size=(B, H // 2, W // 2, C1)).astype(np.float32)
    up = c.repeat(2, axis=1).repeat(2, axis=2)
    cat = np.concatenate([a, up], axis=3)
    pad = np.pad(cat, [(0, 0), (1, 1), (1, 1), (0, 0)])
    ref = np.zeros((B, H, W, k * k * (C0 + C1)), dtype=np.float32)
    for kh in range(k):
        for kw in range(k):
            ref[..., (kh * k + kw) * (C0 + C1):(kh * k + kw + 1) * (C0 + C1)] = pad[:, kh:kh + H, kw:kw + W, :]
    kc = -(-k * k * (C0 + C1) // 4) * 4
    a4 = torch.zeros(B * H * W, 4, device='cuda'); a4[:, :C0] = torch.from_numpy(a).cuda().view(-1, C0)
    c8 = torch.zeros(B * (H // 2) * (W // 2), 8, device='cuda'); c8[:, :C1] = torch.from_numpy(c).cuda().view(-1, C1)
    col = torch.empty(B * H * W, kc, device='cuda')
    ops.im2col(a4, C0, c8, C1, B, H, W, k, col, kc)
    got = col.cpu().numpy()
    np.testing.assert_array_equal(got[:, :k * k * (C0 + C1)], ref.reshape(B * H * W, -1))
    assert np.all(got[:, k * k * (C0 + C1):] == 0)
    # adjoint: <im2col(x), y> == <x, col2im(y)>
    y = torch.randn(B * H * W, kc, device='cuda')
    d0 = torch.empty_like(a4); d1 = torch.empty_like(c8)
    ops.col2im(y, kc, d0, C0, d1, C1, B, H, W, k)
    lhs = float((col.double() * y.double()).sum())
    rhs = float((a4.double() * d0.double()).sum() + (c8.double() * d1.double()).sum())
    assert lhs == pytest.approx(rhs, rel=1e-5)
    d0b = d0.clone()
    ops.col2im(y, kc, d0b, C0, None, C1, B, H, W, k, accumulate0=True)        # accumulate, second gradient skipped
    assert torch.allclose(d0b, 2 * d0)


def test_bn_act_and_pool_kernels(mods):
    models, ops = mods
    rng = np.random.default_rng(1)
    R, C, ld = 4096, 6, 8
    x = torch.zeros(R, ld, device='cuda'); x[:, :C] = torch.from_numpy(rng.normal(1.0, 2.0, size=(R, C)).astype(np.float32)).cuda()
    gamma = torch.zeros(ld, device='cuda'); gamma[:C] = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda()
    beta = torch.zeros(ld, device='cuda'); beta[:C] = torch.from_numpy(rng.normal(size=C).astype(np.float32)).cuda()
    mean, rstd = torch.empty(ld, device='cuda'), torch.empty(ld, device='cuda')
    ops.colstats(x, C, mean, rstd)
    xr = x[:, :C].double().cpu()
    np.testing.assert_allclose(mean[:C].cpu().numpy(), xr.mean(0).numpy(), rtol=1e-5)
    np.testing.assert_allclose(rstd[:C].cpu().numpy(), (1 / torch.sqrt(xr.var(0, unbiased=False) + 1e-3)).numpy(), rtol=1e-5)
    for act, fn in ((1, torch.relu), (2, lambda t: torch.nn.functional.leaky_relu(t, 0.2)), (0, lambda t: t)):
        xt = xr.clone().requires_grad_(True)
        g_t, b_t = gamma[:C].double().cpu().requires_grad_(True), beta[:C].double().cpu().requires_grad_(True)
        yt = fn(g_t * (xt - xt.mean(0)) / torch.sqrt(xt.var(0, unbiased=False) + 1e-3) + b_t)
        dy = torch.from_numpy(rng.normal(size=(R, C))).double()
        yt.backward(dy)
        y = torch.empty_like(x)
        ops.bn_act(x, C, y, mean, rstd, gamma, beta, act)
        np.testing.assert_allclose(y[:, :C].cpu().numpy(), yt.detach().numpy(), atol=2e-5)
        dyd = torch.zeros(R, ld, device='cuda'); dyd[:, :C] = dy.float().cuda()
        dx, dg, db = torch.empty_like(x), torch.empty(ld, device='cuda'), torch.empty(ld, device='cuda')
        ops.bn_act_bwd(x, dyd, C, dx, mean, rstd, gamma, beta, act, dg, db)
        np.testing.assert_allclose(dx[:, :C].cpu().numpy(), xt.grad.numpy(), atol=2e-5)
        np.testing.assert_allclose(dg[:C].cpu().numpy(), g_t.grad.numpy(), rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(db[:C].cpu().numpy(), b_t.grad.numpy(), rtol=2e-4, atol=1e-3)
        assert torch.all(dx[:, C:] == 0)
    # max pooling and its gradient
    B, H, W = 2, 8, 4
    xp = torch.zeros(B * H * W, ld, device='cuda'); xp[:, :C] = torch.randn(B * H * W, C, device='cuda')
    yp = torch.empty(B * (H // 2) * (W // 2), ld, device='cuda')
    ops.maxpool2(xp, yp, B, H, W, C)
    xt = xp[:, :C].cpu().view(B, H, W, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    yt = torch.nn.functional.max_pool2d(xt, 2)
    np.testing.assert_array_equal(yp[:, :C].cpu().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2).numpy(), yt.detach().numpy())
    dyp = torch.zeros_like(yp); dyp[:, :C] = torch.randn(yp.shape[0], C, device='cuda')
    yt.backward(dyp[:, :C].cpu().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2))
    dxp = torch.empty_like(xp)
    ops.maxpool2_bwd(xp, dyp, dxp, B, H, W, C)
    np.testing.assert_array_equal(dxp[:, :C].cpu().view(B, H, W, C).permute(0, 3, 1, 2).numpy(), xt.grad.numpy())


@pytest.mark.parametrize("C,ld,bn,act", [(6, 8, True, 1), (16, 16, False, 1), (32, 32, True, 2), (128, 128, True, 1)])
def test_pooled_bn_act_backward_is_the_chain_of_its_parts(mods, C, ld, bn, act):
    """ops.bn_act_pool_bwd (the encoder layers' backward from the POOLED gradient, window recomputed from the convolution
    output) against torch autograd in float64 on the same chain: batch norm with batch statistics -> activation -> 2 x 2 max
    pooling; and against the three kernels it replaces (maxpool2_bwd on the kept activation, bn_act_bwd, colsum)."""
    models, ops = mods
    rng = np.random.default_rng(C)
    B, H, W = 3, 8, 12
    R = B * H * W
    x = torch.zeros(R, ld, device='cuda'); x[:, :C] = torch.from_numpy(rng.normal(0.3, 1.5, size=(R, C)).astype(np.float32)).cuda()
    gamma = torch.zeros(ld, device='cuda'); gamma[:C] = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda()
    beta = torch.zeros(ld, device='cuda'); beta[:C] = torch.from_numpy(rng.normal(0, 0.5, size=C).astype(np.float32)).cuda()
    mean, rstd = torch.empty(ld, device='cuda'), torch.empty(ld, device='cuda')
    ops.colstats(x, C, mean, rstd)
    bn_args = (mean, rstd, gamma, beta) if bn else (None, None, None, None)
    P = R // 4
    dpool = torch.zeros(P, ld, device='cuda'); dpool[:, :C] = torch.from_numpy(rng.normal(size=(P, C)).astype(np.float32)).cuda()
    # the fused kernel
    dx = torch.full((R, ld), 7.0, device='cuda')
    dg, db, dbias = torch.zeros(ld, device='cuda'), torch.zeros(ld, device='cuda'), torch.zeros(ld, device='cuda')
    ops.bn_act_pool_bwd(x, dpool, B, H, W, C, dx, *bn_args, act, dg if bn else None, db if bn else None, dbias=None if bn else dbias)
    # the three kernels it replaces, on the kept activation
    y, pooled = torch.empty_like(x), torch.empty(P, ld, device='cuda')
    ops.bn_act_pool(x, B, H, W, C, pooled, y, *bn_args, act)
    dy = torch.empty_like(x)
    ops.maxpool2_bwd(y, dpool, dy, B, H, W, C)
    dx2, dg2, db2 = torch.empty_like(x), torch.zeros(ld, device='cuda'), torch.zeros(ld, device='cuda')
    ops.bn_act_bwd(x, dy, C, dx2, *bn_args, act, dg2 if bn else None, db2 if bn else None)
    np.testing.assert_allclose(dx.cpu().numpy(), dx2.cpu().numpy(), atol=1e-5)
    assert torch.all(dx[:, C:] == 0)
    if bn:
        np.testing.assert_allclose(dg[:C].cpu().numpy(), dg2[:C].cpu().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(db[:C].cpu().numpy(), db2[:C].cpu().numpy(), rtol=1e-4, atol=1e-4)
    else:
        np.testing.assert_allclose(dbias[:C].cpu().numpy(), dx2[:, :C].double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-4)
    # torch autograd, float64
    xt = x[:, :C].double().cpu().view(B, H, W, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    g_t, b_t = gamma[:C].double().cpu().requires_grad_(True), beta[:C].double().cpu().requires_grad_(True)
    z = xt
    if bn:
        mu = xt.mean((0, 2, 3), keepdim=True)
        var = xt.var((0, 2, 3), unbiased=False, keepdim=True)
        z = g_t.view(1, C, 1, 1) * (xt - mu) / torch.sqrt(var + 1e-3) + b_t.view(1, C, 1, 1)
    a = torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, 0.2)
    torch.nn.functional.max_pool2d(a, 2).backward(dpool[:, :C].double().cpu().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2))
    np.testing.assert_allclose(dx[:, :C].cpu().view(B, H, W, C).permute(0, 3, 1, 2).numpy(), xt.grad.numpy(), atol=3e-5)
    if bn:
        np.testing.assert_allclose(dg[:C].cpu().numpy(), g_t.grad.numpy(), rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(db[:C].cpu().numpy(), b_t.grad.numpy(), rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("N", [16384, 8192])
def test_unet_forward_backward_matches_oracle(mods, N):
    """N = 16384 is the reference's own size (scripts/config/unet.config:6): 128 frames x 128 bins (config 5);
    8192 adds a non-square map."""
    models, ops = mods
    B = 2
    wav, masks, mean, std, T, _ = _inputs(B, N, 2)
    seq = np.array([T, T - 3])
    params = OU.init_params(3)
    rng = np.random.default_rng(4)
    for k in params:                      # non-trivial bn parameters and biases
        if k.endswith('gamma'):
            params[k] = rng.uniform(0.7, 1.3, params[k].shape).astype(np.float32)
        elif k.endswith('beta') or k.endswith('/b'):
            params[k] = rng.normal(0, 0.1, params[k].shape).astype(np.float32)
    m = models.UNetFConvModel(seq, wav, masks, mean, std, 0.0, _cfg(N))
    m.variables.load_flat(m.layout.flatten_params(params))
    x_in = m.net_inputs.cpu().numpy()
    tgt = m.target_spec_norm.cpu().numpy()
    st = OF.get_stft(wav, window_size=16, step_size=8, n_fft=256)[:, :, :128]
    ref_norm = (OF.get_spectrogram(st, log=True) - mean) / std
    assert np.sqrt(np.mean((tgt - ref_norm) ** 2)) < 1e-4
    assert np.sqrt(np.mean((x_in - ref_norm * masks) ** 2)) < 1e-4
    ref = OU.forward_backward(ref_norm * masks, ref_norm, seq, params)
    pred = m.prediction.cpu().numpy()
    assert pred.shape == ref['prediction'].shape == (B, T, 128)
    assert np.sqrt(np.mean((pred - ref['prediction']) ** 2)) < 2e-4
    assert np.all(pred[1, T - 3:] == 0)
    assert float(m.loss_func) == pytest.approx(ref['loss_func'], rel=5e-4)
    got = m.gradients.cpu().numpy().astype(np.float64)
    for name, shape, off in m.layout.ref_entries:
        g = got[off:off + int(np.prod(shape))].reshape(shape)
        r = ref['grads'][name]
        scale = np.abs(r).max() + 1e-12
        if name.endswith('/b') and (name[:-2] + '/bn/gamma') in params:
            assert np.abs(g).max() < 1e-6          # a bias in front of batch norm has zero gradient
            continue
        # 128 x 128 maps: the batch-norm gradients are sums of 32768 signed L1 terms per channel in float32
        assert np.abs(g - r).max() <= (1e-2 if N == 16384 else 5e-3) * scale, (name, np.abs(g - r).max(), scale)
    # three Adam steps reduce the loss and track the oracle's trajectory
    losses = []
    for _ in range(3):
        m.feed(sequence_lengths=seq, target_sources=wav, masks=masks)
        losses.append(float(m.loss))
        m.train_op
    assert losses[2] < losses[0]
    # waveforms (models.py:664-680; 16 / 8 ms inverse STFT, fft length 256, the 129th bin zero) from the GPU's own
    # prediction, masked and oracle phase, against the oracle's restatement of the same ops
    from oracle import blstm as OB
    m.feed(sequence_lengths=seq, target_sources=wav, masks=masks)
    pred = m.prediction.cpu().numpy().astype(np.float64)
    for oracle_phase in (True, False):
        got = (m.enhanced_sources_oracle_phase if oracle_phase else m.enhanced_sources).cpu().numpy()
        ref_w = OB.enhanced_sources(pred, mean, std, st, None if oracle_phase else masks, num_samples=N, window_size=16,
                                    step_size=8)
        assert got.shape == ref_w.shape == (B, N)
        assert np.sqrt(np.mean((got - ref_w) ** 2)) < 1e-4 * np.abs(ref_w).max()
        assert np.abs(got - ref_w).max() < 1e-3 * np.abs(ref_w).max()


@pytest.mark.parametrize("c0,c1,k,cout,B,H,W", [(16, 0, 5, 32, 2, 12, 10), (32, 64, 3, 32, 3, 8, 16), (128, 128, 3, 128, 1, 4, 6),
                                                (16, 32, 3, 16, 2, 128, 8), (32, 0, 5, 64, 2, 16, 12), (64, 128, 3, 64, 1, 16, 16),
                                                (16, 0, 3, 40, 1, 9, 7)])
def test_implicit_gemm_conv_matches_im2col_gemm(c0, c1, k, cout, B, H, W):
    """avsi_conv2d_f32 (operand rows gathered by the GEMM's DMA loads) against im2col + GEMM."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(c0 + c1 + k)
    R = B * H * W
    src0 = torch.randn(R, c0 + 4, generator=g, device='cuda')                        # pitch > channels on purpose
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    kc = k * k * (c0 + c1)
    filt = torch.randn(kc, cout, generator=g, device='cuda') * 0.1
    bias = torch.randn(cout, generator=g, device='cuda')
    col = torch.empty(R, kc, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
    want = torch.empty(R, cout, device='cuda')
    ops.gemm(col, filt, out=want, n=cout, bias=bias)
    got = torch.full((R, cout), 7.0, device='cuda')
    assert ops.conv2d_supported(c0, c1)
    ops.conv2d(src0, c0, src1, c1, B, H, W, k, filt, bias, got, cout)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-4)
    assert not ops.conv2d_supported(1, 16) and not ops.conv2d_supported(0, 0)


@pytest.mark.parametrize("c0,c1,k,cout,B,H,W", [(128, 0, 3, 128, 32, 4, 4), (128, 128, 3, 128, 32, 8, 8), (64, 0, 3, 128, 32, 16, 16)])
def test_split_reduction_conv_equals_the_plain_launch(c0, c1, k, cout, B, H, W, monkeypatch):
    """The 128-channel layers of the U-Net at the reference's batch of 32 are 4 .. 64 output tiles: avsi_conv2d_splitk_f32 cuts
    their reduction over (tap, channel) into chunks that fill the chip and sums the slabs in order -- the same result as
    the plain launch up to the order of the sum, and the policy (avsi_conv2d_splitk_suggest) does split them."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib, ops
    L = _lib.lib()
    splits = L.avsi_conv2d_splitk_suggest(B, H, W, k, c0, c1, cout)
    assert splits >= 2 and (k * k * (c0 + c1) // 16) // splits >= 8
    assert L.avsi_conv2d_splitk_suggest(512, 64, 64, 3, 32, 0, 64) == 1          # a launch that fills the chip is left alone
    g = torch.Generator(device='cuda')
    g.manual_seed(c0 + c1 + H)
    R = B * H * W
    src0 = torch.randn(R, c0, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    filt = torch.randn(k * k * (c0 + c1), cout, generator=g, device='cuda') * 0.05
    bias = torch.randn(cout, generator=g, device='cuda')
    outs = []
    for on in (False, True):
        monkeypatch.setattr(ops, '_CONV_SPLITK', on)
        out = torch.full((R, cout), 7.0, device='cuda')
        ops.conv2d(src0, c0, src1, c1, B, H, W, k, filt, bias, out, cout)
        outs.append(out.cpu().numpy())
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-5, atol=2e-5 * np.abs(outs[0]).max())
    assert not np.array_equal(outs[1], outs[0]) or splits == 1                   # (another summation order: it really split)


@pytest.mark.parametrize("k,c0,c1,cout", [(7, 1, 0, 16), (3, 1, 16, 1), (1, 1, 0, 1)])
def test_thin_direct_conv_matches_im2col_gemm(k, c0, c1, cout):
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(k * 10 + cout)
    B, H, W = 3, 20, 14
    R = B * H * W
    ld = 4
    src0 = torch.randn(R, ld, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    kc = -(-(k * k * (c0 + c1)) // 4) * 4
    ldf = -(-cout // 4) * 4
    filt = torch.zeros(kc, ldf, device='cuda')
    filt[:k * k * (c0 + c1), :cout] = torch.randn(k * k * (c0 + c1), cout, generator=g, device='cuda') * 0.2
    bias = torch.randn(ldf, generator=g, device='cuda')
    col = torch.empty(R, kc, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
    want = torch.zeros(R, ldf, device='cuda')
    ops.gemm(col, filt, out=want, n=cout, bias=bias)
    got = torch.zeros(R, ldf, device='cuda')
    assert ops.conv2d_thin_supported(k, c0, c1, cout) and not ops.conv2d_thin_supported(3, 1, 0, 16)
    ops.conv2d_thin(src0, c0, src1, c1, B, H, W, k, filt, bias, got, cout)
    np.testing.assert_allclose(got[:, :cout].cpu().numpy(), want[:, :cout].cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_tiled_output_layer_matches_im2col_gemm():
    """The 3 x 3 (1 ++ 16 up-sampled) -> 1 layer at sizes whose tiles fit (H % 8 == 0, W % 32 == 0) takes
    conv3_c17_out1_kernel (LDS patch of fine and coarse pixels): several tiles per image in both directions,
    so every image border and every tile seam is exercised."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(77)
    B, H, W, k, c0, c1, cout = 3, 24, 64, 3, 1, 16, 1
    R = B * H * W
    src0 = torch.randn(R, 4, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda')
    kc = -(-(k * k * (c0 + c1)) // 4) * 4
    filt = torch.zeros(kc, 4, device='cuda')
    filt[:k * k * (c0 + c1), :cout] = torch.randn(k * k * (c0 + c1), cout, generator=g, device='cuda') * 0.2
    bias = torch.randn(4, generator=g, device='cuda')
    col = torch.empty(R, kc, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
    want = torch.zeros(R, 4, device='cuda')
    ops.gemm(col, filt, out=want, n=cout, bias=bias)
    got = torch.zeros(R, 4, device='cuda')
    ops.conv2d_thin(src0, c0, src1, c1, B, H, W, k, filt, bias, got, cout)
    np.testing.assert_allclose(got[:, :cout].cpu().numpy(), want[:, :cout].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("c0,c1,k,cout,B,H,W", [(16, 32, 3, 16, 3, 8, 32), (16, 0, 5, 32, 2, 12, 64), (32, 64, 3, 32, 5, 4, 32),
                                                (16, 32, 3, 16, 70, 64, 64), (32, 64, 3, 32, 9, 32, 32)])
def test_thin_mfma_filter_gradient_matches_im2col_gemm(c0, c1, k, cout, B, H, W):
    """avsi_conv2d_thin_mfma_wgrad_f32 (patch and dY through LDS, filter gradient in 16-wide MFMA accumulators over all the
    tiles of a persistent workgroup) against the explicit im2col^T . dY product; the larger cases give every workgroup
    several tiles (B = 70 at 64 x 64: 2240 tiles on 512 workgroups) and put image borders inside a workgroup's walk."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(c0 + c1 + k + B)
    R = B * H * W
    src0 = torch.randn(R, c0 + 4, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    kc = k * k * (c0 + c1)
    dy = torch.randn(R, cout, generator=g, device='cuda')
    col = torch.empty(R, kc, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
    want = (col.double().t() @ dy.double()).cpu().numpy()
    got = torch.full((kc, cout), 3.0, device='cuda')
    assert ops.conv2d_thin_mfma_wgrad_supported(k, c0, c1, cout, H, W, got)
    assert not ops.conv2d_thin_mfma_wgrad_supported(k, c0, c1, cout, H, W + 16) and not ops.conv2d_thin_mfma_wgrad_supported(3, 16, 16, 16, H, W)
    ops.conv2d_thin_mfma_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, got)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-4 * np.abs(want).max())
    # the same bits on a second run (fixed order of the partial sums)
    again = torch.empty_like(got)
    ops.conv2d_thin_mfma_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, again)
    assert torch.equal(got, again)


@pytest.mark.parametrize("k,c0,c1,cout,B,H,W", [(7, 1, 0, 16, 2, 16, 64), (7, 1, 0, 16, 40, 64, 64), (7, 1, 0, 16, 3, 20, 14),
                                                (3, 1, 16, 1, 3, 20, 14), (3, 1, 16, 1, 2, 16, 64), (3, 1, 16, 1, 40, 64, 64),
                                                (1, 1, 0, 1, 2, 16, 64)])
def test_thin_filter_gradients_match_im2col_gemm(k, c0, c1, cout, B, H, W):
    """avsi_conv2d_thin_wgrad_f32: the one-channel layers' filter gradients -- the first layer on the 16-wide MFMA with the taps
    as the M side, the last 3 x 3 layer with the taps as the N side and dY shifted instead of the input (H % 8 == 0, W % 32 == 0;
    40 x 64 x 64 gives the 80 workgroups eight tiles each), the direct kernels otherwise -- against im2col^T . dY."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(k + B)
    R = B * H * W
    src0 = torch.randn(R, 4, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    kc = k * k * (c0 + c1)
    ld = -(-cout // 4) * 4
    dy = torch.zeros(R, ld, device='cuda')
    dy[:, :cout] = torch.randn(R, cout, generator=g, device='cuda')
    kcp = -(-kc // 4) * 4
    col = torch.empty(R, kcp, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kcp)
    want = (col[:, :kc].double().t() @ dy[:, :cout].double()).cpu().numpy()
    got = torch.full((kcp, ld), 3.0, device='cuda')
    ops.conv2d_thin_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, got)
    np.testing.assert_allclose(got[:kc, :cout].cpu().numpy(), want, rtol=2e-4, atol=2e-4 * np.abs(want).max())


@pytest.mark.parametrize("c0,c1,k,cout,B,H,W", [(32, 0, 5, 16, 2, 8, 32), (16, 0, 3, 48, 3, 12, 64), (32, 0, 5, 16, 40, 64, 64)])
def test_thin_mfma_input_gradient_convolutions_match_implicit_gemm(c0, c1, k, cout, B, H, W):
    """The plain 16-wide-MFMA convolution on the shapes of the two input-gradient convolutions it takes since round 5
    (avsi_conv2d_thin_mfma_plain_supported), against the implicit GEMM; statistics are refused for them."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(c0 + k + B)
    R = B * H * W
    src0 = torch.randn(R, c0, generator=g, device='cuda')
    filt = torch.randn(k * k * c0, cout, generator=g, device='cuda') * 0.2
    assert ops.conv2d_thin_mfma_plain_supported(k, c0, c1, cout, H, W) and not ops.conv2d_thin_mfma_supported(k, c0, c1, cout, H, W)
    want = torch.empty(R, cout, device='cuda')
    ops.conv2d(src0, c0, None, 0, B, H, W, k, filt, None, want, cout)
    got = torch.full((R, cout), 5.0, device='cuda')
    ops.conv2d_thin_mfma(src0, c0, None, 0, B, H, W, k, filt, None, got, cout)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5 * float(want.abs().max()))


@pytest.mark.parametrize("c0,c1,k,cout,B,H,W,splits", [(16, 0, 5, 32, 2, 12, 8, 3), (32, 64, 3, 32, 3, 8, 16, 1), (128, 128, 3, 128, 1, 4, 4, 2),
                                                       (16, 32, 3, 16, 2, 64, 8, 5), (4, 0, 3, 7, 1, 8, 8, 1)])
def test_implicit_filter_gradient_matches_im2col_gemm(c0, c1, k, cout, B, H, W, splits):
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(c0 + c1 + k + 1)
    R = B * H * W
    src0 = torch.randn(R, c0 + 4, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    kc, ld = k * k * (c0 + c1), -(-cout // 4) * 4
    dy = torch.zeros(R, ld, device='cuda')
    dy[:, :cout] = torch.randn(R, cout, generator=g, device='cuda')
    col = torch.empty(R, kc, device='cuda')
    ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
    want = torch.empty(kc, ld, device='cuda')
    ops.gemm_splitk(col, dy, want, trans_a=True, m=kc, n=ld, k=R, splits=splits)
    got = torch.full((kc, ld), 7.0, device='cuda')
    assert ops.conv2d_wgrad_supported(c0, c1, R) and not ops.conv2d_wgrad_supported(1, 16, R)
    ops.conv2d_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, got, splits)
    scale = want.abs().max().item()
    np.testing.assert_allclose(got[:, :cout].cpu().numpy(), want[:, :cout].cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)


def test_captured_hip_graph_replays_the_inference_step(mods):
    """capture_graph(): one HIP graph per feed() instead of ~70 launches; same numbers as the eager path."""
    import torch
    from avsi_amd import _lib, models
    B, N = 4, 16384
    cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
               lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    wavs = [torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000) for _ in range(3)]
    masks = torch.ones(B, 128, 128, device='cuda')
    masks[:, 30:55] = 0
    seq = np.full(B, 128)
    mean, std = torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2
    eager = models.UNetFConvModel(seq, wavs[0], masks, mean, std, 0.0, cfg, is_training=False, seed=2)
    want = []
    for w in wavs:
        eager.feed(seq, w, masks)
        want.append((eager.prediction.clone(), float(eager.loss_func)))
    m = models.UNetFConvModel(seq, wavs[0], masks, mean, std, 0.0, cfg, is_training=False, variables=eager.variables)
    m.capture_graph()
    for w, (pred, loss) in zip(wavs[::-1], want[::-1]):          # different order: results follow the fed data
        m.feed(seq, w, masks)
        assert torch.equal(m.prediction, pred) and float(m.loss_func) == loss
    with pytest.raises(_lib.AvsiError):
        m.feed(np.full(B - 1, 128), wavs[0][:B - 1], masks[:B - 1])
    m.release_graph()
    m.feed(np.full(B - 1, 128), wavs[0][:B - 1], masks[:B - 1])
    assert m.prediction.shape == (B - 1, 128, 128)


def test_inference_model_fused_layers_match_training_form(mods):
    """A model built with is_training=False takes the fused kernels (first layer: 7 x 7 convolution + ReLU + pooling in
    one pass; encoder layers: batch norm + activation + pooling in one pass without the full-resolution activation;
    single-launch statistics): same prediction and loss as the keeping form and as the oracle; asking it for gradients
    re-runs the forward pass in the keeping form."""
    models, ops = mods
    B, N = 3, 16384
    wav, masks, mean, std, T, _ = _inputs(B, N, 5)
    seq = np.array([T, T, T - 5])
    params = OU.init_params(6)
    rng = np.random.default_rng(7)
    for k in params:
        if k.endswith('gamma'):
            params[k] = rng.uniform(0.7, 1.3, params[k].shape).astype(np.float32)
        elif k.endswith('beta') or k.endswith('/b'):
            params[k] = rng.normal(0, 0.1, params[k].shape).astype(np.float32)
    mt = models.UNetFConvModel(seq, wav, masks, mean, std, 0.0, _cfg(N), is_training=True)
    mt.variables.load_flat(mt.layout.flatten_params(params))
    mi = models.UNetFConvModel(seq, wav, masks, mean, std, 0.0, _cfg(N), is_training=False, variables=mt.variables)
    pt, pi = mt.prediction.cpu().numpy(), mi.prediction.cpu().numpy()
    assert 'e1' not in mi._cache['saved'] and mi._cache['saved']['e2']['y'] is None      # nothing kept
    assert np.abs(pt - pi).max() < 1e-5
    assert 'inference' not in mi._cache                       # the fused tail wrote the masked prediction only ...
    assert np.abs(mi.inference.cpu().numpy() - mt.inference.cpu().numpy()).max() < 1e-5      # ... the logits on request
    assert np.all(pi[2, T - 5:] == 0) and np.abs(mi.inference.cpu().numpy()[2, T - 5:]).max() > 0
    assert float(mi.loss_func) == pytest.approx(float(mt.loss_func), rel=1e-6)
    st = OF.get_stft(wav, window_size=16, step_size=8, n_fft=256)[:, :, :128]
    ref_norm = (OF.get_spectrogram(st, log=True) - mean) / std
    ref = OU.forward_backward(ref_norm * masks, ref_norm, seq, params, want_grads=False)
    assert np.sqrt(np.mean((pi - ref['prediction']) ** 2)) < 2e-4
    gi, gt = mi.gradients.cpu().numpy(), mt.gradients.cpu().numpy()
    assert np.abs(gi - gt).max() <= 1e-6 * np.abs(gt).max() + 1e-9


@pytest.mark.parametrize("k,c0,c1,cout,B,H,W", [(3, 16, 32, 16, 3, 64, 64), (5, 16, 0, 32, 2, 64, 64), (3, 16, 32, 16, 1, 8, 32),
                                                (5, 16, 0, 32, 2, 4, 96)])
def test_thin_mfma_conv_matches_implicit_gemm(k, c0, c1, cout, B, H, W):
    """avsi_conv2d_thin_mfma_f32 (LDS patch + v_mfma_f32_16x16x4_f32) against the implicit-GEMM convolution, incl. the
    image border, the fused 2x up-sampling + concat and tiles at the image corner."""
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    assert ops.conv2d_thin_mfma_supported(k, c0, c1, cout, H, W)
    g = torch.Generator(device='cuda')
    g.manual_seed(k + c1)
    R = B * H * W
    src0 = torch.randn(R, c0, generator=g, device='cuda')
    src1 = torch.randn(B * (H // 2) * (W // 2), c1, generator=g, device='cuda') if c1 else None
    filt = torch.randn(k * k * (c0 + c1), cout, generator=g, device='cuda') * 0.1
    bias = torch.randn(cout, generator=g, device='cuda')
    ref = torch.empty(R, cout, device='cuda')
    ops.conv2d(src0, c0, src1, c1, B, H, W, k, filt, bias, ref, cout)
    got = torch.full((R, cout), 7.0, device='cuda')
    ops.conv2d_thin_mfma(src0, c0, src1, c1, B, H, W, k, filt, bias, got, cout)
    assert (got - ref).abs().max().item() < 2e-4 * ref.abs().max().item()


def test_inference_model_captures_its_step_by_itself(monkeypatch):
    """An inference U-Net fed the same shapes three times captures its step into a HIP graph on its own (launch-bound at the
    reference's batch of 32: 0.81 -> 0.62 ms per step); predictions and losses are those of plain launches, bit for bit, and a
    change of shape takes it back to plain launches instead of failing."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    monkeypatch.delenv('AVSI_UNET_GRAPH', raising=False)
    B, N = 4, 16384
    cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
               lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    wavs = [torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000) for _ in range(6)]
    masks = torch.ones(B, 128, 128, device='cuda')
    masks[:, 40:52] = 0
    mean, std = torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2
    seq = np.full(B, 128)

    def run(auto):
        if not auto:
            monkeypatch.setenv('AVSI_UNET_GRAPH', '0')
        else:
            monkeypatch.delenv('AVSI_UNET_GRAPH', raising=False)
        m = models.UNetFConvModel(seq, wavs[0], masks, mean, std, 0.0, cfg, is_training=False, seed=5)
        out = []
        for w in wavs:
            m.feed(seq, w, masks)
            out.append((m.prediction.clone(), float(m.loss_func)))
        return m, out
    m_plain, plain = run(False)
    m_auto, auto = run(True)
    assert getattr(m_plain, '_graph', None) is None and getattr(m_auto, '_graph', None) is not None
    for (p0, l0), (p1, l1) in zip(plain, auto):
        assert torch.equal(p0, p1) and l0 == l1
    # another batch size: back to plain launches, same results as a fresh model
    m_auto.feed(seq[:2], wavs[1][:2], masks[:2])
    assert getattr(m_auto, '_graph', None) is None
    ref = models.UNetFConvModel(seq[:2], wavs[1][:2], masks[:2], mean, std, 0.0, dict(cfg, batch_size=2), is_training=False,
                                variables=m_auto.variables)
    assert torch.equal(m_auto.prediction, ref.prediction)


def test_training_resumes_after_a_validation_pass_that_captured_a_graph(monkeypatch):
    """train() validates with model.is_training = False and then goes back to training (training.py).  Five same-shape
    validation feeds make the model capture its inference step; the captured graph must be gone when training resumes:
    the training steps after the validation pass equal those of a model that never validated in between."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    monkeypatch.delenv('AVSI_UNET_GRAPH', raising=False)
    B, N = 4, 16384
    cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
               lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    g = torch.Generator(device='cuda')
    g.manual_seed(4)
    wavs = [torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000) for _ in range(3)]
    masks = torch.ones(B, 128, 128, device='cuda')
    masks[:, 40:52] = 0
    mean, std = torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2
    seq = np.full(B, 128)

    def run(validate):
        m = models.UNetFConvModel(seq, wavs[0], masks, mean, std, 0.0, cfg, is_training=True, seed=5)
        losses = []
        for epoch in range(2):
            for w in wavs[:2]:
                m.feed(seq, w, masks)
                losses.append(float(m.loss_func))
                m.train_op
            if validate:
                m.is_training = False
                for _ in range(5):
                    m.feed(seq, wavs[2], masks)
                    float(m.loss_func)
                assert getattr(m, '_graph', None) is not None          # the validation pass did capture its step
                m.is_training = True
        return m, losses
    m_plain, plain = run(False)
    m_val, val = run(True)
    assert getattr(m_val, '_graph', None) is not None                  # still held after the LAST validation pass ...
    m_val.feed(seq, wavs[0], masks)                                    # ... and dropped by the next training feed
    assert getattr(m_val, '_graph', None) is None
    assert plain == val and m_val.global_step == 4
    assert torch.equal(m_plain.variables.flat, m_val.variables.flat)


@pytest.mark.parametrize("k,c0,c1,cout,B,H,W", [(3, 16, 32, 16, 3, 64, 64), (5, 16, 0, 32, 2, 64, 64),       # 16-wide MFMA route
                                               (3, 32, 64, 32, 3, 8, 16), (5, 32, 0, 64, 2, 32, 32), (3, 128, 128, 128, 5, 4, 6),
                                               (3, 64, 0, 128, 1, 16, 16)])                                # implicit GEMM, N tiles 32 / 64 / 128
def test_conv_with_statistics_from_its_epilogue(k, c0, c1, cout, B, H, W, monkeypatch):
    """avsi_conv2d_bn_f32: the same output as the plain convolution of the same route, bit for bit, and the batch statistics
    of that output (tf.layers.batch_normalization(training=True), unet_layers.py:14,33) equal to a float64 mean / variance
    over it -- taken from per-tile partial sums in the convolution's epilogue, not from a pass over the output."""
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(k * 100 + cout)
    ld0, ld1 = -(-c0 // 4) * 4, -(-max(c1, 1) // 4) * 4
    s0 = torch.randn(B * H * W, ld0, generator=g, device='cuda')
    s1 = torch.randn(B * (H // 2) * (W // 2), ld1, generator=g, device='cuda') if c1 else None
    filt = torch.randn(k * k * (c0 + c1), cout, generator=g, device='cuda') * 0.05
    bias = torch.randn(cout, generator=g, device='cuda')
    plain = torch.zeros(B * H * W, cout, device='cuda')
    if ops.conv2d_thin_mfma_supported(k, c0, c1, cout, H, W):
        ops.conv2d_thin_mfma(s0, c0, s1, c1, B, H, W, k, filt, bias, plain, cout)
    else:
        monkeypatch.setattr(ops, '_CONV_SPLITK', False)          # the plain launch: these small shapes would split the reduction
        ops.conv2d(s0, c0, s1, c1, B, H, W, k, filt, bias, plain, cout)
    out = torch.zeros_like(plain)
    mean, rstd = torch.zeros(cout, device='cuda'), torch.zeros(cout, device='cuda')
    ops.conv2d_bn(s0, c0, s1, c1, B, H, W, k, filt, bias, out, cout, mean, rstd, eps=1e-3)
    assert torch.equal(out, plain)
    x = plain.double()
    m = x.mean(dim=0)
    var = (x * x).mean(dim=0) - m * m
    np.testing.assert_allclose(mean.cpu().numpy(), m.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), (1.0 / torch.sqrt(var + 1e-3)).cpu().numpy(), rtol=1e-5)
    mean2, rstd2 = torch.zeros(cout, device='cuda'), torch.zeros(cout, device='cuda')
    ops.conv2d_bn(s0, c0, s1, c1, B, H, W, k, filt, bias, out, cout, mean2, rstd2, eps=1e-3)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)           # deterministic
