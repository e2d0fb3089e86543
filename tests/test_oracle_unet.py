"""Pin the U-Net oracle's layer semantics against explicit numpy restatements (no GPU)."""
import numpy as np
import torch

from oracle import unet as OU


def test_parameter_count_and_shapes():
    p = OU.init_params(0)
    assert sum(v.size for v in p.values()) == 1171118
    assert p['e1/w'].shape == (7, 7, 1, 16) and p['d1/w'].shape == (3, 3, 256, 128) and p['out/w'].shape == (1, 1, 1, 1)
    assert 'e1/bn/gamma' not in p and 'e2/bn/gamma' in p and 'd6/bn/beta' in p and 'out/bn/gamma' not in p
    assert np.all(p['e3/b'] == np.float32(0.1))
    sd = np.sqrt(2.0 / (5 * 5 * 32))
    assert np.abs(p['e2/w']).max() <= 2 * sd + 1e-6


def test_conv_same_equals_numpy_im2col():
    """tf.nn.conv2d(SAME, stride 1) with an HWIO filter == zero-padded patch matrix times the reshaped filter."""
    rng = np.random.default_rng(1)
    B, H, W, ci, co, k = 2, 6, 5, 3, 4, 5
    x = rng.normal(size=(B, H, W, ci))
    w = rng.normal(size=(k, k, ci, co))
    b = rng.normal(size=co)
    got = OU._conv(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w), torch.tensor(b)).permute(0, 2, 3, 1).numpy()
    pad = np.pad(x, [(0, 0), (k // 2, k // 2), (k // 2, k // 2), (0, 0)])
    cols = np.concatenate([pad[:, kh:kh + H, kw:kw + W, :] for kh in range(k) for kw in range(k)], axis=3)
    ref = cols.reshape(B * H * W, -1) @ w.reshape(-1, co) + b
    np.testing.assert_allclose(got.reshape(-1, co), ref, atol=1e-12)


def test_batch_norm_uses_biased_batch_statistics():
    x = torch.tensor(np.random.default_rng(2).normal(3.0, 2.0, size=(4, 3, 5, 5)))
    y = OU._bn(x, torch.ones(3, dtype=torch.float64) * 2, torch.ones(3, dtype=torch.float64)).numpy()
    xn = x.numpy()
    mean = xn.mean(axis=(0, 2, 3), keepdims=True)
    var = xn.var(axis=(0, 2, 3), keepdims=True)               # biased
    np.testing.assert_allclose(y, 2 * (xn - mean) / np.sqrt(var + 1e-3) + 1, atol=1e-12)


def test_shapes_through_the_network_and_gradients_exist():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 64, 64))
    out = OU.forward_backward(x, x * 0.5, [64, 60], OU.init_params(4))
    assert out['prediction'].shape == (2, 64, 64)
    assert np.all(out['prediction'][1, 60:] == 0)
    assert set(out['grads']) == set(OU.init_params(4))
    # the bias in front of a batch norm cannot change the output
    assert np.abs(out['grads']['e2/b']).max() < 1e-12 and np.abs(out['grads']['e1/b']).max() > 0
