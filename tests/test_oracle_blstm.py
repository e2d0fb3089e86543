"""Pin the BLSTM oracle against torch.nn.LSTM / torch.autograd on CPU.
No GPU."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as F


def _perm_ijfo_to_ifgo(w, H):
    """TF LSTMBlockCell column blocks (i, j, f, o) -> torch row blocks (i, f, g, o)."""
    i, j, f, o = (w[k * H:(k + 1) * H] for k in range(4))
    return np.concatenate([i, f, j, o], axis=0)


def _torch_lstm_from_params(params, D, H):
    L = len(params['layers'])
    net = torch.nn.LSTM(D, H, num_layers=L, bidirectional=True, batch_first=True).double()
    with torch.no_grad():
        for li, layer in enumerate(params['layers']):
            d_in = D if li == 0 else 2 * H
            for sfx, dname in (('', 'fw'), ('_reverse', 'bw')):
                K = np.asarray(layer[dname]['kernel'], dtype=np.float64)
                b = np.asarray(layer[dname]['bias'], dtype=np.float64)
                getattr(net, 'weight_ih_l%d%s' % (li, sfx)).copy_(
                    torch.from_numpy(_perm_ijfo_to_ifgo(K[:d_in].T, H)))
                getattr(net, 'weight_hh_l%d%s' % (li, sfx)).copy_(
                    torch.from_numpy(_perm_ijfo_to_ifgo(K[d_in:].T, H)))
                getattr(net, 'bias_ih_l%d%s' % (li, sfx)).copy_(
                    torch.from_numpy(_perm_ijfo_to_ifgo(b, H)))
                getattr(net, 'bias_hh_l%d%s' % (li, sfx)).zero_()
    return net


def _rand_bias(params, seed):
    rng = np.random.default_rng(seed)
    for layer in params['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = rng.normal(0, 0.1, size=layer[d]['bias'].shape)
    params['proj']['biases'] = rng.normal(0, 0.1, size=params['proj']['biases'].shape)
    return params


def test_param_counts_anchor():
    assert O.num_params(O.init_params(0, 257)) == 4148757
    assert O.num_params(O.init_params(0, 393)) == 4420757
    p = O.init_params(0, 257)
    assert p['layers'][0]['fw']['kernel'].shape == (507, 1000)
    assert p['layers'][1]['bw']['kernel'].shape == (750, 1000)
    assert p['proj']['weights'].shape == (500, 257)
    assert np.abs(p['proj']['weights']).max() <= 2.0 / np.sqrt(500) + 1e-7


@pytest.mark.parametrize('D,H,T,B', [(7, 5, 9, 3), (13, 6, 4, 2)])
def test_stack_matches_torch_lstm(D, H, T, B):
    params = _rand_bias(O.cast_params(O.init_params(1, D, (H, H, H), 11), np.float64), 2)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(B, T, D))
    ours = O.blstm_stack(x, params)
    ref, _ = _torch_lstm_from_params(params, D, H)(torch.from_numpy(x))
    np.testing.assert_allclose(ours, ref.detach().numpy(), atol=1e-12)


def test_backward_matches_torch_autograd():
    D, H, T, B, F = 6, 5, 7, 3, 6
    params = _rand_bias(O.cast_params(O.init_params(4, D, (H, H), F), np.float64), 5)
    rng = np.random.default_rng(6)
    x = rng.normal(size=(B, T, D))
    target = rng.normal(size=(B, T, F))
    masks = np.ones((B, T, F))
    masks[:, 2:4, :] = 0
    seq_len = np.array([T, T - 2, T])

    logits, rnn, caches = O.inference(x, params, keep=True)
    pred = O.prediction(logits, seq_len)
    fwd = {'params': params, 'target_spec_norm': target, 'prediction': pred,
           'rnn_outputs': rnn, 'caches': caches}
    g = O.model_backward(fwd, masks, seq_len)

    net = _torch_lstm_from_params(params, D, H)
    W = torch.tensor(params['proj']['weights'], requires_grad=True)
    bb = torch.tensor(params['proj']['biases'], requires_grad=True)
    xt = torch.tensor(x, requires_grad=True)
    out, _ = net(xt)
    lg = out.reshape(B * T, 2 * H) @ W + bb
    sm = torch.from_numpy(O.sequence_mask(seq_len, T, np.float64))[:, :, None]
    loss = (torch.from_numpy(target) - sm * lg.reshape(B, T, F)).abs().mean()
    loss.backward()
    assert loss.item() == pytest.approx(O.losses(target, pred, masks)['loss_func'], rel=1e-12)
    np.testing.assert_allclose(g['proj']['weights'], W.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(g['proj']['biases'], bb.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(g['net_inputs'], xt.grad.numpy(), atol=1e-12)
    for li in range(2):
        d_in = D if li == 0 else 2 * H
        for sfx, dname in (('', 'fw'), ('_reverse', 'bw')):
            gih = getattr(net, 'weight_ih_l%d%s' % (li, sfx)).grad.numpy()
            ghh = getattr(net, 'weight_hh_l%d%s' % (li, sfx)).grad.numpy()
            gb = getattr(net, 'bias_ih_l%d%s' % (li, sfx)).grad.numpy()
            dk = g['layers'][li][dname]['kernel']
            np.testing.assert_allclose(_perm_ijfo_to_ifgo(dk[:d_in].T, H), gih, atol=1e-12)
            np.testing.assert_allclose(_perm_ijfo_to_ifgo(dk[d_in:].T, H), ghh, atol=1e-12)
            np.testing.assert_allclose(_perm_ijfo_to_ifgo(g['layers'][li][dname]['bias'], H), gb,
                                       atol=1e-12)


def test_losses_definitions():
    rng = np.random.default_rng(7)
    t, p = rng.normal(size=(2, 5, 4)), rng.normal(size=(2, 5, 4))
    m = np.ones((2, 5, 4))
    m[:, 1:3] = 0
    L = O.losses(t, p, m)
    e = np.abs(t - p)
    assert L['loss_func'] == pytest.approx(e.mean())
    assert L['loss_hole'] == pytest.approx(e[:, 1:3].mean())
    assert L['loss_valid'] == pytest.approx(np.concatenate([e[:, :1], e[:, 3:]], 1).mean())
    assert L['loss'] == L['loss_func']


def test_adam_tf_differs_from_torch_adam_only_by_epsilon_placement():
    rng = np.random.default_rng(8)
    p0 = rng.normal(size=50)
    grads = [rng.normal(size=50) * s for s in (1.0, 1e-4, 3.0, 1e-6, 0.5)]
    p, m, v = p0.copy(), np.zeros(50), np.zeros(50)
    q, qm, qv = p0.copy(), np.zeros(50), np.zeros(50)
    for step, g in enumerate(grads, 1):
        O.adam_tf_step(p, g, m, v, step, lr=1e-3)
        # 5-line restatement straight from App. A.7
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        qm = 0.9 * qm + 0.1 * g
        qv = 0.999 * qv + 0.001 * g * g
        q = q - lr_t * qm / (np.sqrt(qv) + 1e-8)
    np.testing.assert_allclose(p, q, rtol=1e-14)
    # with eps -> 0 the TF form equals the textbook (torch) form
    tp = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-3, eps=1e-30)
    p2, m2, v2 = p0.copy(), np.zeros(50), np.zeros(50)
    for step, g in enumerate(grads, 1):
        tp.grad = torch.tensor(g)
        opt.step()
        O.adam_tf_step(p2, g, m2, v2, step, lr=1e-3, eps=1e-30)
    np.testing.assert_allclose(p2, tp.detach().numpy(), rtol=1e-9)


def test_exponential_decay_staircase():
    assert O.exponential_decay(0.1, 9999, 10000, 0.5) == pytest.approx(0.1)
    assert O.exponential_decay(0.1, 10000, 10000, 0.5) == pytest.approx(0.05)
    assert O.exponential_decay(0.1, 25000, 10000, 0.5) == pytest.approx(0.025)


def test_model_forward_shapes_and_padded_frames_run_through_recurrence():
    """padded frames are NOT masked inside the recurrence, only at the output."""
    rng = np.random.default_rng(9)
    B, N = 2, 192 * 12
    wav = np.round(rng.normal(0, 3000, size=(B, N)))
    masks = np.ones((B, 12, 257))
    masks[:, 4:7] = 0
    params = O.init_params(10, 257, (8, 8), 257)
    mean, std = np.zeros(257), np.ones(257)
    full = O.model_forward(wav, masks, mean, std, np.array([12, 12]), params)
    short = O.model_forward(wav, masks, mean, std, np.array([12, 9]), params)
    assert full['prediction'].shape == (B, 12, 257)
    np.testing.assert_array_equal(short['prediction'][1, 9:], 0)
    # frames before the cut are identical: the backward direction still started at t = T-1
    np.testing.assert_allclose(short['prediction'][1, :9], full['prediction'][1, :9], atol=0)


def test_dropout_factor_gradient_matches_autograd():
    """model_forward(drop_scale=...) / model_backward: the gradient through tf.nn.dropout (models.py:117) is the same
    factor; checked against torch.autograd on a tiny model."""
    import torch
    rng = np.random.default_rng(21)
    B, N, T = 2, 960, 5
    wav = np.round(rng.normal(0, 3000, size=(B, N))).astype(np.float32)
    masks = np.ones((B, T, 257), dtype=np.float32)
    masks[:, 2:4] = 0
    spec = F.get_spectrogram(F.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = F.feature_stats(list(spec))
    p = O.init_params(2, 257, (6, 6), 257)
    seq = np.full(B, T)
    drop = (rng.uniform(size=(B, T, 12)) > 0.3) / 0.7
    fwd = O.model_forward(wav, masks, mean, std, seq, p, keep=True, drop_scale=drop)
    grads = O.model_backward(fwd, masks.astype(np.float64), seq)
    # torch: the projection part with autograd on the dropped activations
    w = torch.tensor(O.cast_params(p, np.float64)['proj']['weights'], requires_grad=True)
    b = torch.tensor(O.cast_params(p, np.float64)['proj']['biases'], requires_grad=True)
    h = torch.tensor(fwd['rnn_outputs'])
    pred = (h.reshape(B * T, -1) @ w + b).reshape(B, T, -1)
    loss = (torch.tensor(fwd['target_spec_norm']) - pred).abs().mean()
    loss.backward()
    np.testing.assert_allclose(grads['proj']['weights'], w.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(grads['proj']['biases'], b.grad.numpy(), atol=1e-12)
    assert np.all(fwd['rnn_outputs'][drop == 0] == 0)
