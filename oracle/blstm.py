"""Oracle: stacked BLSTM inpainter forward / BPTT / loss / TF-Adam (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Restates the graph of ``StackedBLSTMModel`` (reference
``av_speech_inpainting/models.py:20-197``) in its CPU-compatible form
(``CudnnCompatibleLSTMCell`` under ``stack_bidirectional_dynamic_rnn``,
models.py:106-115) following SURVEY.md Appendix A.5, A.7, A.8.  The explicit
per-step loop with one ``[B, D+H] . [D+H, 4H]`` product per step is the
schedule of the reference's ``tf.while_loop``; it is kept on purpose.
"""
import math

import numpy as np

from . import frontend

GATE_ORDER = ('i', 'j', 'f', 'o')   # LSTMBlockCell column blocks (App. A.5)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def init_params(seed, input_dim, net_dim=(250, 250, 250), audio_feat_dim=257, dtype=np.float32):
    """TF default initialisers (App. A.5 / A.8; models.py:107,119-121):
    LSTM kernel glorot-uniform [(D_l+H), 4H], bias zeros; projection
    truncated-normal(stddev 1/sqrt(2H)) [2H, F], bias zeros."""
    rng = np.random.default_rng(seed)
    params = {'layers': []}
    d = input_dim
    for H in net_dim:
        layer = {}
        for direction in ('fw', 'bw'):
            fan_in, fan_out = d + H, 4 * H
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            layer[direction] = {
                'kernel': rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype),
                'bias': np.zeros(4 * H, dtype=dtype),
            }
        params['layers'].append(layer)
        d = 2 * H
    sd = 1.0 / math.sqrt(float(d))
    w = rng.normal(0.0, sd, size=(d, audio_feat_dim))
    bad = np.abs(w) > 2 * sd
    while bad.any():                      # truncated_normal: re-sample outside +-2 sigma
        w[bad] = rng.normal(0.0, sd, size=int(bad.sum()))
        bad = np.abs(w) > 2 * sd
    params['proj'] = {'weights': w.astype(dtype), 'biases': np.zeros(audio_feat_dim, dtype=dtype)}
    return params


def cast_params(params, dtype):
    return {
        'layers': [{d: {k: np.asarray(v, dtype=dtype) for k, v in layer[d].items()}
                    for d in ('fw', 'bw')} for layer in params['layers']],
        'proj': {k: np.asarray(v, dtype=dtype) for k, v in params['proj'].items()},
    }


def flatten_params(params):
    """Variable order of the flat parameter/gradient buffer (layer, fw/bw, kernel/bias, proj)."""
    out = []
    for li, layer in enumerate(params['layers']):
        for d in ('fw', 'bw'):
            out.append(('cell_%d/%s/kernel' % (li, d), layer[d]['kernel']))
            out.append(('cell_%d/%s/bias' % (li, d), layer[d]['bias']))
    out.append(('logits/weights', params['proj']['weights']))
    out.append(('logits/biases', params['proj']['biases']))
    return out


def num_params(params):
    return sum(int(v.size) for _, v in flatten_params(params))


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def lstm_direction(x, kernel, bias, reverse=False, keep=False):
    """One direction of one layer: LSTMBlockCell(H, forget_bias=0) unrolled over the FULL
    padded length (no sequence_length; models.py:111-115).

    x [B, T, D] -> h [B, T, H].  With keep=True also returns the per-step
    (i, j, f, o, c, c_prev, h_prev) needed by BPTT."""
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    dt = x.dtype
    h = np.zeros((B, H), dtype=dt)
    c = np.zeros((B, H), dtype=dt)
    out = np.zeros((B, T, H), dtype=dt)
    cache = [] if keep else None
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = np.concatenate([x[:, t, :], h], axis=1) @ kernel + bias
        i = sigmoid(z[:, 0 * H:1 * H])
        j = np.tanh(z[:, 1 * H:2 * H])
        f = sigmoid(z[:, 2 * H:3 * H])
        o = sigmoid(z[:, 3 * H:4 * H])
        c_new = f * c + i * j
        h_new = o * np.tanh(c_new)
        if keep:
            cache.append((t, i, j, f, o, c_new, c, h))
        c, h = c_new, h_new
        out[:, t, :] = h
    return (out, cache) if keep else out


def blstm_stack(x, params, keep=False):
    """stack_bidirectional_dynamic_rnn: layer output = concat(fw, bw) feeds the next layer."""
    caches = []
    inp = x
    for layer in params['layers']:
        if keep:
            fw, cf = lstm_direction(inp, layer['fw']['kernel'], layer['fw']['bias'], False, True)
            bw, cb = lstm_direction(inp, layer['bw']['kernel'], layer['bw']['bias'], True, True)
            caches.append({'input': inp, 'fw': cf, 'bw': cb})
        else:
            fw = lstm_direction(inp, layer['fw']['kernel'], layer['fw']['bias'], False)
            bw = lstm_direction(inp, layer['bw']['kernel'], layer['bw']['bias'], True)
        inp = np.concatenate([fw, bw], axis=2)
    return (inp, caches) if keep else inp


def sequence_mask(seq_len, T, dtype):
    return (np.arange(T)[None, :] < np.asarray(seq_len)[:, None]).astype(dtype)


def inference(net_inputs, params, keep=False, drop_scale=None):
    """models.py:89-125.  ``drop_scale`` [B, T, 2H]: the factor tf.nn.dropout(rnn_outputs, rate) multiplies each
    element with (0 for a dropped element, 1 / (1 - rate) for a kept one; models.py:117); None = rate 0, the identity
    (App. A.8).  The random draw itself cannot be restated (TensorFlow's generator), so the factor is an input."""
    if keep:
        rnn, caches = blstm_stack(net_inputs, params, True)
    else:
        rnn = blstm_stack(net_inputs, params)
    if drop_scale is not None:
        rnn = rnn * np.asarray(drop_scale, dtype=rnn.dtype)
    B, T, C = rnn.shape
    logits = rnn.reshape(B * T, C) @ params['proj']['weights'] + params['proj']['biases']
    logits = logits.reshape(B, T, -1)
    return (logits, rnn, caches) if keep else logits


def prediction(logits, seq_len):
    """models.py:127-138."""
    return sequence_mask(seq_len, logits.shape[1], logits.dtype)[:, :, None] * logits


def losses(target_norm, pred, masks, params=None, l2=0.0):
    """models.py:140-159 -> dict(loss, loss_func, loss_hole, loss_valid, reg_loss)."""
    err = np.abs(target_norm - pred)
    out = {
        'loss_hole': (err * (1 - masks)).sum() / (1 - masks).sum(),
        'loss_valid': (err * masks).sum() / masks.sum(),
        'loss_func': err.mean(),
    }
    reg = 0.0
    if l2 and params is not None:
        reg = sum(float((v.astype(np.float64) ** 2).sum()) / 2.0 for _, v in flatten_params(params))
    out['reg_loss'] = reg
    out['loss'] = out['loss_func'] + l2 * reg
    return out


def net_inputs(audio_features, video_features, input_type):
    """models.py:38-45."""
    if input_type == 'a':
        return audio_features
    if input_type == 'v':
        return np.asarray(video_features, dtype=audio_features.dtype)
    return np.concatenate([audio_features,
                           np.asarray(video_features, dtype=audio_features.dtype)], axis=2)


def model_forward(wav, masks, mean, std, seq_len, params, video=None, input_type='a',
                  dtype=np.float64, l2=0.0, keep=False, drop_scale=None):
    """Whole StackedBLSTMModel forward from the feed boundary (training.py:67-74):
    wav [B, N], masks [B, T, F], mean/std [F], seq_len [B] -> dict of every tensor the
    reference drivers fetch."""
    p = cast_params(params, dtype)
    masks = np.asarray(masks, dtype=dtype)
    T = int(np.max(seq_len))
    stft_c, norm, feats = frontend.inpainter_frontend(wav, mean, std, masks, dtype,
                                                      audio_feat_dim=masks.shape[2], max_len=T)
    x = net_inputs(feats, video, input_type)
    if keep:
        logits, rnn, caches = inference(x, p, True, drop_scale)     # rnn: AFTER dropout (what the projection saw)
    else:
        logits = inference(x, p, drop_scale=drop_scale)
    pred = prediction(logits, seq_len)
    out = {'target_stft': stft_c, 'target_spec_norm': norm, 'audio_features': feats,
           'net_inputs': x, 'inference': logits, 'prediction': pred}
    out.update(losses(norm, pred, masks, p, l2))
    if keep:
        out['rnn_outputs'] = rnn
        out['caches'] = caches
        out['params'] = p
        out['drop_scale'] = None if drop_scale is None else np.asarray(drop_scale, dtype=dtype)
    return out


def enhanced_sources(pred, mean, std, target_stft, masks=None, num_samples=48000, dtype=np.float64, window_size=24,
                     step_size=12):
    """models.py:181-197: exp(pred*std+mean) with the masked (masks given) or oracle phase.  ``window_size`` /
    ``step_size``: the U-Net's 16 / 8 ms geometry (models.py:664-680 with the App. B9 handling)."""
    mag = np.exp(pred * np.asarray(std, dtype=dtype) + np.asarray(mean, dtype=dtype))
    if masks is None:
        phase = np.angle(target_stft)
    else:
        # models.py:186 multiplies by tf.cast(masks, complex64): a FULL complex product
        # (a + bj)(m + 0j) = (a m - b 0) + (a 0 + b m) j.  In a gap (m = 0) the result is a signed
        # zero pair and tf.angle = atan2 turns (-0, +0) into pi: gap bins whose target has Re < 0
        # and Im > 0 get phase pi, all other gap bins phase 0.  Restated literally, not "fixed".
        a = np.real(target_stft).astype(dtype)
        b = np.imag(target_stft).astype(dtype)
        m = np.asarray(masks, dtype=dtype)
        zero = np.zeros((), dtype=dtype)
        phase = np.arctan2(a * zero + b * m, a * m - b * zero)
    return frontend.get_sources(mag, phase, num_samples=num_samples, window_size=window_size, step_size=step_size,
                                dtype=dtype)


# --------------------------------------------------------------------------------------
# backward (manual BPTT; checked against torch.autograd in tests/test_oracle_blstm.py)
# --------------------------------------------------------------------------------------
def _lstm_direction_bwd(x, kernel, cache, dh_out, return_dz=False):
    """Gradient of lstm_direction.  dh_out [B, T, H] -> (dx [B,T,D], dkernel, dbias); with return_dz also the gradient
    of the gate pre-activations, dz [B, T, 4H] (column blocks i, j, f, o) -- what the BPTT kernels write."""
    B, T, D = x.shape
    H = kernel.shape[1] // 4
    dt = x.dtype
    dx = np.zeros_like(x)
    dk = np.zeros_like(kernel)
    db = np.zeros(4 * H, dtype=dt)
    dh_next = np.zeros((B, H), dtype=dt)
    dc_next = np.zeros((B, H), dtype=dt)
    dz_all = np.zeros((B, T, 4 * H), dtype=dt) if return_dz else None
    for (t, i, j, f, o, c_new, c_prev, h_prev) in reversed(cache):
        dh = dh_out[:, t, :] + dh_next
        tc = np.tanh(c_new)
        do = dh * tc
        dc = dh * o * (1 - tc * tc) + dc_next
        di = dc * j
        dj = dc * i
        df = dc * c_prev
        dc_next = dc * f
        dz = np.concatenate([di * i * (1 - i), dj * (1 - j * j), df * f * (1 - f),
                             do * o * (1 - o)], axis=1)
        if return_dz:
            dz_all[:, t, :] = dz
        xin = np.concatenate([x[:, t, :], h_prev], axis=1)
        dk += xin.T @ dz
        db += dz.sum(axis=0)
        dxin = dz @ kernel.T
        dx[:, t, :] = dxin[:, :D]
        dh_next = dxin[:, D:]
    return (dx, dk, db, dz_all) if return_dz else (dx, dk, db)


def model_backward(fwd, masks, seq_len, l2=0.0):
    """d(loss)/d(params) for a model_forward(keep=True) result; L1 subgradient sign(0)=0
    as in tf.abs.  Returns a params-shaped dict of gradients."""
    p = fwd['params']
    norm, pred, rnn = fwd['target_spec_norm'], fwd['prediction'], fwd['rnn_outputs']
    B, T, F = pred.shape
    dt = pred.dtype
    dpred = np.sign(pred - norm) / dt.type(B * T * F)
    dlogits = dpred * sequence_mask(seq_len, T, dt)[:, :, None]
    dl2 = dlogits.reshape(B * T, F)
    grads = {'layers': [None] * len(p['layers']), 'proj': {}}
    grads['proj']['weights'] = rnn.reshape(B * T, -1).T @ dl2
    grads['proj']['biases'] = dl2.sum(axis=0)
    dout = (dl2 @ p['proj']['weights'].T).reshape(B, T, -1)
    if fwd.get('drop_scale') is not None:
        dout = dout * fwd['drop_scale']              # gradient of tf.nn.dropout: the same factor
    for li in range(len(p['layers']) - 1, -1, -1):
        layer, cache = p['layers'][li], fwd['caches'][li]
        H = layer['fw']['kernel'].shape[1] // 4
        dx_f, dk_f, db_f = _lstm_direction_bwd(cache['input'], layer['fw']['kernel'], cache['fw'],
                                               dout[:, :, :H])
        dx_b, dk_b, db_b = _lstm_direction_bwd(cache['input'], layer['bw']['kernel'], cache['bw'],
                                               dout[:, :, H:])
        grads['layers'][li] = {'fw': {'kernel': dk_f, 'bias': db_f},
                               'bw': {'kernel': dk_b, 'bias': db_b}}
        dout = dx_f + dx_b
    if l2:
        for (_, g), (_, v) in zip(flatten_params(grads), flatten_params(p)):
            g += l2 * v
    grads['net_inputs'] = dout
    return grads


# --------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------
def adam_tf_step(param, grad, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (App. A.7; models.py:168): epsilon is added to the
    UN-corrected sqrt(v).  ``step`` is the 1-based step count.  In place."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m *= beta1
    m += (1.0 - beta1) * grad
    v *= beta2
    v += (1.0 - beta2) * grad * grad
    param -= lr_t * m / (np.sqrt(v) + eps)


def sgd_tf_step(param, grad, lr):
    """tf.train.GradientDescentOptimizer update (models.py:171): var -= lr * grad.  In place."""
    param -= lr * grad


def momentum_tf_step(param, grad, accum, lr, momentum=0.9):
    """tf.train.MomentumOptimizer(lr, momentum=0.9) update (models.py:173; TF's ApplyMomentum, use_nesterov=False):
    accum = momentum * accum + grad; var -= lr * accum -- the rate multiplies the accumulator when it is applied, so a
    decayed rate rescales the whole history (unlike the form that folds lr into the accumulator).  In place."""
    accum *= momentum
    accum += grad
    param -= lr * accum


def exponential_decay(lr0, global_step, decay_steps, decay_rate, staircase=True):
    """tf.train.exponential_decay (models.py:165-166)."""
    e = global_step / decay_steps
    if staircase:
        e = math.floor(e)
    return lr0 * decay_rate ** e
