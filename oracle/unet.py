"""Oracle: U-Net spectrogram inpainter (torch CPU float64 + autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Restates ``UNetFConvModel.inference`` (reference models.py:582-607) with
``encoder_layer_fconv`` / ``decoder_layer_fconv`` (unet_layers.py:6-37) layer by layer:
tf.nn.conv2d(SAME, stride 1) + bias, tf.layers.batch_normalization(training=True: batch
statistics, biased variance, eps 1e-3, gamma/beta), relu / leaky_relu(0.2), 2x2 max pooling,
UpSampling2D(2,2) nearest, concat([skip, up]).  As committed the reference encoder never
down-samples while the decoder up-samples; the intended design -- 2x2 max pool
after every encoder layer, which the channel counts 256/256/192/96/48/17 and the experiment name
'unet_maxpool' imply -- is what is restated here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ENCODER = [(7, 1, 16, False), (5, 16, 32, True), (5, 32, 64, True), (3, 64, 128, True), (3, 128, 128, True),
           (3, 128, 128, True)]                       # (k, cin, cout, batch_norm)
DECODER = [(3, 256, 128), (3, 256, 128), (3, 192, 64), (3, 96, 32), (3, 48, 16), (3, 17, 1)]


def layer_names():
    return ['e%d' % (i + 1) for i in range(6)] + ['d%d' % (i + 1) for i in range(6)] + ['out']


def layer_specs():
    """name -> (k, cin, cout, batch_norm)"""
    specs = {}
    for i, (k, ci, co, bn) in enumerate(ENCODER):
        specs['e%d' % (i + 1)] = (k, ci, co, bn)
    for i, (k, ci, co) in enumerate(DECODER):
        specs['d%d' % (i + 1)] = (k, ci, co, True)
    specs['out'] = (1, 1, 1, False)
    return specs


def init_params(seed, dtype=np.float32):
    """unet_layers.py:7-9: w ~ truncated_normal(stddev sqrt(2 / (k^2 cout))), b = 0.1; BN gamma 1, beta 0."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, (k, ci, co, bn) in layer_specs().items():
        sd = math.sqrt(2.0 / (k * k * co))
        w = rng.normal(0, sd, size=(k, k, ci, co))
        bad = np.abs(w) > 2 * sd
        while bad.any():
            w[bad] = rng.normal(0, sd, size=int(bad.sum()))
            bad = np.abs(w) > 2 * sd
        params[name + '/w'] = w.astype(dtype)
        params[name + '/b'] = np.full(co, 0.1, dtype=dtype)
        if bn:
            params[name + '/bn/gamma'] = np.ones(co, dtype=dtype)
            params[name + '/bn/beta'] = np.zeros(co, dtype=dtype)
    return params


def _conv(x, w, b):
    k = w.shape[0]
    return F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=k // 2)        # HWIO -> OIHW, SAME for odd k


def _bn(x, gamma, beta, eps=1e-3):
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    return gamma.view(1, -1, 1, 1) * (x - mean) / torch.sqrt(var + eps) + beta.view(1, -1, 1, 1)


def inference_torch(net_inputs, p):
    """net_inputs [B, T, F] (torch) -> logits [B, T, F]; p: name -> torch tensor."""
    x = net_inputs.unsqueeze(1)                                          # NCHW with C = 1
    skips = [x]
    h = x
    for i in range(6):
        n = 'e%d' % (i + 1)
        h = _conv(h, p[n + '/w'], p[n + '/b'])
        if n + '/bn/gamma' in p:
            h = _bn(h, p[n + '/bn/gamma'], p[n + '/bn/beta'])
        h = F.max_pool2d(F.relu(h), 2)
        skips.append(h)
    for i in range(6):
        n = 'd%d' % (i + 1)
        up = F.interpolate(h, scale_factor=2, mode='nearest')
        h = torch.cat([skips[5 - i], up], dim=1)
        h = _conv(h, p[n + '/w'], p[n + '/b'])
        h = F.leaky_relu(_bn(h, p[n + '/bn/gamma'], p[n + '/bn/beta']), 0.2)
    h = _conv(h, p['out/w'], p['out/b'])
    return h.squeeze(1)


def forward_backward(net_inputs, target_norm, seq_len, params, want_grads=True):
    """-> dict(inference, prediction, loss_func, grads {name: array})."""
    p = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=want_grads) for k, v in params.items()}
    x = torch.tensor(np.asarray(net_inputs, dtype=np.float64))
    logits = inference_torch(x, p)
    T = logits.shape[1]
    sm = (torch.arange(T)[None, :] < torch.tensor(np.asarray(seq_len))[:, None]).to(torch.float64)[:, :, None]
    pred = sm * logits
    loss = (torch.tensor(np.asarray(target_norm, dtype=np.float64)) - pred).abs().mean()
    out = {'inference': logits.detach().numpy(), 'prediction': pred.detach().numpy(), 'loss_func': float(loss.detach())}
    if want_grads:
        loss.backward()
        out['grads'] = {k: v.grad.numpy() for k, v in p.items()}
    return out
