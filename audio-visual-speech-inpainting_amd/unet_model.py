"""U-Net spectrogram inpainter on MI355X (host side) -- ``UNetFConvModel`` of the reference
(``av_speech_inpainting/models.py:519-715`` with ``unet_layers.py:6-37``), BASELINE.json configs[4].

Same constructor signature and attribute names as the reference class.  Front end: 16 ms / 8 ms /
n_fft 256 (129 bins sliced to ``audio_feat_dim`` = 128), log-magnitude, z-norm, times mask.
Network: six encoder layers conv(k = 7,5,5,3,3,3; 1-16-32-64-128-128-128) + bias (+ batch norm,
not on the first) + ReLU + 2x2 max pooling, six decoder layers [2x nearest up-sampling ++ skip]
conv 3x3 + bias + batch norm + LeakyReLU(0.2), a final 1x1 convolution.  As committed the
reference encoder never down-samples while its decoder up-samples; the 2x2
max pooling its channel counts and experiment name ('unet_maxpool') imply is what runs here.
Batch normalisation always uses batch statistics (the reference never runs UPDATE_OPS and builds
the model with is_training=True everywhere).

Forward convolutions: implicit GEMM on the fp32-MFMA GEMM kernel for the layers whose channel
counts are multiples of 16 (``ops.conv2d``), direct kernels for the three thin full-resolution
layers (``ops.conv2d_thin``), im2col + GEMM otherwise; gradients are a split-K GEMM over the
im2col matrix (filter) and a GEMM + gather-form col2im (input).  Activations are NHWC
``[B*H*W][C]`` with the channel pitch padded to a multiple of 4.
"""
import math
import os
import sys

import numpy as np
import torch

from . import _lib, ops, parallel
from . import audio_processing as ap
from .blstm_layout import round_up

_DEFER_BN = os.environ.get('AVSI_UNET_DEFER_BN', '1') != '0'
ENCODER = [(7, 1, 16, False), (5, 16, 32, True), (5, 32, 64, True), (3, 64, 128, True), (3, 128, 128, True),
           (3, 128, 128, True)]
DECODER = [(3, 256, 128), (3, 256, 128), (3, 192, 64), (3, 96, 32), (3, 48, 16), (3, 17, 1)]


def layer_specs():
    """Ordered (name, k, cin, cout, batch_norm, activation) in variable-creation order."""
    out = [('e%d' % (i + 1), k, ci, co, bn, 1) for i, (k, ci, co, bn) in enumerate(ENCODER)]
    out += [('d%d' % (i + 1), k, ci, co, True, 2) for i, (k, ci, co) in enumerate(DECODER)]
    out.append(('out', 1, 1, 1, False, 0))
    return out


class UNetLayout:
    """Reference variables (w [k,k,cin,cout], b, bn gamma/beta per layer) <-> GEMM-ready packed form
    (filter as [Kc][ld] with Kc = k*k*cin and ld = cout, both rounded up to 4)."""

    def __init__(self):
        self.specs = layer_specs()
        self.ref_entries, self.packed, self.gpacked = [], {}, {}
        off = poff = 0
        for name, k, ci, co, bn, _ in self.specs:
            for vname, shape in ([('w', (k, k, ci, co)), ('b', (co,))]
                                 + ([('bn/gamma', (co,)), ('bn/beta', (co,))] if bn else [])):
                self.ref_entries.append(('%s/%s' % (name, vname), shape, off))
                off += int(np.prod(shape))
            kc, ld = round_up(k * k * ci, 4), round_up(co, 4)
            for vname, shape in ([('w', (kc, ld)), ('b', (ld,))] + ([('bn/gamma', (ld,)), ('bn/beta', (ld,))] if bn else [])):
                self.packed['%s/%s' % (name, vname)] = (poff, shape)
                poff += round_up(int(np.prod(shape)), 64)
        self.ref_size, self.packed_size, self.gpacked_size = off, poff, poff
        self.gpacked = self.packed                       # gradients come out in the packed shapes
        idx = np.full(self.packed_size, self.ref_size, dtype=np.int64)
        gi = np.full(self.ref_size, -1, dtype=np.int64)
        for name, shape, roff in self.ref_entries:
            poff, pshape = self.packed[name]
            if name.endswith('/w'):
                k, _, ci, co = shape
                rows = np.arange(k * k * ci)[:, None]
                cols = np.arange(co)[None, :]
                pos = poff + rows * pshape[1] + cols
                src = roff + rows * co + cols
            else:
                pos = poff + np.arange(shape[0])
                src = roff + np.arange(shape[0])
            idx[pos.reshape(-1)] = src.reshape(-1)
            gi[src.reshape(-1)] = pos.reshape(-1)
        assert (gi >= 0).all()
        self.pack_index, self.grad_index = idx, gi

    def signature(self):
        return [4, self.ref_size, len(self.specs), 0]

    def ref_view(self, flat, name):
        for n, shape, off in self.ref_entries:
            if n == name:
                return flat[off: off + int(np.prod(shape))].reshape(shape)
        raise KeyError(name)

    def packed_view(self, flat, name):
        off, shape = self.packed[name]
        return flat[off: off + int(np.prod(shape))].reshape(shape)

    gpacked_view = packed_view

    def flatten_params(self, params):
        flat = np.zeros(self.ref_size, dtype=np.float32)
        for n, _, _ in self.ref_entries:
            self.ref_view(flat, n)[...] = params[n]
        return flat


def _unet_default_init(layout, seed):
    """unet_layers.py:7-9,24-26: w ~ truncated_normal(sqrt(2 / (k^2 cout))), b = 0.1; BN gamma 1, beta 0."""
    rng = np.random.default_rng(seed)
    flat = np.zeros(layout.ref_size, dtype=np.float32)
    for name, shape, off in layout.ref_entries:
        n = int(np.prod(shape))
        if name.endswith('/w'):
            sd = math.sqrt(2.0 / (shape[0] * shape[1] * shape[3]))
            w = rng.normal(0.0, sd, size=n)
            bad = np.abs(w) > 2 * sd
            while bad.any():
                w[bad] = rng.normal(0.0, sd, size=int(bad.sum()))
                bad = np.abs(w) > 2 * sd
            flat[off:off + n] = w
        elif name.endswith('/b'):
            flat[off:off + n] = 0.1
        elif name.endswith('/gamma'):
            flat[off:off + n] = 1.0
    return flat


class UNetFConvModel(object):
    """
    Speech inpainting U-Net model with full convolutions (reference models.py:519-715).
    Input: log-compressed linear spectrogram of corrupted audio.  Output: restored spectrogram.
    Loss: L1 (target_spectrogram - reconstructed_spectrogram).
    """

    def __init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                 is_training=True, variables=None, seed=0):
        from .models import BLSTMVariables, _as_device
        _lib.require_cuda()
        self._as_device = _as_device
        self.audio_feat_dim = config['audio_feat_dim']
        self.audio_len = config['audio_len']
        self.dropout_rate = dropout_rate
        self.net_dim = config.get('net_dim')
        self.optimizer_choice = config['optimizer_type']
        self.starter_learning_rate = config['starter_learning_rate']
        self.updating_step = config['lr_updating_steps']
        self.learning_decay = config['lr_decay']
        self.is_training = is_training
        self.batch_size = config.get('batch_size', 1)
        self.regularization = config['l2']
        self.var_scope = None
        self.layout = variables.layout if variables is not None else UNetLayout()
        self.variables = variables if variables is not None else BLSTMVariables(self.layout, seed=seed,
                                                                                  init=_unet_default_init)
        self.device = self.variables.device
        self._ws, self._cache, self._flip_idx = {}, {}, {}
        self.audio_feat_mean = _as_device(audio_feat_mean, device=self.device)
        self.audio_feat_std = _as_device(audio_feat_std, device=self.device)
        self.sequence_lengths = None
        self.feed(sequence_lengths=sequence_lengths, target_sources=target_sources, masks=masks)

    # ------------------------------------------------------------------ feed boundary
    def set_dropout_rate(self, rate):
        """The reference's U-Net takes the dropout_rate placeholder and never uses it (models.py:519-715)."""

    def feed(self, sequence_lengths=None, target_sources=None, masks=None, audio_feat_mean=None, audio_feat_std=None,
             **_unused):
        self._auto_graph(sequence_lengths, target_sources, masks)
        self._cache = {}
        graph = getattr(self, '_graph', None)
        if sequence_lengths is not None:
            lens = np.asarray(sequence_lengths.cpu() if isinstance(sequence_lengths, torch.Tensor) else sequence_lengths,
                              dtype=np.int64)
            if graph is not None and (lens.shape != self.sequence_lengths.shape or int(lens.max()) != int(self.sequence_lengths.max())):
                raise _lib.AvsiError("the captured graph was built for another batch / frame count: call release_graph() first")
            self.sequence_lengths = lens
            if graph is not None:
                self._seq_dev.copy_(torch.as_tensor(lens))
            else:
                self._seq_dev = torch.as_tensor(lens, device=self.device)
        if graph is not None:
            # replay mode: new data goes INTO the buffers the graph reads; results are those of the replay
            if target_sources is not None:
                self.target_sources.copy_(self._as_device(target_sources, device=self.device))
            if masks is not None:
                self.masks.copy_(self._as_device(masks, device=self.device))
            graph.replay()
            self._cache = dict(self._graph_cache)
            return
        self.target_sources = self._as_device(target_sources, device=self.device)
        self.masks = self._as_device(masks, device=self.device)
        if audio_feat_mean is not None:
            self.audio_feat_mean = self._as_device(audio_feat_mean, device=self.device)
        if audio_feat_std is not None:
            self.audio_feat_std = self._as_device(audio_feat_std, device=self.device)

    def build_graph(self, var_scope=''):
        self.var_scope = var_scope

    # Inference at the reference's batch sizes is launch-bound once the convolutions fill the chip (32 clips: 0.81 ms per
    # step issued launch by launch, 0.62 ms as one graph), so an inference model captures its step BY ITSELF after it has
    # been fed the same shapes three times in a row (up to AUTO_GRAPH_MAX_CLIPS clips; AVSI_UNET_GRAPH=0: never), and goes
    # back to plain launches when the shapes change.  Results of a step then live in buffers the next feed() overwrites.
    AUTO_GRAPH_AFTER = 3
    AUTO_GRAPH_MAX_CLIPS = 128        # (r6: captured at 512 clips as well, the step takes 3.318 against 3.320 ms: not launch-bound there)

    def _auto_graph(self, sequence_lengths, target_sources, masks):
        if self.is_training:
            # a model that goes back to training (train() validates with is_training = False, then resumes) must not
            # replay the inference-form step: its cache keeps no activation for the backward pass
            if getattr(self, '_graph', None) is not None:
                self.release_graph()
            self._auto_shape, self._auto_count = None, 0
            return
        if os.environ.get('AVSI_UNET_GRAPH', '1') == '0' or getattr(self, '_graph_manual', False):
            return
        shape = (None if sequence_lengths is None else (len(sequence_lengths), int(np.max(np.asarray(
                     sequence_lengths.cpu() if isinstance(sequence_lengths, torch.Tensor) else sequence_lengths)))),
                 None if target_sources is None else tuple(target_sources.shape),
                 None if masks is None else tuple(masks.shape))
        if None in shape:
            return
        captured = getattr(self, '_graph', None) is not None
        if shape != getattr(self, '_auto_shape', None):
            self._auto_shape, self._auto_count = shape, 0
            if captured:
                self.release_graph()
            return
        if captured:
            return
        # count only steps whose results were asked for: those workspaces and tables exist
        self._auto_count += 1 if 'loss3' in self._cache else 0
        if self._auto_count >= self.AUTO_GRAPH_AFTER and shape[0][0] <= self.AUTO_GRAPH_MAX_CLIPS:
            try:
                self.capture_graph(_auto=True)
            except Exception as e:         # never a reason to fail a step: stay with plain launches -- but say so, once
                self.release_graph()
                self._auto_count = -(1 << 30)
                print('avsi: the U-Net inference step was not captured into a HIP graph (%s: %s); plain launches from here on'
                      % (type(e).__name__, str(e)[:200]), file=sys.stderr, flush=True)

    # ------------------------------------------------------------------ HIP graph of the inference step
    def capture_graph(self, _auto=False):
        """Capture front end -> 13 layers -> prediction -> loss (about 70 launches) into one HIP graph
        (torch.cuda.CUDAGraph).  At the reference's batch of 32 a step is launch-bound -- the kernels
        take about half of the 1.5 ms -- so replaying one graph per feed() instead of issuing every
        launch from Python is what is left to gain there.  After capture, feed() copies the new clips
        into the captured input buffers and replays; batch size and frame count are fixed until
        release_graph().  Inference only (training updates the packed weights between steps, which a
        graph would have to re-capture)."""
        if self.is_training:
            raise _lib.AvsiError("capture_graph() is for inference models")
        if self.target_sources is None or self.masks is None or self.sequence_lengths is None:
            raise _lib.AvsiError("feed() the model once before capture_graph()")
        self.release_graph()
        self._graph_manual = not _auto
        self.target_sources, self.masks = self.target_sources.clone(), self.masks.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up: every workspace and table exists before capture
            for _ in range(2):
                self._cache = {}
                self._loss()
        torch.cuda.current_stream().wait_stream(side)
        self._cache = {}
        graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of this process (the reader's upload thread, a collective's watchdog) keep making
        # capture-unsafe calls (allocations, pinned memory) while this thread captures; in the default global mode those
        # calls fail THERE and invalidate the capture
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            self._loss()
        self._graph, self._graph_cache = graph, dict(self._cache)
        graph.replay()
        return graph

    def release_graph(self):
        self._graph, self._graph_cache = None, None
        self._graph_manual = False

    def _buf(self, name, shape, zero=True):
        key = (name, tuple(shape))
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(shape, dtype=torch.float32, device=self.device)
            self._ws[key] = t
        return t

    def _dims(self):
        B = int(self.target_sources.shape[0])
        T = int(self.sequence_lengths.max())
        F = self.audio_feat_dim
        if T % 64 or F % 64:
            raise _lib.AvsiError("the U-Net pools 6 times: frames (%d) and bins (%d) must be multiples of 64" % (T, F))
        return B, T, F

    # ------------------------------------------------------------------ front end (models.py:536-540)
    def _frontend(self):
        c = self._cache
        if 'x0' in c:
            return
        B, T, F = self._dims()
        m = self.masks[:, :T].contiguous()
        fe = ap.frontend(self.target_sources, window_size=16, step_size=8, n_fft=256, num_frames_out=T, num_bins=F,
                         mean=self.audio_feat_mean, std=self.audio_feat_std, masks=m, want_spec=True, want_feat=True)
        # the one-channel layers -- forward, filter gradients, the skip into the last 3 x 3 layer -- read the front end's output
        # where it lies (channel pitch 1): no copy (until round 5 the training form kept a pitch-4 copy, a strided copy per step)
        x0 = fe['feat'].reshape(B * T * F, 1)
        c['target_spec_norm'], c['net_inputs'], c['x0'], c['mask_t'] = fe['spec'], fe['feat'], x0, m

    @property
    def target_spec_norm(self):
        self._frontend()
        return self._cache['target_spec_norm']

    @property
    def net_inputs(self):
        self._frontend()
        return self._cache['net_inputs']

    # ------------------------------------------------------------------ network (models.py:582-607)
    def _conv_fwd(self, name, k, cout, bn, act, src0, c0, src1, c1, B, H, W, pool=False, src1_bn=None, defer=False):
        """One layer: convolution, batch statistics, normalisation + activation -- and, for the encoder layers
        (``pool``), the 2 x 2 max pooling in the same pass as the activation.  A model built for inference
        (``is_training=False``) keeps nothing it does not need: the pooled layers never write their full-resolution
        activation, and the first layer (7 x 7, one input channel, no batch norm) is ONE kernel from input to pooled
        output.  Returns the layer's output (pooled if ``pool``)."""
        v = self.variables
        R, kc, ld = B * H * W, round_up(k * k * (c0 + c1), 4), round_up(cout, 4)
        keep = bool(self.is_training) or bool(getattr(self, '_keep_for_backward', False))
        pooled = self._buf(name + '/pool', (B * (H // 2) * (W // 2), ld)) if pool else None
        if pool and not keep and not bn and act == 1 and (k, c0, c1, cout) == (7, 1, 0, 16):
            ops.conv2d_thin_relu_pool(src0, B, H, W, k, v.p(name + '/w'), v.p(name + '/b'), pooled, cout)
            return pooled
        conv = self._buf(name + '/conv', (R, ld))
        st = None
        if bn and ops.conv2d_bn_supported(k, c0, c1, cout, B, H, W, ld):
            # batch statistics from the convolution's own epilogue (per-tile partial sums + one small finishing launch): no
            # pass over the output
            st = (self._buf(name + '/mean', (ld,)), self._buf(name + '/rstd', (ld,)))
            ops.conv2d_bn(src0, c0, src1, c1, B, H, W, k, v.p(name + '/w'), v.p(name + '/b'), conv, cout, st[0], st[1],
                          src1_bn=src1_bn)
            if defer:
                # inference form: this layer's normalise + activate pass does not run -- the layer above applies them while it
                # stages its operand (ops.conv2d_bn / ops.unet_tail with src1_bn)
                self._cache['saved'][name] = dict(k=k, cout=cout, bn=bn, act=act, conv=conv, stats=st, y=None, kc=kc, ld=ld)
                return conv, (st[0], st[1], v.p(name + '/bn/gamma'), v.p(name + '/bn/beta'))
        elif ops.conv2d_thin_mfma_supported(k, c0, c1, cout, H, W):
            # few channels at high resolution: 16-wide MFMA from an LDS patch (the 128 x 32 GEMM tile wastes half of it)
            ops.conv2d_thin_mfma(src0, c0, src1, c1, B, H, W, k, v.p(name + '/w'), v.p(name + '/b'), conv, cout)
        elif ops.conv2d_supported(c0, c1):
            # implicit GEMM: the operand rows are gathered from the activations by the GEMM's DMA loads
            ops.conv2d(src0, c0, src1, c1, B, H, W, k, v.p(name + '/w'), v.p(name + '/b'), conv, cout)
        elif ops.conv2d_thin_supported(k, c0, c1, cout):
            ops.conv2d_thin(src0, c0, src1, c1, B, H, W, k, v.p(name + '/w'), v.p(name + '/b'), conv, cout)
        else:
            col = self._buf('col', (self._col_floats,))[: R * kc].view(R, kc)
            ops.im2col(src0, c0, src1, c1, B, H, W, k, col, kc)
            ops.gemm(col, v.p(name + '/w'), out=conv, n=cout, bias=v.p(name + '/b'))
        if bn and st is None:
            st = (self._buf(name + '/mean', (ld,)), self._buf(name + '/rstd', (ld,)))
            ops.colstats(conv, cout, st[0], st[1])
        bn_args = (*(st or (None, None)), v.p(name + '/bn/gamma') if bn else None, v.p(name + '/bn/beta') if bn else None)
        # (pooled layers: the backward pass recomputes the window's activations from `conv`, ops.bn_act_pool_bwd -- the
        # full-resolution activation is not written in training either)
        y = self._buf(name + '/act', (R, ld)) if not pool else None
        if pool:
            ops.bn_act_pool(conv, B, H, W, cout, pooled, y, *bn_args, act)
        elif not bn and act == 0 and not keep:
            y = conv                     # the output layer (no batch norm, no activation): nothing to apply, no copy either
        else:
            ops.bn_act(conv, cout, y, *bn_args, act)
        self._cache['saved'][name] = dict(k=k, cout=cout, bn=bn, act=act, src0=src0, c0=c0, src1=src1, c1=c1, B=B, H=H,
                                          W=W, conv=conv, stats=st, y=y, kc=kc, ld=ld)
        return pooled if pool else y

    def _forward(self):
        c = self._cache
        if 'pred' in c:
            return
        self._frontend()
        B, T, F = self._dims()
        self._col_floats = B * T * F * 160                              # largest im2col matrix: d6 (17 ch x 9 taps)
        self._dx_floats = B * T * F * 16                                # >= the largest [R][C0 + C1] input gradient (d5: 64 x 64 x 48)
        c['saved'], c['pool'] = {}, {}
        h, H, W, ch = c['x0'], T, F, 1
        skips = [(c['x0'], 1, T, F)]
        for i, (k, ci, co, bn) in enumerate(ENCODER):
            name = 'e%d' % (i + 1)
            pooled = self._conv_fwd(name, k, co, bn, 1, h, ch, None, 0, B, H, W, pool=True)
            c['pool'][name] = pooled
            h, H, W, ch = pooled, H // 2, W // 2, co
            skips.append((pooled, co, H, W))
        keep = bool(self.is_training) or bool(getattr(self, '_keep_for_backward', False))
        fused_tail = not keep and ops.unet_tail_supported(T, F) and c['x0'].stride(0) == 1
        pending = None            # (mean, rstd, gamma, beta) of a layer whose batch norm + activation the NEXT layer applies
        for i, (k, ci, co) in enumerate(DECODER):
            name = 'd%d' % (i + 1)
            skip, cs, Hs, Ws = skips[5 - i]
            if fused_tail and name == 'd6':
                break
            # inference form: d4 and d5 leave their output raw when the layer above reads it through registers (the
            # 16-wide-MFMA route of d5, the fused tail of d6) -- two of the largest normalise / activate passes disappear
            nk, _, nco = DECODER[i + 1] if i + 1 < len(DECODER) else (0, 0, 0)
            Hn, Wn = (skips[4 - i][2], skips[4 - i][3]) if i + 1 < len(DECODER) else (0, 0)
            defer = (not keep and _DEFER_BN and ops.conv2d_bn_supported(k, cs, ch, co, B, Hs, Ws, round_up(co, 4))
                     and ((name == 'd5' and fused_tail)
                          or (name == 'd4' and ops.conv2d_thin_mfma_supported(nk, skips[4 - i][1], co, nco, Hn, Wn)
                              and ops.conv2d_bn_supported(nk, skips[4 - i][1], co, nco, B, Hn, Wn, round_up(nco, 4)))))
            if pending is not None and not ops.conv2d_bn_supported(k, cs, ch, co, B, Hs, Ws, round_up(co, 4)):
                raise _lib.AvsiError("internal: a deferred batch norm has no consumer")      # (the test above guarantees one)
            out = self._conv_fwd(name, k, co, True, 2, skip, cs, h, ch, B, Hs, Ws, src1_bn=pending, defer=defer)
            h, pending = out if defer else (out, None)
            H, W, ch = Hs, Ws, co
        seq = self._seq_dev
        if fused_tail:
            # inference form: the last two layers and the sequence mask are one call -- the 17 -> 1 convolution leaves its
            # batch statistics behind, then ONE pass applies batch norm, LeakyReLU, the 1 x 1 output convolution and the mask
            # (five passes over the full-resolution tensor before: normalise, convolve, mask, two copies)
            c['tail'] = (h, B, T, F, pending)
            c['pred'] = self._tail(False)
            return
        logits = self._conv_fwd('out', 1, 1, False, 0, h, 1, None, 0, B, T, F)
        rowmask = (torch.arange(T, device=self.device)[None, :] < seq[:, None]).to(torch.float32)
        c['rowmask'] = rowmask
        c['inference'] = logits[:, 0].reshape(B, T, F)
        c['pred'] = (c['inference'] * rowmask[:, :, None]).contiguous()

    def _tail(self, want_logits):
        c, v = self._cache, self.variables
        h, B, T, F, pending = c['tail']
        pred = torch.empty((B, T, F), dtype=torch.float32, device=self.device)       # a result: never a recycled buffer
        logits = torch.empty((B, T, F), dtype=torch.float32, device=self.device) if want_logits else None
        ops.unet_tail(c['x0'], h, B, T, F, v.p('d6/w'), v.p('d6/b'), v.p('d6/bn/gamma'), v.p('d6/bn/beta'), v.p('out/w'),
                      v.p('out/b'), self._seq_dev, self._buf('tail/conv', (B * T * F,), zero=False), pred, logits,
                      src1_bn=pending)
        if want_logits:
            c['inference'] = logits
        return pred

    @property
    def inference(self):
        self._forward()
        if 'inference' not in self._cache:          # the fused inference tail writes the un-masked logits on request only
            self._tail(True)
        return self._cache['inference']

    @property
    def prediction(self):
        self._forward()
        return self._cache['pred']

    # ------------------------------------------------------------------ loss (models.py:617-634)
    def _loss(self, want_grad=False):
        c = self._cache
        want_grad = want_grad or bool(self.is_training)
        if 'loss3' in c and (not want_grad or c.get('dpred') is not None):
            return
        self._forward()
        c['loss3'], c['dpred'] = ops.l1_loss(self.target_spec_norm, c['pred'], c['mask_t'], want_grad=want_grad)

    @property
    def loss_func(self):
        self._loss()
        return self._cache['loss3'][0]

    @property
    def loss_hole(self):
        self._loss()
        return self._cache['loss3'][1]

    @property
    def loss_valid(self):
        self._loss()
        return self._cache['loss3'][2]

    @property
    def reg_loss(self):
        if self.regularization:
            return (self.variables.flat.double() ** 2).sum().float() / 2.0
        return torch.zeros((), device=self.device)

    @property
    def loss(self):
        return self.loss_func + self.regularization * self.reg_loss if self.regularization else self.loss_func

    # ------------------------------------------------------------------ gradients / optimiser
    def _conv_bwd(self, name, dy, gp, dsrc0, acc0, dsrc1, acc1, pooled=False):
        """``pooled``: `dy` is the gradient of the layer's 2 x 2 max-POOLED output (encoder layers): pooling, activation
        and batch norm go backwards in one kernel pair that recomputes the window from the convolution output."""
        s, v, lay = self._cache['saved'][name], self.variables, self.layout
        R, kc, ld, cout = s['B'] * s['H'] * s['W'], s['kc'], s['ld'], s['cout']
        dconv = self._buf(name + '/dconv', (R, ld))
        st = s['stats'] or (None, None)
        bn_args = (st[0], st[1], v.p(name + '/bn/gamma') if s['bn'] else None, v.p(name + '/bn/beta') if s['bn'] else None,
                   s['act'], lay.gpacked_view(gp, name + '/bn/gamma') if s['bn'] else None,
                   lay.gpacked_view(gp, name + '/bn/beta') if s['bn'] else None)
        # The bias gradient, sum of dconv over the pixels: under batch norm dconv = gamma rstd (g - mean g - xhat mean(g xhat))
        # sums to zero identically (the normalisation removes any bias), so it IS zero -- the buffer's zeros stay; a column
        # sum would return rounding noise (13 passes over the gradients per step).  Layers without batch norm (e1, out) sum.
        if not s['bn'] and s['act'] == 0 and not pooled and dy.shape == dconv.shape:
            dconv = dy                  # the output layer: no batch norm, no activation -- its gradient IS the incoming one
            ops.colsum(dconv, lay.gpacked_view(gp, name + '/b'), m=R, n=ld)
        elif pooled:
            ops.bn_act_pool_bwd(s['conv'], dy, s['B'], s['H'], s['W'], cout, dconv, *bn_args,
                                dbias=None if s['bn'] else lay.gpacked_view(gp, name + '/b'))
        else:
            ops.bn_act_bwd(s['conv'], dy, cout, dconv, *bn_args)
            if not s['bn']:
                ops.colsum(dconv, lay.gpacked_view(gp, name + '/b'), m=R, n=ld)
        # reduction slabs: enough workgroups (output tiles x slabs ~ 1024) even when the filter is one 128 x 128 tile
        tiles = -(-kc // 128) * -(-ld // 128)
        splits = max(1, min(R // 2048, max(64, 1024 // tiles)))
        implicit_w = ops.conv2d_wgrad_supported(s['c0'], s['c1'], R)
        if ops.conv2d_thin_supported(s['k'], s['c0'], s['c1'], cout):
            # the thin full-resolution layers: direct kernels for the filter gradient and the one input gradient needed
            ops.conv2d_thin_wgrad(s['src0'], s['c0'], s['src1'], s['c1'], s['B'], s['H'], s['W'], s['k'], dconv, cout,
                                  lay.gpacked_view(gp, name + '/w'))
            if dsrc0 is not None:        # 'out' (1 x 1, one channel): dX = w dY, the forward kernel itself
                assert s['k'] == 1 and not acc0
                ops.conv2d_thin(dconv, 1, None, 0, s['B'], s['H'], s['W'], 1, v.p(name + '/w'), None, dsrc0, 1)
            if dsrc1 is not None:
                ops.conv2d_thin_dx_coarse(dconv, v.p(name + '/w'), dsrc1, acc1, s['B'], s['H'], s['W'])
            return
        dwv = lay.gpacked_view(gp, name + '/w')
        if ops.conv2d_thin_mfma_wgrad_supported(s['k'], s['c0'], s['c1'], cout, s['H'], s['W'], dwv):
            # few channels at high resolution: patch + dY through LDS once, the filter gradient in MFMA accumulators
            ops.conv2d_thin_mfma_wgrad(s['src0'], s['c0'], s['src1'], s['c1'], s['B'], s['H'], s['W'], s['k'], dconv, cout, dwv)
        elif implicit_w:
            ops.conv2d_wgrad(s['src0'], s['c0'], s['src1'], s['c1'], s['B'], s['H'], s['W'], s['k'], dconv, cout,
                             lay.gpacked_view(gp, name + '/w'), splits)
        else:
            col = self._buf('col', (self._col_floats,))[: R * kc].view(R, kc)
            ops.im2col(s['src0'], s['c0'], s['src1'], s['c1'], s['B'], s['H'], s['W'], s['k'], col, kc)
            ops.gemm_splitk(col, dconv, lay.gpacked_view(gp, name + '/w'), trans_a=True, m=kc, n=ld, k=R, splits=splits)
        if dsrc0 is None and dsrc1 is None:
            return
        if cout % 16 == 0 and s['c0'] % 4 == 0 and s['c1'] % 4 == 0:
            # dX = conv2d(dY, tap-flipped transposed filter): implicit GEMM again, then split / 2x2-sum into the sources
            ct = s['c0'] + s['c1']
            dxc = self._buf('dxcat', (self._dx_floats,))[: R * ct].view(R, ct)
            if ops.conv2d_thin_mfma_plain_supported(s['k'], cout, 0, ct, s['H'], s['W']) and dxc.stride(0) == ct:
                ops.conv2d_thin_mfma(dconv, cout, None, 0, s['B'], s['H'], s['W'], s['k'], self._flipped_filter(name, s), None,
                                     dxc, ct)
            else:
                ops.conv2d(dconv, cout, None, 0, s['B'], s['H'], s['W'], s['k'], self._flipped_filter(name, s), None, dxc, ct)
            ops.split_sumpool(dxc, dsrc0, s['c0'], acc0, dsrc1, s['c1'], acc1, s['B'], s['H'], s['W'])
            return
        col = self._buf('col', (self._col_floats,))[: R * kc].view(R, kc)
        ops.gemm(dconv, v.p(name + '/w'), out=col, trans_b=True, m=R, n=kc, k=ld)          # d(im2col matrix)
        ops.col2im(col, kc, dsrc0, s['c0'], dsrc1, s['c1'], s['B'], s['H'], s['W'], s['k'], acc0, acc1)

    def _flipped_filter(self, name, s):
        """[k*k*cout][ct] filter of the input-gradient convolution: Wt[(tap', n)][c] = W[(k*k-1-tap'), c][n]."""
        k, cout, ct, ld = s['k'], s['cout'], s['c0'] + s['c1'], s['ld']
        idx = self._flip_idx.get(name)
        if idx is None:
            tap = np.arange(k * k)[:, None, None]
            n = np.arange(cout)[None, :, None]
            c = np.arange(ct)[None, None, :]
            src = ((k * k - 1 - tap) * ct + c) * ld + n                      # position inside the packed [kc][ld] filter
            idx = torch.from_numpy(src.reshape(k * k * cout, ct).astype(np.int64)).to(self.device)
            self._flip_idx[name] = idx
        return self.variables.p(name + '/w').reshape(-1)[idx]

    def _backward(self):
        c = self._cache
        if 'grads' in c:
            return c['grads']
        if not self.is_training and not getattr(self, '_keep_for_backward', False):
            # gradients asked of a model built for inference: its forward pass kept nothing (fused pooling layers);
            # run it again in the keeping form
            self._keep_for_backward = True
            for key in ('x0', 'pred', 'inference', 'saved', 'pool', 'loss3', 'dpred', 'rowmask'):
                c.pop(key, None)
        self._loss(want_grad=True)
        B, T, F = self._dims()
        lay, saved = self.layout, c['saved']
        gp = self._buf('gpacked', (lay.gpacked_size,))
        dlog = self._buf('dlog', (B * T * F, 4))
        dlog[:, 0] = (c['dpred'] * c['rowmask'][:, :, None]).reshape(-1)
        g = {n: self._buf(n + '/dy', saved[n]['y'].shape) for n in saved if saved[n]['y'] is not None}   # d loss / d activated output (decoder)
        gpool = {n: self._buf(n + '/dpool', c['pool'][n].shape) for n in c['pool']}  # d loss / d pooled encoder outputs
        self._conv_bwd('out', dlog, gp, g['d6'], False, None, False)
        prev = None
        for i in range(5, -1, -1):                                      # d6 .. d1
            name = 'd%d' % (i + 1)
            skip_name = 'e%d' % (5 - i) if 5 - i >= 1 else None         # skip source: e5..e1, then the net input
            dsrc0 = gpool[skip_name] if skip_name else None
            coarse = g['d%d' % i] if i >= 1 else gpool['e6']             # the up-sampled source: d(i) act, or e6 pooled
            self._conv_bwd(name, g[name], gp, dsrc0, False, coarse, False)
        for i in range(5, -1, -1):                                      # e6 .. e1
            name = 'e%d' % (i + 1)
            dsrc0 = gpool['e%d' % i] if i >= 1 else None                 # e(i) pooled output also fed a decoder: accumulate
            self._conv_bwd(name, gpool[name], gp, dsrc0, True, None, False, pooled=True)
        grads = self.variables.unpack_grads(gp, out=self._buf('grads', (lay.ref_size,)))
        c['grads'] = grads
        return grads

    @property
    def gradients(self):
        g = self._backward()
        return g + self.regularization * self.variables.flat if self.regularization else g

    @property
    def step_guard(self):
        """See StackedBLSTMModel.step_guard (this model launches no cooperative kernel: word 1 is always 0)."""
        c = self._cache
        if c.get('guard') is not None:
            return c['guard']
        self._loss()
        return ops.step_guard(c['loss3'][0:1], torch.empty(2, device=self.device), coop=False)

    @property
    def nonfinite_flag(self):
        return self.step_guard[0:1]

    @property
    def global_step(self):
        return self.variables.global_step

    @property
    def learning_rate(self):
        return self.starter_learning_rate * self.learning_decay ** math.floor(
            self.variables.global_step / self.updating_step)

    @property
    def train_op(self):
        c = self._cache
        if c.get('trained'):
            return None
        g = self._backward()
        v = self.variables
        world = parallel.world_size()
        if parallel.dp_active():
            # one flat all-reduce; the two words behind the gradients are this rank's step guard ("my loss is not
            # finite": NaN survives the sum): see StackedBLSTMModel.step_guard
            gf = self._buf('grads+guard', (self.layout.ref_size + 2,))
            gf[:-2].copy_(g)
            ops.step_guard(c['loss3'][0:1], gf[-2:], coop=False)
            parallel.all_reduce_sum_(gf)
            g = gf[:-2]
            c['guard'] = gf[-2:].clone()
        else:
            c['guard'] = ops.step_guard(c['loss3'][0:1], torch.empty(2, device=self.device), coop=False)
        step = v.global_step + 1
        if self.optimizer_choice != 'adam':
            print('Optimizer must be adam on the MI355X U-Net path. Closing...')
            sys.exit(1)
        if v.adam_m is None:
            v.adam_m, v.adam_v = torch.zeros_like(v.flat), torch.zeros_like(v.flat)
        ops.adam_tf(v.flat, g, v.adam_m, v.adam_v, step, self.starter_learning_rate, grad_scale=1.0 / world,
                    l2=float(self.regularization or 0.0), skip=c['guard'])
        v.global_step = step
        v.repack()
        c['trained'] = True
        return None

    # ------------------------------------------------------------------ waveforms (models.py:664-680)
    @property
    def target_stft(self):
        c = self._cache
        if 'target_stft' not in c:
            _, T, F = self._dims()
            c['target_stft'] = ap.frontend(self.target_sources, window_size=16, step_size=8, n_fft=256,
                                           num_frames_out=T, num_bins=F, want_stft=True)['stft']
        return c['target_stft']

    def _enhanced(self, oracle_phase):
        key = 'enh_oracle' if oracle_phase else 'enh'
        c = self._cache
        if key not in c:
            # the reference calls get_sources with its 24 / 12 ms defaults here (App. B9); the STFT geometry of
            # this model (16 / 8 ms, n_fft 256) is used instead
            mask = None if oracle_phase else self._cache['mask_t']
            if 'target_stft' in c or os.environ.get('AVSI_ISTFT_FROM_WAV', '1') == '0':
                c[key] = ap.enhanced_from_prediction(self.prediction, self.audio_feat_mean, self.audio_feat_std,
                                                     self.target_stft, mask, num_samples=self.audio_len, window_size=16,
                                                     step_size=8, n_fft=256)
            else:       # phase from the target waveform inside the kernel (see StackedBLSTMModel._enhanced)
                c[key] = ap.enhanced_from_prediction_wav(self.prediction, self.audio_feat_mean, self.audio_feat_std,
                                                         self.target_sources, mask, num_samples=self.audio_len,
                                                         window_size=16, step_size=8, n_fft=256)
        return c[key]

    @property
    def enhanced_sources(self):
        return self._enhanced(False)

    @property
    def enhanced_sources_oracle_phase(self):
        return self._enhanced(True)
