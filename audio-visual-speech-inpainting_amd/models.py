"""Speech-inpainting models on MI355X (host side).

``StackedBLSTMModel`` mirrors the reference class of the same name
(``av_speech_inpainting/models.py:11-237``): same constructor arguments, same attribute names
(``inference, prediction, loss, loss_func, loss_hole, loss_valid, train_op, enhanced_sources,
target_spec_norm, global_step, learning_rate ...``).  The reference builds a TF graph over
placeholders and evaluates it with ``sess.run(fetches, feed_dict)``; here the constructor
arguments are the fed values themselves (torch device tensors or numpy arrays), ``feed()``
replaces them for the next batch, and reading an attribute runs the gfx950 kernels needed for
it (results are cached until the next ``feed()``).

Device data layout: activations are TIME-MAJOR ``[T][Bp][C]`` with the batch
padded to a multiple of 32 (whole MFMA row tiles) and channels padded (257 -> 272 inputs,
250 -> 256 hidden units per direction); padded hidden units are exactly zero by construction.
"""
import math
import os
import sys

import numpy as np
import torch

from . import _lib, ops, parallel
from . import audio_processing as ap
from .blstm_layout import GP, HP, ParamLayout, round_up

SIDE_DELAY_US = int(os.environ.get('AVSI_SIDE_DELAY_US', '60'))   # head start of a BPTT grid over the side-stream GEMMs
DX_SPLITS = int(os.environ.get('AVSI_DX_SPLITS', '2'))      # reduction slabs of the dX product at small batches (1: unsplit)
_PROJ_SPLIT = os.environ.get('AVSI_PROJ_SPLIT', '1') != '0'    # 257-bin projection as 256 bins + 1 bin (see _forward)


def _as_device(x, dtype=torch.float32, device=None):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.to(device=device or 'cuda', dtype=dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(device or 'cuda')


class BLSTMVariables:
    """The trainable variables of one model + optimiser slots, on one GPU.

    ``flat`` is the reference-layout buffer (what checkpoints / Adam / all-reduce see);
    ``packed`` is the kernels' view, refreshed by ``repack()`` after every update.  Works for any
    layout object with ref_size / ref_entries / pack_index / grad_index / signature()."""

    def __init__(self, layout, device='cuda', seed=0, init=None):
        _lib.require_cuda()
        self.layout = layout
        self.device = torch.device(device)
        # the initial values are drawn when something first READS the variables: a model that is restored from a checkpoint
        # right after its construction (infer(), a resumed train()) never draws them -- 14 ms of host time for 4.4 M values
        # and two gathers, a tenth of a 4096-utterance infer() call
        self._init, self._flat, self._packed = (init or self._tf_default_init, seed), None, None
        self._pack_index = torch.from_numpy(layout.pack_index).to(self.device)
        self._grad_index = torch.from_numpy(layout.grad_index).to(self.device)
        self.adam_m = None
        self.adam_v = None
        self.global_step = 0
        self.version = 0

    @property
    def flat(self):
        if self._flat is None:
            self._flat = torch.from_numpy(self._init[0](self.layout, self._init[1])).to(self.device)
        return self._flat

    @flat.setter
    def flat(self, value):
        self._flat = value

    @property
    def packed(self):
        if self._packed is None:
            self.repack()
        return self._packed

    @staticmethod
    def _tf_default_init(layout, seed):
        """TF default initialisers: LSTM kernels glorot-uniform, biases
        zero, projection truncated-normal(stddev 1/sqrt(2H)) -- reference models.py:107,119-121."""
        rng = np.random.default_rng(seed)
        flat = np.zeros(layout.ref_size, dtype=np.float32)
        for name, shape, off in layout.ref_entries:
            n = int(np.prod(shape))
            if name.endswith('/kernel'):
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                flat[off:off + n] = rng.uniform(-lim, lim, size=n)
            elif name in ('logits/weights', 'asr/weights') or name.startswith('speaker_embedding/weights_'):
                # logits: stddev 1/sqrt(2H) (models.py:119); speaker-embedding MLP: 1/sqrt(F) for
                # weights_1 [2F, W], 1/sqrt(W) for the square ones (models.py:804-808)
                sd = 1.0 / math.sqrt(float(shape[0] // 2 if name.endswith('weights_1') else shape[0]))
                w = rng.normal(0.0, sd, size=n)
                bad = np.abs(w) > 2 * sd
                while bad.any():
                    w[bad] = rng.normal(0.0, sd, size=int(bad.sum()))
                    bad = np.abs(w) > 2 * sd
                flat[off:off + n] = w
        return flat

    def repack(self):
        """packed <- gather(flat): one index_select (padding positions read the appended 0)."""
        self.version = getattr(self, 'version', 0) + 1
        if self._packed is None:
            self._packed = torch.empty(self.layout.packed_size, dtype=torch.float32, device=self.device)
        ext = torch.cat([self.flat, self.flat.new_zeros(1)])
        torch.index_select(ext, 0, self._pack_index, out=self._packed)

    def load_flat(self, flat):
        flat = torch.as_tensor(np.asarray(flat, dtype=np.float32))
        if flat.numel() != self.layout.ref_size:
            raise ValueError("expected %d parameters, got %d" % (self.layout.ref_size, flat.numel()))
        if self._flat is None:
            self._flat = flat.to(self.device)           # (nothing has read the initial values: they are never drawn)
        else:
            self._flat.copy_(flat.to(self.device))
        self.repack()

    def p(self, name):
        return self.layout.packed_view(self.packed, name)

    # ---- checkpoints: the role of tf.train.Saver over all global variables (training.py:114)
    def save(self, path):
        """Write <path>.npz: every variable under its reference name + optimiser slots + global_step."""
        flat = self.flat.cpu().numpy()
        arrays = {n: self.layout.ref_view(flat, n) for n, _, _ in self.layout.ref_entries}
        arrays['global_step'] = np.int64(self.global_step)
        arrays['__layout__'] = np.array(self.layout.signature())
        if self.adam_m is not None:
            arrays['__adam_m__'] = self.adam_m.cpu().numpy()
        if self.adam_v is not None:
            arrays['__adam_v__'] = self.adam_v.cpu().numpy()
        np.savez(path + '.npz', **arrays)
        return path + '.npz'

    def save_tf(self, prefix, scope):
        """Write a TensorFlow tensor-bundle checkpoint (``prefix.index`` / ``.data-00000-of-00001``)
        with the reference's variable names under variable scope ``scope`` (= config['model'],
        training.py:82) -- what tf.train.Saver(model.all_vars).save writes (training.py:334-340)."""
        from . import tf_checkpoint
        flat = self.flat.cpu().numpy()
        m = self.adam_m.cpu().numpy() if self.adam_m is not None else None
        v = self.adam_v.cpu().numpy() if self.adam_v is not None else None
        return tf_checkpoint.write_bundle(prefix, tf_checkpoint.export_variables(
            self.layout, flat, scope, m, v, self.global_step))

    def restore_tf(self, prefix):
        """Load a TensorFlow tensor-bundle checkpoint written by the reference (or by save_tf)."""
        from . import tf_checkpoint
        flat, m, v, step = tf_checkpoint.import_variables(tf_checkpoint.read_bundle(prefix), self.layout)
        self.load_flat(flat)
        self.global_step = step
        self.adam_m = torch.from_numpy(m).to(self.device) if m is not None else None
        self.adam_v = torch.from_numpy(v).to(self.device) if v is not None else None

    def restore(self, path):
        """Load a checkpoint: ``<path>.npz`` written by save(), or -- when ``<path>.index`` exists --
        a TensorFlow tensor bundle.  ValueError if it is neither (reference: exit(2))."""
        from . import tf_checkpoint
        if not path.endswith('.npz') and not os.path.isfile(path + '.npz') and tf_checkpoint.is_bundle(path):
            return self.restore_tf(path)
        fname = path if path.endswith('.npz') else path + '.npz'
        try:
            ck = np.load(fname)
        except Exception as e:
            raise ValueError("%s is not a valid checkpoint (%s)" % (path, e))
        want = list(self.layout.signature())
        if '__layout__' not in ck or ck['__layout__'].tolist() != want:
            raise ValueError("%s was saved for a different model shape" % path)
        flat = np.zeros(self.layout.ref_size, dtype=np.float32)
        for n, _, _ in self.layout.ref_entries:
            self.layout.ref_view(flat, n)[...] = ck[n]
        self.load_flat(flat)
        self.global_step = int(ck['global_step'])
        self.adam_m = torch.from_numpy(ck['__adam_m__']).to(self.device) if '__adam_m__' in ck else None
        self.adam_v = torch.from_numpy(ck['__adam_v__']).to(self.device) if '__adam_v__' in ck else None

    def rewind_step(self):
        """Take back the count of ONE optimiser step that the device-side step guard voided (the variables and slots
        were left alone by it; the host counted it before the verdict arrived)."""
        self.global_step -= 1

    def unpack_grads(self, gpacked, out=None):
        """reference-layout gradient <- gather(gradient buffer written by the backward kernels)."""
        return torch.index_select(gpacked, 0, self._grad_index, out=out)


class StackedBLSTMModel(object):
    """
    Speech inpainting BLSTM model (reference models.py:11-237).
    Input: log-compressed linear spectrogram of corrupted audio (and/or face-landmark motion).
    Model: stacked BLSTM.  Output: log-compressed linear spectrogram of restored audio.
    Loss: L1 (target_spectrogram - reconstructed_spectrogram).
    """

    def __init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                 audio_features=None, video_features=None, input='a', is_training=True, variables=None, seed=0,
                 side=None, blend=False):
        """``side`` / ``blend`` are the hooks the model variants of this module use (not reference
        arguments): ``side = (layer, dim)`` adds a per-utterance side input (speaker embedding,
        see ParamLayout) delivered by ``_side_input()``; ``blend`` selects the variants'
        prediction (known bins restored) and loss (loss_hole)."""
        _lib.require_cuda()
        self.blend = bool(blend)
        self.audio_feat_dim = config['audio_feat_dim']
        self.audio_len = config['audio_len']
        self.video_feat_dim = config.get('video_feat_dim', 136)
        self.input_type = input
        if input not in ('a', 'v', 'av'):
            print('Model input must be "a", "v" or "av". Closing...')
            sys.exit(1)
        # tf.nn.dropout on the last layer's output (models.py:117): the fed value, replaceable per feed() like the
        # reference's dropout_rate placeholder (training feeds config['dropout_rate'], validation 0.0)
        self.dropout_rate = float(dropout_rate or 0.0)
        if not 0.0 <= self.dropout_rate < 1.0:
            raise _lib.AvsiError("dropout_rate must be in [0, 1)")
        self._dropout_seed = (int(seed) + 1) * 0x9E3779B1
        self._dropout_calls = 0
        self.net_dim = config['net_dim']
        self.num_layers = len(self.net_dim)
        self.optimizer_choice = config['optimizer_type']
        self.starter_learning_rate = config['starter_learning_rate']
        self.updating_step = config['lr_updating_steps']
        self.learning_decay = config['lr_decay']
        self.is_training = is_training
        self.batch_size = config.get('batch_size', 1)
        self.regularization = config['l2']
        self.var_scope = None
        self.rows_per_wg = int(config.get('rows_per_wg', 0))   # 0 = kernel picks 32/64 from the batch
        # 'f32' (default: exact fp32 fma chains on the fp32 matrix cores) or 'bf16x3' (EXPLORATORY: the layer input
        # projections with split-bf16 operands, ops.gemm_bf16x3; inference only)
        self.precision = config.get('precision', 'f32')
        if self.precision not in ('f32', 'bf16x3'):
            raise _lib.AvsiError("precision must be 'f32' or 'bf16x3'")
        in_dim = {'a': self.audio_feat_dim, 'v': self.video_feat_dim,
                  'av': self.audio_feat_dim + self.video_feat_dim}[input]
        self.layout = variables.layout if variables is not None else ParamLayout(in_dim, self.net_dim,
                                                                                 self.audio_feat_dim, side=side)
        if side is not None and self.layout.side != (int(side[0]), int(side[1])):
            raise ValueError("variables were built for side input %r, model needs %r" % (self.layout.side, tuple(side)))
        if self.layout.input_dim != in_dim:
            raise ValueError("variables were built for input dim %d, model needs %d" % (self.layout.input_dim, in_dim))
        self.variables = variables if variables is not None else BLSTMVariables(self.layout, seed=seed)
        self.device = self.variables.device
        self._ws = {}
        self.audio_feat_mean = _as_device(audio_feat_mean, device=self.device)
        self.audio_feat_std = _as_device(audio_feat_std, device=self.device)
        self._cache = {}
        self.sequence_lengths = None
        self.feed(sequence_lengths=sequence_lengths, target_sources=target_sources, masks=masks,
                  audio_features=audio_features, video_features=video_features)

    # ---------------------------------------------------------------- feed boundary
    def feed(self, sequence_lengths=None, target_sources=None, masks=None, video_features=None,
             audio_features=None, audio_feat_mean=None, audio_feat_std=None, dropout_rate=None):
        """Replace the fed values (the reference's feed_dict, training.py:67-74) and drop cached results."""
        self._cache = {}
        if dropout_rate is not None:
            if not 0.0 <= float(dropout_rate) < 1.0:
                raise _lib.AvsiError("dropout_rate must be in [0, 1)")
            self.dropout_rate = float(dropout_rate)
        if sequence_lengths is not None:
            new_len = np.asarray(sequence_lengths.cpu() if isinstance(sequence_lengths, torch.Tensor) else sequence_lengths,
                                 dtype=np.int64)
            # on the device once per feed: a host-to-device copy issued later, behind the recurrent kernels
            # of the stream, would block the host until they finish (pageable memory) -- and not at all when the
            # lengths are the ones already there (every batch of a fixed-length dataset): that copy is the one
            # point where the host would otherwise wait for the previous step's kernels
            if self.sequence_lengths is None or not np.array_equal(new_len, self.sequence_lengths) \
                    or getattr(self, '_seq_dev', None) is None:
                self.sequence_lengths = new_len
                self._seq_dev = torch.as_tensor(self.sequence_lengths, device=self.device)
        self.target_sources = _as_device(target_sources, device=self.device)
        self.masks = _as_device(masks, device=self.device)
        self.video_features = _as_device(video_features, device=self.device)
        self.fed_audio_features = _as_device(audio_features, device=self.device)
        if audio_feat_mean is not None:
            self.audio_feat_mean = _as_device(audio_feat_mean, device=self.device)
        if audio_feat_std is not None:
            self.audio_feat_std = _as_device(audio_feat_std, device=self.device)

    def set_dropout_rate(self, rate):
        """The fed value of the reference's dropout_rate placeholder for the following steps (training_emb.py:251
        feeds config['dropout_rate'] when training, :314,326 feed 0.0 when validating)."""
        if not 0.0 <= float(rate) < 1.0:
            raise _lib.AvsiError("dropout_rate must be in [0, 1)")
        self.dropout_rate = float(rate)

    def build_graph(self, var_scope=''):
        """Kept for API parity (reference models.py:74-87): variables already exist."""
        self.var_scope = var_scope

    # ---------------------------------------------------------------- workspaces
    def _buf(self, name, shape, zero=False):
        key = (name, tuple(shape))
        t = self._ws.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=self.device)
            self._ws[key] = t
        return t

    def _dims(self):
        B = int(self.target_sources.shape[0]) if self.target_sources is not None else int(self.video_features.shape[0])
        T = int(self.sequence_lengths.max())
        # whole 32-row MFMA tiles; beyond 4096 utterances whole 64-row ones, so that the 64-row recurrent kernel is
        # taken (an odd number of 32-row tiles there would send the layer through two rounds of the 32-row kernel)
        return B, T, round_up(B, 64 if B > 4096 else 32)

    # ---------------------------------------------------------------- front end (models.py:30-45)
    def _frontend(self):
        c = self._cache
        if 'x0' in c:
            return
        B, T, Bp = self._dims()
        Kp = self.layout.kp[0]
        F = self.audio_feat_dim
        # zero-initialised once: padded batch rows / padded columns stay zero across calls
        x0 = self._buf('x0', (T, Bp, Kp), zero=True)
        if self.input_type in ('a', 'av') or self.target_sources is not None:
            want_feat = self.input_type != 'v' and self.fed_audio_features is None
            # AVSI_LOSS_FROM_WAV=1: a plain model whose loss recomputes the target from the waveform (`_loss_from_wav`) does not
            # store it here -- the front end then writes the masked features only (706 kB per utterance: exactly its
            # algorithmic bytes); `target_spec_norm` is produced on demand (property below)
            want_spec = not (want_feat and self._loss_from_wav())
            fe = ap.frontend(self.target_sources, window_size=24, step_size=12, n_fft=512, num_frames_out=T,
                             num_bins=F, mean=self.audio_feat_mean, std=self.audio_feat_std, masks=self.masks,
                             want_spec=want_spec, want_feat=want_feat,
                             time_major=True, feat_cols=Kp if self.input_type == 'a' else F,
                             _feat_out=x0 if self.input_type != 'v' else None)
            if want_spec:
                c['target_spec_norm'] = fe['spec']
        def place(src, col, width):
            """x0[t, b, col : col + width] = src[b, t, :width]: batch-major fed features into the time-major padded input
            (avsi_relayout_rows_f32 -- the one device step of the AV models that used to be a torch strided copy)."""
            src = src.to(torch.float32)
            if src.stride(2) != 1:
                src = src.contiguous()
            ops.relayout_rows(src, x0.view(-1)[col:], B, T, width, width, (src.stride(0), src.stride(1)), (Kp, Bp * Kp))
        if self.input_type != 'v' and self.fed_audio_features is not None:
            place(self.fed_audio_features, 0, F)
        if self.input_type == 'v':
            place(self.video_features, 0, self.video_feat_dim)
        elif self.input_type == 'av':
            place(self.video_features, F, self.video_feat_dim)
        c['x0'] = x0

    def _loss_from_wav(self):
        """AVSI_LOSS_FROM_WAV=1 (opt-in): the step's L1 loss is taken by the front-end kernel itself from the target WAVEFORM
        (ap.frontend_l1_loss) and the normalised target is never stored -- 257 kB per utterance less written and read back
        (2.1 GB less resident at 8192 utterances), the front end moves exactly its algorithmic 706 kB.  Not the default: the
        second transform costs more than the re-read saves (front end 1.68 -> 1.54 ms, loss 1.02 -> 1.56 ms at 8192 utterances;
        DESIGN.md 4.1).  Plain models only (the blend variants need the stored target for their prediction), 257 bins."""
        return (not self.blend and self.target_sources is not None and self.audio_feat_dim == 257
                and self.masks is not None and os.environ.get('AVSI_LOSS_FROM_WAV', '0') == '1')

    @property
    def target_spec_norm(self):
        self._frontend()
        c = self._cache
        if 'target_spec_norm' not in c:         # not stored by the step's front-end call (see _frontend): on demand
            _, T, _ = self._dims()
            c['target_spec_norm'] = ap.frontend(self.target_sources, window_size=24, step_size=12, n_fft=512, num_frames_out=T,
                                                num_bins=self.audio_feat_dim, mean=self.audio_feat_mean,
                                                std=self.audio_feat_std, want_spec=True)['spec']
        return c['target_spec_norm']

    @property
    def net_inputs(self):
        """[B, T, D] view of the (time-major, padded) network input."""
        self._frontend()
        B, T, _ = self._dims()
        return self._cache['x0'][:, :B, :self.layout.input_dim].transpose(0, 1)

    # ---------------------------------------------------------------- BLSTM + projection (models.py:89-138)
    def _forward(self, keep=False):
        # a training model keeps the BPTT reserve on its first forward, so fetching the loss and
        # then train_op (the reference's [loss, ..., train_op] fetch list) runs the network once
        keep = keep or bool(self.is_training)
        c = self._cache
        if 'pred' in c and (not keep or c.get('kept')):
            return
        self._frontend()
        B, T, Bp = self._dims()
        v = self.variables
        x = c['x0']
        xproj = self._buf('xproj', (T, Bp, 2 * GP))
        c['layer_in'] = []
        c['reserve'] = []
        # zero padding inside the reductions (weight rows that are zero by construction, ParamLayout): layer 0 behind its
        # real inputs, the other layers and the projection behind the H real units of each direction
        H = self.layout.H
        kz_hidden = ((H, HP), (HP + H, 2 * HP))            # behind the top layer's units (the projection's input)
        for li in range(self.num_layers):
            kp = self.layout.kp[li]
            Hin = self.layout.Hs[li - 1] if li else 0      # units per direction of the layer below
            kz = ((self.layout.input_dim, kp),) if li == 0 else ((Hin, HP), (HP + Hin, 2 * HP))
            if self.layout.side_dim(li):
                # tile(side) . W_side is the same row for every frame of an utterance: one small GEMM
                # [Bp, E] . [E, 2048] (+ bias), broadcast over time, and the layer GEMM accumulates
                # onto it -- the tiled [B, T, E] tensor of the reference is never formed
                E = self.layout.side_dim(li)
                sp = self._buf('side_p', (Bp, self.layout.side_p), zero=True)
                sp[:B, :E] = self._side_input()
                eb = self._buf('side_bias', (Bp, 2 * GP))
                ops.gemm(sp, v.p('we'), out=eb, bias=v.p('b%d' % li))
                xproj.copy_(eb.unsqueeze(0).expand(T, Bp, 2 * GP))
                ops.gemm(x.view(T * Bp, kp), v.p('wx%d' % li), out=xproj.view(T * Bp, 2 * GP), beta=1.0, k_zero=kz)
                c['side_p'] = sp
            elif self.precision == 'bf16x3' and not keep:
                # the split weights are packed once per set of variables (repack() bumps the version)
                key = ('wx_bf16x3', li)
                hit = self._ws.get(key)
                if hit is None or hit[0] != v.version:
                    hit = self._ws[key] = (v.version, ops.pack_bf16x3_b(v.p('wx%d' % li)))
                ops.gemm_bf16x3(x.view(T * Bp, kp), hit[1], xproj.view(T * Bp, 2 * GP), kp, bias=v.p('b%d' % li))
            else:
                ops.gemm(x.view(T * Bp, kp), v.p('wx%d' % li), out=xproj.view(T * Bp, 2 * GP), bias=v.p('b%d' % li), k_zero=kz)
            hout = self._buf('h%d' % li, (T, Bp, 2 * HP))
            resv = self._buf('resv%d' % li, (T, Bp, 2, 5, HP)) if keep else None
            ops.blstm_rec_fwd(xproj, v.p('wh%d' % li), hout, resv, self.rows_per_wg)
            c['layer_in'].append(x)
            c['reserve'].append(resv)
            if keep and self.layout.ones_col[li] >= 0:
                # the constant-1 input column of this layer (ParamLayout.ones_col): zero weights forward, bias
                # gradient inside the dWx GEMM backward.  hout columns are written by the recurrent kernel, so the
                # fill comes after it (nothing reads the column before the backward pass)
                x[:, :B, self.layout.ones_col[li]] = 1.0
            x = hout
        if self.dropout_rate > 0.0:
            # rnn_outputs_res = tf.nn.dropout(rnn_outputs, rate) (models.py:117): the projection (and its weight
            # gradient) sees the dropped activations, the BPTT below receives dh * scale; a fresh draw per forward pass
            xd = self._buf('rnn_drop', (T, Bp, 2 * HP), zero=True)
            sc = self._buf('drop_scale', (T, Bp, 2 * HP), zero=True)
            self._dropout_calls += 1
            ops.dropout(x.view(T * Bp, 2 * HP), xd.view(T * Bp, 2 * HP), sc.view(T * Bp, 2 * HP), 2 * HP, self.dropout_rate,
                        self._dropout_seed + self._dropout_calls * 0x632BE5AB)
            c['drop_scale'] = sc
            x = xd
        if keep and self.layout.ones_col_top >= 0:
            x[:, :B, self.layout.ones_col_top] = 1.0
        c['rnn_out'] = x
        if not self.rows_per_wg and any(sp for _, _, sp in ops.rec_fwd_parts(Bp)):
            ops.coop_poll(self.device)       # small batches: a bounded wait that gave up surfaces here, without a sync
        # prediction = sequence_mask * (rnn_out . W + b), stored batch-major [B, T, F]
        seq = self._seq_dev
        row_scale = self._buf('row_scale', (T, Bp), zero=True)
        if self._ws.get(('row_scale_of', T, Bp, B)) is not seq:      # same lengths as last time (feed keeps the tensor): five small
            row_scale[:, :B] = (torch.arange(T, device=self.device)[:, None] < seq[None, :]).to(torch.float32)   # kernels saved
            self._ws[('row_scale_of', T, Bp, B)] = seq
        pred = torch.empty((B, T, self.audio_feat_dim), dtype=torch.float32, device=self.device)
        F = self.audio_feat_dim
        x2, p2 = x.view(T * Bp, 2 * HP), pred.view(B * T, F)
        # 257 bins over many rows = ONE launch on the wide (128 x 256) tile whose workgroups also take the 257th bin, as a dot
        # product on the VALU over the rows they have staged anyway (round 6, avsi_gemm_f32 does this by itself for N = 257;
        # AVSI_GEMM_FOLD_TAIL=0: the round-3 form -- 256 bins on the wide tile + the last bin in a second launch on a narrow
        # one, which reads all of A again; AVSI_PROJ_SPLIT=0 on top: five 64-column tiles)
        fold = F == 257 and T * Bp >= 65536 and os.environ.get('AVSI_GEMM_FOLD_TAIL', '1') != '0'
        n_main = F - F % 256 if (_PROJ_SPLIT and not fold and F > 256 and 0 < F % 256 <= 32 and T * Bp >= 65536) else F
        ops.gemm(x2, v.p('pw'), out=p2, n=n_main, bias=v.p('pb'), row_scale=row_scale.view(-1), row_map=(Bp, T, B),
                 k_zero=kz_hidden)
        if n_main < F:
            ops.gemm(x2, v.p('pw')[:, n_main:F], out=p2[:, n_main:], n=F - n_main, bias=v.p('pb')[n_main:F],
                     row_scale=row_scale.view(-1), row_map=(Bp, T, B), k_zero=kz_hidden)
        c['row_scale'] = row_scale
        c['pred'] = pred
        c['kept'] = keep
        if self.layout.asr:
            # second head (models.py:1910-1916): the asr columns of the same packed matrix, no sequence mask
            lay = self.layout
            asr = torch.empty((B, T, lay.asr), dtype=torch.float32, device=self.device)
            ops.gemm(x.view(T * Bp, 2 * HP), v.p('pw')[:, lay.asr_col:lay.asr_col + lay.asr], out=asr.view(B * T, lay.asr),
                     n=lay.asr, bias=v.p('pb')[lay.asr_col:lay.asr_col + lay.asr], row_map=(Bp, T, B), k_zero=kz_hidden)
            c['asr_logits'] = asr
        if self.blend:
            # prediction = seq_mask * (target * mask + logits * (1 - mask)); loss_func = loss_hole
            tgt = self.target_spec_norm
            mask = self.masks[:, :T].contiguous()
            rs_bm = row_scale[:, :B].t().contiguous()
            out3, dlog, inv_gap = ops.l1_loss_blend(tgt, pred, mask, rs_bm, want_grad=keep)
            c['loss3'] = torch.stack([out3[1], out3[1], out3[2]])
            if keep and parallel.dp_active():
                # Data parallel: the objective is the ratio of GLOBAL sums, sum_r num_r / sum_r gap_r, and its
                # gradient sum_r d num_r / G.  The kernel left d num_r / gap_r; with the ranks' gap element counts summed (one
                # float, on the device) each rank rescales by gap_r * world / G, and the 1 / world of the fused Adam over the
                # summed buckets gives exactly the single-process gradient of the global batch -- whatever the ranks' shares
                # of the gap frames (tests/test_dp_gpu.py::test_two_ranks_with_unequal_gaps_train_the_global_loss_hole),
                # a share of ZERO included: that rank's kernel left inv = inf and a zero gradient (loss.hip), its
                # gap_r and num_r are 0 here, and its LOCAL loss_hole (0 / 0) is reported as 0 so that the step guard's
                # 'my loss is not finite' word does not void a step whose global objective is well defined
                has_gap = torch.isfinite(inv_gap)
                gap = torch.where(has_gap, 1.0 / inv_gap, torch.zeros_like(inv_gap))
                total = parallel.all_reduce_sum_(gap.clone())
                dlog.mul_(gap * float(parallel.world_size()) / total)
                num = torch.where(has_gap[0], out3[1] * gap[0], torch.zeros_like(out3[1]))
                # (no gap element on ANY rank: 0 / 0 as in the reference -- the NaN stays and the trainer leaves)
                c['loss3'] = torch.where(has_gap[0] | (total[0] == 0), c['loss3'], torch.stack([num, num, out3[2]]))
                c['hole_sums'] = (num, gap, total)          # num_r, gap_r, G: loss_hole_global
            c['dpred'] = dlog
            self._extra_loss(keep)

    def _extra_loss(self, want_grad):
        """Hook of the multi-task variants: add further loss terms to ``_cache['loss3'][0]`` and leave
        the gradient of extra heads' logits in ``_cache['dasr']`` [B, T, classes]."""

    # ---------------------------------------------------------------- side input hooks (variants)
    def _side_input(self):
        """[B, side_dim] device tensor concatenated (tiled over time) to the side layer's input."""
        raise NotImplementedError("this model declares a side input but does not provide it")

    def _side_backward(self, dside, gp):
        """Receives d loss / d side input [B, side_dim] and the packed gradient buffer; models whose
        side input is trainable override and add their gradients to ``gp``."""

    @property
    def inference(self):
        """Logits before the sequence mask (models.py:89-125).  Equal to `prediction` on frames
        inside each utterance; the reference's un-masked tail frames are not materialised."""
        return self.prediction

    @property
    def prediction(self):
        self._forward()
        return self._cache['pred']

    @property
    def rnn_outputs(self):
        """[B, T, 2H] view of the last BLSTM layer output (padding stripped)."""
        self._forward()
        B, T, _ = self._dims()
        H = self.layout.H
        h = self._cache['rnn_out'][:, :B]
        return torch.cat([h[:, :, :H], h[:, :, HP:HP + H]], dim=2).transpose(0, 1)

    # ---------------------------------------------------------------- loss (models.py:140-159)
    def _loss(self, want_grad=False):
        want_grad = want_grad or bool(self.is_training)
        c = self._cache
        if 'loss3' in c and (not want_grad or c.get('dpred') is not None):
            return
        self._forward(keep=want_grad)
        if self.blend:
            return      # computed with the prediction
        if 'target_spec_norm' not in c and self._loss_from_wav():
            # the target recomputed from the waveform inside the front-end kernel: nothing was written for it, nothing is read
            got = ap.frontend_l1_loss(self.target_sources, c['pred'], self.masks, self.audio_feat_mean, self.audio_feat_std,
                                      want_grad=want_grad)
            if got is not None:
                c['loss3'], c['dpred'] = got
                return
        tgt = self.target_spec_norm
        mask = self.masks[:, :tgt.shape[1]].contiguous()
        out3, dpred = ops.l1_loss(tgt, c['pred'], mask, want_grad=want_grad)
        c['loss3'] = out3
        c['dpred'] = dpred

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
    def loss_hole_global(self):
        """loss_hole of the GLOBAL batch of a data-parallel step: sum over ranks of sum|err|(1 - m) over sum over ranks of
        sum(1 - m), numerator and denominator all-reduced separately.  Equal to loss_hole in a single process."""
        self._loss()
        c = self._cache
        if not parallel.dp_active():
            return c['loss3'][1]
        if 'hole_sums' in c:
            num, gap, total = c['hole_sums']
            return parallel.all_reduce_sum_(num.clone().reshape(1))[0] / total.reshape(-1)[0]
        gap = (1.0 - self.masks[:, :self._dims()[1]]).sum().reshape(1)
        both = parallel.all_reduce_sum_(torch.cat([c['loss3'][1:2] * gap, gap]))
        return both[0] / both[1]

    @property
    def reg_loss(self):
        if self.regularization:
            return (self.variables.flat.double() ** 2).sum().float() / 2.0
        return torch.zeros((), device=self.device)

    @property
    def loss(self):
        if self.regularization:
            return self.loss_func + self.regularization * self.reg_loss
        return self.loss_func

    # ---------------------------------------------------------------- waveforms (models.py:181-197)
    @property
    def target_stft(self):
        c = self._cache
        if 'target_stft' not in c:
            _, T, _ = self._dims()
            c['target_stft'] = ap.frontend(self.target_sources, window_size=24, step_size=12, n_fft=512,
                                           num_frames_out=T, num_bins=self.audio_feat_dim, want_stft=True)['stft']
        return c['target_stft']

    def _enhanced(self, oracle_phase):
        key = 'enh_oracle' if oracle_phase else 'enh'
        c = self._cache
        if key not in c:
            pred = self.prediction
            masks = None if oracle_phase else self.masks[:, :pred.shape[1]]
            if 'target_stft' in c or os.environ.get('AVSI_ISTFT_FROM_WAV', '1') == '0':
                c[key] = ap.enhanced_from_prediction(pred, self.audio_feat_mean, self.audio_feat_std, self.target_stft,
                                                     masks, num_samples=self.audio_len)
            else:
                # the phase of the target comes from its WAVEFORM, transformed tile by tile inside the inverse-STFT kernel:
                # the complex spectrogram is neither written by the front end nor read here
                c[key] = ap.enhanced_from_prediction_wav(pred, self.audio_feat_mean, self.audio_feat_std, self.target_sources,
                                                         masks, num_samples=self.audio_len)
        return c[key]

    @property
    def enhanced_sources(self):
        """exp(prediction*std+mean) with the phase of the masked target STFT, inverse STFT."""
        return self._enhanced(False)

    @property
    def enhanced_sources_oracle_phase(self):
        return self._enhanced(True)

    # ---------------------------------------------------------------- gradients (tf.gradients of models.py:178)
    def _backward(self):
        """d loss / d variables in the reference layout (cached).  Chain: L1 -> projection ->
        per layer (top down): BPTT kernel -> dWx, dWh, db as split-K GEMMs / column sums -> dX."""
        c = self._cache
        if 'grads' in c:
            return c['grads']
        self._loss(want_grad=True)
        B, T, Bp = self._dims()
        v, lay = self.variables, self.layout
        F, ldp = self.audio_feat_dim, self.layout.ldp
        M = T * Bp
        # two slots behind the gradients carry this rank's STEP GUARD through the LAST all-reduce bucket: "my loss is not
        # finite" (NaN, and NaN survives the sum over ranks) and "a cooperative recurrent launch of mine timed out" (1;
        # the sum counts the ranks).  The fused Adam reads the summed words on the device and leaves the variables
        # alone when either is set; the trainer reads them one step late together with the loss, so the
        # rank-synchronous abort / fall-back costs neither a collective of its own nor a host synchronisation
        gp = self._buf('gpacked', (lay.gpacked_size + 2,), zero=True)
        # d logits, time-major + padded, sequence mask folded in (rows of padded utterances stay 0)
        dlog = self._buf('dlog', (T, Bp, ldp), zero=True)
        ops.relayout_rows(c['dpred'], dlog, B, T, F, ldp, (T * F, F), (ldp, Bp * ldp),
                          row_scale=c['row_scale'], scale_strides=(1, Bp))
        if lay.asr:
            ops.relayout_rows(c['dasr'], dlog.view(-1)[lay.asr_col:], B, T, lay.asr, ldp - lay.asr_col,
                              (T * lay.asr, lay.asr), (ldp, Bp * ldp))
        dlog2 = dlog.view(M, ldp)
        h_top = c['rnn_out'].view(M, 2 * HP)
        splits = ops.splitk_for(M)
        # Small batches: the BPTT kernels occupy 64 .. 256 CUs and everything on the chain dz_l -> dX -> BPTT_{l-1} is
        # latency-bound, while the weight gradients of layer l (dWx, dWh, db: split-K GEMMs and column sums over dz_l)
        # feed nothing on that chain -- they run on a second stream, behind an event, beside the BPTT of the layers
        # below (dz then needs a buffer per layer).  Measured (ms per step, one stream -> two): B = 8: 11.9 -> 10.4,
        # B = 32: 11.1 -> 10.3, B = 128: 15.0 -> 15.7, B = 256: 23.2 -> 22.8 -- so only the smallest batches do it.
        overlap = Bp <= 64 and os.environ.get('AVSI_TRAIN_OVERLAP', '1') != '0'
        main = torch.cuda.current_stream(self.device)
        side = main
        if overlap:
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            side = self._side_stream

        # Data parallel (train_op sets _reduce_in_backward): the packed gradient buffer is all-reduced in buckets --
        # one per layer, started right behind that layer's weight-gradient kernels, and the head at the end -- so
        # the collectives of the upper layers run while the BPTT of the lower ones is still going
        reduce = bool(getattr(self, '_reduce_in_backward', False)) and parallel.dp_active()
        works = []

        def reduce_from(name, upto=None):
            if reduce:
                lo = lay.gpacked[name][0]
                hi = lay.gpacked_size + 2 if upto is None else lay.gpacked[upto][0]
                works.append(parallel.all_reduce_sum_async(gp[lo:hi]))

        # Three side streams: dWx and the two dWh of a layer are independent few-tile GEMMs, and those of layer 0 are the
        # tail of the step -- one after the other they were 0.40 ms with the chip otherwise idle.  Under data parallelism
        # too (round 6: a rank used to put them on ONE stream so that the layer's all-reduce bucket could follow them in
        # stream order -- 0.9 ms of GEMMs per layer in a row in front of every bucket): the bucket is started from the
        # first side stream once it has waited for the other two
        sides = [side]
        if overlap:
            if getattr(self, '_side_streams', None) is None:
                self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(2)]
            sides = [side] + self._side_streams

        def on_side(fn, which=0, delay=0):
            """Run fn on a side stream once everything enqueued on the main stream so far has finished (`delay` us
            later: the next cooperative BPTT grid on the main stream becomes ready at the same instant and must
            get its CUs first -- behind three GEMM grids it waited 250 us for residency)."""
            if not overlap:
                return fn()
            st = sides[which % len(sides)]
            ev = torch.cuda.Event()
            ev.record(main)
            st.wait_event(ev)
            with torch.cuda.stream(st):
                if delay:
                    ops.stream_delay(delay)
                fn()

        def head_grads():
            ops.gemm_splitk(h_top, dlog2, lay.gpacked_view(gp, 'dpw'), trans_a=True, m=2 * HP, n=ldp, k=M, splits=splits)
            if lay.ones_col_top < 0:
                ops.colsum(dlog2, lay.gpacked_view(gp, 'dpb'), m=M, n=ldp)
        on_side(head_grads)
        dh = self._buf('dh', (T, Bp, 2 * HP))
        ops.gemm(dlog2, v.p('pw'), out=dh.view(M, 2 * HP), trans_b=True, m=M, n=2 * HP, k=ldp)
        if c.get('drop_scale') is not None:
            ops.scale_elements(dh.view(M, 2 * HP), c['drop_scale'].view(M, 2 * HP), 2 * HP)      # gradient of the dropout
        for li in range(self.num_layers - 1, -1, -1):
            kp = lay.kp[li]
            dz = self._buf('dz%d' % li if overlap else 'dz', (T, Bp, 2 * GP))
            ops.blstm_rec_bwd(dh, c['reserve'][li], v.p('whb%d' % li), dz)
            dz2 = dz.view(M, 2 * GP)

            # dX first: it is on the chain to the BPTT of the layer below, the weight gradients are not -- started
            # together with it (they need the same dz) four GEMMs shared the chip and the dX product took 427 us
            # instead of 230; behind it they run beside the next BPTT kernel, which occupies 64 CUs
            if li > 0:
                if DX_SPLITS > 1 and M * 2 * HP < 2 * 256 * 128 * 128:
                    # too few 128 x 128 output tiles to fill the chip once (8000 rows: 252): cut the 2048-deep reduction
                    ops.gemm_splitk(dz2, v.p('wx%d' % li), dh.view(M, 2 * HP), trans_b=True, m=M, n=2 * HP, k=2 * GP,
                                    splits=DX_SPLITS)
                else:
                    ops.gemm(dz2, v.p('wx%d' % li), out=dh.view(M, 2 * HP), trans_b=True, m=M, n=2 * HP, k=2 * GP)

            def input_grads(li=li, kp=kp, dz=dz, dz2=dz2):
                x = c['layer_in'][li].view(M, kp)
                ops.gemm_splitk(x, dz2, lay.gpacked_view(gp, 'dwx%d' % li), trans_a=True, m=kp, n=2 * GP, k=M, splits=splits)
                if lay.ones_col[li] < 0:
                    ops.colsum(dz2, lay.gpacked_view(gp, 'db%d' % li), m=M, n=2 * GP)
                if lay.side_dim(li):
                    # the side input saw every frame's dz: sum over time first, then two small GEMMs
                    E = lay.side_dim(li)
                    dsb = self._buf('dside_bias', (Bp, 2 * GP))
                    ops.colsum(dz.view(T, Bp * 2 * GP), dsb.view(-1), m=T, n=Bp * 2 * GP)
                    ops.gemm(c['side_p'], dsb, out=lay.gpacked_view(gp, 'dwe'), trans_a=True, m=lay.side_p, n=2 * GP, k=Bp)
                    dside = self._buf('dside', (Bp, lay.side_p))
                    ops.gemm(dsb, v.p('we'), out=dside, trans_b=True, m=Bp, n=lay.side_p, k=2 * GP)
                    self._side_backward(dside[:B, :E], gp)

            def recurrent_grads(d, li=li, dz2=dz2):
                # dWh[d] = H_prev^T . dZ_d : fw pairs h[t-1] with dz[t], bw pairs h[t+1] with dz[t]
                hout = self._ws[('h%d' % li, (T, Bp, 2 * HP))].view(M, 2 * HP)
                dwh = lay.gpacked_view(gp, 'dwh%d' % li)
                if T > 1:
                    Mr = (T - 1) * Bp
                    if d == 0:
                        ops.gemm_splitk(hout[:Mr, :HP], dz2[Bp:, :GP], dwh[0], trans_a=True, m=HP, n=GP, k=Mr, splits=splits)
                    else:
                        ops.gemm_splitk(hout[Bp:, HP:], dz2[:Mr, GP:], dwh[1], trans_a=True, m=HP, n=GP, k=Mr, splits=splits)
                else:
                    dwh[d].zero_()

            if len(sides) > 1:
                hold = SIDE_DELAY_US if li > 0 else 0
                on_side(input_grads, 0, hold)
                on_side(lambda: recurrent_grads(0), 1, hold)
                on_side(lambda: recurrent_grads(1), 2, hold)
                if reduce:
                    # the layer's bucket: behind all three products (the collective is ordered after the work enqueued on the
                    # stream it is started from)
                    sides[0].wait_stream(sides[1])
                    sides[0].wait_stream(sides[2])
                    with torch.cuda.stream(sides[0]):
                        reduce_from('dwx%d' % li, 'dwx%d' % (li + 1) if li + 1 < self.num_layers else 'dpw')
            else:
                def weight_grads(li=li):
                    input_grads()
                    recurrent_grads(0)
                    recurrent_grads(1)
                    reduce_from('dwx%d' % li, 'dwx%d' % (li + 1) if li + 1 < self.num_layers else 'dpw')
                on_side(weight_grads)
        if overlap:
            for st in sides:
                main.wait_stream(st)
        if reduce:
            ops.step_guard(c['loss3'][0:1], gp[lay.gpacked_size:])
        reduce_from('dpw')                       # projection head (+ the speaker-embedding MLP of the SSNN variant) + the guard
        for w in works:
            if w is not None:
                w.wait()
        c['grads_reduced'] = reduce
        if reduce:
            c['guard'] = gp[lay.gpacked_size:].clone()        # gp is this model's buffer for the next step too
        if ops.coop_split(Bp):
            ops.coop_poll(self.device)
        grads = v.unpack_grads(gp, out=self._buf('grads', (lay.ref_size,)))
        c['grads'] = grads
        return grads

    @property
    def gradients(self):
        """Flat reference-layout gradient of `loss_func` (+ l2 term) w.r.t. the variables."""
        g = self._backward()
        if self.regularization:
            return g + self.regularization * self.variables.flat
        return g

    @property
    def train_op(self):
        """One optimiser step (reference models.py:161-179): Adam with the CONSTANT starter learning
        rate, or sgd / momentum 0.9 with the staircase-decayed rate.  Under
        torch.distributed the flat gradient is all-reduced (sum) first and averaged in the update,
        which reproduces the single-GPU gradient of the global batch (equal per-rank batches)."""
        c = self._cache
        if c.get('trained'):
            return None
        self._reduce_in_backward = 'grads' not in c
        try:
            g = self._backward()
        finally:
            self._reduce_in_backward = False
        v = self.variables
        world = parallel.world_size()
        if not c.get('grads_reduced'):
            # single process, or gradients that were fetched (unreduced) before train_op: one flat all-reduce + the guard's
            parallel.all_reduce_sum_(g)
            c['grads_reduced'] = parallel.dp_active()
            c['guard'] = parallel.all_reduce_sum_(ops.step_guard(c['loss3'][0:1], torch.empty(2, device=self.device)))
        self.apply_gradients(g, world, c['guard'])
        c['trained'] = True
        return None

    def apply_gradients(self, g, world=1, guard=None):
        """The optimiser update of ``train_op`` (reference models.py:168-179) with a gradient the caller hands in: ``g`` is
        the flat reference-layout gradient SUMMED over ``world`` equal shares (the update divides by it), ``guard`` the step
        guard's device words (any non-zero word leaves variables and slots alone).  Every optimizer_type runs as one guarded
        HIP launch: TF-Adam with the CONSTANT starter rate, sgd / momentum 0.9 with the staircase-decayed rate."""
        v = self.variables
        step = v.global_step + 1
        l2 = float(self.regularization or 0.0)
        if self.optimizer_choice == 'adam':
            if v.adam_m is None:
                v.adam_m = torch.zeros_like(v.flat)
                v.adam_v = torch.zeros_like(v.flat)
            ops.adam_tf(v.flat, g, v.adam_m, v.adam_v, step, self.starter_learning_rate, grad_scale=1.0 / world, l2=l2,
                        skip=guard)
        elif self.optimizer_choice in ('sgd', 'momentum'):
            # learning_rate is taken BEFORE the count moves: exponential_decay(global_step) as the step's train_op sees it
            accum = None
            if self.optimizer_choice == 'momentum':
                if v.adam_m is None:
                    v.adam_m = torch.zeros_like(v.flat)         # the 'Momentum' slot
                accum = v.adam_m
            ops.sgd_momentum(v.flat, g, accum, self.learning_rate, momentum=0.9, grad_scale=1.0 / world, l2=l2, skip=guard)
        else:
            print('Optimizer must be either sgd, momentum or adam. Closing...')
            sys.exit(1)
        v.global_step = step
        v.repack()

    @property
    def step_guard(self):
        """Two-element device tensor, the verdict on this step that the optimiser update itself obeyed (no reference
        counterpart; the reference aborts on a NaN / Inf loss AFTER the update, training_emb.py:244-249):
        [0] 0 while the loss is finite on EVERY data-parallel rank, NaN otherwise; [1] the number of ranks on which a
        cooperative recurrent launch gave up waiting for residency (results void).  After ``train_op`` the words have
        been summed over the ranks inside the last gradient bucket and the update was SKIPPED on the device if either is
        set; before it they are this rank's own.  The trainer reads them one step late with the loss."""
        c = self._cache
        if c.get('guard') is not None:
            return c['guard']
        self._loss()
        return ops.step_guard(c['loss3'][0:1], torch.empty(2, device=self.device))

    @property
    def nonfinite_flag(self):
        """step_guard[0:1] (kept for callers of round 3)."""
        return self.step_guard[0:1]

    @property
    def global_step(self):
        return self.variables.global_step

    @property
    def learning_rate(self):
        """tf.train.exponential_decay(staircase=True) (models.py:165-166); Adam ignores it (F9)."""
        return self.starter_learning_rate * self.learning_decay ** math.floor(
            self.variables.global_step / self.updating_step)

    @property
    def summaries(self):
        """The tensors the reference hands to TensorBoard (models.py:199-218), as a dict of device
        tensors for the first 10 utterances: spectrogram images [n, F, T, 1] flipped upside down
        (frequency up), the mask likewise, peak-normalised target / enhanced audio [n, samples]."""
        n = 10

        def image(x):
            return x[:n].transpose(1, 2).flip(1).unsqueeze(3)
        T = self.prediction.shape[1]
        tgt, enh = self.target_sources[:n], self.enhanced_sources[:n]
        return {'Target_spectrogram': image(self.target_spec_norm), 'Enhanced_spectrogram': image(self.prediction),
                'Mask': image(self.masks[:, :T]),
                'Target_audio': tgt / tgt.abs().amax(dim=1, keepdim=True),
                'Enhanced_audio': enh / enh.abs().amax(dim=1, keepdim=True)}

    @property
    def train_vars(self):
        return [(n, self.layout.ref_view(self.variables.flat, n)) for n, _, _ in self.layout.ref_entries]

    all_vars = train_vars


from .unet_model import UNetFConvModel  # noqa: E402,F401  (the reference keeps it in models.py:519-715)
