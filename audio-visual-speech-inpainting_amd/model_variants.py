"""Model variants built on the same gfx950 kernels as StackedBLSTMModel.

* ``StackedBLSTM2StepsModel``     reference models.py:240-317 -- a video-only BLSTM predicts a
  spectrogram, an audio-visual BLSTM takes that prediction (in place of the masked audio features)
  plus the video features; only the audio-visual network trains.
* ``StackedBLSTMEmbeddingModel``  reference models.py:1120-1474 -- an external speaker embedding
  (512-d, from the TFRecords) is concatenated, tiled over time, to the input of BLSTM layer
  ``integration_layer``; the prediction restores the known bins and the loss is ``loss_hole``.

MI355X-first: the tiled embedding is never formed.  ``tile(e) . W_e`` is one row per utterance, so
it enters the gate pre-activations as a per-utterance bias computed by a [B, E] x [E, 2048] GEMM,
and its gradient is the time-sum of ``dz`` through two small GEMMs (see StackedBLSTMModel._forward
and ._backward, ``side`` hooks).  That removes E / (D + E) of the layer's input-projection FLOPs
(512 of 769 columns for the audio model at integration layer 0).

Deliberately not inherited (SURVEY App. B style): in the CPU-compatible branch of the reference's
embedding model with ``integration_layer >= 1`` the second BLSTM is fed ``net_inputs`` instead of
the concatenation it has just built (models.py:1276-1281), so the first BLSTM and the embedding are
dead code there; this build implements the training (CudnnLSTM) branch's dataflow (:1236-1259) for
both.  Variable names follow the ``integration_layer = 0`` scoping (``cudnn_lstm/...cell_<l>``).

* ``StackedBLSTMSSNNModel``       reference models.py:718-1118 -- the speaker embedding is computed
  from the masked audio features by a 3-layer MLP (514 -> 200 -> 200 -> 200, leaky_relu 0.3) over
  every frame, masked by ``masks[:, :, 0]`` and averaged with ``sum / (count + 1)`` (:800-838); it
  trains end to end.  Here the MLP runs on the time-major input buffer in place (the
  [features, delta] concatenation is two accumulating GEMMs), and because its last layer is linear
  the masked time average is taken BEFORE it: ``mean_t(w_t (a_t W3 + b3)) = (sum_t w_t a_t) W3 +
  (sum_t w_t) b3`` -- a [B, 200] x [200, 200] GEMM instead of a [B T, 200] x [200, 200] one.
"""
import numpy as np
import torch

from . import _lib, ops
from . import audio_processing as ap
from .blstm_layout import ParamLayout, input_pitch
from .models import BLSTMVariables, StackedBLSTMModel, _as_device


class StackedBLSTMEmbeddingModel(StackedBLSTMModel):
    """Speech inpainting BLSTM with an external speaker embedding (reference models.py:1120-1474)."""

    def __init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                 audio_features=None, video_features=None, embeddings=None, input='a', is_training=True,
                 variables=None, seed=0):
        self.int_layer = int(config.get('integration_layer', 0))
        self.emb_dim = int(config.get('emb_dim', 512))           # training_emb.py:71: [None, 512]
        self.embeddings = None
        super().__init__(sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                         audio_features=audio_features, video_features=video_features, input=input,
                         is_training=is_training, variables=variables, seed=seed,
                         side=(self.int_layer, self.emb_dim), blend=True)
        self.feed_embeddings(embeddings)

    def feed_embeddings(self, embeddings):
        self.embeddings = _as_device(embeddings, device=self.device)
        if self.embeddings is not None and (self.embeddings.dim() != 2 or self.embeddings.shape[1] != self.emb_dim):
            raise ValueError("embeddings must be [batch, %d], got %r" % (self.emb_dim, tuple(self.embeddings.shape)))

    def feed(self, sequence_lengths=None, target_sources=None, masks=None, video_features=None, audio_features=None,
             audio_feat_mean=None, audio_feat_std=None, embeddings=None):
        super().feed(sequence_lengths, target_sources, masks, video_features, audio_features, audio_feat_mean,
                     audio_feat_std)
        if embeddings is not None:
            self.feed_embeddings(embeddings)

    def _side_input(self):
        if self.embeddings is None:
            raise _lib.AvsiError("the embedding model needs `embeddings` [batch, %d] to be fed" % self.emb_dim)
        return self.embeddings


class TwoStepVariables(object):
    """What the drivers' tf.train.Saver sees of the two-step model: both networks
    (training_emb.py:122-125).  npz checkpoints are a pair (``<path>.npz`` for the audio-visual
    network, ``<path>.vnet.npz`` for the video one); a TensorFlow bundle holds both variable scopes."""

    def __init__(self, av, video):
        self.av, self.video = av, video

    def __getattr__(self, name):            # flat, layout, global_step, adam_m, ... of the trained network
        return getattr(self.av, name)

    def rewind_step(self):
        self.av.rewind_step()

    def save(self, path):
        self.video.save(path + '.vnet')
        return self.av.save(path)

    def save_tf(self, prefix, scope=None):
        from . import tf_checkpoint as tc
        cpu = lambda t: t.cpu().numpy() if t is not None else None
        merged = tc.export_variables(self.video.layout, cpu(self.video.flat), 'v-blstm')
        del merged['v-blstm/Variable']
        merged.update(tc.export_variables(self.av.layout, cpu(self.av.flat), 'av-blstm-twosteps', cpu(self.av.adam_m),
                                          cpu(self.av.adam_v), self.av.global_step))
        return tc.write_bundle(prefix, merged)

    def restore(self, path):
        import os
        from . import tf_checkpoint as tc
        if not path.endswith('.npz') and not os.path.isfile(path + '.npz') and tc.is_bundle(path):
            bundle = tc.read_bundle(path)
            for var, scope in ((self.video, 'v-blstm'), (self.av, 'av-blstm-twosteps')):
                flat, m, v, step = tc.import_variables(bundle, var.layout, scope=scope)
                var.load_flat(flat)
                var.global_step = step
                var.adam_m = torch.from_numpy(m).to(var.device) if m is not None else None
                var.adam_v = torch.from_numpy(v).to(var.device) if v is not None else None
            return
        self.av.restore(path)
        base = path[:-4] if path.endswith('.npz') else path
        if os.path.isfile(base + '.vnet.npz'):
            self.video.restore(base + '.vnet')


class StackedBLSTM2StepsModel(object):
    """2-steps speech inpainting BLSTM model (reference models.py:240-317).

    ``variables`` / ``video_variables`` are the BLSTMVariables of the audio-visual and of the
    video-only network (variable scopes ``av-blstm-twosteps`` and ``v-blstm``).  As in the
    reference, ``train_op`` updates the audio-visual network only (its ``train_vars``); the video
    network is what ``model_ckp_vnet`` restores (training_emb.py:125,162-168)."""

    def __init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                 video_features, is_training=True, variables=None, video_variables=None, seed=0):
        self.audio_feat_dim = config['audio_feat_dim']
        self.audio_len = config['audio_len']
        self.is_training = is_training
        # the video network never trains here: build it in inference form (no BPTT reserve kept)
        self.video_model = StackedBLSTMModel(sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std,
                                             dropout_rate, config, video_features=video_features, input='v',
                                             is_training=False, variables=video_variables, seed=seed)
        self.video_model.build_graph(var_scope='v-blstm')
        self.av_model = StackedBLSTMModel(sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std,
                                          dropout_rate, config, audio_features=None, video_features=video_features,
                                          input='av', is_training=is_training, variables=variables, seed=seed + 1)
        self.av_model.build_graph(var_scope='av-blstm-twosteps')
        self.variables = TwoStepVariables(self.av_model.variables, self.video_model.variables)
        self.video_variables = self.video_model.variables
        self.device = self.av_model.device
        self.var_scope = None
        self._chained = False

    def build_graph(self, var_scope=''):
        self.var_scope = var_scope

    def set_dropout_rate(self, rate):
        self.video_model.set_dropout_rate(rate)
        self.av_model.set_dropout_rate(rate)

    def feed(self, sequence_lengths=None, target_sources=None, masks=None, video_features=None, **kw):
        self.video_model.feed(sequence_lengths, target_sources, masks, video_features, **kw)
        self.av_model.feed(sequence_lengths, target_sources, masks, video_features, **kw)
        self._chained = False

    def _chain(self):
        """audio_features of the second step = prediction of the first (models.py:261-263)."""
        if not self._chained and self.av_model.target_sources is not None:
            self.av_model.fed_audio_features = self.video_model.prediction
            self.av_model._cache = {}
            self._chained = True
        return self.av_model

    @property
    def video_prediction(self):
        return self.video_model.prediction

    def __getattr__(self, name):
        # everything else the drivers fetch is the second step's (models.py:280-294)
        if name in ('target_spec_norm', 'inference', 'prediction', 'loss', 'loss_func', 'loss_hole', 'loss_valid',
                    'train_op', 'learning_rate', 'global_step', 'enhanced_sources', 'enhanced_sources_oracle_phase',
                    'all_vars', 'train_vars', 'gradients', 'nonfinite_flag', 'step_guard', 'net_inputs', 'target_stft',
                    'sequence_lengths', 'masks', 'target_sources'):
            return getattr(self._chain(), name)
        raise AttributeError(name)


class StackedBLSTMSSNNModel(StackedBLSTMModel):
    """Speech inpainting BLSTM model with SSNN (reference models.py:718-1118)."""
    EMB = 200    # models.py:804-809

    def __init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                 audio_features=None, video_features=None, input='a', is_training=True, variables=None, seed=0):
        self.int_layer = int(config.get('integration_layer', 0))
        F = config['audio_feat_dim']
        in_dim = {'a': F, 'v': config.get('video_feat_dim', 136), 'av': F + config.get('video_feat_dim', 136)}[input]
        # the MLP reads the audio features where the network input holds them (pitch = input pitch);
        # the video-only model keeps them in a buffer of their own
        pitch = input_pitch(F if input == 'v' else in_dim)
        if variables is None:
            layout = ParamLayout(in_dim, config['net_dim'], F, side=(self.int_layer, self.EMB), mlp=self.EMB,
                                 mlp_in_pitch=pitch)
            variables = BLSTMVariables(layout, seed=seed)
        elif variables.layout.mlp != self.EMB or variables.layout.mlp_in_pitch != pitch:
            raise ValueError("variables were not built for the SSNN model (mlp=%d, pitch=%d)" % (self.EMB, pitch))
        super().__init__(sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std, dropout_rate, config,
                         audio_features=audio_features, video_features=video_features, input=input,
                         is_training=is_training, variables=variables, seed=seed,
                         side=(self.int_layer, self.EMB), blend=True)

    # ------------------------------------------------------------------ forward (models.py:800-838)
    def _feat_tm(self):
        """Time-major [T][Bp][pitch] buffer whose first F columns are the masked audio features."""
        c = self._cache
        self._frontend()
        if self.input_type != 'v':
            return c['x0']
        if 'ssnn_feat' not in c:
            B, T, Bp = self._dims()
            P = self.layout.mlp_in_pitch
            buf = self._buf('ssnn_feat', (T, Bp, P), zero=True)
            if self.fed_audio_features is not None:
                buf[:, :B, :self.audio_feat_dim] = self.fed_audio_features[:, :T].transpose(0, 1)
            else:
                ap.frontend(self.target_sources, window_size=24, step_size=12, n_fft=512, num_frames_out=T,
                            num_bins=self.audio_feat_dim, mean=self.audio_feat_mean, std=self.audio_feat_std,
                            masks=self.masks, want_spec=False, want_feat=True, time_major=True, feat_cols=P,
                            _feat_out=buf)
            c['ssnn_feat'] = buf
        return c['ssnn_feat']

    def _side_input(self):
        c = self._cache
        B, T, Bp = self._dims()
        if 'spk' in c:
            return c['spk'][:B]
        v, W, P = self.variables, self.EMB, self.layout.mlp_in_pitch
        M = T * Bp
        feat = self._feat_tm()
        delta = self._buf('ssnn_delta', (T, Bp, P))
        # regression deltas run along time; in time-major storage every (utterance, bin) is a "bin"
        _lib.check(_lib.lib().avsi_delta_f32(_lib.ptr(feat), _lib.ptr(delta), 1, T, Bp * P, 2, _lib.stream_ptr()),
                   "avsi_delta_f32")
        l1 = self._buf('ssnn_l1', (M, W))
        ops.gemm(feat.view(M, P), v.p('mw1a'), out=l1, bias=v.p('mb1'))
        ops.gemm(delta.view(M, P), v.p('mw1b'), out=l1, beta=1.0)
        a1 = ops.bn_act(l1, W, self._buf('ssnn_a1', (M, W)), act=3)
        l2 = self._buf('ssnn_l2', (M, W))
        ops.gemm(a1, v.p('mw2'), out=l2, bias=v.p('mb2'))
        a2 = ops.bn_act(l2, W, self._buf('ssnn_a2', (M, W)), act=3)
        # frame weights w[t][b] = mask[b][t][0] / (sum_t mask[b][t][0] + 1); zero for padded rows
        m0 = self.masks[:, :T, 0]
        w = self._buf('ssnn_w', (T, Bp), zero=True)
        w[:, :B] = (m0 / (m0.sum(dim=1, keepdim=True) + 1.0)).t()
        a2w = self._buf('ssnn_a2w', (T, Bp, W))
        ops.relayout_rows(a2, a2w, Bp, T, W, W, (W, Bp * W), (W, Bp * W), row_scale=w, scale_strides=(1, Bp))
        S = self._buf('ssnn_S', (Bp, W))
        ops.colsum(a2w.view(T, Bp * W), S.view(-1), m=T, n=Bp * W)
        cw = w.sum(dim=0)
        spk = self._buf('ssnn_spk', (Bp, W))
        ops.gemm(S, v.p('mw3'), out=spk)
        spk.addcmul_(cw[:, None], v.p('mb3')[None, :])
        c.update(ssnn=(feat, delta, l1, a1, l2, w, S, cw), spk=spk)
        return spk[:B]

    @property
    def speaker_embedding(self):
        self._frontend()
        return self._side_input()

    # ------------------------------------------------------------------ backward
    def _side_backward(self, dside, gp):
        c, v, lay, W, P = self._cache, self.variables, self.layout, self.EMB, self.layout.mlp_in_pitch
        B, T, Bp = self._dims()
        M = T * Bp
        feat, delta, l1, a1, l2, w, S, cw = c['ssnn']
        g = lambda name: lay.gpacked_view(gp, name)
        demb = self._buf('ssnn_demb', (Bp, W), zero=True)
        demb[:B] = dside
        ops.gemm(S, demb, out=g('dmw3'), trans_a=True, m=W, n=W, k=Bp)
        g('dmb3').copy_((cw[:, None] * demb).sum(dim=0))
        dS = self._buf('ssnn_dS', (Bp, W))
        ops.gemm(demb, v.p('mw3'), out=dS, trans_b=True, m=Bp, n=W, k=W)
        # d a2[t][b][:] = w[t][b] dS[b][:]  (broadcast over time: source time stride 0)
        da2 = self._buf('ssnn_da2', (T, Bp, W))
        ops.relayout_rows(dS, da2, Bp, T, W, W, (W, 0), (W, Bp * W), row_scale=w, scale_strides=(1, Bp))
        splits = ops.splitk_for(M)
        dl2 = ops.bn_act_bwd(l2, da2.view(M, W), W, self._buf('ssnn_dl2', (M, W)), act=3)
        ops.gemm_splitk(a1, dl2, g('dmw2'), trans_a=True, m=W, n=W, k=M, splits=splits)
        ops.colsum(dl2, g('dmb2'), m=M, n=W)
        da1 = self._buf('ssnn_da1', (M, W))
        ops.gemm(dl2, v.p('mw2'), out=da1, trans_b=True, m=M, n=W, k=W)
        dl1 = ops.bn_act_bwd(l1, da1, W, self._buf('ssnn_dl1', (M, W)), act=3)
        ops.gemm_splitk(feat.view(M, P), dl1, g('dmw1a'), trans_a=True, m=P, n=W, k=M, splits=splits)
        ops.gemm_splitk(delta.view(M, P), dl1, g('dmw1b'), trans_a=True, m=P, n=W, k=M, splits=splits)
        ops.colsum(dl1, g('dmb1'), m=M, n=W)


class StackedBLSTMSSNNCTCLossModel(StackedBLSTMSSNNModel):
    """Multi-task model: speech inpainting + phone recognition with a CTC loss (reference
    models.py:1741-2047; the model the shipped ``blstm_ctc.config`` trains, ``v-blstm-ssnn-ctc``).

    A plain stacked BLSTM with two heads on its output: ``inpainting`` (prediction and loss_hole as in
    the embedding variants) and ``asr`` (``num_asr_labels + 1`` classes, the last one blank);
    ``loss_func = loss_hole + ctc_loss * mean_b CTC_b``.  The class also creates the speaker-embedding
    MLP variables and ``speaker_embedding`` can be fetched, but -- as in the reference, :1874-1918 --
    nothing of it reaches ``inference``: those variables get no gradient.

    Here both heads are column windows of ONE packed projection matrix (ParamLayout ``asr``), so the
    backward pass is the plain model's with a wider d-logits buffer; the CTC loss and its gradient come
    from avsi_ctc_loss_f32, ``decoding`` / ``per`` run the beam search (width 20) on the host like TF."""
    WITH_MLP = True

    def __init__(self, sequence_lengths, labels_lengths, target_sources, masks, labels, audio_feat_mean, audio_feat_std,
                 dropout_rate, config, audio_features=None, video_features=None, input='a', apply_mask=False,
                 is_training=True, variables=None, seed=0):
        self.num_classes = int(config['num_asr_labels'])         # blank included (config_utils adds it)
        self.ctc_loss_weight = float(config['ctc_loss'])
        self.int_layer = 0
        F = config['audio_feat_dim']
        in_dim = {'a': F, 'v': config.get('video_feat_dim', 136), 'av': F + config.get('video_feat_dim', 136)}[input]
        pitch = input_pitch(F if input == 'v' else in_dim)
        if variables is None:
            layout = ParamLayout(in_dim, config['net_dim'], F, asr=self.num_classes, mlp=self.EMB if self.WITH_MLP else None,
                                 mlp_in_pitch=pitch if self.WITH_MLP else None)
            variables = BLSTMVariables(layout, seed=seed)
        elif variables.layout.asr != self.num_classes or bool(variables.layout.mlp) != self.WITH_MLP:
            raise ValueError("variables were not built for this multi-task model (asr=%d)" % self.num_classes)
        self.labels = self.labels_lengths = None
        StackedBLSTMModel.__init__(self, sequence_lengths, target_sources, masks, audio_feat_mean, audio_feat_std,
                                   dropout_rate, config, audio_features=audio_features, video_features=video_features,
                                   input=input, is_training=is_training, variables=variables, seed=seed, blend=True)
        self.feed_labels(labels, labels_lengths)

    # ------------------------------------------------------------------ feed boundary
    def feed_labels(self, labels, labels_lengths):
        """Dense labels float/int [B, L] + labels_lengths [B] (training_ctc.py:68-73; the model keeps the
        first labels_lengths[b] entries of row b, ctc_label_dense_to_sparse :1760).  Labels outside
        [0, num_classes - 1) raise ValueError (TensorFlow: InvalidArgumentError)."""
        self._cache = {}
        if labels is None or labels_lengths is None:
            self.labels = self.labels_lengths = None
            return
        lab = np.asarray(labels.cpu() if isinstance(labels, torch.Tensor) else labels)
        lens = np.asarray(labels_lengths.cpu() if isinstance(labels_lengths, torch.Tensor) else labels_lengths).astype(np.int64)
        if lab.ndim != 2 or lens.shape != (lab.shape[0],) or (lens < 0).any() or (lens > lab.shape[1]).any():
            raise ValueError("labels must be [batch, L] with 0 <= labels_lengths <= L")
        lab = lab.astype(np.int32)                              # tf.cast(..., tf.int32) :1760
        used = np.arange(lab.shape[1])[None, :] < lens[:, None]
        if used.any() and (lab[used].min() < 0 or lab[used].max() >= self.num_classes - 1):
            raise ValueError("labels must lie in [0, %d)" % (self.num_classes - 1))
        self.labels, self.labels_lengths = lab, lens.astype(np.int32)
        self._labels_dev = torch.as_tensor(self.labels, device=self.device)
        self._lab_len_dev = torch.as_tensor(self.labels_lengths, device=self.device)

    def feed(self, sequence_lengths=None, target_sources=None, masks=None, video_features=None, audio_features=None,
             audio_feat_mean=None, audio_feat_std=None, labels=None, labels_lengths=None):
        StackedBLSTMModel.feed(self, sequence_lengths, target_sources, masks, video_features, audio_features,
                               audio_feat_mean, audio_feat_std)
        if sequence_lengths is not None:
            self._seq_dev32 = self._seq_dev.to(torch.int32)
        if labels is not None:
            self.feed_labels(labels, labels_lengths)

    # ------------------------------------------------------------------ loss (models.py:1944-1964)
    def _extra_loss(self, want_grad):
        c = self._cache
        if self.labels is None:
            # prediction-only use (the `infer` driver feeds no labels): loss_func stays loss_hole
            if want_grad:
                raise _lib.AvsiError("the multi-task model needs `labels` and `labels_lengths` to be fed for training")
            c['ctc_loss'] = None
            return
        logits = c['asr_logits']
        B = logits.shape[0]
        if self.labels.shape[0] != B:
            raise ValueError("labels are for %d utterances, the batch has %d" % (self.labels.shape[0], B))
        per_utt, dasr = ops.ctc_loss(logits, self._labels_dev, self._lab_len_dev, self._seq_dev32,
                                     grad_scale=self.ctc_loss_weight / B, want_grad=want_grad,
                                     max_label_len=max(int(self.labels_lengths.max()), 1))
        c['ctc_loss'] = per_utt.mean()
        c['dasr'] = dasr
        l3 = c['loss3']
        c['loss3'] = torch.stack([l3[1] + self.ctc_loss_weight * c['ctc_loss'], l3[1], l3[2]])

    @property
    def ctc_loss(self):
        self._loss()
        return self._cache['ctc_loss']

    @property
    def inference(self):
        """(inpainting head, asr logits [B, T, num_classes]) -- models.py:1903-1918; the first is the
        prediction (the raw logits of the known bins are not materialised, as in the base class)."""
        self._forward()
        return self._cache['pred'], self._cache['asr_logits']

    # ------------------------------------------------------------------ diagnostics (models.py:1934-1942, 2026-2031)
    def _decode(self):
        c = self._cache
        if 'decoded' not in c:
            self._forward()
            c['decoded'] = ops.ctc_beam_search(c['asr_logits'], self.sequence_lengths, beam_width=20)
        return c['decoded']

    @property
    def decoding(self):
        """Dense int32 [B, longest] padded with -1 (tf.sparse.to_dense(default_value=-1)), numpy."""
        return self._decode()[0]

    @property
    def per(self):
        """Normalised edit distance of every decoded sequence to its labels (tf.edit_distance), numpy [B]."""
        dec, dlen, _ = self._decode()
        if self.labels is None:
            raise _lib.AvsiError("`per` needs the labels to be fed")
        return np.array([ops.edit_distance(dec[b, :dlen[b]], self.labels[b, :self.labels_lengths[b]])
                         for b in range(len(dlen))], dtype=np.float32)


class StackedBLSTMCTCLossModel(StackedBLSTMSSNNCTCLossModel):
    """reference models.py:1475-1739.  There, ``inference`` concatenates ``self.speaker_embedding``,
    which the class never defines (:1565-1566): the reference cannot build this
    model.  Provided here as the multi-task model without the (unused) speaker-embedding variables."""
    WITH_MLP = False

    @property
    def speaker_embedding(self):
        raise AttributeError("StackedBLSTMCTCLossModel has no speaker embedding (reference models.py:1565)")
