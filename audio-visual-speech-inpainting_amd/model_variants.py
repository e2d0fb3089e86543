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
"""
from . import _lib
from .models import StackedBLSTMModel, _as_device


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
        self.variables = self.av_model.variables
        self.video_variables = self.video_model.variables
        self.device = self.av_model.device
        self.var_scope = None
        self._chained = False

    def build_graph(self, var_scope=''):
        self.var_scope = var_scope

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
                    'all_vars', 'train_vars', 'gradients', 'net_inputs', 'target_stft', 'sequence_lengths'):
            return getattr(self._chain(), name)
        raise AttributeError(name)
