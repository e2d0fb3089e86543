"""Oracle: the BLSTM model variants (torch CPU float64 + autograd).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Restates, literally (tiled embeddings, concatenations, the full MLP over every frame), the graphs of
* ``StackedBLSTMEmbeddingModel`` (reference av_speech_inpainting/models.py:1120-1474): embedding
  tiles concatenated to the input of BLSTM layer ``integration_layer`` (:1205-1209, :1247-1259),
  prediction = seq_mask * (target * mask + inference * (1 - mask)) (:1367-1372), loss = loss_hole (:1378-1394);
* ``StackedBLSTMSSNNModel`` (:718-1118): the speaker embedding is a 3-layer MLP over
  [features, delta features], masked by ``masks[:, :, 0]`` and averaged with ``sum / (count + 1)``
  (:800-838), then used like the external embedding;
* ``StackedBLSTM2StepsModel`` (:240-317): the second network's audio features are the first's prediction.
The LSTM cell is oracle.blstm's, gradients come from torch.autograd.
"""
import numpy as np
import torch

from . import blstm as OB
from . import frontend as OF

DT = torch.float64


def _t(x, grad=False):
    return torch.tensor(np.asarray(x, dtype=np.float64), dtype=DT, requires_grad=grad)


def params_to_torch(params, grad=True):
    out = {'layers': [{d: {k: _t(layer[d][k], grad) for k in ('kernel', 'bias')} for d in ('fw', 'bw')}
                      for layer in params['layers']],
           'proj': {k: _t(params['proj'][k], grad) for k in ('weights', 'biases')}}
    if 'mlp' in params:
        out['mlp'] = {k: _t(v, grad) for k, v in params['mlp'].items()}
    if 'asr' in params:
        out['asr'] = {k: _t(v, grad) for k, v in params['asr'].items()}
    return out


def grads_to_numpy(tp):
    g = {'layers': [{d: {k: layer[d][k].grad.numpy() for k in ('kernel', 'bias')} for d in ('fw', 'bw')}
                    for layer in tp['layers']],
         'proj': {k: tp['proj'][k].grad.numpy() for k in ('weights', 'biases')}}
    if 'mlp' in tp:
        g['mlp'] = {k: v.grad.numpy() for k, v in tp['mlp'].items()}
    if 'asr' in tp:
        g['asr'] = {k: v.grad.numpy() for k, v in tp['asr'].items()}
    return g


def lstm_direction(x, kernel, bias, reverse):
    B, T, _ = x.shape
    H = kernel.shape[1] // 4
    h = torch.zeros(B, H, dtype=DT)
    c = torch.zeros(B, H, dtype=DT)
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = torch.cat([x[:, t], h], dim=1) @ kernel + bias
        i, j, f, o = torch.sigmoid(z[:, :H]), torch.tanh(z[:, H:2 * H]), torch.sigmoid(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
        c = f * c + i * j
        h = o * torch.tanh(c)
        out[t] = h
    return torch.stack(out, dim=1)


def blstm_stack(x, layers, side_layer=None, side=None):
    """stack_bidirectional_dynamic_rnn; before layer ``side_layer`` the per-utterance vector ``side``
    [B, E] is tiled over time and concatenated to the layer input."""
    for li, layer in enumerate(layers):
        if side_layer is not None and li == side_layer:
            x = torch.cat([x, side[:, None, :].expand(-1, x.shape[1], -1)], dim=2)
        fw = lstm_direction(x, layer['fw']['kernel'], layer['fw']['bias'], False)
        bw = lstm_direction(x, layer['bw']['kernel'], layer['bw']['bias'], True)
        x = torch.cat([fw, bw], dim=2)
    return x


def delta_features(x):
    """add_delta_features(n_delta=1, N=2) of audio_processing.py:85-104 on [B, T, F] (torch)."""
    T = x.shape[1]
    pad = torch.cat([x[:, :1], x[:, :1], x, x[:, -1:], x[:, -1:]], dim=1)     # cumulative SYMMETRIC pad = edge repeat
    d = sum(i * (pad[:, 2 + i:2 + i + T] - pad[:, 2 - i:2 - i + T]) for i in (1, 2)) / 10.0
    return torch.cat([x, d], dim=2)


def speaker_embedding(feats, masks, mlp):
    """models.py:800-838."""
    B, T, F = feats.shape
    inp = delta_features(feats).reshape(B * T, 2 * F)
    lrelu = torch.nn.functional.leaky_relu
    a1 = lrelu(inp @ mlp['weights_1'] + mlp['biases_1'], 0.3)
    a2 = lrelu(a1 @ mlp['weights_2'] + mlp['biases_2'], 0.3)
    out = (a2 @ mlp['weights_3'] + mlp['biases_3']).reshape(B, T, -1)
    m = masks[:, :, 0]
    return (out * m[:, :, None]).sum(dim=1) / (m.sum(dim=1) + 1)[:, None]


def frontend(wav, masks, mean, std, seq_len):
    T = int(np.max(seq_len))
    stft_c, norm, feats = OF.inpainter_frontend(wav, mean, std, np.asarray(masks, np.float64), np.float64,
                                                audio_feat_dim=masks.shape[2], max_len=T)
    return _t(norm), _t(feats), T


def variant_forward(wav, masks, mean, std, seq_len, tparams, int_layer, embeddings=None, video=None, input_type='a',
                    fed_audio_features=None):
    """Embedding model (``embeddings`` given) or SSNN model (``tparams['mlp']`` present)."""
    norm, feats, T = frontend(wav, masks, mean, std, seq_len)
    m = _t(masks)[:, :T]
    if fed_audio_features is not None:
        feats = _t(fed_audio_features)
    if input_type == 'a':
        x = feats
    elif input_type == 'v':
        x = _t(video)[:, :T]
    else:
        x = torch.cat([feats, _t(video)[:, :T]], dim=2)
    side = _t(embeddings) if embeddings is not None else speaker_embedding(feats, m, tparams['mlp'])
    rnn = blstm_stack(x, tparams['layers'], int_layer, side)
    B = rnn.shape[0]
    logits = (rnn.reshape(B * T, -1) @ tparams['proj']['weights'] + tparams['proj']['biases']).reshape(B, T, -1)
    seq = _t(OB.sequence_mask(seq_len, T, np.float64))[:, :, None]
    pred = seq * (norm * m + logits * (1 - m))
    err = (norm - pred).abs()
    out = {'prediction': pred, 'inference': logits, 'speaker_embedding': side,
           'loss_hole': (err * (1 - m)).sum() / (1 - m).sum(), 'loss_valid': (err * m).sum() / m.sum()}
    out['loss_func'] = out['loss'] = out['loss_hole']
    return out


def init_variant_params(seed, input_dim, side_layer, side_dim, net_dim=(250, 250, 250), audio_feat_dim=257, mlp=False):
    """TF default initialisers (as oracle.blstm.init_params) with the side rows in place."""
    rng = np.random.default_rng(seed)
    H = net_dim[0]
    layers, d = [], input_dim
    for li in range(len(net_dim)):
        rows = d + (side_dim if li == side_layer else 0) + H
        lim = np.sqrt(6.0 / (rows + 4 * H))
        layers.append({dn: {'kernel': rng.uniform(-lim, lim, size=(rows, 4 * H)).astype(np.float32),
                            'bias': rng.normal(0, 0.1, size=4 * H).astype(np.float32)} for dn in ('fw', 'bw')})
        d = 2 * H
    params = {'layers': layers,
              'proj': {'weights': rng.normal(0, 1 / np.sqrt(2 * H), size=(2 * H, audio_feat_dim)).astype(np.float32),
                       'biases': rng.normal(0, 0.1, size=audio_feat_dim).astype(np.float32)}}
    if mlp:
        F = audio_feat_dim
        params['mlp'] = {'weights_1': rng.normal(0, 1 / np.sqrt(F), size=(2 * F, side_dim)).astype(np.float32),
                         'biases_1': rng.normal(0, 0.1, size=side_dim).astype(np.float32),
                         'weights_2': rng.normal(0, 1 / np.sqrt(side_dim), size=(side_dim, side_dim)).astype(np.float32),
                         'biases_2': rng.normal(0, 0.1, size=side_dim).astype(np.float32),
                         'weights_3': rng.normal(0, 1 / np.sqrt(side_dim), size=(side_dim, side_dim)).astype(np.float32),
                         'biases_3': rng.normal(0, 0.1, size=side_dim).astype(np.float32)}
    return params



def ctc_multitask_forward(wav, masks, mean, std, seq_len, tparams, labels, labels_lengths, ctc_weight, video=None,
                          input_type='a'):
    """StackedBLSTMSSNNCTCLossModel (reference av_speech_inpainting/models.py:1741-2047): a plain stacked
    BLSTM with two heads on its output -- ``inpainting`` (prediction / loss_hole as in the embedding
    variants, :1921-1931) and ``asr`` (un-masked logits over num_asr_labels + 1 classes, :1910-1916) --
    and ``loss_func = loss_hole + ctc_loss_weight * mean_b ctc_loss_b`` (:1944-1955).  The speaker-embedding
    MLP the class also builds (:1830-1871) feeds nothing and is left out.  ``tparams['asr']`` holds the
    second head.  The CTC term is torch's ctc_loss in float64 (tests/test_oracle_ctc.py shows it equal to
    oracle.ctc's explicit recursions, which restate tf.nn.ctc_loss)."""
    norm, feats, T = frontend(wav, masks, mean, std, seq_len)
    m = _t(masks)[:, :T]
    if input_type == 'a':
        x = feats
    elif input_type == 'v':
        x = _t(video)[:, :T]
    else:
        x = torch.cat([feats, _t(video)[:, :T]], dim=2)
    rnn = blstm_stack(x, tparams['layers'])
    B = rnn.shape[0]
    flat = rnn.reshape(B * T, -1)
    logits = (flat @ tparams['proj']['weights'] + tparams['proj']['biases']).reshape(B, T, -1)
    asr = (flat @ tparams['asr']['weights'] + tparams['asr']['biases']).reshape(B, T, -1)
    seq = _t(OB.sequence_mask(seq_len, T, np.float64))[:, :, None]
    pred = seq * (norm * m + logits * (1 - m))
    loss_hole = ((norm - pred).abs() * (1 - m)).sum() / (1 - m).sum()
    C = asr.shape[2]
    ctc = torch.nn.functional.ctc_loss(torch.log_softmax(asr, dim=2).transpose(0, 1),
                                       torch.as_tensor(np.asarray(labels)).long(),
                                       torch.as_tensor(np.asarray(seq_len)).long(),
                                       torch.as_tensor(np.asarray(labels_lengths)).long(), blank=C - 1,
                                       reduction='none').mean()
    return {'prediction': pred, 'asr_logits': asr, 'loss_hole': loss_hole, 'ctc_loss': ctc,
            'loss_func': loss_hole + ctc_weight * ctc}
