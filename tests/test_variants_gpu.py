"""Model variants on the GPU against the literal torch-float64 restatement in
oracle/variants.py: embedding model (side input at layer 0 and after layer 1), two-step model."""
import numpy as np
import pytest
import torch

from oracle import blstm as OB
from oracle import frontend as OF
from oracle import variants as OV

pytestmark = pytest.mark.gpu


def _config(**kw):
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=2880, net_dim=[250, 250, 250],
               optimizer_type='adam', starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0,
               batch_size=3, l2=0.0, integration_layer=0)
    cfg.update(kw)
    return cfg


def _inputs(B, N, seed, ragged=False):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, T - 4)
        masks[b, s:s + 4] = 0
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(spec))
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    seq = np.full(B, T, np.int32)
    if ragged:
        seq[1] = T - 3
    return wav, masks, mean.astype(np.float32), std.astype(np.float32), video, seq, T


def _flat_grads(layout, g):
    return layout.flatten_oracle_params({'layers': g['layers'], 'proj': g['proj']}).astype(np.float64)


@pytest.mark.parametrize('int_layer,input_type', [(0, 'a'), (1, 'av'), (2, 'a')])
def test_embedding_model_forward_backward(int_layer, input_type):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    from avsi_amd.blstm_layout import ParamLayout
    from avsi_amd.model_variants import StackedBLSTMEmbeddingModel
    B, N, E = 3, 2880, 512
    wav, masks, mean, std, video, seq, T = _inputs(B, N, 10 + int_layer, ragged=True)
    rng = np.random.default_rng(5)
    emb = rng.normal(0, 1, size=(B, E)).astype(np.float32)
    D = 257 if input_type == 'a' else 393
    params = OV.init_variant_params(3, D, int_layer, E)
    layout = ParamLayout(D, (250, 250, 250), 257, side=(int_layer, E))
    assert layout.ref_size == sum(int(np.prod(v.shape)) for _, v in OB.flatten_params(params))
    variables = models.BLSTMVariables(layout)
    variables.load_flat(layout.flatten_oracle_params(params))
    cfg = _config(integration_layer=int_layer, audio_len=N)
    vid = video if input_type == 'av' else None
    m = StackedBLSTMEmbeddingModel(seq, wav, masks, mean, std, 0.0, cfg, video_features=vid, embeddings=emb,
                                   input=input_type, is_training=True, variables=variables)
    m.build_graph('a-blstm-emb')
    tp = OV.params_to_torch(params)
    ref = OV.variant_forward(wav, masks, mean, std, seq, tp, int_layer, embeddings=emb, video=vid, input_type=input_type)
    ref['loss'].backward()
    pred = m.prediction.cpu().numpy()
    np.testing.assert_allclose(pred, ref['prediction'].detach().numpy(), rtol=2e-4, atol=2e-4)
    # the known bins are restored exactly, frames beyond an utterance's length are zero
    tn = m.target_spec_norm.cpu().numpy()
    keep = masks[:, :T] == 1
    keep[1, seq[1]:] = False
    np.testing.assert_array_equal(pred[keep], tn[keep])
    assert np.all(pred[1, seq[1]:] == 0)
    for name in ('loss', 'loss_func', 'loss_hole', 'loss_valid'):
        np.testing.assert_allclose(float(getattr(m, name)), float(ref[name].detach()), rtol=1e-4, err_msg=name)
    g = m.gradients.cpu().numpy().astype(np.float64)
    gref = _flat_grads(layout, OV.grads_to_numpy(tp))
    scale = np.abs(gref).max()
    np.testing.assert_allclose(g, gref, rtol=2e-3, atol=2e-5 * scale)
    # the side rows of the kernel did receive a gradient
    k = layout.ref_view(g, 'cell_%d/fw/kernel' % int_layer)
    d_in = D if int_layer == 0 else 500
    assert np.abs(k[d_in:d_in + E]).max() > 0
    # one Adam step moves every variable by about lr
    before = variables.flat.clone()
    m.train_op
    step = (variables.flat - before).abs()
    assert float(step.max()) <= 1.01e-3 and float((step > 0).float().mean()) > 0.95


def test_embedding_model_requires_embeddings():
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib
    from avsi_amd.model_variants import StackedBLSTMEmbeddingModel
    wav, masks, mean, std, video, seq, T = _inputs(2, 1920, 1)
    m = StackedBLSTMEmbeddingModel(seq, wav, masks, mean, std, 0.0, _config(audio_len=1920), input='a', is_training=False)
    with pytest.raises(_lib.AvsiError):
        m.prediction
    with pytest.raises(ValueError):
        m.feed_embeddings(np.zeros((2, 7), np.float32))


def test_two_steps_model():
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    from avsi_amd.blstm_layout import ParamLayout
    from avsi_amd.model_variants import StackedBLSTM2StepsModel
    B, N = 3, 2880
    wav, masks, mean, std, video, seq, T = _inputs(B, N, 21)
    pv = OB.init_params(7, 136)
    pav = OB.init_params(8, 393)
    lv, lav = ParamLayout(136), ParamLayout(393)
    vv, vav = models.BLSTMVariables(lv), models.BLSTMVariables(lav)
    vv.load_flat(lv.flatten_oracle_params(pv))
    vav.load_flat(lav.flatten_oracle_params(pav))
    cfg = _config(audio_len=N)
    m = StackedBLSTM2StepsModel(seq, wav, masks, mean, std, 0.0, cfg, video, is_training=True, variables=vav,
                                video_variables=vv)
    m.build_graph('av-blstm-twosteps')
    # oracle: two plain models chained (models.py:255-263)
    f1 = OB.model_forward(wav, masks, mean, std, seq, pv, video=video, input_type='v', dtype=np.float64)
    x2 = np.concatenate([f1['prediction'], video.astype(np.float64)], axis=2)
    p2 = OB.cast_params(pav, np.float64)
    logits2, rnn2, caches2 = OB.inference(x2, p2, True)
    pred2 = OB.prediction(logits2, seq)
    np.testing.assert_allclose(m.video_prediction.cpu().numpy(), f1['prediction'], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(m.prediction.cpu().numpy(), pred2, rtol=5e-4, atol=5e-4)
    l2 = OB.losses(f1['target_spec_norm'], pred2, masks.astype(np.float64))
    np.testing.assert_allclose(float(m.loss), l2['loss'], rtol=1e-4)
    fwd = {'params': p2, 'target_spec_norm': f1['target_spec_norm'], 'prediction': pred2, 'rnn_outputs': rnn2,
           'caches': caches2}
    gref = lav.flatten_oracle_params(OB.model_backward(fwd, masks, seq)).astype(np.float64)
    g = m.gradients.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(g, gref, rtol=3e-3, atol=3e-5 * np.abs(gref).max())
    # train_op updates the audio-visual network only
    v_before, av_before = vv.flat.clone(), vav.flat.clone()
    m.train_op
    assert torch.equal(vv.flat, v_before) and not torch.equal(vav.flat, av_before)
    assert m.global_step == 1
    # a new feed re-chains the two steps
    m.feed(seq, wav * 0.5, masks, video)
    f1b = OB.model_forward(wav * 0.5, masks, mean, std, seq, pv, video=video, input_type='v', dtype=np.float64)
    np.testing.assert_allclose(m.video_prediction.cpu().numpy(), f1b['prediction'], rtol=2e-4, atol=2e-4)
    assert m.prediction.shape == (B, T, 257)


@pytest.mark.parametrize('int_layer,input_type', [(0, 'a'), (1, 'av'), (0, 'v')])
def test_ssnn_model_forward_backward(int_layer, input_type):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    from avsi_amd.blstm_layout import ParamLayout, input_pitch
    from avsi_amd.model_variants import StackedBLSTMSSNNModel
    B, N, E = 3, 2880, 200
    wav, masks, mean, std, video, seq, T = _inputs(B, N, 30 + int_layer, ragged=True)
    D = {'a': 257, 'av': 393, 'v': 136}[input_type]
    params = OV.init_variant_params(4, D, int_layer, E, mlp=True)
    pitch = input_pitch(257 if input_type == 'v' else D)
    layout = ParamLayout(D, (250, 250, 250), 257, side=(int_layer, E), mlp=E, mlp_in_pitch=pitch)
    variables = models.BLSTMVariables(layout)
    variables.load_flat(layout.flatten_oracle_params(params))
    cfg = _config(integration_layer=int_layer, audio_len=N)
    vid = video if input_type != 'a' else None
    m = StackedBLSTMSSNNModel(seq, wav, masks, mean, std, 0.0, cfg, video_features=vid, input=input_type,
                              is_training=True, variables=variables)
    m.build_graph(input_type + '-blstm-ssnn')
    tp = OV.params_to_torch(params)
    ref = OV.variant_forward(wav, masks, mean, std, seq, tp, int_layer, video=vid, input_type=input_type)
    ref['loss'].backward()
    np.testing.assert_allclose(m.speaker_embedding.cpu().numpy(), ref['speaker_embedding'].detach().numpy(),
                               rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(m.prediction.cpu().numpy(), ref['prediction'].detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(m.loss), float(ref['loss'].detach()), rtol=1e-4)
    g = m.gradients.cpu().numpy().astype(np.float64)
    gn = OV.grads_to_numpy(tp)
    gref = layout.flatten_oracle_params(gn).astype(np.float64)
    np.testing.assert_allclose(g, gref, rtol=2e-3, atol=2e-5 * np.abs(gref).max())
    # the MLP's own gradients, separately (they are small next to the BLSTM's)
    for k in ('weights_1', 'biases_1', 'weights_2', 'biases_2', 'weights_3', 'biases_3'):
        got, want = layout.ref_view(g, 'speaker_embedding/' + k), gn['mlp'][k]
        assert np.abs(want).max() > 0
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=5e-4 * np.abs(want).max(), err_msg=k)
    before = variables.flat.clone()
    m.train_op
    assert float((variables.flat - before).abs().max()) <= 1.01e-3
    # TF-style default initialisation covers the MLP (truncated normal, zero biases)
    fresh = models.BLSTMVariables(layout, seed=1)
    w1 = layout.ref_view(fresh.flat.cpu().numpy(), 'speaker_embedding/weights_1')
    assert 0.5 / np.sqrt(257) < w1.std() < 1.0 / np.sqrt(257) and np.abs(w1).max() <= 2.0 / np.sqrt(257) + 1e-6
    assert np.all(layout.ref_view(fresh.flat.cpu().numpy(), 'speaker_embedding/biases_2') == 0)


def _ctc_labels(B, T, C, rng, Lp=50):
    lab_len = rng.integers(1, 6, size=B).astype(np.int32)
    labels = np.zeros((B, Lp), dtype=np.float32)
    for b in range(B):
        labels[b, :lab_len[b]] = rng.integers(0, C - 1, size=lab_len[b])
    return labels, lab_len


@pytest.mark.parametrize('input_type,with_mlp', [('a', True), ('av', False)])
def test_ctc_multitask_model_forward_backward(input_type, with_mlp):
    """Both heads, loss_func = loss_hole + ctc_loss * mean CTC, gradient w.r.t. every variable
    (the unused speaker-embedding variables of the SSNN class get exactly zero)."""
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    from avsi_amd.blstm_layout import ParamLayout, input_pitch
    from avsi_amd import model_variants as mv
    B, N, C, W = 3, 2880, 34, 0.05
    wav, masks, mean, std, video, seq, T = _inputs(B, N, 31, ragged=True)
    rng = np.random.default_rng(8)
    labels, lab_len = _ctc_labels(B, T, C, rng)
    D = 257 if input_type == 'a' else 393
    params = OV.init_variant_params(4, D, None, 0)
    params['asr'] = {'weights': rng.normal(0, 1 / np.sqrt(500), size=(500, C)).astype(np.float32),
                     'biases': rng.normal(0, 0.1, size=C).astype(np.float32)}
    layout = ParamLayout(D, (250, 250, 250), 257, asr=C, mlp=200 if with_mlp else None,
                         mlp_in_pitch=input_pitch(D) if with_mlp else None)
    variables = models.BLSTMVariables(layout, seed=2)            # MLP variables: random, must not matter
    flat = variables.flat.cpu().numpy()
    base = layout.flatten_oracle_params(params)
    known = base != 0
    flat[known] = base[known]
    for name in ('logits/biases', 'asr/biases'):
        layout.ref_view(flat, name)[...] = layout.ref_view(base, name)
    for li in range(3):
        for d in ('fw', 'bw'):
            layout.ref_view(flat, 'cell_%d/%s/bias' % (li, d))[...] = layout.ref_view(base, 'cell_%d/%s/bias' % (li, d))
    variables.load_flat(flat)
    cfg = _config(audio_len=N, num_asr_labels=C, ctc_loss=W)
    vid = video if input_type == 'av' else None
    cls = mv.StackedBLSTMSSNNCTCLossModel if with_mlp else mv.StackedBLSTMCTCLossModel
    m = cls(seq, lab_len, wav, masks, labels, mean, std, 0.0, cfg, video_features=vid, input=input_type,
            is_training=True, variables=variables)
    m.build_graph('a-blstm-ssnn-ctc')
    tp = OV.params_to_torch(params)
    ref = OV.ctc_multitask_forward(wav, masks, mean, std, seq, tp, labels, lab_len, W, video=vid, input_type=input_type)
    ref['loss_func'].backward()
    pred, asr = m.inference
    np.testing.assert_allclose(pred.cpu().numpy(), ref['prediction'].detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(asr.cpu().numpy(), ref['asr_logits'].detach().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(m.loss_hole), float(ref['loss_hole'].detach()), rtol=1e-4)
    np.testing.assert_allclose(float(m.ctc_loss), float(ref['ctc_loss'].detach()), rtol=2e-4)
    np.testing.assert_allclose(float(m.loss_func), float(ref['loss_func'].detach()), rtol=2e-4)
    np.testing.assert_allclose(float(m.loss), float(ref['loss_func'].detach()), rtol=2e-4)
    g = m.gradients.cpu().numpy().astype(np.float64)
    go = OV.grads_to_numpy(tp)
    gref = layout.flatten_oracle_params({'layers': go['layers'], 'proj': go['proj'], 'asr': go['asr']}).astype(np.float64)
    scale = np.abs(gref).max()
    np.testing.assert_allclose(g, gref, rtol=2e-3, atol=2e-5 * scale)
    if with_mlp:
        for n_, shape, off in layout.ref_entries:
            if n_.startswith('speaker_embedding/'):
                assert not g[off:off + int(np.prod(shape))].any()
        assert tuple(m.speaker_embedding.shape) == (B, 200)
    # diagnostics: beam search on the host equals the oracle's decoder on the same logits
    from oracle import ctc as OC
    outs, _ = OC.beam_search(asr.cpu().numpy(), seq, beam_width=20)
    dec = m.decoding
    for b in range(B):
        got = [int(v) for v in dec[b] if v >= 0]
        assert got == outs[b]
        want_per = OC.edit_distance(outs[b], labels[b, :lab_len[b]].astype(int).tolist())
        assert abs(float(m.per[b]) - want_per) < 1e-6
    # one optimiser step moves both heads, leaves the unused variables alone
    before = m.variables.flat.clone()
    m.train_op
    after = m.variables.flat
    for n_, shape, off in layout.ref_entries:
        moved = bool((after[off:off + int(np.prod(shape))] != before[off:off + int(np.prod(shape))]).any())
        assert moved == (not n_.startswith('speaker_embedding/')), n_


def test_ctc_model_predicts_without_labels_and_rejects_bad_ones():
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib
    from avsi_amd import model_variants as mv
    B, N, C = 2, 2880, 34
    wav, masks, mean, std, video, seq, T = _inputs(B, N, 32)
    cfg = _config(audio_len=N, num_asr_labels=C, ctc_loss=0.001)
    m = mv.StackedBLSTMSSNNCTCLossModel(seq, None, wav, masks, None, mean, std, 0.0, cfg, input='a', is_training=False)
    pred = m.prediction
    keep = masks[:, :T] == 1
    np.testing.assert_array_equal(pred.cpu().numpy()[keep], m.target_spec_norm.cpu().numpy()[keep])
    assert np.isfinite(float(m.loss_hole)) and m.ctc_loss is None
    assert m.enhanced_sources.shape == (B, N)
    with pytest.raises(ValueError):
        m.feed_labels(np.full((B, 4), C - 1.0, np.float32), np.full(B, 2))     # the blank is not a label
    with pytest.raises(ValueError):
        m.feed_labels(np.zeros((B, 4), np.float32), np.full(B, 5))
    m.is_training = True
    m.feed(seq, wav, masks)
    with pytest.raises(_lib.AvsiError):
        m.loss_func
