"""Committed golden vectors (tests/golden/hotpath_golden.npz, made by make_hotpath_golden.py):
the CPU oracle must keep reproducing them (no GPU), and the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

from oracle import blstm as O
from oracle import frontend as OF

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "hotpath_golden.npz"))


def _params(D):
    rng_seed, bias_seed = int(G['param_seed']), int(G['bias_seed'])
    p = O.init_params(rng_seed, D)
    rng = np.random.default_rng(bias_seed)
    for layer in p['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = rng.normal(0, 0.1, size=layer[d]['bias'].shape).astype(np.float32)
    p['proj']['biases'] = rng.normal(0, 0.1, size=p['proj']['biases'].shape).astype(np.float32)
    return p


def _cfg():
    return dict(audio_feat_dim=257, video_feat_dim=136, audio_len=int(G['wav'].shape[1]), net_dim=[250, 250, 250],
                optimizer_type='adam', starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0,
                batch_size=3, l2=0.0)


@pytest.mark.parametrize("kind,D", [("a", 257), ("av", 393)])
def test_oracle_reproduces_golden(kind, D):
    fwd = O.model_forward(G['wav'], G['masks'], G['mean'], G['std'], G['seq_len'], _params(D), video=G['video'],
                          input_type=kind)
    np.testing.assert_allclose(fwd['prediction'], G[kind + '_prediction'], atol=2e-6)
    np.testing.assert_allclose([fwd['loss_func'], fwd['loss_hole'], fwd['loss_valid']], G[kind + '_losses'], rtol=1e-9)
    if kind == 'a':
        np.testing.assert_allclose(fwd['target_spec_norm'], G['target_spec_norm'], atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D", [("a", 257), ("av", 393)])
def test_hip_path_matches_golden(kind, D):
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import audio_processing as ap
    from avsi_amd import models
    m = models.StackedBLSTMModel(G['seq_len'], G['wav'], G['masks'], G['mean'], G['std'], 0.0, _cfg(),
                                 video_features=G['video'], input=kind)
    m.variables.load_flat(m.layout.flatten_oracle_params(_params(D)))
    pred = m.prediction.cpu().numpy()
    assert np.sqrt(np.mean((pred - G[kind + '_prediction']) ** 2)) < 1e-4
    got_lm = OF.logmel_of_prediction(pred.astype(np.float64), G['mean'], G['std'])
    assert np.sqrt(np.mean((got_lm - G[kind + '_logmel']) ** 2)) < 1e-3          # BASELINE.json tolerance
    np.testing.assert_allclose([float(m.loss_func), float(m.loss_hole), float(m.loss_valid)], G[kind + '_losses'],
                               rtol=2e-4)
    grads = m.gradients.cpu().numpy().astype(np.float64)
    l2 = np.array([np.sqrt((grads[off:off + int(np.prod(shape))] ** 2).sum()) for _, shape, off in m.layout.ref_entries])
    np.testing.assert_allclose(l2, G[kind + '_grad_l2'], rtol=2e-3)
    np.testing.assert_allclose(m.layout.ref_view(grads, 'logits/biases'), G[kind + '_grad_proj_bias'], atol=2e-6)
    if kind == 'a':
        assert np.sqrt(np.mean((m.target_spec_norm.cpu().numpy() - G['target_spec_norm']) ** 2)) < 1e-4
        lm = ap.frontend(torch.from_numpy(G['wav']).cuda(), want_logmel=True)['logmel'].cpu().numpy()
        assert np.sqrt(np.mean((lm - G['target_logmel']) ** 2)) < 1e-4
        scale = np.abs(G['enhanced_oracle']).max()
        assert np.abs(m.enhanced_sources_oracle_phase.cpu().numpy() - G['enhanced_oracle']).max() < 1e-3 * scale
        assert np.sqrt(np.mean((m.enhanced_sources.cpu().numpy() - G['enhanced_masked']) ** 2)) < 1e-3 * scale
        losses = []
        for _ in range(3):
            m.feed(sequence_lengths=G['seq_len'], target_sources=G['wav'], masks=G['masks'])
            losses.append(float(m.loss))
            m.train_op
        np.testing.assert_allclose(losses, G['a_adam_losses'], rtol=5e-4)
