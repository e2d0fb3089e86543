"""The variants oracle (torch autograd) against the plain-model oracle (numpy, manual BPTT)."""
import numpy as np

from oracle import blstm as OB
from oracle import frontend as OF
from oracle import variants as OV


def _case(B=2, N=1920, seed=0):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    masks = np.ones((B, T, 257), dtype=np.float32)
    masks[:, 3:6] = 0
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(spec))
    return wav, masks, mean, std, np.full(B, T, np.int32), T


def test_embedding_at_layer0_equals_plain_model_on_concatenated_input():
    wav, masks, mean, std, seq, T = _case()
    E, H = 16, 12
    rng = np.random.default_rng(1)
    emb = rng.normal(size=(2, E))
    params = OV.init_variant_params(2, 257, 0, E, net_dim=(H, H))
    tp = OV.params_to_torch(params)
    out = OV.variant_forward(wav, masks, mean, std, seq, tp, 0, embeddings=emb)
    out['loss'].backward()
    # plain oracle on [features | tiled embedding]
    _, norm, feats = OF.inpainter_frontend(wav, mean, std, masks.astype(np.float64), np.float64, audio_feat_dim=257, max_len=T)
    x = np.concatenate([feats, np.repeat(emb[:, None, :], T, axis=1)], axis=2)
    p = OB.cast_params(params, np.float64)
    logits, rnn, caches = OB.inference(x, p, True)
    np.testing.assert_allclose(out['inference'].detach().numpy(), logits, rtol=1e-10, atol=1e-12)
    # loss_hole gradient through the blended prediction, by hand, into the manual BPTT of the plain oracle
    m = masks.astype(np.float64)
    pred = norm * m + logits * (1 - m)
    np.testing.assert_allclose(out['prediction'].detach().numpy(), pred, rtol=1e-10, atol=1e-12)
    dlogits = np.sign(pred - norm) * (1 - m) * (1 - m) / (1 - m).sum()
    B, F = 2, 257
    dl2 = dlogits.reshape(B * T, F)
    gk = OV.grads_to_numpy(tp)
    np.testing.assert_allclose(gk['proj']['weights'], rnn.reshape(B * T, -1).T @ dl2, rtol=1e-8, atol=1e-12)
    dout = (dl2 @ p['proj']['weights'].T).reshape(B, T, -1)
    _, dk_f, db_f = OB._lstm_direction_bwd(caches[1]['input'], p['layers'][1]['fw']['kernel'], caches[1]['fw'], dout[:, :, :H])
    np.testing.assert_allclose(gk['layers'][1]['fw']['kernel'], dk_f, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(gk['layers'][1]['fw']['bias'], db_f, rtol=1e-7, atol=1e-12)


def test_delta_features_match_frontend_oracle():
    import torch
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 9, 5))
    got = OV.delta_features(torch.tensor(x)).numpy()
    want = OF.add_delta_features(x, n_delta=1, N=2)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_speaker_embedding_average_uses_count_plus_one():
    import torch
    rng = np.random.default_rng(4)
    B, T, F, E = 2, 6, 7, 5
    feats = torch.tensor(rng.normal(size=(B, T, F)))
    masks = torch.ones(B, T, F, dtype=torch.float64)
    masks[0, 2:4] = 0
    mlp = {k: torch.tensor(v, dtype=torch.float64) for k, v in
           OV.init_variant_params(0, F, 0, E, net_dim=(4,), audio_feat_dim=F, mlp=True)['mlp'].items()}
    e = OV.speaker_embedding(feats, masks, mlp).numpy()
    # by hand for utterance 0: masked frames dropped, divisor = kept frames + 1 (models.py:832-833)
    inp = OV.delta_features(feats)[0].numpy()
    lr = lambda z: np.where(z > 0, z, 0.3 * z)
    w = {k: v.numpy() for k, v in mlp.items()}
    out = lr(lr(inp @ w['weights_1'] + w['biases_1']) @ w['weights_2'] + w['biases_2']) @ w['weights_3'] + w['biases_3']
    keep = np.array([1, 1, 0, 0, 1, 1.0])
    np.testing.assert_allclose(e[0], (out * keep[:, None]).sum(0) / (keep.sum() + 1), rtol=1e-12)
