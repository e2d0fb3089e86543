"""tf.nn.dropout on the last BLSTM layer's output (reference models.py:117).  TensorFlow's random stream cannot be
reproduced, so the kernel writes out the factor it drew per element and the oracle is fed the same factors."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as OF

pytestmark = pytest.mark.gpu


def test_dropout_kernel_statistics_and_reproducibility():
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    x = torch.randn(4000, 512, device='cuda')
    y, sc = torch.empty_like(x), torch.empty_like(x)
    ops.dropout(x, y, sc, 500, 0.3, seed=1234)
    live = sc[:, :500]
    kept = (live > 0).float().mean().item()
    assert abs(kept - 0.7) < 5e-3                                              # 2 M draws
    assert torch.all((live == 0) | ((live - 1 / 0.7).abs() < 1e-6))
    assert torch.equal(y[:, :500], x[:, :500] * live)
    y2, sc2 = torch.empty_like(x), torch.empty_like(x)
    ops.dropout(x, y2, sc2, 500, 0.3, seed=1234)
    assert torch.equal(sc2[:, :500], live)                                      # same seed, same draw
    ops.dropout(x, y2, sc2, 500, 0.3, seed=1235)
    assert not torch.equal(sc2[:, :500], live)
    # neighbouring elements are not correlated
    a, b = (live[:, :-1] > 0).float(), (live[:, 1:] > 0).float()
    assert abs(((a * b).mean() - a.mean() * b.mean()).item()) < 2e-3


@pytest.mark.parametrize("input_type", ["a", "av"])
def test_model_with_dropout_matches_oracle(input_type):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    B, N = 5, 3840
    T = N // 192
    rng = np.random.default_rng(8)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    masks = np.ones((B, T, 257), dtype=np.float32)
    masks[:, 7:12] = 0
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = (a.astype(np.float32) for a in OF.feature_stats(list(spec)))
    D = {'a': 257, 'av': 393}[input_type]
    p = O.init_params(9, D)
    seq = np.full(B, T)
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
               starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    m = models.StackedBLSTMModel(seq, wav, masks, mean, std, 0.4, cfg, video_features=video, input=input_type)
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    pred = m.prediction.cpu().numpy()
    sc = m._cache['drop_scale'].cpu().numpy()                                    # [T][Bp][512]: fw 0..249, bw 256..505
    drop = np.concatenate([sc[:, :B, :250], sc[:, :B, 256:506]], axis=2).transpose(1, 0, 2)
    assert 0.5 < (drop > 0).mean() < 0.7
    ref = O.model_forward(wav, masks, mean, std, seq, p, video=video, input_type=input_type, keep=True, drop_scale=drop)
    assert np.sqrt(np.mean((pred - ref['prediction']) ** 2)) < 1e-4
    assert float(m.loss_func) == pytest.approx(ref['loss_func'], rel=2e-4)
    g_ref = m.layout.flatten_oracle_params(O.model_backward(ref, masks.astype(np.float64), seq)).astype(np.float64)
    g = m.gradients.cpu().numpy()
    assert np.abs(g - g_ref).max() < 2e-3 * np.abs(g_ref).max()
    # rate 0 for the next feed (validation): the identity again, and no stale factors
    m.set_dropout_rate(0.0)
    m.feed(sequence_lengths=seq, target_sources=wav, masks=masks, video_features=video)
    ref0 = O.model_forward(wav, masks, mean, std, seq, p, video=video, input_type=input_type)
    assert np.sqrt(np.mean((m.prediction.cpu().numpy() - ref0['prediction']) ** 2)) < 1e-4
    assert m._cache.get('drop_scale') is None
