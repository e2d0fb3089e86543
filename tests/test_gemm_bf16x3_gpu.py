"""EXPLORATORY split-bf16 GEMM (avsi_gemm_bf16x3_f32: hi.hi + hi.lo + lo.hi on the bf16 matrix cores, fp32 accumulation)
against numpy float64, and the model with config['precision'] = 'bf16x3' against the oracle at the BASELINE tolerance
(1e-3 RMS on the reconstructed log-mel).  Never the default path."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as OF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K", [(128, 32), (1000, 272), (8200, 512), (300, 400), (16384, 512)])
def test_gemm_bf16x3_matches_numpy(M, K):
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    N = 2048 if M > 500 else 256
    rng = np.random.default_rng(M + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = (rng.normal(size=(K, N)) * 0.1).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    out = torch.full((M, N), 7.0, device='cuda')
    ops.gemm_bf16x3(torch.from_numpy(A).cuda(), ops.pack_bf16x3_b(torch.from_numpy(B).cuda()), out, K, bias=torch.from_numpy(bias).cuda())
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    err = np.abs(out.cpu().numpy() - ref)
    scale = np.sqrt(K) * 0.1                      # typical magnitude of a dot product here
    # products carry 16+ significant bits: relative error ~2^-16 per term, random signs
    assert err.max() < 2e-4 * scale and np.sqrt(np.mean(err ** 2)) < 2e-5 * scale
    # and it IS less exact than the fp32 path (a guard against silently testing the wrong kernel)
    exact = ops.gemm(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), bias=torch.from_numpy(bias).cuda())
    assert np.sqrt(np.mean((exact.cpu().numpy() - ref) ** 2)) < np.sqrt(np.mean(err ** 2))


def test_model_with_bf16x3_projections_meets_the_baseline_tolerance():
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    B, N = 33, 48000
    rng = np.random.default_rng(3)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = 250
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, T - 33)
        masks[b, s:s + 33] = 0
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = (a.astype(np.float32) for a in OF.feature_stats(list(spec)))
    p = O.init_params(7, 257)
    seq = np.full(B, T)
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
               starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0, precision='bf16x3')
    m = models.StackedBLSTMModel(seq, wav, masks, mean, std, 0.0, cfg, input='a', is_training=False)
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    ref = O.model_forward(wav, masks, mean, std, seq, p)
    pred = m.prediction.cpu().numpy().astype(np.float64)
    lm = OF.logmel_of_prediction(pred, mean, std) - OF.logmel_of_prediction(ref['prediction'], mean, std)
    rms = float(np.sqrt(np.mean(lm ** 2)))
    assert rms < 1e-3, rms                                  # BASELINE.json tolerance
    assert np.sqrt(np.mean((pred - ref['prediction']) ** 2)) < 1e-3
