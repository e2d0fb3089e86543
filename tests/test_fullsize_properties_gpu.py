"""Full-size checks that need no oracle: the BASELINE.json configurations (3 s clips, T = 250 frames,
257 bins, 3 x BLSTM-250, batches of 256 and more) are too large for the CPU restatement to finish
in seconds, so the GPU path is held to size-independent properties of the maths instead:
round trips, linearity, permutation / padding invariance, time-reversal symmetry of the BLSTM,
agreement of independent kernel implementations, and a directional-derivative check of the
gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, T, F = 48000, 250, 257


def _cfg(B):
    return dict(audio_feat_dim=F, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
                starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)


def _batch(B, seed):
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    wav = torch.clamp(torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000.0), -32768, 32767)
    masks = torch.ones(B, T, F, device='cuda')
    starts = torch.randint(0, T - 33, (B,), generator=g, device='cuda')
    t = torch.arange(T, device='cuda')[None, :]
    masks[(t >= starts[:, None]) & (t < starts[:, None] + 33)] = 0.0
    return wav, masks


@pytest.fixture(scope="module")
def ap():
    import avsi_amd  # noqa: F401
    from avsi_amd import audio_processing
    return audio_processing


def test_stft_istft_round_trip_full_size(ap):
    """inverse_stft(stft(x)) = x on samples [192, 48000) (the first hop is covered by one frame only)."""
    wav, _ = _batch(256, 0)
    stft = ap.get_stft(wav, window_size=24, step_size=12, n_fft=512)
    assert tuple(stft.shape) == (256, T, F)
    back = ap.get_sources(stft.abs(), torch.angle(stft), num_samples=N)
    err = (back[:, 192:] - wav[:, 192:]).abs().max().item()
    assert err < 0.05, err                      # int16-scale samples, float32 FFT pair


def test_front_end_is_linear_full_size(ap):
    wav, _ = _batch(128, 1)
    x, y = wav[:64], wav[64:]
    s = ap.get_stft(1.5 * x - 0.25 * y, window_size=24, step_size=12, n_fft=512)
    sx = ap.get_stft(x, window_size=24, step_size=12, n_fft=512)
    sy = ap.get_stft(y, window_size=24, step_size=12, n_fft=512)
    ref = 1.5 * sx - 0.25 * sy
    assert (s - ref).abs().max().item() < 2e-3 * ref.abs().max().item()


def _model(B, wav, masks, seed=3, is_training=False, input='a', video=None):
    from avsi_amd import audio_processing as ap_mod
    from avsi_amd import models
    spec = ap_mod.frontend(wav[:64], want_spec=True)['spec']
    mean, std = spec.mean(dim=(0, 1)), spec.std(dim=(0, 1), unbiased=False)
    return models.StackedBLSTMModel(np.full(B, T), wav, masks, mean, std, 0.0, _cfg(B), video_features=video, input=input,
                                    seed=seed, is_training=is_training)


def test_utterances_are_independent_full_size():
    """Permuting the batch permutes the predictions; an utterance's result does not depend on its
    batch-mates, on the batch size (32-row padding, kernel choice: cooperative at 64, batch-stationary
    tiles at 4096) or on its position."""
    import avsi_amd  # noqa: F401
    B = 4096
    wav, masks = _batch(B, 2)
    m = _model(B, wav, masks)
    pred = m.prediction.clone()
    perm = torch.randperm(B, device='cuda')
    m.feed(np.full(B, T), wav[perm], masks[perm])
    pred_p = m.prediction
    # same kernels, same tiles of 32 rows but different neighbours: bit-identical per utterance
    assert torch.equal(pred_p, pred[perm])
    # a 50-utterance sub-batch (pads to 64, takes the cooperative kernels): same numbers up to summation order
    m.feed(np.full(50, T), wav[:50], masks[:50])
    small = m.prediction
    assert (small - pred[:50]).abs().max().item() < 2e-4


def test_projection_in_two_products_is_the_single_product(monkeypatch):
    """From 263 utterances on the 257-bin projection is issued as 256 bins (128 x 256 tiles) + 1 bin (32-wide tile)
    instead of five 64-column tiles (models.py:117-123 is one matmul): the same sums in the same order, so the
    prediction -- ragged lengths, sequence mask and the batch-major scatter included -- is bit-identical."""
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    B = 300
    wav, masks = _batch(B, 5)
    lens = np.full(B, T)
    lens[::7] = 113
    lens[3] = 1
    m = _model(B, wav, masks)
    m.feed(lens, wav, masks)
    assert models._PROJ_SPLIT
    split = m.prediction.clone()
    loss = float(m.loss_func)
    monkeypatch.setattr(models, '_PROJ_SPLIT', False)
    m.feed(lens, wav, masks)
    assert torch.equal(m.prediction, split)
    assert float(m.loss_func) == loss
    assert float(split[3, 1:].abs().max()) == 0.0 and float(split[0, :, 256].abs().max()) > 0.0


def test_blstm_time_reversal_symmetry_full_size():
    """Swapping the forward and backward cells of every layer and reversing the input in time
    reverses the output in time with its two halves swapped -- a property of
    stack_bidirectional_dynamic_rnn that involves every kernel of the forward path."""
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    Bp, H = 512, 250
    g = torch.Generator(device='cuda')
    g.manual_seed(11)
    lay = models.ParamLayout(F)
    v = models.BLSTMVariables(lay, seed=5)
    flat = v.flat.clone()
    # swapped model: fw <-> bw kernels and biases; layer inputs of layers >= 1 are [fw, bw] halves, so their
    # kernel rows swap too
    swapped = flat.clone()
    for li in range(3):
        kf, kb = lay.ref_view(flat, 'cell_%d/fw/kernel' % li), lay.ref_view(flat, 'cell_%d/bw/kernel' % li)
        if li > 0:
            def swap_rows(k):
                return torch.cat([k[H:2 * H], k[:H], k[2 * H:]], dim=0)
            kf, kb = swap_rows(kf), swap_rows(kb)
        lay.ref_view(swapped, 'cell_%d/fw/kernel' % li)[...] = kb
        lay.ref_view(swapped, 'cell_%d/bw/kernel' % li)[...] = kf
        lay.ref_view(swapped, 'cell_%d/fw/bias' % li)[...] = lay.ref_view(flat, 'cell_%d/bw/bias' % li)
        lay.ref_view(swapped, 'cell_%d/bw/bias' % li)[...] = lay.ref_view(flat, 'cell_%d/fw/bias' % li)
    pw = lay.ref_view(flat, 'logits/weights')
    lay.ref_view(swapped, 'logits/weights')[...] = torch.cat([pw[H:], pw[:H]], dim=0)
    wav, masks = _batch(Bp, 4)
    feats = None

    def run(params, reverse):
        nonlocal feats
        v.load_flat(params.cpu().numpy())
        m = models.StackedBLSTMModel(np.full(Bp, T), wav, masks, torch.zeros(F, device='cuda'), torch.ones(F, device='cuda'),
                                     0.0, _cfg(Bp), input='a', variables=v, is_training=False)
        if feats is None:
            feats = (m.target_spec_norm * masks).clone()
        m.feed(np.full(Bp, T), wav, masks, audio_features=feats.flip(1) if reverse else feats)
        return m.inference.clone()
    a = run(flat, False)
    b = run(swapped, True)
    assert (b.flip(1) - a).abs().max().item() < 5e-4 * max(1.0, a.abs().max().item())


def test_gradient_is_the_directional_derivative_full_size():
    """(L(theta + eps d) - L(theta - eps d)) / (2 eps) = <grad, d> for a direction d: ties BPTT, the
    split-K weight-gradient GEMMs and the column sums to the forward path at T = 250, B = 256."""
    import avsi_amd  # noqa: F401
    B = 256
    wav, masks = _batch(B, 6)
    g = torch.Generator(device='cuda')
    g.manual_seed(12)
    video = torch.randn(B, T, 136, generator=g, device='cuda')
    m = _model(B, wav, masks, is_training=True, input='av', video=video)
    grad = m.gradients.double().clone()
    theta = m.variables.flat.clone()
    # a direction with a component along the gradient (a purely random one gives a derivative of
    # |grad| / sqrt(4.4 M), below what a float32 loss can resolve) and a random component
    r = torch.randn(theta.shape, generator=g, device='cuda')
    d = (grad / grad.norm()).float() + r / r.norm()
    d /= d.norm()
    eps = 2e-2

    def loss_at(p):
        m.variables.load_flat(p.cpu().numpy())
        m.feed(np.full(B, T), wav, masks, video_features=video)
        return float(m.loss_func)
    fd = (loss_at(theta + eps * d) - loss_at(theta - eps * d)) / (2 * eps)
    an = float((grad * d.double()).sum())
    assert abs(fd - an) <= 0.05 * abs(an) + 2e-5, (fd, an)
