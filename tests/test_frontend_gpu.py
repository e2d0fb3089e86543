"""Parity of the fused gfx950 front-end kernel (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import frontend as OF

pytestmark = pytest.mark.gpu


def _wav(B, N, seed):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)


def _rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - b) ** 2)))


@pytest.fixture(scope="module")
def ap():
    import avsi_amd
    from avsi_amd import audio_processing
    return audio_processing


@pytest.mark.parametrize("B,N", [(3, 48000), (2, 47999), (1, 5000), (5, 192 * 16), (2, 192 * 17 + 1)])
def test_stft_matches_oracle(ap, B, N):
    wav = _wav(B, N, 0)
    got = ap.get_stft(torch.from_numpy(wav).cuda(), window_size=24, step_size=12).cpu().numpy()
    ref = OF.get_stft(wav, window_size=24, step_size=12)
    assert got.shape == ref.shape and got.dtype == np.complex64
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() / scale < 2e-6


def test_default_geometry_25ms_10ms(ap):
    wav = _wav(2, 16000, 1)
    got = ap.get_stft(torch.from_numpy(wav).cuda()).cpu().numpy()     # 400 / 160 / 512
    ref = OF.get_stft(wav)
    assert got.shape == ref.shape == (2, 100, 257)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


def test_out_shape_slice(ap):
    wav = _wav(2, 9600, 2)
    got = ap.get_stft(torch.from_numpy(wav).cuda(), window_size=24, step_size=12,
                      out_shape=[2, 40, 128]).cpu().numpy()
    ref = OF.get_stft(wav, window_size=24, step_size=12, out_shape=(2, 40, 128))
    assert got.shape == (2, 40, 128)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


def test_inpainter_frontend_all_outputs(ap):
    """models.py:30-35 chain + log-mel, tolerance of BASELINE.json: RMS <= 1e-3 (we ask 1e-4)."""
    B, N = 4, 48000
    wav = _wav(B, N, 3)
    rng = np.random.default_rng(4)
    masks = np.ones((B, 250, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, 250 - 33)
        masks[b, s:s + 33] = 0
    feats64 = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(feats64))
    out = ap.frontend(torch.from_numpy(wav).cuda(), mean=torch.from_numpy(mean).cuda(),
                      std=torch.from_numpy(std).cuda(), masks=torch.from_numpy(masks).cuda(),
                      want_stft=True, want_spec=True, want_feat=True, want_logmel=True)
    _, norm, feat = OF.inpainter_frontend(wav, mean, std, masks)
    assert _rms(out['spec'].cpu().numpy(), norm) < 1e-4
    assert _rms(out['feat'].cpu().numpy(), feat) < 1e-4
    assert np.abs(out['spec'].cpu().numpy() - norm).max() < 5e-3     # near-silent bins amplify
    p = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), power=2)
    lm = OF.get_log_mel_spectrogram(p)
    assert out['logmel'].shape == (B, 250, 80)
    assert _rms(out['logmel'].cpu().numpy(), lm) < 1e-4
    # gap frames are exactly zero in the masked features
    got = out['feat'].cpu().numpy()
    assert np.all(got[masks == 0] == 0)


def test_time_major_padded_layout(ap):
    B, N = 3, 9600
    wav = _wav(B, N, 5)
    masks = np.ones((B, 50, 257), dtype=np.float32)
    masks[:, 10:20] = 0
    mean = np.zeros(257, dtype=np.float32)
    std = np.ones(257, dtype=np.float32)
    w = torch.from_numpy(wav).cuda()
    a = ap.frontend(w, mean=torch.from_numpy(mean).cuda(), std=torch.from_numpy(std).cuda(),
                    masks=torch.from_numpy(masks).cuda(), want_feat=True)['feat']
    b = ap.frontend(w, mean=torch.from_numpy(mean).cuda(), std=torch.from_numpy(std).cuda(),
                    masks=torch.from_numpy(masks).cuda(), want_feat=True, time_major=True,
                    feat_cols=264)['feat']
    assert b.shape == (50, B, 264)
    assert torch.equal(b[:, :, :257].transpose(0, 1), a)
    assert torch.all(b[:, :, 257:] == 0)


def test_real_clip_shape_silence_is_finite(ap):
    """All-zero input: log(0 + 1e-6) everywhere, no NaN/Inf (reference eps semantics)."""
    w = torch.zeros(1, 4800, device='cuda')
    sp = ap.frontend(w, want_spec=True)['spec']
    assert torch.isfinite(sp).all()
    assert torch.allclose(sp, torch.full_like(sp, float(np.log(1e-6))), atol=1e-5)


def test_bad_arguments_raise(ap):
    import avsi_amd
    w = torch.zeros(1, 4800, device='cuda')
    with pytest.raises(avsi_amd._lib.AvsiError):
        ap.frontend(w, n_fft=1024, want_spec=True)
    with pytest.raises(avsi_amd._lib.AvsiError):
        ap.frontend(w, num_frames_out=1000, want_spec=True)


def test_fft_length_256_unet_geometry(ap):
    """U-Net front end (models.py:537): 16 ms / 8 ms / n_fft 256 -> 129 bins, sliced to 128."""
    wav = _wav(2, 16384, 7)
    w = torch.from_numpy(wav).cuda()
    got = ap.get_stft(w, window_size=16, step_size=8, n_fft=256).cpu().numpy()
    ref = OF.get_stft(wav, window_size=16, step_size=8, n_fft=256)
    assert got.shape == ref.shape == (2, 128, 129)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6
    rng = np.random.default_rng(8)
    mean, std = rng.normal(5, 1, 128).astype(np.float32), rng.uniform(1, 2, 128).astype(np.float32)
    masks = (rng.uniform(size=(2, 128, 128)) > 0.2).astype(np.float32)
    out = ap.frontend(w, window_size=16, step_size=8, n_fft=256, num_bins=128, mean=torch.from_numpy(mean).cuda(),
                      std=torch.from_numpy(std).cuda(), masks=torch.from_numpy(masks).cuda(), want_spec=True, want_feat=True)
    spec = (OF.get_spectrogram(ref[:, :, :128], log=True) - mean) / std
    assert _rms(out['spec'].cpu().numpy(), spec) < 1e-4
    assert _rms(out['feat'].cpu().numpy(), spec * masks) < 1e-4
    # inverse: fft length inferred from 129 bins
    rec = ap.reconstruct_sources(torch.from_numpy(ref.astype(np.complex64)).cuda(), 16384, window_size=16, step_size=8)
    ref_rec = OF.reconstruct_sources(ref, 16384, window_size=16, step_size=8)
    assert np.abs(rec.cpu().numpy() - ref_rec).max() < 2e-2


@pytest.mark.parametrize("B,N,T", [(3, 48000, 250), (2, 47999, 250), (5, 192 * 16, 16), (2, 192 * 17 + 1, 18), (70, 9600, 50), (3, 48000, 201)])
@pytest.mark.parametrize("want_grad", [False, True])
def test_l1_loss_from_the_waveform_equals_the_loss_on_the_stored_target(ap, B, N, T, want_grad):
    """ap.frontend_l1_loss (avsi_frontend_l1_loss_f32: the front-end transform run again, its epilogue comparing with the
    prediction instead of storing the normalised target) against ops.l1_loss on the target the front end stores: the same
    three losses to summation order, the same gradient signs element by element (a prediction equal to the target gives 0 on
    both paths; a few elements may differ where |p - t| is a rounding of the two transforms' last bit: none here, same
    kernel code); batch-major prediction, masks longer than T frames, partial last tiles and samples ending mid-tile."""
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    rng = np.random.default_rng(B * 1000 + T)
    wav = torch.from_numpy(_wav(B, N, 3)).cuda()
    Tm = T + 3
    masks = torch.ones(B, Tm, 257, device='cuda')
    for b in range(B):
        s0 = int(rng.integers(0, max(1, T - 5)))
        masks[b, s0:s0 + 4] = 0
    mean = torch.from_numpy(rng.normal(5, 1, 257).astype(np.float32)).cuda()
    std = torch.from_numpy((1 + rng.random(257)).astype(np.float32)).cuda()
    fe = ap.frontend(wav, num_frames_out=T, mean=mean, std=std, masks=masks, want_spec=True, want_feat=True)
    tgt = fe['spec']
    pred = (tgt + torch.from_numpy(rng.normal(0, 0.3, size=(B, T, 257)).astype(np.float32)).cuda()).contiguous()
    pred[0, 0, :5] = tgt[0, 0, :5]                       # exact zeros of p - t
    ref3, refd = ops.l1_loss(tgt, pred, masks[:, :T].contiguous(), want_grad=want_grad)
    got = ap.frontend_l1_loss(wav, pred, masks, mean, std, want_grad=want_grad)
    assert got is not None
    out3, dpred = got
    np.testing.assert_allclose(out3.cpu().numpy(), ref3.cpu().numpy(), rtol=2e-5)
    if want_grad:
        # sign(p - t) / n element by element; the two instantiations of the kernel may round the target's last bit differently
        # (another fused-multiply-add contraction), which can only flip a sign where p and t agree to that bit
        diff = dpred != refd
        assert float(diff.float().mean()) < 1e-3          # (the five planted zeros of p - t are such places)
        assert bool(((pred - tgt).abs()[diff] < 1e-5).all())
        assert float(dpred.abs().max()) == float(refd.abs().max())
    else:
        assert dpred is None
    # the features-only launch (no target output) writes the same masked features as the two-output launch
    only = ap.frontend(wav, num_frames_out=T, mean=mean, std=std, masks=masks, want_feat=True)
    assert 'spec' not in only and torch.equal(only['feat'], fe['feat'])
    # geometries the loss form does not take: None, the caller falls back to the stored target
    assert ap.frontend_l1_loss(wav, pred, masks, mean, std, window_size=16, step_size=8, n_fft=256) is None
