"""Pin the front-end oracle against independent CPU implementations and known answers
(SURVEY.md 8(c) anchors).  No GPU."""
import numpy as np
import pytest
import torch

from oracle import frontend as F


def _wav(B=3, N=48000, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767)


def test_frame_count_and_padding_anchor():
    assert F.num_frames(48000, 192) == 250
    assert (250 - 1) * 192 + 384 - 48000 == 192      # pad_end adds exactly 192 zeros
    assert F.ms_to_samples(24, 16000) == 384 and F.ms_to_samples(12, 16000) == 192
    assert F.ms_to_samples(25, 16000) == 400 and F.ms_to_samples(10, 16000) == 160


def test_stft_shape_and_direct_dft():
    x = _wav(2, 4000)
    X = F.stft(x, 384, 192, 512)
    T = F.num_frames(4000, 192)
    assert X.shape == (2, T, 257)
    # brute-force DFT of one frame, including the zero-padded tail frame
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(384) / 384)
    for t in (0, 5, T - 1):
        seg = np.zeros(384)
        src = x[1, t * 192: t * 192 + 384]
        seg[: len(src)] = src
        n = np.arange(384)
        k = np.arange(257)[:, None]
        ref = ((seg * w)[None, :] * np.exp(-2j * np.pi * k * n / 512)).sum(axis=1)
        np.testing.assert_allclose(X[1, t], ref, rtol=0, atol=1e-6 * np.abs(ref).max())


def test_stft_matches_torch_stft_with_offset_trick():
    """App. A.1 cross-check: torch centres the 384 window inside the 512 buffer (offset 64)."""
    x = _wav(2, 48000, seed=1)
    X = F.stft(x, 384, 192, 512)
    T = X.shape[1]
    padded = np.zeros((2, 64 + (T - 1) * 192 + 384 + 64))
    padded[:, 64:64 + 48000] = x
    Y = torch.stft(torch.from_numpy(padded), n_fft=512, hop_length=192, win_length=384,
                   window=torch.hann_window(384, periodic=True, dtype=torch.float64),
                   center=False, return_complex=True).numpy().transpose(0, 2, 1)[:, :T]
    np.testing.assert_allclose(np.abs(X), np.abs(Y), rtol=0, atol=1e-9 * np.abs(X).max())
    k = np.arange(257)
    np.testing.assert_allclose(X, Y * np.exp(2j * np.pi * k * 64 / 512)[None, None, :],
                               rtol=0, atol=1e-9 * np.abs(X).max())


def test_mel_matrix_anchors():
    w = F.mel_weight_matrix()
    assert w.shape == (257, 80)
    assert int((w > 0).sum()) == 470
    assert np.all(w[0] == 0)
    assert abs(w.max() - 0.99771) < 1e-5
    assert np.all(w >= 0) and np.all(w <= 1)


def test_logspec_and_logmel_f32_vs_f64():
    x = _wav(2, 9600, seed=2)
    s64 = F.get_spectrogram(F.get_stft(x, window_size=24, step_size=12), log=True)
    s32 = F.get_spectrogram(F.get_stft(x, window_size=24, step_size=12, dtype=np.float32),
                            log=True, dtype=np.float32)
    assert s32.dtype == np.float32
    assert np.sqrt(np.mean((s32 - s64) ** 2)) < 1e-5
    p64 = F.get_spectrogram(F.get_stft(x, window_size=24, step_size=12), power=2)
    m64 = F.get_log_mel_spectrogram(p64)
    assert m64.shape == (2, 50, 80)
    # dense product equals the sparse triangular accumulation
    w = F.mel_weight_matrix()
    ref = np.log(np.einsum('btf,fm->btm', p64, w) + 1e-6)
    np.testing.assert_allclose(m64, ref, rtol=1e-12)


def test_mfcc_matches_scipy_dct():
    from scipy.fft import dct
    rng = np.random.default_rng(3)
    lm = rng.normal(size=(2, 7, 80))
    ref = dct(lm, type=2, axis=-1) / np.sqrt(2 * 80)
    np.testing.assert_allclose(F.get_mfcc(lm, 13), ref[..., :13], atol=1e-12)


def test_delta_small_case():
    x = np.arange(6, dtype=np.float64).reshape(1, 6, 1) ** 2
    d = F.delta(x, N=2)
    # interior point t=2: (1*(x3-x1) + 2*(x4-x0)) / 10
    assert d[0, 2, 0] == pytest.approx((1 * (9 - 1) + 2 * (16 - 0)) / 10)
    # edge t=0: SYMMETRIC padding applied cumulatively repeats the edge sample, so the
    # padded row is [x0, x0, x0, x1, x2, ...] for i=2 -> (1*(x1-x0) + 2*(x2-x0)) / 10
    assert d[0, 0, 0] == pytest.approx((1 * (1 - 0) + 2 * (4 - 0)) / 10)
    assert F.add_delta_features(x, 2).shape == (1, 6, 3)


def test_preemphasis():
    x = np.array([[1.0, 2.0, 3.0]])
    np.testing.assert_allclose(F.preemphasis(x, 0.95), [[1.0, 2 - 0.95, 3 - 1.9]])


def test_istft_roundtrip_anchor():
    """App. A.6 / SURVEY 8(c): identity on [192, 48000), NOT on [0, 192)."""
    x = _wav(1, 48000, seed=4)
    X = F.stft(x, 384, 192, 512)
    y = F.reconstruct_sources(X, 48000, window_size=24, step_size=12)
    assert y.shape == (1, 48000)
    assert np.abs(y[:, 192:] - x[:, 192:]).max() < 1e-7
    assert np.abs(y[:, 1:192] - x[:, 1:192]).max() > 1.0
    w = F.hann_periodic(384)
    np.testing.assert_allclose(y[0, :192], x[0, :192] * w[:192] ** 2 / (w[:192] ** 2 + w[192:] ** 2),
                               atol=1e-7)


def test_inverse_stft_fft_length_is_enclosing_power_of_two_of_the_frame():
    """TF 1.x inverse_stft(fft_length=None) -> _enclosing_power_of_two(frame_length), and irfft crops / pads the
    bins to fft_length / 2 + 1.  Checked against torch.fft.irfft + explicit overlap-add for the two cases where
    this differs from (F - 1) * 2: the 16 / 8 ms default of reconstruct_sources on a 257-bin spectrogram
    (audio_processing.py:145) and a sliced 128-bin U-Net spectrogram."""
    import torch
    assert [F.enclosing_power_of_two(n) for n in (1, 2, 3, 256, 257, 384, 512)] == [1, 2, 4, 256, 512, 512, 512]
    rng = np.random.default_rng(11)
    for bins, L, S, nfft in ((257, 256, 128, 256), (128, 256, 128, 256), (257, 384, 192, 512), (100, 384, 192, 512)):
        X = rng.normal(size=(2, 9, bins)) + 1j * rng.normal(size=(2, 9, bins))
        y = F.inverse_stft(X, L, S)
        nb = nfft // 2 + 1
        Xc = np.zeros((2, 9, nb), dtype=complex)
        Xc[..., :min(bins, nb)] = X[..., :nb]
        fr = torch.fft.irfft(torch.from_numpy(Xc), n=nfft, dim=-1).numpy()[..., :L] * F.inverse_stft_window(L, S)
        ref = np.zeros((2, 8 * S + L))
        for t in range(9):
            ref[:, t * S: t * S + L] += fr[:, t]
        np.testing.assert_allclose(y, ref, atol=1e-12)
    # the default call on 257 bins is a 256-point transform of bins 0..128
    X = rng.normal(size=(1, 5, 257)) + 1j * rng.normal(size=(1, 5, 257))
    np.testing.assert_allclose(F.reconstruct_sources(X), F.reconstruct_sources(X[..., :129]), atol=0)


def test_feature_stats():
    rng = np.random.default_rng(5)
    feats = [rng.normal(2.0, 3.0, size=(40, 5)) for _ in range(4)]
    mean, std = F.feature_stats(feats)
    allf = np.concatenate(feats)
    np.testing.assert_allclose(mean, allf.mean(0))
    np.testing.assert_allclose(std, allf.std(0))
