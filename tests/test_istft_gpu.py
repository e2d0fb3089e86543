"""Inverse STFT / waveform reconstruction kernel against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import blstm as OB
from oracle import frontend as OF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ap():
    import avsi_amd
    from avsi_amd import audio_processing
    return audio_processing


def _wav(B, N, seed):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)


@pytest.mark.parametrize("B,N", [(3, 48000), (2, 5000), (1, 192 * 15), (2, 192 * 31 + 7)])
def test_stft_istft_roundtrip_and_oracle(ap, B, N):
    wav = _wav(B, N, 1)
    X = OF.get_stft(wav, window_size=24, step_size=12)
    ref = OF.reconstruct_sources(X, N, window_size=24, step_size=12)
    st = ap.get_stft(torch.from_numpy(wav).cuda(), window_size=24, step_size=12)
    got = ap.reconstruct_sources(st, N, window_size=24, step_size=12).cpu().numpy()
    assert got.shape == (B, N)
    assert np.abs(got - ref).max() < 2e-2            # samples are ~1e4: 2e-6 relative
    assert np.abs(got[:, 192:] - wav[:, 192:]).max() < 2e-2   # identity past the first hop (App. A.6)


def test_full_length_output_when_num_samples_zero(ap):
    wav = _wav(1, 3840, 2)
    X = OF.get_stft(wav, window_size=24, step_size=12)
    ref = OF.reconstruct_sources(X, 0, window_size=24, step_size=12)
    st = torch.from_numpy(X.astype(np.complex64)).cuda()
    got = ap.reconstruct_sources(st, 0, window_size=24, step_size=12).cpu().numpy()
    assert got.shape == ref.shape == (1, 19 * 192 + 384)
    assert np.abs(got - ref).max() < 2e-2


def test_get_sources_magnitude_phase(ap):
    rng = np.random.default_rng(3)
    mag = np.abs(rng.normal(size=(2, 40, 257))).astype(np.float32) * 100
    ang = rng.uniform(-np.pi, np.pi, size=(2, 40, 257)).astype(np.float32)
    ref = OF.get_sources(mag, ang, num_samples=7000)
    got = ap.get_sources(torch.from_numpy(mag).cuda(), torch.from_numpy(ang).cuda(), num_samples=7000).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("oracle_phase", [False, True])
def test_enhanced_sources_fused_matches_oracle(ap, oracle_phase):
    """models.py:181-197 from prediction to waveform, masked and oracle phase."""
    B, N, T = 2, 9600, 50
    wav = _wav(B, N, 4)
    rng = np.random.default_rng(5)
    masks = np.ones((B, T, 257), dtype=np.float32)
    masks[:, 20:31] = 0
    mean = rng.normal(6, 1, 257).astype(np.float32)
    std = rng.uniform(1, 2, 257).astype(np.float32)
    pred = rng.normal(0, 0.5, size=(B, T, 257)).astype(np.float32)
    X = OF.get_stft(wav, window_size=24, step_size=12)
    ref = OB.enhanced_sources(pred.astype(np.float64), mean, std, X, None if oracle_phase else masks, num_samples=N)
    st = ap.get_stft(torch.from_numpy(wav).cuda(), window_size=24, step_size=12)
    got = ap.enhanced_from_prediction(torch.from_numpy(pred).cuda(), torch.from_numpy(mean).cuda(),
                                      torch.from_numpy(std).cuda(), st,
                                      None if oracle_phase else torch.from_numpy(masks).cuda(), num_samples=N)
    got = got.cpu().numpy()
    assert got.shape == (B, N)
    assert np.sqrt(np.mean((got - ref) ** 2)) < 1e-4 * np.abs(ref).max()
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()


@pytest.mark.parametrize("bins,ws,ss", [(257, 16, 8), (128, 16, 8), (129, 16, 8), (200, 24, 12)])
def test_fft_length_follows_the_frame_length_not_the_bin_count(ap, bins, ws, ss):
    """tf.contrib.signal.inverse_stft(fft_length=None): enclosing power of two of the frame length, bins cropped /
    zero-padded to fit -- e.g. reconstruct_sources' own 16 / 8 ms default on a 257-bin spectrogram is a 256-point
    transform of bins 0..128 (reference audio_processing.py:145-151)."""
    rng = np.random.default_rng(12)
    X = (rng.normal(size=(2, 21, bins)) + 1j * rng.normal(size=(2, 21, bins))) * 100
    ref = OF.reconstruct_sources(X, 0, window_size=ws, step_size=ss)
    got = ap.reconstruct_sources(torch.from_numpy(X.astype(np.complex64)).cuda(), 0, window_size=ws, step_size=ss)
    got = got.cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max() + 1e-4
