"""The LWS oracle (oracle/lws.py) against first principles: its consistency weights reproduce STFT o iSTFT exactly,
its transforms are a perfect-reconstruction pair, and its iterations do what LWS is for -- make a spectrogram with
missing phases consistent.  (UNPINNED against the `lws` package, which is not installable here.)"""
import numpy as np
import pytest

from oracle import lws as OL


def _speechlike(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    f0 = 180 + 40 * np.sin(2 * np.pi * 2.0 * t / 16000)
    x = sum(2000 / h * np.sin(2 * np.pi * h * np.cumsum(f0) / 16000) for h in range(1, 9))
    return x * (0.6 + 0.4 * np.sin(2 * np.pi * 4 * t / 16000)) + rng.normal(0, 100, n)


def test_consistency_weights_reproduce_stft_of_istft():
    """F = STFT o iSTFT written as the local sum of oracle/lws.py's docstring, untruncated (|p| <= N/2), on a small
    geometry: equal to transforming back and forth to 1e-14."""
    N, R = 16, 6
    lw = OL.LWS(12, R, L=3, fftsize=N)
    Q = lw.Q
    A = OL.consistency_weights(lw.awin, lw.swin, R, N // 2)
    rng = np.random.default_rng(0)
    S = lw.stft(rng.normal(size=200))
    M = S.shape[0]
    W = S * np.exp(1j * rng.normal(size=S.shape) * 0.5)
    W[:, 0], W[:, -1] = W[:, 0].real, W[:, -1].real
    y = np.fft.irfft(W, n=N, axis=1) * lw.swin
    sig = np.zeros((M - 1) * R + N)
    for m in range(M):
        sig[m * R:m * R + N] += y[m]
    idx = np.arange(M)[:, None] * R + np.arange(N)[None, :]
    F = np.fft.rfft(sig[idx] * lw.awin, axis=1)
    full = np.concatenate([W, np.conj(W[:, -2:0:-1])], axis=1)
    G = np.zeros_like(W)
    for m in range(M):
        for k in range(N // 2 + 1):
            for q in range(-(Q - 1), Q):
                if 0 <= m + q < M:
                    for p in range(-N // 2 + 1, N // 2 + 1):
                        G[m, k] += A[q + Q - 1, p + N // 2] * np.exp(-2j * np.pi * (k + p) * q * R / N) * full[m + q, (k + p) % N]
    assert np.abs(G - F).max() < 1e-13 * np.abs(F).max()


def test_reference_geometry():
    """lws.lws(384, 192, fftsize=512, mode='speech') (inference.py:119): 512-sample frames, 252 of them for a 3 s
    clip, rows |q| = 2 of the consistency sum vanish (the windows' supports are 384 wide), 102 sweeps."""
    lw = OL.LWS(384, 192, fftsize=512, mode='speech')
    assert lw.N == 512 and lw.Q == 3 and lw.num_frames(48000) == 252
    assert np.all(lw.awin[:64] == 0) and np.all(lw.awin[448:] == 0) and lw.awin[64 + 191] > 0.99
    assert np.abs(lw.alpha[0]).max() == 0 and np.abs(lw.alpha[4]).max() == 0
    assert lw.alpha[2, 5].real == pytest.approx(0.375, abs=2e-3) and abs(lw.alpha[2, 5].imag) < 1e-12
    sched = lw.sweep_schedule()
    assert len(sched) == 102 and sched[0] == (True, 1.0) and sched[1] == (False, 1.0)
    assert sched[2][1] == pytest.approx(100.0) and sched[-1][1] == pytest.approx(100 * np.exp(-9.9))


def test_transforms_are_a_perfect_reconstruction_pair():
    lw = OL.LWS(384, 192, fftsize=512, mode='speech')
    x = _speechlike(5000, 1)
    S = lw.stft(x)
    assert S.shape == (lw.num_frames(5000), 257)
    y = lw.istft(S)
    assert len(y) >= 5000 and np.abs(y[:5000] - x).max() < 1e-9 and np.abs(y[5000:]).max() < 1e-9
    assert lw.inconsistency(S) < 1e-25


def test_lws_makes_a_gapped_spectrogram_consistent():
    """Phases zeroed in a gap of 6 frames: the iterations bring the inconsistency down by orders of magnitude, and
    (sampled every 10 sweeps of the batch stage) never up."""
    lw = OL.LWS(384, 192, fftsize=512, mode='speech')
    S = lw.stft(_speechlike(3840, 2))
    S0 = S.copy()
    S0[8:14] = np.abs(S0[8:14])
    trace = []
    S1 = lw.run_lws(S0, trace=trace)
    np.testing.assert_allclose(np.abs(S1), np.abs(S0), rtol=1e-12)          # magnitudes are kept
    before = lw.inconsistency(S0)
    assert trace[-1] < 0.02 * before
    batch = trace[2:]
    assert all(b <= a * 1.0001 for a, b in zip(batch[::10], batch[10::10]))


def test_refine_enhanced_keeps_known_phases_and_length():
    """inference.py:141-154: outside the mask's gap the final spectrogram has the phase the waveform came with."""
    lw = OL.LWS(384, 192, fftsize=512, batch_iterations=8, nofuture_iterations=1, online_iterations=1)
    x = _speechlike(3840, 3)
    mask = np.ones((20, 257))
    mask[7:12] = 0
    out = OL.refine_enhanced(lw, x, mask)
    assert len(out) == (lw.num_frames(3840) - 1) * 192 + 512 - 640 and len(out) >= 3840
    S_in, S_out = lw.stft(x), lw.stft(out[:3840])
    keep = np.zeros(S_in.shape[0], dtype=bool)
    keep[:20] = mask[:, 0] > 0
    # frames far from the gap are untouched (their neighbourhood kept its phases)
    far = [m for m in range(S_in.shape[0]) if keep[m] and all(abs(m - g) > 3 for g in range(7, 12)) and m < 18]
    assert np.abs(S_out[far] - S_in[far]).max() < 1e-6 * np.abs(S_in).max()
