"""Oracle: STFT / log-spectrogram / log-mel / MFCC / inverse-STFT front end (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  STFT / |.| / inverse STFT are PINNED by the reference's
own docs/files/*/masked.wav (tests/test_ref_docs_golden.py); mel / MFCC / deltas remain unpinned.

Restates, op for op, the TensorFlow-1.x semantics behind the reference's
``av_speech_inpainting/audio_processing.py``.
Every function takes ``dtype`` (np.float64 = "truth", np.float32 = the
arithmetic type of the reference graph) and is batched over the leading axis.
"""
import math

import numpy as np


def _cdtype(dtype):
    return np.complex128 if np.dtype(dtype) == np.float64 else np.complex64


def ms_to_samples(ms, sample_rate):
    """audio_processing.py:27-28 -- int(round(ms / 1e3 * sample_rate))."""
    return int(round(ms / 1e3 * sample_rate))


def hann_periodic(length, dtype=np.float64):
    """tf.contrib.signal.hann_window(periodic=True): 0.5 - 0.5 cos(2 pi n / L)."""
    n = np.arange(length, dtype=dtype)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / length)).astype(dtype)


def num_frames(num_samples, frame_step):
    """tf.contrib.signal.frame(pad_end=True): T = ceil(N / step)."""
    return -(-num_samples // frame_step)


def stft(signals, frame_length, frame_step, fft_length, dtype=np.float64):
    """tf.contrib.signal.stft(..., pad_end=True) (audio_processing.py:35-36; App. A.1).

    signals [B, N] -> complex [B, T, fft_length // 2 + 1].
    """
    x = np.asarray(signals, dtype=dtype)
    B, N = x.shape
    T = num_frames(N, frame_step)
    padded = np.zeros((B, (T - 1) * frame_step + frame_length), dtype=dtype)
    padded[:, :N] = x
    idx = np.arange(T)[:, None] * frame_step + np.arange(frame_length)[None, :]
    frames = padded[:, idx] * hann_periodic(frame_length, dtype)[None, None, :]
    # rfft(n=fft_length) zero-pads the windowed frame at the END to fft_length.
    return np.fft.rfft(frames, n=fft_length, axis=-1).astype(_cdtype(dtype))


def _maybe_slice(x, out_shape):
    """tf.cond(all(out_shape == 0), identity, tf.slice(begin 0, size out_shape))."""
    if out_shape is None or all(int(s) == 0 for s in out_shape):
        return x
    return x[: out_shape[0], : out_shape[1], : out_shape[2]]


def get_stft(sources, sample_rate=16000, window_size=25, step_size=10, n_fft=512,
             out_shape=(0, 0, 0), dtype=np.float64):
    """audio_processing.py:25-42."""
    L = ms_to_samples(window_size, sample_rate)
    S = ms_to_samples(step_size, sample_rate)
    return _maybe_slice(stft(sources, L, S, n_fft, dtype), out_shape)


def get_spectrogram(stfts, power=1, log=False, out_shape=(0, 0, 0), dtype=np.float64):
    """audio_processing.py:45-56 -- |X| (**power) (log(. + 1e-6))."""
    spec = np.abs(stfts).astype(dtype)
    if power != 1:
        spec = spec ** dtype(power)
    if log:
        spec = np.log(spec + dtype(1e-6))
    return _maybe_slice(spec.astype(dtype), out_shape)


def hertz_to_mel(f):
    """HTK mel scale used by tf.signal.linear_to_mel_weight_matrix."""
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_weight_matrix(num_mel_bins=80, num_spec_bins=257, sample_rate=16000,
                      lower_edge_freq=125.0, upper_edge_freq=7600.0, dtype=np.float64):
    """tf.signal.linear_to_mel_weight_matrix (audio_processing.py:63-64; App. A.3).

    Returns [num_spec_bins, num_mel_bins]; un-normalised triangles, DC row zero.
    """
    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spec_bins)[1:]
    spec_mel = hertz_to_mel(lin)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_freq), hertz_to_mel(upper_edge_freq),
                        num_mel_bins + 2)
    lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[1, 0], [0, 0]]).astype(dtype)


def get_log_mel_spectrogram(spectrograms, sample_rate=16000, num_spec_bins=257, num_mel_bins=80,
                            lower_edge_freq=125, upper_edge_freq=7600, eps=1e-6,
                            out_shape=(0, 0, 0), dtype=np.float64):
    """audio_processing.py:59-72.  NB the reference discards its out_shape slice (B6)."""
    if upper_edge_freq is None:
        upper_edge_freq = sample_rate / 2
    w = mel_weight_matrix(num_mel_bins, num_spec_bins, sample_rate, lower_edge_freq,
                          upper_edge_freq, dtype)
    mel = np.tensordot(np.asarray(spectrograms, dtype=dtype), w, axes=1)
    return np.log(mel + dtype(eps)).astype(dtype)


def get_mfcc(log_mel, num_mfccs=13, out_shape=(0, 0, 0), dtype=np.float64):
    """tf.signal.mfccs_from_log_mel_spectrograms(...)[..., :num_mfccs]
    (audio_processing.py:75-82; App. A.4): DCT-II (un-normalised, factor 2) * rsqrt(2 M)."""
    x = np.asarray(log_mel, dtype=dtype)
    M = x.shape[-1]
    n = np.arange(M, dtype=np.float64)
    k = np.arange(M, dtype=np.float64)[:, None]
    basis = 2.0 * np.cos(np.pi * k * (2.0 * n + 1.0) / (2.0 * M))      # [k, n]
    dct = x @ basis.T.astype(dtype)
    mfcc = dct * dtype(1.0 / math.sqrt(2.0 * M))
    return _maybe_slice(mfcc[..., :num_mfccs].astype(dtype), out_shape)


def delta(features, N=2):
    """audio_processing.py:85-94 -- regression deltas with cumulative SYMMETRIC padding."""
    x = np.asarray(features)
    denom = 2 * sum(i ** 2 for i in range(1, N + 1))
    acc = np.zeros_like(x)
    padded = x
    for i in range(1, N + 1):
        padded = np.pad(padded, [[0, 0], [1, 1], [0, 0]], mode='symmetric')
        acc = acc + i * (padded[:, i * 2:, :] - padded[:, :-i * 2, :])
    return acc / x.dtype.type(denom)


def add_delta_features(features, n_delta=2, N=2):
    """audio_processing.py:97-104."""
    feats = [np.asarray(features)]
    cur = feats[0]
    for _ in range(n_delta):
        cur = delta(cur, N)
        feats.append(cur)
    return np.concatenate(feats, axis=2)


def preemphasis(sources, alpha=0.95, dtype=np.float64):
    """audio_processing.py:19-22 -- y[t] = x[t] - alpha x[t-1], x[-1] = 0."""
    x = np.asarray(sources, dtype=dtype)
    prev = np.concatenate([np.zeros((x.shape[0], 1), dtype=dtype), x[:, :-1]], axis=1)
    return x - dtype(alpha) * prev


def inverse_stft_window(frame_length, frame_step, dtype=np.float64):
    """tf.contrib.signal.inverse_stft_window_fn(frame_step)(frame_length) (App. A.6):
    w[n] / sum_k w[n mod S + k S]^2 with the periodic Hann forward window."""
    w = hann_periodic(frame_length, np.float64)
    denom = np.zeros(frame_step)
    for k in range(-(-frame_length // frame_step)):
        seg = w[k * frame_step:(k + 1) * frame_step]
        denom[: len(seg)] += seg ** 2
    denom = np.tile(denom, -(-frame_length // frame_step))[:frame_length]
    return (w / denom).astype(dtype)


def enclosing_power_of_two(n):
    """tf.contrib.signal (spectral_ops._enclosing_power_of_two): 2 ** ceil(log2(n))."""
    return 1 << max(0, int(n) - 1).bit_length()


def inverse_stft(stfts, frame_length, frame_step, dtype=np.float64, fft_length=None):
    """tf.contrib.signal.inverse_stft (audio_processing.py:149-151; App. A.6).

    complex [B, T, F] -> real [B, (T-1) step + frame_length].  ``fft_length=None`` is TF 1.x's default: the
    enclosing power of two of ``frame_length`` -- NOT (F - 1) * 2 -- and ``irfft`` crops (or zero-pads) the
    bin axis to fft_length // 2 + 1.  So the reference's 384 / 192 call transforms 257 bins at 512 points, and
    ``reconstruct_sources`` with its 16 / 8 ms default on a 257-bin spectrogram uses the first 129 bins at 256."""
    X = np.asarray(stfts)
    B, T, F = X.shape
    nfft = enclosing_power_of_two(frame_length) if fft_length is None else int(fft_length)
    nb = nfft // 2 + 1
    X = X[..., :nb] if F >= nb else np.pad(X, [[0, 0], [0, 0], [0, nb - F]])
    frames = np.fft.irfft(X, n=nfft, axis=-1).astype(dtype)[..., :frame_length]
    if frames.shape[-1] < frame_length:       # fft shorter than frame: TF pads with zeros
        frames = np.pad(frames, [[0, 0], [0, 0], [0, frame_length - frames.shape[-1]]])
    frames = frames * inverse_stft_window(frame_length, frame_step, dtype)[None, None, :]
    out = np.zeros((B, (T - 1) * frame_step + frame_length), dtype=dtype)
    for t in range(T):
        out[:, t * frame_step: t * frame_step + frame_length] += frames[:, t]
    return out


def reconstruct_sources(stfts, num_samples=0, sample_rate=16000, window_size=16, step_size=8,
                        dtype=np.float64):
    """audio_processing.py:145-157."""
    L = ms_to_samples(window_size, sample_rate)
    S = ms_to_samples(step_size, sample_rate)
    y = inverse_stft(stfts, L, S, dtype)
    return y[:, :num_samples] if num_samples > 0 else y


def get_sources(mag, phase, num_samples=48000, sample_rate=16000, window_size=24, step_size=12,
                dtype=np.float64):
    """audio_processing.py:160-164 -- mag (cos phi + j sin phi) -> inverse STFT."""
    mag = np.asarray(mag, dtype=dtype)
    phase = np.asarray(phase, dtype=dtype)
    X = (mag * np.cos(phase)) + 1j * (mag * np.sin(phase))
    return reconstruct_sources(X.astype(_cdtype(dtype)), num_samples, sample_rate, window_size,
                               step_size, dtype)


def inpainter_frontend(wav, mean, std, masks, dtype=np.float64, audio_feat_dim=257, max_len=None):
    """StackedBLSTMModel.__init__ front end (models.py:30-35).

    Returns (target_stft, target_spec_norm, audio_features)."""
    x = np.asarray(wav, dtype=dtype)
    T = num_frames(x.shape[1], 192) if max_len is None else max_len
    st = get_stft(x, window_size=24, step_size=12, n_fft=512,
                  out_shape=(x.shape[0], T, audio_feat_dim), dtype=dtype)
    spec = get_spectrogram(st, log=True, dtype=dtype)
    norm = (spec - np.asarray(mean, dtype=dtype)) / np.asarray(std, dtype=dtype)
    feats = norm * np.asarray(masks, dtype=dtype)
    return st, norm.astype(dtype), feats.astype(dtype)


def feature_stats(feature_list, masks=None):
    """compute_mean_std_features accumulation (audio_feat_preprocessing.py:77-116):
    float64 sum / sum-of-squares over frames; std = sqrt(E[x^2] - mean^2)."""
    tot = None
    tot2 = None
    count = 0
    for i, feat in enumerate(feature_list):
        feat = np.asarray(feat, dtype=np.float64)
        if masks is not None:
            m = np.asarray(masks[i], dtype=np.float64)
            feat = feat[: len(m), : m.shape[1]] * m
            count += int(m[:, 0].sum())
        else:
            count += len(feat)
        tot = feat.sum(axis=0) if tot is None else tot + feat.sum(axis=0)
        tot2 = (feat ** 2).sum(axis=0) if tot2 is None else tot2 + (feat ** 2).sum(axis=0)
    mean = tot / count
    return mean, np.sqrt(tot2 / count - mean ** 2)


def logmel_of_prediction(pred_norm, mean, std, dtype=np.float64):
    """Metric of BASELINE.json ('reconstructed log-mel'), SURVEY.md F3:
    log(mel_W . exp(pred*std+mean)^2 + 1e-6), the op chain of models_asr.py:33-36
    applied to the de-normalised predicted magnitude (models.py:185)."""
    mag = np.exp(np.asarray(pred_norm, dtype=dtype) * np.asarray(std, dtype=dtype)
                 + np.asarray(mean, dtype=dtype))
    return get_log_mel_spectrogram(mag ** 2, dtype=dtype)
