"""The reference's own output pins the STFT -> mask -> inverse-STFT chain (SURVEY §8 rows a1, a2, a12 / f1 and
``masking.py``): tests/golden/ref_docs/ holds the ``target.wav`` / ``masked.wav`` pairs of the reference's
docs/files/{800ms,1600ms}/ex{1,2}/ -- ``masked.wav`` was written by the reference's TensorFlow graph
(av_speech_inpainting/masking.py:42-46,87-89) -- and ``gaps.json`` the whole-frame gap of each pair
(tests/golden/make_ref_docs_golden.py derives it).

Bar: every one of the 48,000 int16 samples within ONE LSB.  One LSB is the floor, not slack: outside the gap (past the
first hop) the chain reconstructs the integer-valued target exactly, so the float result sits ON an integer and
``astype(int16)`` (truncation) lands either side of it depending on the last-bit rounding of the arithmetic; inside
the gap the output must be exactly zero, and in float the reproduction must be within 1 + 2e-2 of the stored integer.
"""
import json
import os

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import frontend as OF

HERE = os.path.dirname(os.path.abspath(__file__))
DOCS = os.path.join(HERE, "golden", "ref_docs")
GAPS = json.load(open(os.path.join(DOCS, "gaps.json")))["gaps"]
KEYS = sorted(GAPS)
T = 250


def load_pair(key):
    sr, target = wavfile.read(os.path.join(DOCS, key + "_target.wav"))
    sr2, masked = wavfile.read(os.path.join(DOCS, key + "_masked.wav"))
    assert sr == sr2 == 16000 and target.dtype == masked.dtype == np.int16
    assert target.shape == masked.shape == (48000,)
    mask = np.ones((T, 257), dtype=np.float32)
    g0, g1 = GAPS[key]["gap_frames"]
    mask[g0:g1] = 0
    return target.astype(np.float32), masked, mask, (g0, g1)


def check_against_masked_wav(y, masked, gap):
    """y: float waveform [48000] produced by the chain under test."""
    g0, g1 = gap
    assert y.shape == (48000,)
    as_int = y.astype(np.int16).astype(np.int64)           # wavfile.write(..., masked.astype(np.int16)), masking.py:89
    assert np.abs(as_int - masked.astype(np.int64)).max() <= 1
    assert np.abs(y.astype(np.float64) - masked).max() <= 1.0 + 2e-2
    # samples covered only by gap frames: frames g0 .. g1-1 span [g0*192, (g1-1)*192+384); their neighbours reach
    # 192 samples in from either side
    inner = slice((g0 + 1) * 192, g1 * 192)
    assert np.all(masked[inner] == 0) and np.all(as_int[inner] == 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("key", KEYS)
def test_oracle_reproduces_reference_masked_wav(key, dtype):
    target, masked, mask, gap = load_pair(key)
    st = OF.get_stft(target[None], window_size=24, step_size=12, n_fft=512, out_shape=(1, T, 257), dtype=dtype)
    ms = st * mask[None].astype(st.dtype)                  # masking.py:43
    mag = OF.get_spectrogram(ms, dtype=dtype)              # masking.py:44  tf.abs
    ang = np.angle(st).astype(dtype)                       # masking.py:45  oracle_phase=True
    y = OF.get_sources(mag, ang, num_samples=48000, dtype=dtype)[0]
    check_against_masked_wav(y[: T * 192], masked, gap)


@pytest.mark.parametrize("key", KEYS)
def test_gap_length_follows_the_folder_name(key):
    """dataset_generator.py:16-17,73: a gap of g ms is round(250 g / 3000) whole frames."""
    g0, g1 = GAPS[key]["gap_frames"]
    assert g1 - g0 == int(round(T * GAPS[key]["gap_ms"] / 3000)) and 0 <= g0 < g1 <= T


# ------------------------------------------------------------------------------------------------ HIP path
@pytest.fixture(scope="module")
def ap():
    import avsi_amd  # noqa: F401
    from avsi_amd import audio_processing
    return audio_processing


@pytest.mark.gpu
@pytest.mark.parametrize("key", KEYS)
def test_hip_get_stft_mask_get_sources_reproduces_masked_wav(ap, key):
    """The reference's literal op chain through the C ABI: frontend kernel, |.|, angle, inverse-STFT kernel."""
    import torch
    target, masked, mask, gap = load_pair(key)
    wav = torch.from_numpy(target[None]).cuda()
    m = torch.from_numpy(mask[None]).cuda()
    st = ap.get_stft(wav, window_size=24, step_size=12, n_fft=512, out_shape=[1, T, 257])
    ms = st * m
    mag = ap.get_spectrogram(ms)
    y = ap.get_sources(mag, torch.angle(st), num_samples=48000)
    check_against_masked_wav(y[0].cpu().numpy(), masked, gap)
    # the same through the complex-input entry (what avsi_amd.masking.mask_app calls)
    y2 = ap.reconstruct_sources(ms, 48000, window_size=24, step_size=12)
    check_against_masked_wav(y2[0].cpu().numpy(), masked, gap)


@pytest.mark.gpu
@pytest.mark.parametrize("key", KEYS)
def test_hip_fused_enhanced_sources_reproduces_masked_wav(ap, key):
    """The fused enhanced_sources kernel (models.py:181-197, oracle phase) fed the masked magnitude as its
    'prediction' (mean 0, std 1, log of the masked magnitude; exp(-100) = 0 inside the gap)."""
    import torch
    target, masked, mask, gap = load_pair(key)
    wav = torch.from_numpy(target[None]).cuda()
    m = torch.from_numpy(mask[None]).cuda()
    st = ap.get_stft(wav, window_size=24, step_size=12, n_fft=512)
    mag = ap.get_spectrogram(st) * m
    pred = torch.where(mag > 0, torch.log(mag), torch.full_like(mag, -100.0))
    zeros, ones = torch.zeros(257, device='cuda'), torch.ones(257, device='cuda')
    y = ap.enhanced_from_prediction(pred, zeros, ones, st, None, num_samples=48000)
    check_against_masked_wav(y[0].cpu().numpy(), masked, gap)


@pytest.mark.gpu
def test_mask_app_driver_writes_the_reference_masked_wavs(tmp_path):
    """avsi_amd.masking.mask_app end to end: TFRecords holding the four targets and masks in, masked.wav files out,
    compared with the files the reference's mask_app wrote."""
    import avsi_amd  # noqa: F401
    from avsi_amd import masking
    from avsi_amd import tfrecord_io as tio
    data = tmp_path / "tfrecords"
    data.mkdir()
    for i, key in enumerate(KEYS):
        target, _, mask, _ = load_pair(key)
        rec = tio.serialize_sample_fixed(T, 3, target, np.zeros((T, 136), np.float32), mask, np.zeros(50), key)
        tio.write_records(str(data / ("data_%05d.tfrecord" % (i + 1))), [rec])
    masking.mask_app(str(data), str(tmp_path / "audio"), batch_size=3)
    for key in KEYS:
        _, masked, _, gap = load_pair(key)
        rate, got = wavfile.read(str(tmp_path / "audio" / key / "masked.wav"))
        assert rate == 16000 and got.dtype == np.int16 and got.shape == (48000,)
        assert np.abs(got.astype(np.int64) - masked.astype(np.int64)).max() <= 1
        g0, g1 = gap
        assert np.all(got[(g0 + 1) * 192: g1 * 192] == 0)
