"""Stand-alone DSP operators, feature-statistics tool, masking app and CLI on the GPU vs the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from oracle import frontend as OF

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ap():
    import avsi_amd
    from avsi_amd import audio_processing
    return audio_processing


def _wav(B, N, seed):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)


def test_per_op_chain_matches_oracle(ap):
    """get_stft -> get_spectrogram(power 2) -> get_log_mel_spectrogram -> get_mfcc -> add_delta_features
    (the asr_preprocessing chain, audio_processing.py:107-142) and preemphasis."""
    wav = _wav(2, 16000, 0)
    w = torch.from_numpy(wav).cuda()
    pre = ap.preemphasis(w, 0.95)
    np.testing.assert_allclose(pre.cpu().numpy(), OF.preemphasis(wav, 0.95), atol=1e-2)
    st = ap.get_stft(pre)
    ref_st = OF.get_stft(OF.preemphasis(wav, 0.95))
    spec = ap.get_spectrogram(st, power=2)
    ref_spec = OF.get_spectrogram(ref_st, power=2)
    assert np.abs(spec.cpu().numpy() - ref_spec).max() < 1e-5 * ref_spec.max()
    logspec = ap.get_spectrogram(st, log=True)
    assert np.sqrt(np.mean((logspec.cpu().numpy() - OF.get_spectrogram(ref_st, log=True)) ** 2)) < 1e-4
    p03 = ap.get_spectrogram(st, power=0.3)
    assert np.abs(p03.cpu().numpy() - OF.get_spectrogram(ref_st, power=0.3)).max() < 1e-3
    fb = ap.get_log_mel_spectrogram(spec)
    ref_fb = OF.get_log_mel_spectrogram(ref_spec)
    assert fb.shape == (2, 100, 80)
    assert np.sqrt(np.mean((fb.cpu().numpy() - ref_fb) ** 2)) < 1e-4
    mf = ap.get_mfcc(fb, 13)
    ref_mf = OF.get_mfcc(ref_fb, 13)
    assert mf.shape == (2, 100, 13)
    assert np.abs(mf.cpu().numpy() - ref_mf).max() < 2e-3
    d = ap.add_delta_features(mf, n_delta=2, N=2)
    ref_d = OF.add_delta_features(ref_mf, 2, 2)
    assert d.shape == (2, 100, 39)
    assert np.abs(d.cpu().numpy() - ref_d).max() < 2e-3
    sl = ap.get_spectrogram(st, out_shape=[1, 50, 100])
    assert sl.shape == (1, 50, 100)


def _make_audio_tree(root, n, N, seed):
    rng = np.random.default_rng(seed)
    wavs, masks = [], []
    for i in range(n):
        d = os.path.join(root, "s%02d" % i)
        os.makedirs(d)
        w = np.clip(np.round(rng.normal(0, 3000, N)), -32768, 32767).astype(np.int16)
        wavfile.write(os.path.join(d, "target.wav"), 16000, w)
        T = -(-N // 192)
        m = np.ones((T, 257), dtype=np.float32)
        s = rng.integers(0, T - 5)
        m[s:s + 5] = 0
        np.save(os.path.join(d, "mask.npy"), m)
        wavs.append(w.astype(np.float32))
        masks.append(m)
    return wavs, masks


@pytest.mark.parametrize("ftype,apply_mask", [("spec", False), ("spec", True), ("fbanks", False)])
def test_compute_mean_std_features(ap, tmp_path, ftype, apply_mask):
    from avsi_amd.audio_feat_preprocessing import compute_mean_std_features
    root = str(tmp_path)
    wavs, masks = _make_audio_tree(root, 5, 9600, 1)
    mean, std = compute_mean_std_features(root, "target", "norm", type=ftype, sample_rate=16000, n_fft=512,
                                          window_size=24, step_size=12, apply_mask=apply_mask, save_feat=True)
    feats = []
    for w in wavs:
        st = OF.get_stft(w[None], window_size=24, step_size=12)
        if ftype == "spec":
            feats.append(OF.get_spectrogram(st, log=True)[0])
        else:
            feats.append(OF.get_log_mel_spectrogram(OF.get_spectrogram(st, power=2))[0])
    # the oracle's accumulation does not depend on directory order (sums)
    rmean, rstd = OF.feature_stats(feats, masks if apply_mask else None)
    np.testing.assert_allclose(mean, rmean, atol=2e-4)
    np.testing.assert_allclose(std, rstd, atol=2e-4)
    assert np.load(os.path.join(root, "norm_mean.npy")).dtype == np.float64
    assert os.path.isfile(os.path.join(root, "s00", "target.npy"))


def test_cli_help_and_out_of_scope_subcommands():
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, "-c", "import avsi_amd; from avsi_amd import speech_inpainting_main as m; "
                                     "import sys; m.main(sys.argv[1:])"] + list(a), capture_output=True, text=True, env=env)
    r = run("inference", "--help")
    assert r.returncode == 0 and "--oracle_phase" in r.stdout and "--out_file_prefix" in r.stdout
    r = run("evaluation", "-ed", "x", "-ef", "y", "-o", "z", "--pesq_path", "p", "--pesq_mode", "nb")
    assert r.returncode == 1 and "not part of the MI355X hot-path package" in r.stdout
    r = run()
    assert r.returncode == 1 and "Bad subcommand name" in r.stdout


@pytest.mark.parametrize("ftype", ["mfcc", "fbanks", "spec"])
def test_asr_preprocessing_matches_oracle_chain(ap, ftype):
    """asr_preprocessing (audio_processing.py:107-142): the composed recognition front end, with the
    per-utterance mean normalisation it applies when only feat_std is given."""
    wav = _wav(2, 16000, 3)
    n_delta = 2
    D = {"mfcc": 13, "fbanks": 80, "spec": 257}[ftype] * (n_delta + 1)
    std = np.linspace(0.5, 2.0, D).astype(np.float32)
    out = ap.asr_preprocessing(torch.from_numpy(wav).cuda(), type=ftype, n_delta=n_delta, feat_std=std)
    st = OF.get_stft(OF.preemphasis(wav, 0.95))
    if ftype == "spec":
        ref = OF.get_spectrogram(st, power=0.3)
    else:
        ref = OF.get_log_mel_spectrogram(OF.get_spectrogram(st, power=2))
        if ftype == "mfcc":
            ref = OF.get_mfcc(ref, 13)
    ref = OF.add_delta_features(ref, n_delta, 2)
    ref = (ref - ref.mean(axis=1, keepdims=True)) / std
    assert out.shape == ref.shape == (2, 100, D)
    assert np.abs(out.cpu().numpy() - ref).max() < 5e-3 * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):
        ap.asr_preprocessing(torch.from_numpy(wav).cuda(), type='stft')
    # preemph <= 0: the sources are used as they are (the reference stops on an undefined name, App. B7)
    plain = ap.asr_preprocessing(torch.from_numpy(wav).cuda(), type='fbanks', preemph=0, n_delta=0)
    ref_plain = OF.get_log_mel_spectrogram(OF.get_spectrogram(OF.get_stft(wav), power=2))
    assert np.sqrt(np.mean((plain.cpu().numpy() - ref_plain) ** 2)) < 1e-4


def test_oracle_masks_and_downsampling(ap):
    wav = _wav(2, 9600, 4)
    noise = _wav(2, 9600, 5)
    t = ap.get_stft(torch.from_numpy(wav).cuda())
    m = ap.get_stft(torch.from_numpy(wav + noise).cuda())
    T, M = OF.get_stft(wav), OF.get_stft(wav + noise)
    iam = ap.get_oracle_iam(t, m).cpu().numpy()
    ref_iam = np.clip(np.abs(T) / np.abs(M), 0, 10)
    np.testing.assert_allclose(iam, ref_iam, rtol=2e-3, atol=2e-4)
    ipsm = ap.get_oracle_ipsm(t, m).cpu().numpy()
    ref_ipsm = np.clip(np.abs(T) * np.cos(np.angle(M) - np.angle(T)) / np.abs(M), 0, 10)
    np.testing.assert_allclose(ipsm, ref_ipsm, rtol=2e-3, atol=2e-3)
    assert iam.dtype == np.float32 and iam.max() <= 10 and ipsm.min() >= 0
    x = np.sin(2 * np.pi * 440 * np.arange(50000) / 50000.0)
    y = ap.downsampling(x, 50000, 16000)
    assert len(y) == 16000 and ap.downsampling(x, 16000, 16000) is x
    np.testing.assert_allclose(y[100:-100], np.sin(2 * np.pi * 440 * np.arange(16000) / 16000.0)[100:-100], atol=1e-6)


def test_save_features_writes_one_npy_per_wav(ap, tmp_path):
    from avsi_amd.audio_feat_preprocessing import save_features
    rng = np.random.default_rng(6)
    for i, n in enumerate((9600, 9600, 4800)):
        w = np.clip(np.round(rng.normal(0, 3000, n)), -32768, 32767).astype(np.int16)
        wavfile.write(str(tmp_path / ("utt%d.wav" % i)), 16000, w)
    save_features(str(tmp_path), type='fbanks', sample_rate=16000, window_size=24, step_size=12, delta=1)
    f0, f2 = np.load(str(tmp_path / "utt0.npy")), np.load(str(tmp_path / "utt2.npy"))
    assert f0.shape == (50, 160) and f2.shape == (25, 160)
    rate, w0 = wavfile.read(str(tmp_path / "utt0.wav"))
    ref = OF.add_delta_features(OF.get_log_mel_spectrogram(OF.get_spectrogram(
        OF.get_stft(w0[None].astype(np.float32), window_size=24, step_size=12), power=2)), 1, 2)[0]
    assert np.sqrt(np.mean((f0 - ref) ** 2)) < 1e-4
    save_features(str(tmp_path), type='stft', sample_rate=16000, window_size=24, step_size=12)
    assert np.load(str(tmp_path / "utt2.npy")).dtype == np.complex64
