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
