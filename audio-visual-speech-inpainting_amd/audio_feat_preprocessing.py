"""Feature statistics tool: per-bin mean / standard deviation of spec | fbanks | mfcc features over a
directory of WAVs -- the numbers ``StackedBLSTMModel`` normalises with.

Same signature, file contract and printed report as the reference
``av_speech_inpainting/audio_feat_preprocessing.py:23-129``: reads
``<audio_folder>/<sample>/<file_prefix>.<ext>``, optionally ``mask.npy`` beside it, writes
``<audio_folder>/<out_prefix>_mean.npy`` / ``_std.npy`` (float64) and, with ``save_feat``,
``<audio_folder>/<sample>/<file_prefix>.npy``.  The reference runs one ``sess.run`` per file; here
files of equal length go through the fused gfx950 front-end kernel in batches, and the float64
sums of x and x^2 are accumulated on the host exactly as the reference does (:105-116).
"""
import os
from glob import glob

import numpy as np
import torch
from scipy.io import wavfile

from . import audio_processing as ap


downsampling = ap.downsampling       # reference audio_processing.py:9-16, imported there by this module


def _features(wavs, type, sample_rate, n_fft, window_size, step_size, preemph, num_mel_bins, num_mfcc, delta):
    x = torch.from_numpy(np.stack(wavs).astype(np.float32)).cuda()
    if preemph > 0:
        x = ap.preemphasis(x, alpha=preemph)
    if type == 'spec':
        feat = ap.frontend(x, sample_rate, window_size, step_size, n_fft, want_spec=True)['spec']
    elif type in ('fbanks', 'mfcc'):
        feat = ap.frontend(x, sample_rate, window_size, step_size, n_fft, want_logmel=True,
                           num_mel_bins=num_mel_bins)['logmel']
        if type == 'mfcc':
            feat = ap.get_mfcc(feat, num_mfcc)
    else:
        print('Type must be "stft", "spec", "fbanks" or "mfcc". Closing...')
        exit(1)
    if delta > 0:
        feat = ap.add_delta_features(feat, n_delta=delta, N=2)
    return feat.cpu().numpy()


def compute_mean_std_features(audio_folder, file_prefix, out_prefix, type='spec', sample_rate=16e3, n_fft=512,
                              window_size=25, step_size=10, preemph=0, num_mel_bins=80, num_mfcc=13, delta=0,
                              apply_mask=False, save_feat=False, file_ext='wav', batch_size=64):
    sample_rate = int(sample_rate)
    sample_dirs = [d for d in glob(os.path.join(audio_folder, '*')) if os.path.isdir(d)]
    print('Computing features...')
    frame_count = 0
    tot_sum = tot_sq = None

    def flush(group):
        nonlocal frame_count, tot_sum, tot_sq
        feats = _features([w for _, w in group], type, sample_rate, n_fft, window_size, step_size, preemph,
                          num_mel_bins, num_mfcc, delta)
        for (audio_dir, _), feat in zip(group, feats):
            if apply_mask:
                mask = np.load(os.path.join(audio_dir, 'mask.npy'))
                feat = feat[: len(mask), : mask.shape[1]] * mask     # drops the last bins / frames like the reference
            if save_feat:
                np.save(os.path.join(audio_folder, os.path.basename(audio_dir), file_prefix + '.npy'), feat)
            f64 = feat.astype(np.float64)
            tot_sum = f64.sum(axis=0) if tot_sum is None else tot_sum + f64.sum(axis=0)
            tot_sq = (f64 ** 2).sum(axis=0) if tot_sq is None else tot_sq + (f64 ** 2).sum(axis=0)
            frame_count += int(mask[:, 0].sum()) if apply_mask else len(feat)

    group, group_len = [], None
    for audio_dir in sample_dirs:
        rate, samples = wavfile.read(os.path.join(audio_dir, file_prefix + '.' + file_ext))
        samples = np.asarray(downsampling(samples, rate, sample_rate), dtype=np.float32)
        if group and (len(samples) != group_len or len(group) == batch_size):
            flush(group)
            group = []
        group.append((audio_dir, samples))
        group_len = len(samples)
    if group:
        flush(group)
    print('done. Audio files processed:', len(sample_dirs))

    print('Computing mean and standard deviation of features...')
    print('Total number of frames:', frame_count)
    feat_mean = tot_sum / frame_count
    feat_std = np.sqrt(tot_sq / frame_count - feat_mean ** 2)
    print('done.')
    print('')
    print('Features mean:')
    print(feat_mean.shape)
    print(feat_mean)
    print('Features standard deviation:')
    print(feat_std.shape)
    print(feat_std)
    np.save(os.path.join(audio_folder, out_prefix + '_mean.npy'), feat_mean)
    np.save(os.path.join(audio_folder, out_prefix + '_std.npy'), feat_std)
    print('Normalization data files saved.')
    return feat_mean, feat_std


def save_features(audio_folder, type='spec', sample_rate=16e3, n_fft=512, window_size=25, step_size=10, preemph=0,
                  num_mel_bins=80, num_mfcc=13, delta=0, file_ext='wav', batch_size=64):
    """Write ``<file>.npy`` = the features of every ``<audio_folder>/*.<file_ext>`` -- reference
    audio_feat_preprocessing.py:132-198 (types 'stft' -- complex64 --, 'spec', 'fbanks', 'mfcc'; optional
    pre-emphasis and deltas).  Files of equal length share a launch of the front-end kernel."""
    sample_rate = int(sample_rate)
    files = sorted(glob(os.path.join(audio_folder, '*.' + file_ext)))
    if type not in ('stft', 'spec', 'fbanks', 'mfcc'):
        print('Type must be "spec", "fbanks" or "mfcc". Closing...')
        exit(1)
    print('Computing and saving features...')

    def flush(group):
        wavs = [w for _, w in group]
        if type == 'stft':
            x = torch.from_numpy(np.stack(wavs).astype(np.float32)).cuda()
            if preemph > 0:
                x = ap.preemphasis(x, alpha=preemph)
            feats = ap.get_stft(x, sample_rate, window_size, step_size, n_fft)
            if delta > 0:
                raise ValueError("delta features of a complex STFT are not defined")
            feats = feats.cpu().numpy()
        else:
            feats = _features(wavs, type, sample_rate, n_fft, window_size, step_size, preemph, num_mel_bins, num_mfcc, delta)
        for (path, _), feat in zip(group, feats):
            np.save(os.path.splitext(path)[0] + '.npy', feat)

    group, group_len = [], None
    for path in files:
        rate, samples = wavfile.read(path)
        samples = np.asarray(downsampling(samples, rate, sample_rate), dtype=np.float32)
        if group and (len(samples) != group_len or len(group) == batch_size):
            flush(group)
            group = []
        group.append((path, samples))
        group_len = len(samples)
    if group:
        flush(group)
    print('done. Audio files processed:', len(files))
