"""Synthetic gap masks and the per-sample dataset folders (reference dataset_generator.py:11-140).

``get_intrusions_mask`` draws from Python's ``random`` module in the same order as the reference
(number of gaps; Gaussian coverage; the gap lengths, first to last but one; a shuffle; the onsets),
so a seeded run produces the same masks.  Directory contract of ``create_syn_data_speaker``:
``<dest>/s<spk>_<clip>_<gap ms>_<n gaps>/{target.wav, landmarks.npy, transcription.lbl,
video_feat_mean.npy, video_feat_std.npy, mask.npy}``.
"""
import os
import random
import shutil
from glob import glob

import numpy as np

MAX_COVERAGE = 0.8      # dataset_generator.py:16
FRAME_MS = 12           # step size of the spectrogram, :72
FRAME_DIM = 257         # :73


def _gap_lengths(n_gaps, masked_frames, min_len):
    """Split ``masked_frames`` into ``n_gaps`` lengths >= min_len (reference :20-28).  Every gap but the
    last draws uniformly up to a damped share of what is left; the last takes the remainder."""
    damp = np.exp(-(n_gaps - 1) / 6)
    lengths = []
    for i in range(n_gaps - 1):
        reserve = min_len * (n_gaps - i - 1)                 # what the gaps still to come need at least
        room = int((masked_frames - sum(lengths) - reserve) * damp)
        lengths.append(random.randint(min_len, max(min_len, room)))
    lengths.append(masked_frames - sum(lengths))
    random.shuffle(lengths)
    return lengths


def _gap_onsets(lengths, spec_len, masked_frames):
    """Onset of every gap (reference :31-40)."""
    n = len(lengths)
    onsets = []
    for i in range(n):
        if n == 1:
            onsets.append(random.randint(0, spec_len - masked_frames))
        elif i == 0:
            onsets.append(random.randint(0, spec_len - masked_frames - (n - 1)) // 2)
        else:
            after_prev = onsets[-1] + lengths[i - 1] + 1
            if i == n - 1:
                onsets.append(random.randint(onsets[-1], after_prev + spec_len - lengths[i]))
            else:
                onsets.append(random.randint(after_prev, (after_prev + spec_len - sum(lengths[i:]) - (n - i - 1)) // 2))
    return onsets


def get_intrusions_mask(frame_dim, spec_len, cov_mean, cov_std, n_max_intr, min_intr_len=3):
    """-> (mask [spec_len, frame_dim] of ones with zeroed gap frames, true coverage, number of gaps)."""
    n_gaps = random.randint(1, n_max_intr)
    coverage = max(min_intr_len * n_gaps / spec_len, min(random.gauss(cov_mean, cov_std), MAX_COVERAGE))
    masked_frames = int(np.around(spec_len * coverage))
    lengths = _gap_lengths(n_gaps, masked_frames, min_intr_len)
    onsets = _gap_onsets(lengths, spec_len, masked_frames)
    mask = np.ones([spec_len, frame_dim])
    for start, length in zip(onsets, lengths):
        mask[start:start + length] = 0
    return mask, masked_frames / spec_len, n_gaps


def create_syn_data_speaker(dataset_dir, dest_dir, n_speaker, n_samples=0, audio_len=3000, n_max_intr=1,
                            cov_mean=1000, cov_std=300, file_ext='wav'):
    """One sample folder per clean clip of speaker ``n_speaker`` (reference :51-109)."""
    spk = 's' + str(n_speaker)
    clips = glob(os.path.join(dataset_dir, spk, spk + '_16kHz', '*.' + file_ext))
    landmarks_dir = os.path.join(dataset_dir, spk, spk + '.landmarks')
    align_dir = os.path.join(dataset_dir, spk, 'align')
    if n_samples > 0:
        random.seed(30)
        random.shuffle(clips)
        clips = clips[:n_samples]
    spec_len = audio_len // FRAME_MS
    coverages = []
    for n, clip in enumerate(clips):
        print('{:d} - {:s}'.format(n, clip))
        mask, cov, n_gaps = get_intrusions_mask(FRAME_DIM, spec_len, cov_mean / audio_len, cov_std / audio_len, n_max_intr)
        coverages.append(cov)
        stem = os.path.splitext(os.path.basename(clip))[0]
        sample_dir = os.path.join(dest_dir, '{:s}_{:s}_{:d}_{:d}'.format(spk, stem, int(cov * audio_len), n_gaps))
        os.makedirs(sample_dir, exist_ok=True)
        shutil.copy(clip, os.path.join(sample_dir, 'target.wav'))
        shutil.copy(os.path.join(landmarks_dir, stem + '.npy'), os.path.join(sample_dir, 'landmarks.npy'))
        shutil.copy(os.path.join(align_dir, stem + '.lbl'), os.path.join(sample_dir, 'transcription.lbl'))
        for f in ('video_feat_mean.npy', 'video_feat_std.npy'):
            shutil.copy(os.path.join(landmarks_dir, f), os.path.join(sample_dir, f))
        np.save(os.path.join(sample_dir, 'mask.npy'), mask)
    return coverages


def create_syn_dataset(dataset_dir, dest_dir, speakers=(), n_samples=0, audio_len=3000, n_max_intr=1, cov_mean=1000,
                       cov_std=300, file_ext='wav'):
    """Reference :112-131 (same console output)."""
    os.makedirs(dest_dir, exist_ok=True)
    coverages = []
    print('Starting dataset generation...')
    for s in speakers:
        print('Creating masks of speaker {:d}...'.format(s))
        coverages += create_syn_data_speaker(dataset_dir, dest_dir, s, n_samples, audio_len, n_max_intr, cov_mean,
                                             cov_std, file_ext)
        print('done.')
    print('Dataset generation completed.')
    print('Number of generated samples: {:d}. Total length: {:.2f} seconds'.format(
        len(coverages), len(coverages) * audio_len / 1000))
    print('True mask coverage mean: {:.2f} ms - std: {:.2f} ms'.format(
        np.mean(coverages) * audio_len, np.std(coverages) * audio_len))
    return coverages
