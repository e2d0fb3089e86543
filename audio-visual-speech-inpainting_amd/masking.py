"""Masked-input baseline: writes ``masked.wav`` (the gapped signal) for every utterance of a
TFRecord set and reports the L1 'hole' loss of the un-inpainted input.

Signature and outputs of the reference ``av_speech_inpainting/masking.py:18-103``.  The reference
hard-codes the author's normalisation files (:37-38); here they are optional arguments (identity
normalisation when omitted).
"""
import os
from glob import glob

import numpy as np
import torch
from scipy.io import wavfile

from . import audio_processing as ap
from . import ops
from .dataset_reader import DataManager, OutOfRangeError


def mask_app(data_path, audio_path, tfrecord_mode='fixed', oracle_phase=True, audio_feat_dim=257, video_feat_dim=136,
             num_audio_samples=48000, batch_size=1, audio_feat_mean=None, audio_feat_std=None):
    dm = DataManager(num_audio_samples=num_audio_samples, audio_feat_size=audio_feat_dim,
                     video_feat_size=video_feat_dim, buffer_size=4000, mode='fixed')
    files = sorted(glob(os.path.join(data_path, '*.tfrecord')))
    _, it = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=max(1, batch_size), n_epochs=1)
    mean = torch.from_numpy(np.load(audio_feat_mean).astype(np.float32)).cuda() if audio_feat_mean else None
    std = torch.from_numpy(np.load(audio_feat_std).astype(np.float32)).cuda() if audio_feat_std else None

    print('Mask application on dataset: {:s}'.format(data_path))
    total_wavs = 0
    losses = []
    while True:
        try:
            seq_length, _, target_audio, sample_path, _, _, mask = it.get_next()
        except OutOfRangeError:
            print('done.')
            break
        wav = torch.from_numpy(target_audio.astype(np.float32)).cuda()
        m = torch.from_numpy(mask).cuda()
        T = m.shape[1]
        fe = ap.frontend(wav, window_size=24, step_size=12, n_fft=512, num_frames_out=T, num_bins=audio_feat_dim,
                         mean=mean, std=std, want_stft=True, want_spec=True)
        # |S m| with the phase of S (oracle) or of S m: both are S m wherever the magnitude is non-zero
        masked = ap.reconstruct_sources(fe['stft'] * m, num_audio_samples, window_size=24, step_size=12)
        out3, _ = ops.l1_loss(fe['spec'], torch.zeros_like(fe['spec']), m.contiguous())
        masked = masked.cpu().numpy()
        for audio, sample_dir, seq_len in zip(masked, sample_path, seq_length):
            os.makedirs(os.path.join(audio_path, sample_dir.decode()), exist_ok=True)
            wavfile.write(os.path.join(audio_path, sample_dir.decode(), 'masked.wav'), 16000,
                          audio[: int(seq_len) * 192].astype(np.int16))
        total_wavs += len(seq_length)
        losses.append(float(out3[1]))
        print('Written {:d} masked wavs. Total wavs written so far {:d}.'.format(len(seq_length), total_wavs))
    print('Loss hole: {:.5}'.format(np.mean(losses)))
    return float(np.mean(losses)) if losses else float('nan')
