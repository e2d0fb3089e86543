#!/usr/bin/env python3
"""Reference-held golden for the STFT -> mask -> inverse-STFT chain.

/root/reference/docs/files/{800ms,1600ms}/ex{1,2}/ hold ``target.wav`` and ``masked.wav`` pairs that the
reference's own TensorFlow-1 graph produced: ``masked.wav`` is the output of ``mask_app``
(av_speech_inpainting/masking.py:42-46,87-89):

    target_stft   = get_stft(target, window_size=24, step_size=12, n_fft=512)
    masked_stft   = target_stft * cast(mask, complex64)            # whole-frame gap of zeros
    masked_source = get_sources(|masked_stft|, angle(target_stft), num_samples=48000)
    wavfile.write('masked.wav', 16000, masked_source[: seq_len * 192].astype(np.int16))

This script COPIES those eight WAV files (data, 96 KB each) to tests/golden/ref_docs/ and derives, per pair, the
one thing the files do not state: the whole-frame gap [g0, g1) of the mask.  The gap length is fixed by the folder
name (round(250 * ms / 3000) frames: dataset_generator.py:16-17,73); its onset is found by exhaustive search over
every onset, scoring the oracle's reproduction of ``masked.wav`` (sum of |err| over all 48,000 samples).  Both go to
``ref_docs/gaps.json`` with the max |err| of the best fit in int16 LSB (astype(int16) truncates towards zero, so
one LSB of disagreement is the float32-vs-float64 rounding of a sample that sits on an integer boundary).

Run from the repo root, in the build container (needs /root/reference):  python tests/golden/make_ref_docs_golden.py
"""
import json
import os
import shutil
import sys

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import frontend as OF  # noqa: E402

SRC = '/root/reference/docs/files'
DST = os.path.join(HERE, 'ref_docs')
T = 250


def masked_wav(target, g0, g1):
    st = OF.get_stft(target[None].astype(np.float64), window_size=24, step_size=12, n_fft=512)
    mask = np.ones((1, T, 1))
    mask[:, g0:g1] = 0
    ms = st * mask
    y = OF.get_sources(np.abs(ms), np.angle(st), num_samples=48000)[0]
    return y[: T * 192]


def main():
    gaps = {}
    for gap_ms in (800, 1600):
        n_gap = int(round(T * gap_ms / 3000))
        for ex in ('ex1', 'ex2'):
            key = '%dms_%s' % (gap_ms, ex)
            os.makedirs(DST, exist_ok=True)
            for name in ('target', 'masked'):
                shutil.copyfile(os.path.join(SRC, '%dms' % gap_ms, ex, name + '.wav'),
                                os.path.join(DST, '%s_%s.wav' % (key, name)))
            sr, target = wavfile.read(os.path.join(DST, key + '_target.wav'))
            sr2, masked = wavfile.read(os.path.join(DST, key + '_masked.wav'))
            assert sr == sr2 == 16000 and target.shape == masked.shape == (48000,) and masked.dtype == np.int16
            # the masked STFT is exactly linear in the frames that are kept, so score every onset with one STFT
            st = OF.get_stft(target[None].astype(np.float64), window_size=24, step_size=12, n_fft=512)
            per_frame = np.zeros((T, 48000))
            # contribution of frame t to the output: inverse STFT of a spectrogram holding only frame t
            for t in range(T):
                one = np.zeros_like(st)
                one[:, t] = st[:, t]
                per_frame[t] = OF.reconstruct_sources(one, 48000, window_size=24, step_size=12)[0]
            full = per_frame.sum(0)
            csum = np.concatenate([np.zeros((1, 48000)), np.cumsum(per_frame, axis=0)])
            best = None
            for g0 in range(0, T - n_gap + 1):
                y = full - (csum[g0 + n_gap] - csum[g0])
                score = float(np.abs(y - masked).sum())
                if best is None or score < best[0]:
                    best = (score, g0)
            g0 = best[1]
            y = masked_wav(target, g0, g0 + n_gap)
            err = np.abs(y.astype(np.int16).astype(np.int64) - masked.astype(np.int64))
            gaps[key] = {'gap_ms': gap_ms, 'gap_frames': [g0, g0 + n_gap], 'max_err_lsb_f64_oracle': int(err.max()),
                         'n_samples_off_by_one': int((err > 0).sum())}
            print(key, gaps[key])
    with open(os.path.join(DST, 'gaps.json'), 'w') as f:
        json.dump({'source': 'docs/files/{800ms,1600ms}/ex{1,2}/{target,masked}.wav', 'frames': T, 'gaps': gaps},
                  f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
