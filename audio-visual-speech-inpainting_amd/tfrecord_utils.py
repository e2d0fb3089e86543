"""Sample folders -> TFRecords (reference tfrecord_utils.py:72-158, tfrecord_emb_utils.py:72-125;
``fixed`` mode -- the ``var`` mode of the reference is broken).

For every ``<data_path>/<sample>/`` folder written by dataset_generator: read ``target.wav`` as
16-bit samples, ``mask.npy`` [T, 257], ``landmarks.npy`` -> [frames, 136]; align the landmarks to
the T spectrogram frames (av_sync, 75 video frames expected, fewer than 70 rejects the sample);
motion vectors (first difference, first row zero -- face_landmarks.py:30-39); normalise with the
sample's ``video_feat_{mean,std}.npy``; phoneme labels from ``transcription.lbl`` padded to 50
(transcription2phonemes.py:7-22); with ``embeddings=True`` also ``vgg_embeddings/target.npy``.
One record per file ``data_%05d.tfrecord`` and ``seq_lengths.npy`` beside them.
"""
import os
from glob import glob

import numpy as np
from scipy.io import wavfile

from . import tfrecord_io
from .av_sync import sync_audio_visual_features

MAX_LABELS = 50


def get_motion_vector(landmarks, delta=1):
    """face_landmarks.py:30-39 (anchor-free form): row 0 is zero, row t = x[t] - x[t-1]; delta = 2
    differences that once more (one row shorter)."""
    landmarks = np.asarray(landmarks)
    features = landmarks
    if delta > 0:
        features = np.zeros_like(landmarks)
        features[1:] = np.diff(landmarks, axis=0)
        if delta == 2:
            features = np.diff(features, axis=0)
    return features


def load_dictionary(filename):
    """Sorted set of the whitespace-separated phoneme symbols of the dictionary file."""
    with open(filename, 'r') as f:
        return sorted(set(f.read().split()))


def get_labels(phonemes, dictionary):
    """'SP'-free comma-separated transcription -> indices into the dictionary.  ('SP' is removed as
    a substring, as the reference does.)"""
    return np.asarray([dictionary.index(ph) for ph in phonemes.replace('SP', '').split(',') if ph != ''])


def read_wav_int16(path):
    """What pydub's AudioSegment.from_file(path).set_sample_width(2) yields for PCM files: interleaved
    16-bit samples (wider formats are rescaled)."""
    rate, data = wavfile.read(path)
    if data.dtype == np.int16:
        out = data
    elif data.dtype == np.int32:
        out = (data >> 16).astype(np.int16)
    elif data.dtype == np.uint8:
        out = ((data.astype(np.int16) - 128) << 8).astype(np.int16)
    else:                                   # float PCM in [-1, 1)
        out = np.clip(np.round(data * 32768.0), -32768, 32767).astype(np.int16)
    return out.reshape(-1)


def create_tfrecords_training(data_path, dest_dir, ph_dict, tfrecord_mode='fixed', embeddings=False):
    if tfrecord_mode != 'fixed':
        raise ValueError("only the 'fixed' TFRecord mode is supported")
    sample_dirs = sorted(d for d in glob(os.path.join(data_path, '*')) if os.path.isdir(d))
    os.makedirs(dest_dir, exist_ok=True)
    count, seq_lengths = 0, []
    for i, sample_dir in enumerate(sample_dirs):
        print(str(i) + ' - ' + sample_dir)
        wav = read_wav_int16(os.path.join(sample_dir, 'target.wav'))
        mask = np.load(os.path.join(sample_dir, 'mask.npy'))
        landmarks = np.load(os.path.join(sample_dir, 'landmarks.npy')).reshape((-1, 136))
        video = sync_audio_visual_features(mask, landmarks, tot_frames=75, min_frames=70)
        if video is None:
            print('Skipped. Video features corrupted.')
            continue
        video = get_motion_vector(video, delta=1)
        with open(os.path.join(sample_dir, 'transcription.lbl')) as f:
            labels = get_labels(f.read(), ph_dict)
        lab_len = len(labels)
        labels = np.pad(labels, (0, MAX_LABELS - lab_len), mode='constant')
        emb = np.load(os.path.join(sample_dir, 'vgg_embeddings', 'target.npy')).reshape(-1) if embeddings else None
        mean = np.load(os.path.join(sample_dir, 'video_feat_mean.npy')).flatten()
        std = np.load(os.path.join(sample_dir, 'video_feat_std.npy')).flatten()
        video = (video - mean) / std
        seq_lengths.append(len(mask))
        count += 1
        rec = tfrecord_io.serialize_sample_fixed(len(mask), lab_len, wav, video, mask, labels,
                                                 os.path.basename(sample_dir), embedding=emb)
        tfrecord_io.write_records(os.path.join(dest_dir, 'data_{:05d}.tfrecord'.format(count)), [rec])
    np.save(os.path.join(dest_dir, 'seq_lengths.npy'), np.array(seq_lengths))
    return count


def create_dataset(data_path, dest_dir, dictionary_file, tfrecord_mode='fixed', embeddings=False):
    """training-set / validation-set / test-set -> TFRecords (reference tfrecord_utils.py:127-158)."""
    ph_dict = load_dictionary(dictionary_file)
    counts = []
    for title, sub in (('training', 'training-set'), ('validation', 'validation-set'), ('test', 'test-set')):
        print('Creating {:s} TFRecords...'.format(title))
        counts.append(create_tfrecords_training(os.path.join(data_path, sub), os.path.join(dest_dir, sub), ph_dict,
                                                tfrecord_mode, embeddings))
    print('')
    print('Samples successfully generated:')
    print('-> Training:', counts[0])
    print('-> Validation:', counts[1])
    print('-> Test:', counts[2])
    return counts
