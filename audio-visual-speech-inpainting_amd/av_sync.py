"""Audio-visual alignment of the data preparation (reference av_sync.py:7-41).

``inc_fps`` up-samples a [frames, features] landmark track to the spectrogram frame rate.  The
reference does it with ``scipy.interpolate.interp2d(kind='linear')`` evaluated on the original
feature grid, i.e. plain linear interpolation along the frame axis at
``linspace(0, n (1 - 1/target), target)`` with positions past the last frame clamped to it
(interp2d's nearest-edge extrapolation).  interp2d no longer exists in current SciPy; this is the
same arithmetic with numpy.
"""
import numpy as np


def inc_fps(frames, target_len):
    frames = np.asarray(frames, dtype=np.float64)
    n = frames.shape[0]
    pos = np.clip(np.linspace(0, n * (1 - 1 / target_len), target_len), 0, n - 1)
    lo = np.minimum(np.floor(pos).astype(np.int64), max(n - 2, 0))
    hi = np.minimum(lo + 1, n - 1)
    w = (pos - lo)[:, None]
    return frames[lo] * (1 - w) + frames[hi] * w


def sync_audio_visual_features(mask, video_features, tot_frames=None, min_frames=None, pad='start'):
    """Up-sample the video features to ``len(mask)`` frames.  Tracks that are not 2-D or shorter than
    ``min_frames`` are rejected (None); tracks shorter than ``tot_frames`` are completed by repeating
    the FIRST frame at the start (``pad='start'``) or -- as the reference does -- also the first
    frame at the end (``pad='end'``)."""
    video_features = np.asarray(video_features)
    if video_features.ndim != 2 or (min_frames is not None and video_features.shape[0] < min_frames):
        return None
    if tot_frames is not None and video_features.shape[0] < tot_frames:
        fill = np.tile(video_features[0], (tot_frames - video_features.shape[0], 1))
        if pad == 'start':
            video_features = np.vstack((fill, video_features))
        elif pad == 'end':
            video_features = np.vstack((video_features, fill))
    video_features = inc_fps(video_features, len(mask))
    return video_features if len(mask) == len(video_features) else None
