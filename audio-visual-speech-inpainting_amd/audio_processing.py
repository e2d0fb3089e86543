"""DSP operators of the hot path on MI355X (host side).

Same names, arguments and defaults as the reference module
``av_speech_inpainting/audio_processing.py`` (get_stft :25-42, get_spectrogram :45-56,
get_log_mel_spectrogram :59-72), taking and returning torch device tensors instead of TF graph
nodes.  The reference builds these as separate TF ops; here ``frontend()`` runs the whole chain
(frame, window, rFFT, |.|, log, z-norm, mask, log-mel) as ONE gfx950 kernel through the C ABI
(``avsi_frontend_f32``), and the per-op functions are thin views over it.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

_TABLES = {}     # (device index, frame_len, nfft) -> device table
_MEL = {}        # (device index, cfg) -> (start, len, w, stride)


def ms_to_samples(ms, sample_rate):
    """int(round(ms / 1e3 * sample_rate)) -- reference audio_processing.py:27-28."""
    return int(round(ms / 1e3 * sample_rate))


def num_frames(num_samples, step):
    """Frames produced by tf.contrib.signal.stft(pad_end=True): ceil(N / step)."""
    return -(-int(num_samples) // int(step))


def _tables(device, frame_len, nfft):
    key = (device.index, frame_len, nfft)
    tab = _TABLES.get(key)
    if tab is None:
        L = _lib.lib()
        n = L.avsi_frontend_table_floats(frame_len, nfft)
        if n == 0:
            raise _lib.AvsiError("unsupported STFT geometry frame_len=%d nfft=%d" % (frame_len, nfft))
        tab = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(L.avsi_frontend_init_tables(_lib.ptr(tab), frame_len, nfft, _lib.stream_ptr()),
                   "avsi_frontend_init_tables")
        _TABLES[key] = tab
    return tab


def linear_to_mel_weight_matrix(num_mel_bins=80, num_spec_bins=257, sample_rate=16000,
                                lower_edge_freq=125.0, upper_edge_freq=7600.0):
    """HTK-mel triangular filterbank [num_spec_bins, num_mel_bins] (host, float64 -> float32).

    Role of tf.signal.linear_to_mel_weight_matrix at audio_processing.py:63-64: un-normalised
    triangles on the mel scale 1127 ln(1 + f/700), DC row zero."""
    def mel(f):
        return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    bins = mel(np.linspace(0.0, sample_rate / 2.0, num_spec_bins)[1:])[:, None]
    e = np.linspace(mel(lower_edge_freq), mel(upper_edge_freq), num_mel_bins + 2)
    up = (bins - e[None, :-2]) / (e[None, 1:-1] - e[None, :-2])
    down = (e[None, 2:] - bins) / (e[None, 2:] - e[None, 1:-1])
    w = np.clip(np.minimum(up, down), 0.0, None)
    return np.concatenate([np.zeros((1, num_mel_bins)), w], axis=0).astype(np.float32)


def _mel_bands(device, num_mel_bins, num_spec_bins, sample_rate, lower, upper):
    """Band form of the mel matrix for the kernel: per band first bin, tap count, taps."""
    key = (device.index, num_mel_bins, num_spec_bins, sample_rate, float(lower), float(upper))
    hit = _MEL.get(key)
    if hit is None:
        w = linear_to_mel_weight_matrix(num_mel_bins, num_spec_bins, sample_rate, lower, upper)
        start = np.zeros(num_mel_bins, dtype=np.int32)
        length = np.zeros(num_mel_bins, dtype=np.int32)
        for m in range(num_mel_bins):
            nz = np.nonzero(w[:, m])[0]
            if len(nz):
                start[m], length[m] = nz[0], nz[-1] - nz[0] + 1
        stride = max(1, int(length.max()))
        taps = np.zeros((num_mel_bins, stride), dtype=np.float32)
        for m in range(num_mel_bins):
            taps[m, :length[m]] = w[start[m]:start[m] + length[m], m]
        hit = (torch.from_numpy(start).to(device), torch.from_numpy(length).to(device),
               torch.from_numpy(taps).to(device), stride)
        _MEL[key] = hit
    return hit


_FE_LOSS_WS = {}


def frontend_l1_loss(sources, pred, masks, mean, std, sample_rate=16000, window_size=24, step_size=12, n_fft=512,
                     want_grad=False, grad_scale=None, eps=1e-6):
    """(out3, dpred) as ops.l1_loss(target_spec_norm, pred, masks) -- with the normalised target spectrogram recomputed from the
    waveform INSIDE the kernel (avsi_frontend_l1_loss_f32) instead of read from memory: the step's front-end call then has no
    reason to write it.  pred [B, T, 257] (last dimension contiguous), masks [B, >= T, 257].  Returns None when the C side does
    not take the geometry (the caller uses the stored target then)."""
    _lib.require_cuda(sources, pred, masks, mean, std)
    L = _lib.lib()
    if sources.dtype != torch.float32 or sources.dim() != 2 or pred.dtype != torch.float32 or pred.dim() != 3:
        raise _lib.AvsiError("frontend_l1_loss: sources float32 [B, N], pred float32 [B, T, 257]")
    if sources.stride(1) != 1:
        sources = sources.contiguous()
    if pred.stride(2) != 1:
        pred = pred.contiguous()
    dev = sources.device
    B, N = sources.shape
    T, F = int(pred.shape[1]), int(pred.shape[2])
    a = _lib.FrontendArgs()
    a.wav, a.batch, a.num_samples, a.wav_stride = _lib.ptr(sources), B, N, sources.stride(0) if B > 1 else N
    a.frame_len, a.hop = ms_to_samples(window_size, sample_rate), ms_to_samples(step_size, sample_rate)
    a.nfft, a.num_frames, a.num_bins = n_fft, T, F
    tab = _tables(dev, a.frame_len, n_fft)
    a.table = _lib.ptr(tab)
    mean, std = mean.to(torch.float32).contiguous(), std.to(torch.float32).contiguous()
    a.mean, a.stdev = _lib.ptr(mean), _lib.ptr(std)
    if masks.dtype != torch.float32 or masks.stride(2) != 1:
        masks = masks.to(torch.float32).contiguous()
    a.mask, a.mask_stride_b, a.mask_stride_t = _lib.ptr(masks), masks.stride(0), masks.stride(1)
    a.spec_power, a.log_spec, a.eps = 1.0, 1, float(eps)
    if pred.shape[0] != B or masks.shape[0] != B or masks.shape[1] < T or masks.shape[2] != F \
            or not L.avsi_frontend_l1_loss_supported(ctypes.byref(a)):
        return None
    key = (dev.index, _lib.stream_ptr().value)
    ws = _FE_LOSS_WS.get(key)
    need = L.avsi_l1_loss_workspace_bytes(B * T * F)
    if ws is None or ws.numel() * 4 < need:
        ws = _FE_LOSS_WS[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    dpred = torch.empty((B, T, F), dtype=torch.float32, device=dev) if want_grad else None
    if dpred is not None and (dpred.stride(0), dpred.stride(1)) != (pred.stride(0), pred.stride(1)):
        pred = pred.contiguous()               # dpred shares pred's strides in the kernel
    gs = (1.0 / (B * T * F)) if grad_scale is None else float(grad_scale)
    _lib.check(L.avsi_frontend_l1_loss_f32(ctypes.byref(a), _lib.ptr(pred), pred.stride(0), pred.stride(1), _lib.ptr(dpred), gs,
                                           _lib.ptr(out3), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "avsi_frontend_l1_loss_f32")
    return out3, dpred


def frontend(sources, sample_rate=16000, window_size=24, step_size=12, n_fft=512,
             num_frames_out=None, num_bins=None, mean=None, std=None, masks=None,
             want_stft=False, want_spec=False, want_feat=False, want_logmel=False,
             power=1.0, log=True, eps=1e-6, time_major=False, feat_cols=None,
             num_mel_bins=80, lower_edge_freq=125.0, upper_edge_freq=7600.0, _feat_out=None):
    """Fused front end: ONE kernel from waveform to every requested output.

    sources [B, N] float32 (device).  Returns a dict with the requested keys:
      'stft'   complex64 [B, T, F]                      (get_stft, audio_processing.py:25-42)
      'spec'   float32 [B, T, F]  |X|^power -> log -> (.-mean)/std  (models.py:32-33)
      'feat'   'spec' * masks; [B, T, feat_cols] or, time_major, [T, B, feat_cols] with
               columns >= F zero (the BLSTM's padded input layout)              (models.py:35)
      'logmel' float32 [B, T, num_mel_bins]  log(mel_W . |X|^2 + eps)  (audio_processing.py:59-72)
    """
    _lib.require_cuda(sources, mean, std, masks)
    L = _lib.lib()
    if sources.dtype != torch.float32 or sources.dim() != 2:
        raise _lib.AvsiError("sources must be float32 [B, N]")
    if sources.stride(1) != 1:
        sources = sources.contiguous()
    dev = sources.device
    B, N = sources.shape
    frame_len = ms_to_samples(window_size, sample_rate)
    hop = ms_to_samples(step_size, sample_rate)
    T_full = num_frames(N, hop)
    T = T_full if num_frames_out is None else int(num_frames_out)
    F = n_fft // 2 + 1 if num_bins is None else int(num_bins)

    a = _lib.FrontendArgs()
    # a size-1 batch axis can carry any stride (numpy's newaxis gives 0): the pitch is then irrelevant
    a.wav, a.batch, a.num_samples, a.wav_stride = _lib.ptr(sources), B, N, sources.stride(0) if B > 1 else N
    a.frame_len, a.hop, a.nfft, a.num_frames, a.num_bins = frame_len, hop, n_fft, T, F
    tab = _tables(dev, frame_len, n_fft)
    a.table = _lib.ptr(tab)
    keep = [tab]
    if mean is not None:
        mean = mean.to(torch.float32).contiguous()
        std = std.to(torch.float32).contiguous()
        a.mean, a.stdev = _lib.ptr(mean), _lib.ptr(std)
        keep += [mean, std]
    if masks is not None:
        if masks.dtype != torch.float32 or masks.stride(2) != 1:
            masks = masks.to(torch.float32).contiguous()
        a.mask, a.mask_stride_b, a.mask_stride_t = _lib.ptr(masks), masks.stride(0), masks.stride(1)
        keep.append(masks)
    out = {}
    if want_stft:
        st = torch.empty((B, T, F, 2), dtype=torch.float32, device=dev)
        a.out_stft, a.stft_stride_b, a.stft_stride_t = _lib.ptr(st), st.stride(0), st.stride(1)
        out['stft'] = torch.view_as_complex(st)
    if want_spec:
        sp = torch.empty((B, T, F), dtype=torch.float32, device=dev)
        a.out_spec, a.spec_stride_b, a.spec_stride_t = _lib.ptr(sp), sp.stride(0), sp.stride(1)
        out['spec'] = sp
    if want_feat:
        cols = F if feat_cols is None else int(feat_cols)
        if _feat_out is not None:
            # write into the caller's (padded) buffer: [T, Bp, C] if time_major else [Bp, T, C]
            ft = _feat_out
            if ft.dtype != torch.float32 or ft.stride(2) != 1 or ft.shape[2] < cols:
                raise _lib.AvsiError("bad _feat_out buffer")
            a.feat_stride_b, a.feat_stride_t = (ft.stride(1), ft.stride(0)) if time_major else (ft.stride(0), ft.stride(1))
        elif time_major:
            ft = torch.empty((T, B, cols), dtype=torch.float32, device=dev)
            a.feat_stride_b, a.feat_stride_t = ft.stride(1), ft.stride(0)
        else:
            ft = torch.empty((B, T, cols), dtype=torch.float32, device=dev)
            a.feat_stride_b, a.feat_stride_t = ft.stride(0), ft.stride(1)
        a.out_feat, a.feat_cols = _lib.ptr(ft), cols
        out['feat'] = ft
    if want_logmel:
        ms, ml, mw, mstride = _mel_bands(dev, num_mel_bins, n_fft // 2 + 1, sample_rate,
                                         lower_edge_freq, upper_edge_freq)
        lm = torch.empty((B, T, num_mel_bins), dtype=torch.float32, device=dev)
        a.out_logmel, a.logmel_stride_b, a.logmel_stride_t = _lib.ptr(lm), lm.stride(0), lm.stride(1)
        a.num_mel, a.mel_start, a.mel_len, a.mel_w, a.mel_w_stride = (
            num_mel_bins, _lib.ptr(ms), _lib.ptr(ml), _lib.ptr(mw), mstride)
        out['logmel'] = lm
    a.spec_power, a.log_spec, a.eps = float(power), int(bool(log)), float(eps)
    _lib.check(L.avsi_frontend_f32(ctypes.byref(a), _lib.stream_ptr()), "avsi_frontend_f32")
    return out


def _out_dims(out_shape, T_full, F_full):
    """out_shape all-zero = no slice (reference tf.cond at audio_processing.py:38-40)."""
    if out_shape is None or all(int(s) == 0 for s in out_shape):
        return T_full, F_full
    return int(out_shape[1]), int(out_shape[2])


def get_stft(sources, sample_rate=16000, window_size=25, step_size=10, n_fft=512, out_shape=[0, 0, 0]):
    """Compute STFT -- reference audio_processing.py:25-42.  Returns complex64 [B, T, F]."""
    hop = ms_to_samples(step_size, sample_rate)
    T, F = _out_dims(out_shape, num_frames(sources.shape[1], hop), n_fft // 2 + 1)
    return frontend(sources, sample_rate, window_size, step_size, n_fft, T, F, want_stft=True)['stft']


_ITABLES = {}


def _istft_tables(device, frame_len, hop, nfft):
    key = (device.index, frame_len, hop, nfft)
    tab = _ITABLES.get(key)
    if tab is None:
        L = _lib.lib()
        n = L.avsi_istft_table_floats(frame_len, hop, nfft)
        if n == 0:
            raise _lib.AvsiError("unsupported inverse-STFT geometry frame_len=%d hop=%d nfft=%d" % (frame_len, hop, nfft))
        tab = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(L.avsi_istft_init_tables(_lib.ptr(tab), frame_len, hop, nfft, _lib.stream_ptr()),
                   "avsi_istft_init_tables")
        _ITABLES[key] = tab
    return tab


def enclosing_power_of_two(n):
    """2 ** ceil(log2(n)): the fft length tf.contrib.signal.inverse_stft uses when none is given."""
    return 1 << max(0, int(n) - 1).bit_length()


def _istft(mode, in0, in1, in2, mean, std, B, T, F, num_samples, sample_rate, window_size, step_size,
           in_strides, in1_strides=(0, 0), in2_strides=(0, 0), nfft=None, wav=None):
    """``nfft=None`` follows TF 1.x: fft length = enclosing power of two of the frame length (the reference never
    passes one, audio_processing.py:149-151), and irfft crops / zero-pads the bin axis to nfft / 2 + 1 -- the
    kernel reads ``min(F, nfft / 2 + 1)`` bins of each row and treats the rest of the grid as zero."""
    _lib.require_cuda(in0, in1, in2, mean, std, wav)
    frame_len = ms_to_samples(window_size, sample_rate)
    hop = ms_to_samples(step_size, sample_rate)
    if nfft is None:
        nfft = enclosing_power_of_two(frame_len)
    if nfft not in (256, 512):
        raise _lib.AvsiError("inverse STFT: fft length %d (frame of %d samples) is not 256 or 512" % (nfft, frame_len))
    F = min(F, nfft // 2 + 1)
    full = (T - 1) * hop + frame_len
    n_out = full if not num_samples or num_samples <= 0 else min(int(num_samples), full)
    dev = in0.device
    out = torch.empty((B, n_out), dtype=torch.float32, device=dev)
    a = _lib.IstftArgs()
    a.mode = mode
    a.in0, a.in_stride_b, a.in_stride_t = _lib.ptr(in0), in_strides[0], in_strides[1]
    a.in1, a.in1_stride_b, a.in1_stride_t = _lib.ptr(in1), in1_strides[0], in1_strides[1]
    a.in2, a.in2_stride_b, a.in2_stride_t = _lib.ptr(in2), in2_strides[0], in2_strides[1]
    a.mean, a.stdev = _lib.ptr(mean), _lib.ptr(std)
    a.batch, a.num_frames, a.num_bins = B, T, F
    a.frame_len, a.hop, a.nfft = frame_len, hop, nfft
    tab = _istft_tables(dev, frame_len, hop, nfft)
    a.table = _lib.ptr(tab)
    a.out, a.out_stride_b, a.num_samples = _lib.ptr(out), out.stride(0), n_out
    if wav is not None:
        a.wav, a.wav_stride_b, a.wav_samples = _lib.ptr(wav), wav.stride(0), wav.shape[1]
    _lib.check(_lib.lib().avsi_istft_f32(ctypes.byref(a), _lib.stream_ptr()), "avsi_istft_f32")
    return out


def reconstruct_sources(stfts, num_samples=0, sample_rate=16000, window_size=16, step_size=8):
    """Compute inverse STFT -- reference audio_processing.py:145-157.  stfts complex64 [B, T, F]."""
    if stfts.dtype != torch.complex64 or stfts.dim() != 3:
        raise _lib.AvsiError("reconstruct_sources needs complex64 [B, T, F]")
    x = torch.view_as_real(stfts.contiguous())
    B, T, F = stfts.shape
    return _istft(0, x, None, None, None, None, B, T, F, num_samples, sample_rate, window_size, step_size,
                  (x.stride(0), x.stride(1)))


def get_sources(mag_spectrograms, rec_ang_spectrograms, num_samples=48000, sample_rate=16000, window_size=24,
                step_size=12):
    """Get waveform from magnitude and phase of STFT -- reference audio_processing.py:160-164."""
    mag = mag_spectrograms.to(torch.float32).contiguous()
    ang = rec_ang_spectrograms.to(torch.float32).contiguous()
    B, T, F = mag.shape
    return _istft(1, mag, ang, None, None, None, B, T, F, num_samples, sample_rate, window_size, step_size,
                  (mag.stride(0), mag.stride(1)))


def enhanced_from_prediction(prediction, mean, std, target_stft, masks=None, num_samples=48000, sample_rate=16000,
                             window_size=24, step_size=12, n_fft=512):
    """Fused StackedBLSTMModel.enhanced_sources (reference models.py:181-197): waveform of
    exp(prediction*std+mean) with the phase of target_stft*masks (masks=None: oracle phase)."""
    pred = prediction.contiguous()
    st = torch.view_as_real(target_stft.contiguous())
    B, T, F = pred.shape
    m = None if masks is None else masks.to(torch.float32).contiguous()
    return _istft(2, pred, st, m, None if mean is None else mean.contiguous(), None if std is None else std.contiguous(),
                  B, T, F, num_samples, sample_rate, window_size, step_size, (pred.stride(0), pred.stride(1)),
                  (st.stride(0), st.stride(1)), (0, 0) if m is None else (m.stride(0), m.stride(1)), nfft=n_fft)


def enhanced_from_prediction_wav(prediction, mean, std, target_sources, masks=None, num_samples=48000, sample_rate=16000,
                                 window_size=24, step_size=12, n_fft=512):
    """enhanced_from_prediction with the target WAVEFORM [B, n] in place of its STFT (avsi_istft_f32 mode 3): the frames
    of every tile are transformed forward inside the kernel -- get_stft's framing and window, audio_processing.py:25-42 --
    and only their phase is used, so the complex spectrogram (514 kB per utterance written by the front end, then read
    here) never exists.  Same result as enhanced_from_prediction(prediction, mean, std, get_stft(target_sources), masks)."""
    pred = prediction.contiguous()
    wav = target_sources.to(torch.float32)
    if wav.dim() != 2 or wav.stride(1) != 1:
        wav = wav.contiguous()
    B, T, F = pred.shape
    m = None if masks is None else masks.to(torch.float32).contiguous()
    return _istft(3, pred, None, m, None if mean is None else mean.contiguous(), None if std is None else std.contiguous(),
                  B, T, F, num_samples, sample_rate, window_size, step_size, (pred.stride(0), pred.stride(1)),
                  (0, 0), (0, 0) if m is None else (m.stride(0), m.stride(1)), nfft=n_fft, wav=wav)


# ---------------------------------------------------------------------------- per-op API (reference names)
def get_spectrogram(stfts, power=1, log=False, out_shape=[0, 0, 0]):
    """|X| (** power) (log(. + 1e-6)) -- reference audio_processing.py:45-56."""
    _lib.require_cuda(stfts)
    x = torch.view_as_real(stfts.contiguous())
    out = torch.empty(stfts.shape, dtype=torch.float32, device=stfts.device)
    _lib.check(_lib.lib().avsi_spectrogram_f32(_lib.ptr(x), _lib.ptr(out), out.numel(), float(power), int(bool(log)),
                                               1e-6, _lib.stream_ptr()), "avsi_spectrogram_f32")
    if out_shape is not None and any(int(s) != 0 for s in out_shape):
        out = out[: out_shape[0], : out_shape[1], : out_shape[2]]
    return out


def get_log_mel_spectrogram(spectrograms, sample_rate=16000, num_spec_bins=257, num_mel_bins=80, lower_edge_freq=125,
                            upper_edge_freq=7600, eps=1e-6, out_shape=[0, 0, 0]):
    """log(spectrograms . mel_W + eps) -- reference audio_processing.py:59-72 (its out_shape slice is
    discarded there, SURVEY B6, and so it is here)."""
    _lib.require_cuda(spectrograms)
    if upper_edge_freq is None:
        upper_edge_freq = sample_rate / 2
    sp = spectrograms.to(torch.float32).contiguous()
    rows = sp.numel() // sp.shape[-1]
    ms, ml, mw, stride = _mel_bands(sp.device, num_mel_bins, num_spec_bins, sample_rate, lower_edge_freq, upper_edge_freq)
    out = torch.empty(sp.shape[:-1] + (num_mel_bins,), dtype=torch.float32, device=sp.device)
    _lib.check(_lib.lib().avsi_logmel_f32(_lib.ptr(sp), sp.shape[-1], rows, num_mel_bins, _lib.ptr(ms), _lib.ptr(ml),
                                          _lib.ptr(mw), stride, _lib.ptr(out), float(eps), _lib.stream_ptr()),
               "avsi_logmel_f32")
    return out


_DCT = {}


def get_mfcc(log_mel_spectrograms, num_mfccs=13, out_shape=[0, 0, 0]):
    """First num_mfccs of DCT-II(log-mel) * rsqrt(2 M) -- reference audio_processing.py:75-82
    (tf.signal.mfccs_from_log_mel_spectrograms), as one fp32-MFMA GEMM with the DCT basis."""
    from . import ops
    x = log_mel_spectrograms.to(torch.float32).contiguous()
    M = x.shape[-1]
    key = (x.device.index, M)
    basis = _DCT.get(key)
    if basis is None:
        n = np.arange(M, dtype=np.float64)
        k = np.arange(M, dtype=np.float64)[:, None]
        b = (2.0 * np.cos(np.pi * k * (2.0 * n + 1.0) / (2.0 * M)) / math.sqrt(2.0 * M)).T      # [n, k]
        basis = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(x.device)
        _DCT[key] = basis
    rows = x.numel() // M
    xm = x.view(rows, M)
    if M % 4:
        xm = torch.nn.functional.pad(xm, (0, 4 - M % 4))
    out = ops.gemm(xm, basis, m=rows, n=M, k=M).view(x.shape)[..., :num_mfccs]
    if out_shape is not None and any(int(s) != 0 for s in out_shape):
        out = out[: out_shape[0], : out_shape[1], : out_shape[2]]
    return out


def delta(features, N=2):
    """Regression deltas -- reference audio_processing.py:85-94."""
    _lib.require_cuda(features)
    x = features.to(torch.float32).contiguous()
    B, T, F = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().avsi_delta_f32(_lib.ptr(x), _lib.ptr(out), B, T, F, int(N), _lib.stream_ptr()), "avsi_delta_f32")
    return out


def add_delta_features(features, n_delta=2, N=2):
    """features ++ delta ++ delta-delta ... along the last axis -- reference audio_processing.py:97-104."""
    feats = [features]
    cur = features
    for _ in range(n_delta):
        cur = delta(cur, N)
        feats.append(cur)
    return torch.cat(feats, dim=2)


def preemphasis(sources, alpha=0.95):
    """y[t] = x[t] - alpha x[t-1] -- reference audio_processing.py:19-22."""
    _lib.require_cuda(sources)
    x = sources.to(torch.float32)
    if x.stride(1) != 1:
        x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().avsi_preemphasis_f32(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1],
                                               x.stride(0) if x.shape[0] > 1 else x.shape[1],
                                               float(alpha), _lib.stream_ptr()), "avsi_preemphasis_f32")
    return out


def downsampling(samples, sample_rate, downsample_rate):
    """FFT resampling of one waveform to `downsample_rate` on the host -- reference audio_processing.py:9-16
    (scipy.signal.resample; offline data preparation, not part of the device path)."""
    from scipy import signal
    if sample_rate == downsample_rate:
        return samples
    return signal.resample(samples, int(downsample_rate * (len(samples) / float(sample_rate))))


def asr_preprocessing(input_sources, type='mfcc', preemph=0.95, num_spec_bins=257, num_mel_bins=80, num_mfccs=13, n_delta=3,
                      feat_mean=None, feat_std=None, stft_shape=[0, 0, 0], name='features'):
    """Front end of the recognition branch -- reference audio_processing.py:107-142: pre-emphasis, STFT
    (25 / 10 ms, the get_stft defaults), then ``type`` = 'spec' (|X| ** 0.3), 'fbanks' (log-mel of the power
    spectrogram) or 'mfcc'; ``n_delta`` orders of regression deltas appended; normalised with feat_mean /
    feat_std (feat_mean = the per-utterance mean over time when only feat_std is given).  With preemph <= 0 the
    reference hits an undefined name; here the sources are used as they are."""
    sources = preemphasis(input_sources, alpha=preemph) if preemph > 0 else input_sources
    stfts = get_stft(sources, out_shape=stft_shape)
    if type == 'spec':
        features = get_spectrogram(stfts, power=0.3)
    else:
        fbanks = get_log_mel_spectrogram(get_spectrogram(stfts, power=2), num_spec_bins=num_spec_bins,
                                         num_mel_bins=num_mel_bins)
        if type == 'fbanks':
            features = fbanks
        elif type == 'mfcc':
            features = get_mfcc(fbanks, num_mfccs=num_mfccs)
        else:
            raise ValueError("type must be 'spec', 'fbanks' or 'mfcc', got %r" % (type,))
    if n_delta > 0:
        features = add_delta_features(features, n_delta=n_delta)
    if feat_mean is None and feat_std is not None:
        feat_mean = features.mean(dim=1, keepdim=True)
    if feat_mean is not None and feat_std is not None:
        dev = features.device
        features = (features - torch.as_tensor(feat_mean, dtype=torch.float32, device=dev)) / \
            torch.as_tensor(feat_std, dtype=torch.float32, device=dev)
    return features


def get_oracle_iam(target_stft, mixed_stft, clip_value=10):
    """Oracle Ideal Amplitude Mask |target| / |mixed| clipped to [0, clip_value] -- reference :167-173."""
    iam = get_spectrogram(target_stft) / get_spectrogram(mixed_stft)
    return torch.clamp(iam, 0, clip_value).to(torch.float32)


def get_oracle_ipsm(target_stft, mixed_stft, min_clip_value=0, max_clip_value=10):
    """Oracle Ideal Phase Sensitive Mask |target| cos(angle(mixed) - angle(target)) / |mixed| -- reference :176-184."""
    ipsm = get_spectrogram(target_stft) * torch.cos(torch.angle(mixed_stft) - torch.angle(target_stft)) / \
        get_spectrogram(mixed_stft)
    return torch.clamp(ipsm, min_clip_value, max_clip_value)
