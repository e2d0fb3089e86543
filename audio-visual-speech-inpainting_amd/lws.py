"""LWS ("local weighted sums") phase reconstruction on MI355X (host side).

The reference refines the gap phases of every enhanced waveform with the third-party ``lws`` package when
``--oracle_phase`` is not given (``av_speech_inpainting/inference.py:119,141-154``):

    lws_processor = lws.lws(384, 192, fftsize=512, mode='speech')
    stft = lws_processor.stft(enhanced) ... rec_stft = lws_processor.run_lws(...) ... lws_processor.istft(...)

``lws`` here mirrors that class -- same constructor arguments and the ``stft`` / ``run_lws`` / ``istft`` methods --
on top of the gfx950 kernels of ``csrc/lws.hip`` (C ABI ``avsi_lws_*``); arrays may be numpy (host in, host out,
like the package) or torch device tensors, single ``[n]`` / ``[M, F]`` or batched along a leading axis.
``refine_enhanced`` is the whole of inference.py:141-154 for a batch, without leaving the device.

The package itself is not part of the reference tree and not installable here: the kernels implement the published
algorithm under the conventions written down in ``oracle/lws.py`` (UNPINNED against the package).
"""
import os

import numpy as np
import torch

from . import _lib

_TABLES = {}


def _table(device, fsize, fshift, fftsize):
    key = (device.index, fsize, fshift, fftsize)
    tab = _TABLES.get(key)
    if tab is None:
        L = _lib.lib()
        n = L.avsi_lws_table_floats(fsize, fshift, fftsize)
        if n == 0:
            raise _lib.AvsiError("unsupported LWS geometry: window %d, shift %d, fft %d" % (fsize, fshift, fftsize))
        tab = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(L.avsi_lws_init_tables(_lib.ptr(tab), fsize, fshift, fftsize, _lib.stream_ptr()), "avsi_lws_init_tables")
        _TABLES[key] = tab
    return tab


class lws(object):
    """``lws.lws(awin_or_fsize, fshift, L=5, ..., mode=None, fftsize=None)`` of the package, window LENGTH form
    (the reference passes 384; a custom analysis window is not supported)."""

    def __init__(self, awin_or_fsize, fshift, L=5, swin=None, look_ahead=3, nofuture_iterations=0, nofuture_alpha=1,
                 online_iterations=0, online_alpha=1, batch_iterations=100, batch_alpha=100, batch_beta=0.1, batch_gamma=1,
                 symmetric_win=True, mode=None, fftsize=None, utterances_per_wave=0, waves_per_group=0,
                 groups_per_utterance=0, kernel=None):
        """``kernel``: 'auto' (default; AVSI_LWS_KERNEL overrides) = 'duo' from AVSI_LWS_DUO_MIN utterances of the
        reference's geometry, 'skew' below; 'skew' = the sweeps with the frames of ONE utterance in the lanes of a wave
        (csrc/lws_skew.hip), 'duo' = two utterances per wave, every lane busy (csrc/lws_duo.hip; bit-identical to
        'skew'), 'raster' = the frame-by-frame kernel of csrc/lws.hip (``utterances_per_wave`` > 0 selects it too: that
        argument only exists there).  ``waves_per_group`` / ``groups_per_utterance`` (per pair of utterances for 'duo')
        shape the pipeline of sweeps of any kernel; results do not depend on them."""
        if not isinstance(awin_or_fsize, (int, np.integer)) or swin is not None or not symmetric_win:
            raise _lib.AvsiError("lws: only the window-length form with the default sqrt-Hann windows is implemented")
        if mode == 'speech':
            look_ahead, nofuture_iterations, nofuture_alpha = 3, 1, 1
            online_iterations, online_alpha, batch_iterations, batch_alpha = 1, 1, 100, 100
        elif mode == 'music':
            look_ahead, nofuture_iterations, nofuture_alpha = 3, 1, 1
            online_iterations, online_alpha, batch_iterations, batch_alpha = 10, 1, 100, 100
        elif mode is not None:
            raise ValueError("mode must be None, 'speech' or 'music'")
        self.fsize, self.fshift, self.L = int(awin_or_fsize), int(fshift), int(L)
        self.fftsize = int(fftsize) if fftsize else self.fsize
        self.look_ahead = look_ahead
        self.nofuture_iterations, self.nofuture_alpha = int(nofuture_iterations), float(nofuture_alpha)
        self.online_iterations, self.online_alpha = int(online_iterations), float(online_alpha)
        self.batch_iterations = int(batch_iterations)
        self.batch_alpha, self.batch_beta, self.batch_gamma = float(batch_alpha), float(batch_beta), float(batch_gamma)
        self.utterances_per_wave, self.waves_per_group = int(utterances_per_wave), int(waves_per_group)
        self.groups_per_utterance = int(groups_per_utterance)
        self.kernel = kernel or ('raster' if self.utterances_per_wave else os.environ.get('AVSI_LWS_KERNEL', 'auto'))
        if self.kernel not in ('auto', 'skew', 'duo', 'raster'):
            raise ValueError("kernel must be 'auto', 'skew', 'duo' or 'raster'")
        self.duo_min = int(os.environ.get('AVSI_LWS_DUO_MIN', '128'))
        self._status = None
        if _lib.lib().avsi_lws_table_floats(self.fsize, self.fshift, self.fftsize) == 0:
            raise _lib.AvsiError("unsupported LWS geometry: window %d, shift %d, fft %d" % (self.fsize, self.fshift, self.fftsize))

    # ------------------------------------------------------------------------------------------------ helpers
    @staticmethod
    def _to_device(x, dtype):
        host = not isinstance(x, torch.Tensor)
        t = torch.as_tensor(np.asarray(x)) if host else x
        _lib.require_cuda()
        return t.to(device='cuda', dtype=dtype), host

    def num_frames(self, num_samples):
        return _lib.lib().avsi_lws_num_frames(int(num_samples), self.fshift, self.fftsize)

    # ------------------------------------------------------------------------------------------------ package API
    def stft(self, x):
        """waveform [n] or [B, n] -> complex64 [M, F] or [B, M, F] (numpy in -> numpy out)."""
        w, host = self._to_device(x, torch.float32)
        single = w.dim() == 1
        w = (w[None] if single else w).contiguous()
        B, n = w.shape
        M, F = self.num_frames(n), self.fftsize // 2 + 1
        spec = torch.empty((B, M, F, 2), dtype=torch.float32, device=w.device)
        tab = _table(w.device, self.fsize, self.fshift, self.fftsize)
        _lib.check(_lib.lib().avsi_lws_stft_f32(_lib.ptr(w), n, B, n, _lib.ptr(tab), self.fshift, self.fftsize, _lib.ptr(spec),
                                                M, _lib.stream_ptr()), "avsi_lws_stft_f32")
        out = torch.view_as_complex(spec)
        out = out[0] if single else out
        return out.cpu().numpy() if host else out

    def _as_spec(self, S):
        s, host = self._to_device(S, torch.complex64)
        single = s.dim() == 2
        s = torch.view_as_real((s[None] if single else s).contiguous().clone())
        if s.shape[2] != self.fftsize // 2 + 1:
            raise _lib.AvsiError("spectrogram must have %d bins" % (self.fftsize // 2 + 1))
        return s, host, single

    def run_lws(self, S):
        """complex spectrogram (magnitudes kept, phases = starting point) -> complex spectrogram."""
        s, host, single = self._as_spec(S)
        self._run(s)
        self.check()
        out = torch.view_as_complex(s)
        out = out[0] if single else out
        return out.cpu().numpy() if host else out

    def _standard_geometry(self):
        return (self.fsize, self.fshift, self.fftsize, self.L) == (384, 192, 512, 5)

    def kernel_for(self, batch):
        """The kernel a batch of this size takes: 'auto' = two utterances per wave once the batch fills the chip that way."""
        if self.kernel != 'auto':
            return self.kernel
        return 'duo' if batch >= self.duo_min and self._standard_geometry() else 'skew'

    def _run(self, s):
        B, M = s.shape[0], s.shape[1]
        kernel = self.kernel_for(B)
        if kernel in ('skew', 'duo'):
            L = _lib.lib()
            size, run = ((L.avsi_lws_run_duo_workspace_bytes, L.avsi_lws_run_duo_f32) if kernel == 'duo' else
                         (L.avsi_lws_run_skew_workspace_bytes, L.avsi_lws_run_skew_f32))
            need = size(B, M)
            if self._status is None or self._status.device != s.device or self._status.numel() * 4 < need:
                self._status = None                 # (drop the old workspace first: they are 1.4 - 1.7 MB per utterance)
                self._status = torch.zeros((need + 3) // 4, dtype=torch.int32, device=s.device)
            _lib.check(run(_lib.ptr(s), B, M, self.fsize, self.fshift, self.fftsize, self.L, self.nofuture_iterations,
                           self.nofuture_alpha, self.online_iterations, self.online_alpha, self.batch_iterations, self.batch_alpha,
                           self.batch_beta, self.batch_gamma, self.waves_per_group, self.groups_per_utterance,
                           _lib.ptr(self._status), self._status.numel() * 4, _lib.stream_ptr()), "avsi_lws_run_%s_f32" % kernel)
            return
        need = _lib.lib().avsi_lws_run_workspace_bytes(B)
        if self._status is None or self._status.device != s.device or self._status.numel() * 4 < need:
            self._status = torch.zeros((need + 3) // 4, dtype=torch.int32, device=s.device)     # word 0: status
        _lib.check(_lib.lib().avsi_lws_run_f32(_lib.ptr(s), B, M, self.fsize, self.fshift, self.fftsize, self.L,
                                               self.nofuture_iterations, self.nofuture_alpha, self.online_iterations,
                                               self.online_alpha, self.batch_iterations, self.batch_alpha, self.batch_beta,
                                               self.batch_gamma, self.utterances_per_wave, self.waves_per_group,
                                               self.groups_per_utterance, _lib.ptr(self._status), self._status.numel() * 4,
                                               _lib.stream_ptr()), "avsi_lws_run_f32")

    def check(self):
        """Raise if a pipeline stage of the last run gave up waiting for its predecessor (synchronises)."""
        if self._status is not None and int(self._status[0].item()) != 0:
            raise _lib.AvsiError("LWS sweep pipeline timed out waiting for a predecessor stage; results are invalid")

    def istft(self, S, num_samples=None):
        """complex [M, F] or [B, M, F] -> waveform (front / back padding removed; ``num_samples`` truncates)."""
        s, host, single = self._as_spec(S)
        out = self._istft(s, num_samples)
        out = out[0] if single else out
        return out.cpu().numpy() if host else out

    def _istft(self, s, num_samples=None):
        B, M = s.shape[0], s.shape[1]
        avail = (M - 1) * self.fshift + self.fftsize - 2 * (self.fftsize - self.fshift)
        n = avail if num_samples is None else min(int(num_samples), avail)
        L = _lib.lib()
        ws = torch.empty(L.avsi_lws_istft_workspace_bytes(B, M, self.fftsize) // 4, dtype=torch.float32, device=s.device)
        out = torch.empty((B, n), dtype=torch.float32, device=s.device)
        tab = _table(s.device, self.fsize, self.fshift, self.fftsize)
        _lib.check(L.avsi_lws_istft_f32(_lib.ptr(s), B, M, _lib.ptr(tab), self.fshift, self.fftsize, _lib.ptr(out), n, n,
                                        _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "avsi_lws_istft_f32")
        return out

    # ------------------------------------------------------------------------------------------------ inference.py:141-154
    def status_word(self):
        """Device int32[1], a COPY taken on the current stream right behind the runs enqueued so far: non-zero if a
        pipeline stage of the last run on this object gave up waiting (see check()).  A copy, not a view: every run
        resets word 0 of the workspace at its start and the workspace is replaced when a batch outgrows it, so a consumer
        on another stream (inference._WavWriter) that read the word itself could see the NEXT run's reset, or freed
        memory, instead of this run's verdict."""
        if self._status is None:
            return torch.zeros(1, dtype=torch.int32, device='cuda')
        return self._status[:1].clone()

    def refine_enhanced(self, enhanced, masks, num_samples=None, check=True):
        """The reference's per-utterance block, batched on the device: ``enhanced`` [B, n] (output of
        ``enhanced_sources``: masked target phase, zero phase inside the gaps), ``masks`` [B, T, F] ->
        [B, num_samples or all] with the gap phases replaced by LWS estimates."""
        _lib.require_cuda(enhanced, masks)
        w = enhanced.to(torch.float32).contiguous()
        m = masks.to(torch.float32)
        if m.stride(2) != 1:
            m = m.contiguous()
        B, n = w.shape
        M, F = self.num_frames(n), self.fftsize // 2 + 1
        L = _lib.lib()
        tab = _table(w.device, self.fsize, self.fshift, self.fftsize)
        init = torch.empty((B, M, F, 2), dtype=torch.float32, device=w.device)
        _lib.check(L.avsi_lws_stft_f32(_lib.ptr(w), n, B, n, _lib.ptr(tab), self.fshift, self.fftsize, _lib.ptr(init), M,
                                       _lib.stream_ptr()), "avsi_lws_stft_f32")
        msb = m.stride(0) if B > 1 else m.shape[1] * m.stride(1)
        # mag_spec * exp(1j * ang_spec): phases kept where the mask is one, zero in the gaps
        _lib.check(L.avsi_lws_stitch_f32(_lib.ptr(init), None, _lib.ptr(m), msb, m.stride(1), m.shape[1], m.shape[2], B, M,
                                         self.fftsize, _lib.stream_ptr()), "avsi_lws_stitch_f32")
        rec = init.clone()
        self._run(rec)
        # rec_mag * exp(1j * (ang_spec + rec_ang * (1 - mask_adj)))
        _lib.check(L.avsi_lws_stitch_f32(_lib.ptr(rec), _lib.ptr(init), _lib.ptr(m), msb, m.stride(1), m.shape[1], m.shape[2],
                                         B, M, self.fftsize, _lib.stream_ptr()), "avsi_lws_stitch_f32")
        out = self._istft(rec, num_samples)
        if check:               # (synchronises; callers that check status_word() where the result arrives pass False)
            self.check()
        return out
