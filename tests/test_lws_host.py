"""Host side of the LWS phase refinement that needs no GPU: which sweep kernel a batch takes (avsi_amd.lws.lws.kernel_for), the
workspace queries of the C ABI, and the argument checks that come before any launch (inference.py:119,141-154 is the path)."""
import ctypes

import pytest

import avsi_amd  # noqa: F401
from avsi_amd import _lib, lws


def test_kernel_choice_by_batch_and_geometry(monkeypatch):
    monkeypatch.delenv('AVSI_LWS_KERNEL', raising=False)
    monkeypatch.delenv('AVSI_LWS_DUO_MIN', raising=False)
    p = lws.lws(384, 192, fftsize=512, mode='speech')
    assert p.kernel == 'auto' and p.duo_min == 128
    assert [p.kernel_for(b) for b in (1, 32, 127, 128, 1024, 8192)] == ['skew', 'skew', 'skew', 'duo', 'duo', 'duo']
    # other geometries have no compile-time weights: the general form of the skewed kernel, whatever the batch
    assert lws.lws(512, 256, fftsize=512, mode='speech').kernel_for(4096) == 'skew'
    assert lws.lws(384, 192, fftsize=512, L=4).kernel_for(4096) == 'skew'
    # an explicit choice is kept; utterances_per_wave belongs to the frame-by-frame kernel
    assert lws.lws(384, 192, fftsize=512, kernel='skew').kernel_for(4096) == 'skew'
    assert lws.lws(384, 192, fftsize=512, kernel='duo').kernel_for(2) == 'duo'
    assert lws.lws(384, 192, fftsize=512, utterances_per_wave=2).kernel == 'raster'
    monkeypatch.setenv('AVSI_LWS_DUO_MIN', '512')
    assert lws.lws(384, 192, fftsize=512).kernel_for(256) == 'skew'
    monkeypatch.setenv('AVSI_LWS_KERNEL', 'raster')
    assert lws.lws(384, 192, fftsize=512).kernel == 'raster'
    with pytest.raises(ValueError):
        lws.lws(384, 192, fftsize=512, kernel='quad')


def test_workspace_queries():
    L = _lib.lib()
    for fn in (L.avsi_lws_run_skew_workspace_bytes, L.avsi_lws_run_duo_workspace_bytes):
        assert fn(0, 252) == 0 and fn(4, 0) == 0 and fn(-1, 252) == 0
        assert fn(2, 252) > 0 and fn(1024, 252) > fn(512, 252) > fn(2, 252)
    # two utterances share a pair's arrays (rows 9 frames apart instead of 6 per utterance): ~1.0 MB against ~1.4 MB per utterance
    duo, skew = L.avsi_lws_run_duo_workspace_bytes(1024, 252), L.avsi_lws_run_skew_workspace_bytes(1024, 252)
    assert 0.9e6 < duo / 1024 < 1.2e6 and 1.3e6 < skew / 1024 < 1.6e6
    assert L.avsi_lws_run_duo_workspace_bytes(3, 252) == L.avsi_lws_run_duo_workspace_bytes(4, 252)      # a half-empty pair (256-byte granules)


def test_argument_checks_come_before_any_launch():
    L = _lib.lib()
    buf = (ctypes.c_float * 16)()
    common = (384, 192, 512, 5, 1, 1.0, 1, 1.0, 100, 100.0, 0.1, 1.0)
    for run in (L.avsi_lws_run_skew_f32, L.avsi_lws_run_duo_f32):
        assert run(None, 4, 252, *common, 0, 0, None, 0, None) == _lib.AVSI_ERR_INVALID_ARG
        assert run(buf, 0, 252, *common, 0, 0, None, 0, None) == _lib.AVSI_ERR_INVALID_ARG
        assert run(buf, 4, 252, *common, 0, 0, None, 0, None) == _lib.AVSI_ERR_WORKSPACE
        assert run(buf, 4, 252, *common, 5, 0, buf, 1 << 40, None) == _lib.AVSI_ERR_INVALID_ARG          # waves per group: 4, 8, 16
        assert run(buf, 4, 252, 384, 192, 500, 5, *common[4:], 0, 0, None, 0, None) == _lib.AVSI_ERR_UNSUPPORTED   # fft length
    # the reference's geometry only for the two-utterances-per-wave kernel
    assert L.avsi_lws_run_duo_f32(buf, 4, 252, 512, 256, 512, 5, *common[4:], 0, 0, None, 0, None) == _lib.AVSI_ERR_UNSUPPORTED
    assert L.avsi_lws_run_duo_f32(buf, 4, 252, 384, 192, 512, 4, *common[4:], 0, 0, None, 0, None) == _lib.AVSI_ERR_UNSUPPORTED
