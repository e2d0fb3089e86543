"""LWS phase reconstruction on the GPU (csrc/lws.hip through avsi_amd.lws) against the float64 oracle
(oracle/lws.py), plus size-independent properties at the reference's full clip length.  Both sides implement the same
written definition of the published algorithm; neither is pinned against the `lws` package (not installable here)."""
import numpy as np
import pytest
import torch

from oracle import lws as OL

pytestmark = pytest.mark.gpu


def _speechlike(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    f0 = 180 + 40 * np.sin(2 * np.pi * 2.0 * t / 16000)
    x = sum(2000 / h * np.sin(2 * np.pi * h * np.cumsum(f0) / 16000) for h in range(1, 9))
    return (x * (0.6 + 0.4 * np.sin(2 * np.pi * 4 * t / 16000)) + rng.normal(0, 100, n)).astype(np.float32)


@pytest.fixture(scope="module")
def L():
    import avsi_amd  # noqa: F401
    from avsi_amd import lws
    return lws


def test_stft_and_istft_match_oracle(L):
    p = L.lws(384, 192, fftsize=512, mode='speech')
    o = OL.LWS(384, 192, fftsize=512, mode='speech')
    x = np.stack([_speechlike(5000, 1), _speechlike(5000, 2)])
    S = p.stft(x)
    assert S.shape == (2, o.num_frames(5000), 257) and S.dtype == np.complex64
    ref = np.stack([o.stft(x[0]), o.stft(x[1])])
    assert np.abs(S - ref).max() < 2e-6 * np.abs(ref).max()
    y = p.istft(ref.astype(np.complex64))
    assert y.shape == (2, (S.shape[1] - 1) * 192 + 512 - 640)
    assert np.abs(y[:, :5000] - x).max() < 2e-6 * np.abs(x).max() * 10
    single = p.stft(x[0])                       # the package's per-utterance call shape
    assert single.shape == S.shape[1:] and np.array_equal(single, S[0])


@pytest.mark.parametrize("shape", [dict(utterances_per_wave=1), dict(utterances_per_wave=2), dict(utterances_per_wave=4),
                                   dict(kernel='skew'), dict(kernel='skew', waves_per_group=4, groups_per_utterance=1),
                                   dict(kernel='skew', waves_per_group=8, groups_per_utterance=2),
                                   dict(kernel='skew', waves_per_group=16, groups_per_utterance=1),
                                   dict(kernel='duo'), dict(kernel='duo', waves_per_group=4, groups_per_utterance=2)])
def test_run_lws_matches_oracle(L, shape):
    """Same sweeps (one 'no future', one online, 12 batch iterations with a fast-decaying threshold so that every
    bin is visited) on three utterances with different gaps, through all kernels: the frame-by-frame one (U utterances
    per wave), the skewed-frame one (frames in the lanes, csrc/lws_skew.hip) and its two-utterances-per-wave form
    (csrc/lws_duo.hip, what large batches take)."""
    kw = dict(nofuture_iterations=1, online_iterations=1, batch_iterations=12, batch_alpha=100, batch_beta=0.9)
    p = L.lws(384, 192, fftsize=512, **shape, **kw)
    o = OL.LWS(384, 192, fftsize=512, **kw)
    specs = []
    for i, gap in enumerate([(8, 14), (3, 9), (12, 20)]):
        S = o.stft(_speechlike(3840, 10 + i))
        S[gap[0]:gap[1]] = np.abs(S[gap[0]:gap[1]])
        specs.append(S)
    S0 = np.stack(specs)
    got = p.run_lws(S0.astype(np.complex64))
    ref = np.stack([o.run_lws(s.astype(np.complex64)) for s in S0])
    np.testing.assert_allclose(np.abs(got), np.abs(S0), rtol=2e-5, atol=1e-3)      # magnitudes are kept
    # phases: compare the complex values, weighted by what a listener gets (large bins matter, tiny ones are noise)
    err = np.abs(got - ref)
    assert np.sqrt((err ** 2).sum() / (np.abs(ref) ** 2).sum()) < 2e-3
    assert err.max() < 2e-2 * np.abs(ref).max()
    # and the point of it all: as consistent as the oracle's result
    for b in range(3):
        assert o.inconsistency(got[b].astype(np.complex128)) < 1.05 * o.inconsistency(ref[b]) + 1e-6


@pytest.mark.parametrize("frame,hop", [(512, 256), (320, 160), (400, 200)])
def test_other_window_geometries_match_oracle(L, frame, hop):
    """Geometries other than the reference's 384 / 192 (inference.py:119) take the skewed kernel's general form -- the
    consistency weights as kernel arguments instead of compile-time constants (lws_skew_kernel<8, false>) -- and the
    frame-by-frame kernel: both against the oracle, on gapped spectrograms."""
    kw = dict(nofuture_iterations=1, online_iterations=1, batch_iterations=10, batch_alpha=100, batch_beta=0.9)
    o = OL.LWS(frame, hop, fftsize=512, **kw)
    specs = []
    for i, gap in enumerate([(6, 11), (2, 7)]):
        S = o.stft(_speechlike(20 * hop, 30 + i))
        S[gap[0]:gap[1]] = np.abs(S[gap[0]:gap[1]])
        specs.append(S)
    S0 = np.stack(specs)
    ref = np.stack([o.run_lws(s.astype(np.complex64)) for s in S0])
    for kernel in ('skew', 'raster'):
        got = L.lws(frame, hop, fftsize=512, kernel=kernel, **kw).run_lws(S0.astype(np.complex64))
        np.testing.assert_allclose(np.abs(got), np.abs(S0), rtol=2e-5, atol=1e-3)
        err = np.abs(got - ref)
        assert np.sqrt((err ** 2).sum() / (np.abs(ref) ** 2).sum()) < 2e-3, kernel
        for b in range(2):
            assert o.inconsistency(got[b].astype(np.complex128)) < 1.05 * o.inconsistency(ref[b]) + 1e-6, kernel


def test_launch_shape_does_not_change_the_result(L):
    """Utterances per wave (1, 2, 4) and waves per group (the sweeps of an utterance pipelined over 4, 8 or 16 waves,
    each sweep trailing its predecessor by two rows, over one or several workgroups): the raster-order dependences are kept exactly, so the results
    are bit-identical to one wave running the sweeps one after the other."""
    kw = dict(fftsize=512, mode='speech')
    o = OL.LWS(384, 192, **kw)
    S0 = np.stack([np.abs(o.stft(_speechlike(4800, 30 + i))) for i in range(5)]).astype(np.complex64)
    ref = L.lws(384, 192, utterances_per_wave=1, waves_per_group=1, kernel='raster', **kw).run_lws(S0)
    for U, NW, G in ((2, 1, 1), (4, 1, 1), (1, 4, 1), (1, 8, 1), (1, 16, 1), (1, 16, 3), (1, 16, 7), (1, 8, 5), (1, 16, 0),
                     (2, 8, 1), (4, 4, 1), (1, 4, 26), (0, 0, 0)):       # 5 utterances with 0, 0, 0: the policy's own choice
        out = L.lws(384, 192, utterances_per_wave=U, waves_per_group=NW, groups_per_utterance=G, kernel='raster', **kw).run_lws(S0)
        assert np.array_equal(ref, out), (U, NW, G)


def test_skewed_kernel_launch_shape_does_not_change_the_result(L):
    """The skewed-frame kernel: stages per workgroup (4, 8, 16) and workgroups per utterance (1 .. 7, 0 = the policy) only
    decide which wave runs which sweep and whether a hand-over goes through the CU's L1 or through device memory -- every
    bin sees exactly the values the raster order gives it, so the results are bit-identical.  252 frames (four rounds of
    the 64 lanes, the last one partial) and 26 frames (less than one round)."""
    kw = dict(fftsize=512, mode='speech', kernel='skew')
    o = OL.LWS(384, 192, fftsize=512, mode='speech')
    for n, B in ((48000, 3), (4800, 5)):
        S0 = np.stack([np.abs(o.stft(_speechlike(n, 60 + i))) for i in range(B)]).astype(np.complex64)
        ref = L.lws(384, 192, waves_per_group=4, groups_per_utterance=1, **kw).run_lws(S0)
        for NW, G in ((8, 1), (16, 1), (4, 3), (8, 5), (16, 7), (16, 0), (0, 0)):
            out = L.lws(384, 192, waves_per_group=NW, groups_per_utterance=G, **kw).run_lws(S0)
            assert np.array_equal(ref, out), (n, NW, G)
        # The other kernel sums the 33 taps in another order and takes the magnitudes per sweep instead of once: from an
        # all-zero-phase start (nothing known, every bin free) a hundred sweeps amplify that into a different, equally valid
        # solution -- what must agree is how CONSISTENT the results are (tests against the oracle hold both to 2e-3 where
        # the problem is well posed: gaps in a known spectrogram)
        other = L.lws(384, 192, fftsize=512, mode='speech', kernel='raster').run_lws(S0)
        for b_ in range(B):
            i_s, i_r = o.inconsistency(ref[b_].astype(np.complex128)), o.inconsistency(other[b_].astype(np.complex128))
            assert i_s < 1.5 * i_r + 1e-6 and i_r < 1.5 * i_s + 1e-6, (n, b_, i_s, i_r)


def test_duo_kernel_is_bit_identical_to_the_skewed_kernel(L, monkeypatch):
    """Two utterances per wave (csrc/lws_duo.hip): frames nine bins apart instead of six, the neighbour rows' sums through LDS
    instead of DPP rotates, rings of 15 instead of 12 -- every bin still sees exactly the values the raster order gives it and
    every sum is taken in the same order (with every multiply-add that the compiler could fuse or not spelled out as fused),
    so the results equal the skewed kernel's bit for bit, whatever the launch shape.  252 frames (eight rounds of a half's 32
    lanes, the last one partial), 26 frames (less than one round), batches that leave a pair half empty; reference settings
    (102 sweeps, among them a 'no future' one that takes the past-only form); all-zero phases (ill-posed: any difference in
    rounding would be amplified to a different solution) and a known spectrogram with gaps."""
    kw = dict(fftsize=512, mode='speech')
    o = OL.LWS(384, 192, fftsize=512, mode='speech')
    for n, B in ((48000, 3), (4800, 5), (48000, 9)):
        S0 = np.stack([np.abs(o.stft(_speechlike(n, 60 + i))) for i in range(B)]).astype(np.complex64)
        if B == 9:      # a known spectrogram with gaps, and one utterance that is silent but for one bin (idle sweeps)
            S0 = np.stack([o.stft(_speechlike(n, 90 + i)) for i in range(B)]).astype(np.complex64)
            S0[:, 100:133] = np.abs(S0[:, 100:133])
            S0[4] *= 0
            S0[4, 7, 30] = 5.0
        ref = L.lws(384, 192, kernel='skew', **kw).run_lws(S0)
        for NW, G in ((0, 0), (4, 1), (8, 3), (16, 2), (16, 1)):
            out = L.lws(384, 192, kernel='duo', waves_per_group=NW, groups_per_utterance=G, **kw).run_lws(S0)
            assert np.array_equal(ref, out), (n, B, NW, G)
        # more pairs than the chip holds at a time: a workgroup runs several pairs one after the other, each wave going on to
        # the next pair as soon as it is through with this one (here: two / one slot instead of 256)
        for slots, NW, G in ((2, 16, 1), (1, 4, 2), (2, 8, 1)):
            monkeypatch.setenv('AVSI_LWS_DUO_SLOTS', str(slots))
            out = L.lws(384, 192, kernel='duo', waves_per_group=NW, groups_per_utterance=G, **kw).run_lws(S0)
            monkeypatch.delenv('AVSI_LWS_DUO_SLOTS')
            assert np.array_equal(ref, out), (n, B, 'slots', slots, NW, G)


def test_unsupported_geometry_is_reported_by_the_duo_kernel(L):
    """(which kernel a batch takes: tests/test_lws_host.py, no GPU needed)"""
    from avsi_amd import _lib
    with pytest.raises(_lib.AvsiError):
        L.lws(512, 256, fftsize=512, kernel='duo').run_lws(np.zeros((1, 12, 257), np.complex64) + 1)


def test_refine_enhanced_matches_oracle(L):
    """inference.py:141-154 end to end (stft, mask stitching, run_lws, stitching, istft) on a batch."""
    kw = dict(nofuture_iterations=1, online_iterations=1, batch_iterations=10, batch_alpha=100, batch_beta=1.0)
    p = L.lws(384, 192, fftsize=512, **kw)
    o = OL.LWS(384, 192, fftsize=512, **kw)
    x = np.stack([_speechlike(3840, 40), _speechlike(3840, 41)])
    masks = np.ones((2, 20, 257), dtype=np.float32)
    masks[0, 6:11] = 0
    masks[1, 12:17] = 0
    got = p.refine_enhanced(torch.from_numpy(x).cuda(), torch.from_numpy(masks).cuda(), num_samples=3840).cpu().numpy()
    ref = np.stack([OL.refine_enhanced(o, x[b].astype(np.float64), masks[b])[:3840] for b in range(2)])
    assert got.shape == (2, 3840)
    assert np.sqrt(np.mean((got - ref) ** 2)) < 2e-3 * np.abs(ref).max()


def test_full_length_gap_becomes_consistent(L):
    """A 3 s clip (252 frames) with a 400 ms gap of zero phases, reference settings (102 sweeps): the result keeps
    the magnitudes and is far more consistent than the input; more sweeps never make it worse (sampled)."""
    o = OL.LWS(384, 192, fftsize=512, mode='speech')
    S = o.stft(_speechlike(48000, 50))
    S0 = S.copy()
    S0[100:133] = np.abs(S0[100:133])
    before = o.inconsistency(S0)
    S0 = S0.astype(np.complex64)
    vals = []
    for iters in (0, 20, 50, 100):
        p = L.lws(384, 192, fftsize=512, mode='speech')
        p.batch_iterations = iters
        out = p.run_lws(S0)
        np.testing.assert_allclose(np.abs(out), np.abs(S0), rtol=1e-4, atol=1e-2)
        vals.append(o.inconsistency(out.astype(np.complex128)))
    assert vals[-1] < 0.05 * before
    assert vals[1] >= vals[2] * 0.999 and vals[2] >= vals[3] * 0.999
