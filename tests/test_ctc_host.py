"""Host side of the CTC head: the C++ beam-search decoder of libavsi_hip.so (no GPU work) against the
oracle's restatement of TF's decoder, and the edit-distance bookkeeping."""
import numpy as np
import pytest

from oracle import ctc as OC


@pytest.fixture(scope="module")
def ops():
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    return ops


@pytest.mark.parametrize("beam_width", [1, 3, 20])
def test_beam_search_matches_oracle(ops, beam_width):
    rng = np.random.default_rng(beam_width)
    for trial in range(6):
        B, T, C = 4, 40, 9
        logits = rng.normal(0, 1.5, size=(B, T, C)).astype(np.float32)
        seq = rng.integers(1, T + 1, size=B)
        dec, dlen, lp = ops.ctc_beam_search(logits, seq, beam_width=beam_width)
        outs, scores = OC.beam_search(logits, seq, beam_width=beam_width)
        for b in range(B):
            assert dec[b, :dlen[b]].tolist() == outs[b]
            assert (dec[b, dlen[b]:] == -1).all()
            assert abs(lp[b] - scores[b]) < 1e-3


def test_beam_search_grid_sized_and_options(ops):
    """The reference's shape: 250 frames, 34 classes (33 phones + blank), beam 20."""
    rng = np.random.default_rng(7)
    logits = rng.normal(0, 2.0, size=(2, 250, 34)).astype(np.float32)
    dec, dlen, lp = ops.ctc_beam_search(logits, [250, 120])
    outs, scores = OC.beam_search(logits, [250, 120])
    for b in range(2):
        assert dec[b, :dlen[b]].tolist() == outs[b]
        assert abs(lp[b] - scores[b]) < 2e-2
    raw, rlen, _ = ops.ctc_beam_search(logits, [250, 120], merge_repeated=False)
    merged = [v for i, v in enumerate(raw[0, :rlen[0]]) if i == 0 or v != raw[0, i - 1]]
    assert merged == dec[0, :dlen[0]].tolist()
    # a zero-length utterance decodes to nothing
    dec0, dlen0, _ = ops.ctc_beam_search(logits, [0, 3])
    assert dlen0[0] == 0


def test_beam_search_rejects_bad_arguments(ops):
    import avsi_amd
    with pytest.raises(avsi_amd._lib.AvsiError):
        ops.ctc_beam_search(np.zeros((1, 4, 1), np.float32), [4])
    with pytest.raises(avsi_amd._lib.AvsiError):
        ops.ctc_beam_search(np.zeros((1, 4, 3), np.float32), [4], beam_width=0)


def test_edit_distance_matches_oracle(ops):
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.integers(0, 5, size=rng.integers(0, 9)).tolist()
        b = rng.integers(0, 5, size=rng.integers(0, 9)).tolist()
        for norm in (True, False):
            assert ops.edit_distance(a, b, norm) == OC.edit_distance(a, b, norm)


def test_beam_search_and_per_against_the_golden(ops):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_golden.npz'))
    dec, dlen, lp = ops.ctc_beam_search(g['logits'], g['sequence_lengths'], beam_width=20)
    np.testing.assert_array_equal(dlen, g['decoded_lengths'])
    for b in range(len(dlen)):
        assert dec[b, :dlen[b]].tolist() == g['decoded'][b, :dlen[b]].tolist()
        per = ops.edit_distance(dec[b, :dlen[b]], g['labels'][b, :g['labels_lengths'][b]].astype(int))
        assert per == g['per'][b] or abs(per - g['per'][b]) < 1e-12
    np.testing.assert_allclose(lp, g['log_prob'], atol=2e-3)
