"""The CTC oracle against independent implementations: torch's ctc_loss + autograd (float64) and an
exhaustive enumeration of alignments (tiny cases)."""
import itertools

import numpy as np
import pytest
import torch

from oracle import ctc as OC


def _case(seed, B=5, T=12, C=6, Lmax=5):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2.0, size=(B, T, C))
    lab_len = rng.integers(0, Lmax + 1, size=B)
    seq_len = rng.integers(T // 2, T + 1, size=B)
    seq_len[0], lab_len[0] = T, Lmax
    labels = np.zeros((B, Lmax), dtype=np.float32)
    for b in range(B):
        labels[b, :lab_len[b]] = rng.integers(0, C - 1, size=lab_len[b])
    # feasibility: repeated labels need a blank between them
    for b in range(B):
        lab = labels[b, :lab_len[b]]
        need = len(lab) + sum(lab[i] == lab[i - 1] for i in range(1, len(lab)))
        seq_len[b] = max(seq_len[b], min(T, need))
    return logits, labels, lab_len, seq_len


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_loss_and_gradient_match_torch(seed):
    logits, labels, lab_len, seq_len = _case(seed)
    B, T, C = logits.shape
    loss, grad = OC.ctc_loss(logits, labels, lab_len, seq_len)
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, dim=2).transpose(0, 1)
    tl = torch.nn.functional.ctc_loss(lp, torch.tensor(labels, dtype=torch.long), torch.tensor(seq_len), torch.tensor(lab_len),
                                      blank=C - 1, reduction='none')
    tl.sum().backward()
    feasible = np.isfinite(loss)
    assert feasible.all()
    np.testing.assert_allclose(loss, tl.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(grad, x.grad.numpy(), rtol=1e-9, atol=1e-10)
    # frames past the utterance end carry no gradient; gradient rows sum to zero (softmax - posterior)
    for b in range(B):
        assert not grad[b, seq_len[b]:].any()
    np.testing.assert_allclose(grad.sum(axis=2), 0.0, atol=1e-10)


def _enumerate(logits):
    """{labelling: probability} by summing over every alignment of a tiny utterance."""
    T, C = logits.shape
    p = np.exp(OC.log_softmax(logits))
    out = {}
    for path in itertools.product(range(C), repeat=T):
        pr = np.prod([p[t, k] for t, k in enumerate(path)])
        lab = tuple(k for i, k in enumerate(path) if k != C - 1 and (i == 0 or k != path[i - 1]))
        out[lab] = out.get(lab, 0.0) + pr
    return out


def test_loss_matches_exhaustive_enumeration():
    rng = np.random.default_rng(3)
    logits = rng.normal(0, 1.5, size=(5, 4))
    table = _enumerate(logits)
    assert abs(sum(table.values()) - 1.0) < 1e-12
    for lab in [(), (0,), (1, 1), (0, 2, 1), (2, 2, 2)]:
        loss, _ = OC.ctc_loss_one(logits, lab)
        assert abs(loss + np.log(table[lab])) < 1e-10, lab
    loss, grad = OC.ctc_loss_one(logits, (0, 0, 0, 0))          # needs 7 frames, has 5
    assert loss == np.inf and not grad.any()


def test_wide_beam_finds_the_most_probable_labelling():
    rng = np.random.default_rng(4)
    for _ in range(6):
        logits = rng.normal(0, 2.0, size=(5, 4))
        table = _enumerate(logits)
        best = max(table, key=table.get)
        out, score = OC.beam_search_one(logits, beam_width=400, merge_repeated=False)
        assert tuple(out) == best
        assert abs(score - np.log(table[best])) < 1e-9


def test_beam_search_greedy_case_and_merge_repeated():
    C = 4
    path = [0, 0, 3, 0, 1, 1, 3, 2]                              # -> 0 0 1 2
    logits = np.full((len(path), C), -20.0)
    for t, k in enumerate(path):
        logits[t, k] = 20.0
    out, _ = OC.beam_search_one(logits, beam_width=20, merge_repeated=False)
    assert out == [0, 0, 1, 2]
    out, _ = OC.beam_search_one(logits, beam_width=20)            # TF 1.x default collapses the OUTPUT too
    assert out == [0, 1, 2]
    outs, _ = OC.beam_search(logits[None], [3])
    assert outs == [[0]]


def test_narrow_beam_keeps_its_width():
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 1.0, size=(30, 8))
    a, sa = OC.beam_search_one(logits, beam_width=1)
    b, sb = OC.beam_search_one(logits, beam_width=20)
    assert sb >= sa - 1e-12
    assert all(0 <= k < 7 for k in b)


def test_edit_distance():
    assert OC.edit_distance(list(b'kitten'), list(b'sitting'), normalize=False) == 3
    assert OC.edit_distance([1, 2, 3], [1, 2, 3]) == 0
    assert OC.edit_distance([], [1, 2]) == 1.0
    assert OC.edit_distance([1, 2, 3, 4], [1, 3]) == 1.0
    assert OC.edit_distance([], []) == 0.0
    assert OC.edit_distance([1], []) == np.inf


def test_dense_to_sparse_drops_padding():
    lab = np.array([[3, 1, 0, 0], [0, 0, 0, 0]], dtype=np.float32)
    out = OC.dense_to_sparse(lab, [2, 3])
    assert out[0].tolist() == [3, 1] and out[1].tolist() == [0, 0, 0]


def test_oracle_reproduces_the_committed_golden():
    """tests/golden/ctc_golden.npz (made by make_ctc_golden.py): the oracle must not drift."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_golden.npz'))
    loss, grad = OC.ctc_loss(g['logits'], g['labels'], g['labels_lengths'], g['sequence_lengths'])
    np.testing.assert_allclose(loss, g['loss'], rtol=1e-12)
    np.testing.assert_allclose(grad, g['grad'], atol=1e-7)
    dec, scores = OC.beam_search(g['logits'], g['sequence_lengths'], beam_width=20)
    for b, d in enumerate(dec):
        assert d == g['decoded'][b, :g['decoded_lengths'][b]].tolist()
    np.testing.assert_allclose(scores, g['log_prob'], rtol=1e-12)
