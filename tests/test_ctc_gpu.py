"""avsi_ctc_loss_f32 against the float64 oracle (which tests/test_oracle_ctc.py pins to torch's ctc_loss
and to exhaustive enumeration).  The kernel keeps alpha / beta as float32 logarithms, renormalised
every fourth frame, and normalises the state posteriors per frame: loss within 2e-4 relative,
gradient (probabilities in [-1, 1]) within 2e-5 absolute at 250 frames."""
import numpy as np
import pytest
import torch

from oracle import ctc as OC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    return ops


def _case(seed, B, T, C, Lmax, ragged=True):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2.0, size=(B, T, C)).astype(np.float32)
    lab_len = rng.integers(0, Lmax + 1, size=B)
    lab_len[0] = Lmax
    if B > 1:
        lab_len[1] = 0
    labels = np.zeros((B, Lmax), dtype=np.int32)
    for b in range(B):
        labels[b, :lab_len[b]] = rng.integers(0, C - 1, size=lab_len[b])
    if B > 2 and lab_len[2] >= 3:
        labels[2, :3] = labels[2, 0]                      # repeats: blanks are mandatory between them
    seq_len = np.full(B, T)
    if ragged:
        for b in range(B):
            lab = labels[b, :lab_len[b]]
            need = len(lab) + sum(lab[i] == lab[i - 1] for i in range(1, len(lab)))
            seq_len[b] = rng.integers(min(T, max(need, 1)), T + 1)
    return logits, labels, lab_len.astype(np.int32), seq_len.astype(np.int32)


def _run(ops, logits, labels, lab_len, seq_len, scale=1.0, pitch=None):
    B, T, C = logits.shape
    if pitch:
        buf = torch.zeros(B, T, pitch, device='cuda')
        buf[:, :, :C] = torch.from_numpy(logits).cuda()
        x = buf[:, :, :C]
    else:
        x = torch.from_numpy(logits).cuda()
    loss, grad = ops.ctc_loss(x, torch.from_numpy(labels).cuda(), torch.from_numpy(lab_len).cuda(),
                              torch.from_numpy(seq_len).cuda(), grad_scale=scale, want_grad=True)
    torch.cuda.synchronize()
    return loss.cpu().numpy(), grad.cpu().numpy()


@pytest.mark.parametrize("B,T,C,Lmax", [(5, 12, 6, 5), (7, 60, 34, 20), (3, 33, 5, 40), (2, 40, 130, 100)])
def test_loss_and_gradient_match_oracle(ops, B, T, C, Lmax):
    """State counts 11 / 41 / 81 / 201: one, one, two and four states per lane; C = 130 loops the class lanes."""
    logits, labels, lab_len, seq_len = _case(B * 100 + T, B, T, C, Lmax)
    loss, grad = _run(ops, logits, labels, lab_len, seq_len)
    want, wgrad = OC.ctc_loss(logits, labels, lab_len, seq_len)
    fin = np.isfinite(want)
    assert (np.isinf(loss) == ~fin).all()
    np.testing.assert_allclose(loss[fin], want[fin], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(grad, wgrad, atol=2e-5)
    for b in range(B):
        assert not grad[b, seq_len[b]:].any()


def test_reference_shape_strided_logits_and_scale(ops):
    """250 frames, 33 phones + blank, labels padded to 50 (tfrecord_utils), logits living in a wider
    buffer (the fused projection output), gradient scaled by ctc_loss / B as the model does."""
    B, T, C, Lmax = 8, 250, 34, 50
    logits, labels, lab_len, seq_len = _case(1, B, T, C, Lmax, ragged=False)
    lab_len[:] = np.minimum(lab_len, 30)
    scale = 0.001 / B
    loss, grad = _run(ops, logits, labels, lab_len, seq_len, scale=scale, pitch=48)
    want, wgrad = OC.ctc_loss(logits, labels, lab_len, seq_len)
    np.testing.assert_allclose(loss, want, rtol=2e-4)
    np.testing.assert_allclose(grad, wgrad * scale, atol=2e-5 * scale)


def test_infeasible_labelling_gives_inf_and_zero_gradient(ops):
    logits, labels, lab_len, seq_len = _case(3, 3, 10, 5, 8, ragged=False)
    labels[0, :8] = 1
    lab_len[0] = 8                                         # 8 equal labels need 15 frames, there are 10
    seq_len[2] = 0
    lab_len[2] = 0                                         # empty utterance, empty labelling: p = 1
    loss, grad = _run(ops, logits, labels, lab_len, seq_len)
    assert np.isinf(loss[0]) and not grad[0].any()
    assert loss[2] == 0.0 and not grad[2].any()
    want, wgrad = OC.ctc_loss(logits, labels, lab_len, seq_len)
    np.testing.assert_allclose(loss[1], want[1], rtol=2e-4)
    np.testing.assert_allclose(grad[1], wgrad[1], atol=2e-5)


def test_is_deterministic_and_rejects_unsupported(ops):
    import avsi_amd
    logits, labels, lab_len, seq_len = _case(5, 16, 100, 34, 30)
    a = _run(ops, logits, labels, lab_len, seq_len)
    b = _run(ops, logits, labels, lab_len, seq_len)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(avsi_amd._lib.AvsiError):
        big = np.zeros((1, 8, 5), np.float32)
        _run(ops, big, np.zeros((1, 200), np.int32), np.zeros(1, np.int32), np.full(1, 8, np.int32))


def test_kernel_against_the_committed_golden(ops):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_golden.npz'))
    loss, grad = _run(ops, g['logits'], g['labels'].astype(np.int32), g['labels_lengths'], g['sequence_lengths'])
    np.testing.assert_allclose(loss, g['loss'], rtol=2e-4)
    np.testing.assert_allclose(grad, g['grad'], atol=2e-5)
