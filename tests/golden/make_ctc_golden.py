"""Writes tests/golden/ctc_golden.npz: seeded logits / labels and what oracle/ctc.py computes from them
(loss, gradient, beam-search decoding, edit distance).  The reference holds no CTC fixtures (its arithmetic is
TensorFlow's); this file pins the oracle -- itself pinned to torch's ctc_loss and to exhaustive enumeration by
tests/test_oracle_ctc.py -- so that the GPU tests and later rounds compare against fixed numbers.
Run from the repository root: python tests/golden/make_ctc_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ctc as OC  # noqa: E402

rng = np.random.default_rng(20261003)
B, T, C, Lp = 6, 40, 34, 50                       # 33 phones + blank, labels padded to 50 like tfrecord_utils
logits = rng.normal(0, 2.0, size=(B, T, C)).astype(np.float32)
lab_len = np.array([7, 0, 12, 3, 20, 1], dtype=np.int32)
seq_len = np.array([40, 40, 31, 9, 40, 2], dtype=np.int32)
labels = np.zeros((B, Lp), dtype=np.float32)
for b in range(B):
    labels[b, :lab_len[b]] = rng.integers(0, C - 1, size=lab_len[b])
labels[2, 3:6] = labels[2, 3]                     # repeats
loss, grad = OC.ctc_loss(logits, labels, lab_len, seq_len)
dec, scores = OC.beam_search(logits, seq_len, beam_width=20)
dec_len = np.array([len(d) for d in dec], dtype=np.int32)
dense = np.full((B, max(dec_len.max(), 1)), -1, dtype=np.int32)
for b, d in enumerate(dec):
    dense[b, :len(d)] = d
per = np.array([OC.edit_distance(d, labels[b, :lab_len[b]].astype(int).tolist()) for b, d in enumerate(dec)])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ctc_golden.npz')
np.savez_compressed(out, logits=logits, labels=labels, labels_lengths=lab_len, sequence_lengths=seq_len, loss=loss,
                    grad=grad.astype(np.float32), decoded=dense, decoded_lengths=dec_len, log_prob=scores, per=per)
print(out, os.path.getsize(out), 'bytes; loss', loss, 'per', per)
