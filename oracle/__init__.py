"""CPU oracle for the speech-inpainting hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference's TensorFlow-1.x op chain
(STFT / log-spectrogram / log-mel front end, stacked BLSTM forward + BPTT,
projection, L1 loss, TF-flavoured Adam, inverse STFT).  It exists to CHECK the
HIP path; it is never the thing shipped or measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``audio-visual-speech-inpainting_amd``) never
does, and fails loudly when its HIP library is missing.

PARITY STATUS
* PINNED by the reference's own output: the STFT -> mask -> |.| / angle -> inverse-STFT -> int16 chain
  (``frontend.get_stft``, ``get_spectrogram``, ``get_sources`` / ``reconstruct_sources`` / ``inverse_stft``;
  SURVEY §8 rows a1, a2, a12 / f1 and ``masking.py:42-46,87-89``).  The reference's docs/files/*/ex*/masked.wav
  were written by its TensorFlow graph from the sibling target.wav; the pairs are committed under
  tests/golden/ref_docs/ and ``tests/test_ref_docs_golden.py`` holds this oracle (float64 and float32) to
  them: every one of 4 x 48,000 samples within one int16 LSB (the truncation floor), exact zeros in the gap.
  One integer per pair is FITTED, not stated by the reference: the onset of the whole-frame gap (exhaustive search in
  tests/golden/make_ref_docs_golden.py; its length follows from the directory name) -- one free parameter against
  48,000 samples.  Also pinned by the reference's own importable code: the config parser and the label helpers.
* UNPINNED (no reference-held vector exists, TensorFlow 1.13-1.15 -- requirements.txt:5-6 -- and ``lws`` are
  not installable in the build container): the mel matrix / log-mel / MFCC / deltas, the LSTM cell and the
  stacked BLSTM, the losses, TF-Adam, the U-Net layers, CTC, and the LWS phase refinement (``oracle/lws.py``,
  restated from the published algorithm).  Their op semantics follow SURVEY.md Appendix A; each function
  cites the reference call site it mirrors, and ``tests/test_oracle_*.py`` triangulate them against independent
  CPU implementations (numpy.fft, torch.stft, scipy.fft.dct, torch.nn.LSTM, torch.autograd, torch ctc_loss).
"""
from . import frontend, blstm  # noqa: F401
