"""CPU oracle for the speech-inpainting hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference's TensorFlow-1.x op chain
(STFT / log-spectrogram / log-mel front end, stacked BLSTM forward + BPTT,
projection, L1 loss, TF-flavoured Adam, inverse STFT).  It exists to CHECK the
HIP path; it is never the thing shipped or measured.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``audio-visual-speech-inpainting_amd``) never
does, and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
this path, and its arithmetic lives in TensorFlow 1.13-1.15 (requirements.txt:5-6),
which is not installable in the build container.  The op semantics restated here
follow SURVEY.md Appendix A; each function cites the reference call site it
mirrors.  The restatement is triangulated in ``tests/test_oracle_*.py`` against
independent implementations available on CPU (numpy.fft, torch.stft,
torch.nn.LSTM, torch.autograd).
"""
from . import frontend, blstm  # noqa: F401
