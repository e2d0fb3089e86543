"""MI355X-native hot path of dr-pato/audio-visual-speech-inpainting.

Host layer (Python on PyTorch-ROCm, used for device memory / streams / torch.distributed only)
over ``csrc/libavsi_hip.so`` -- hand-written gfx950 HIP kernels behind the C ABI declared in
``include/avsi_hip.h``.  Module names mirror the reference package
(``audio_processing``, ``models``, ``inference``, ``training``, ``config_utils`` ...).

There is NO CPU fallback: any compute entry point raises if the HIP library is missing or no
GPU is visible.  Import as ``import avsi_amd`` (see ``avsi_amd.py`` at the repository root).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
