import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """The CPU suite needs libavsi_hip.so for its host routines (CRC-32C) and for the symbol / prototype checks.
    It is a build product (git-ignored): build it in-tree when a fresh checkout does not have it yet."""
    lib = os.path.join(ROOT, "audio-visual-speech-inpainting_amd", "csrc", "libavsi_hip.so")
    if os.path.isfile(lib):
        return
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        return                      # the tests that need it fail loudly on their own
    subprocess.run(["make", "-C", os.path.dirname(lib), "-j4"], check=False, stdout=subprocess.DEVNULL,
                   stderr=subprocess.STDOUT, timeout=900)
