"""avsi_wav_write_batch_int16_host (host code of the C ABI, no GPU): the files of the inference driver are byte-identical to
what the reference's `wavfile.write(path, 16000, wav[:n].astype(np.int16))` writes (inference.py:159-162), including numpy's
float -> int16 conversion of out-of-range values, and missing directories are created."""
import ctypes
import os
import warnings

import numpy as np
from scipy.io import wavfile


def test_files_equal_scipy_wavfile_write(tmp_path):
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    x = rng.normal(0, 9000, size=(4, 5000)).astype(np.float32)
    x[0, :8] = [40000.5, -40000.5, np.nan, 3e10, -3e10, 0.999, -0.999, 32767.9]
    n = np.array([5000, 4800, 0, 1], dtype=np.int32)
    paths = [str(tmp_path / "out" / ("s%d" % i) / "video" / "enhanced" / "p.wav").encode() for i in range(4)]
    arr = (ctypes.c_char_p * 4)(*paths)
    assert L.avsi_wav_write_batch_int16_host(arr, x.ctypes.data, x.strides[0] // 4, n.ctypes.data, 4, 16000, 1) == 0
    for i in range(4):
        ref = str(tmp_path / ("ref%d.wav" % i))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                      # numpy warns about the out-of-range casts it performs
            wavfile.write(ref, 16000, x[i, :n[i]].astype(np.int16))
        assert open(ref, "rb").read() == open(paths[i].decode(), "rb").read(), i
    # an unwritable path is an error, not a silent skip
    bad = (ctypes.c_char_p * 1)(str(tmp_path / "ref0.wav" / "x" / "p.wav").encode())
    assert L.avsi_wav_write_batch_int16_host(bad, x.ctypes.data, x.strides[0] // 4, n.ctypes.data, 1, 16000, 1) != 0
    assert L.avsi_wav_write_batch_int16_host(bad, x.ctypes.data, x.strides[0] // 4, n.ctypes.data, 1, 16000, 0) != 0
