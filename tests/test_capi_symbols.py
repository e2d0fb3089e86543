"""CPU-side checks of the drop-in boundary: libavsi_hip.so builds, loads and exports every symbol
that include/avsi_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    import avsi_amd
    return avsi_amd._lib.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "avsi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avsi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built_lib):
    syms = _declared_symbols()
    assert "avsi_frontend_f32" in syms
    for s in syms:
        assert hasattr(built_lib, s), "symbol %s declared in avsi_hip.h but not exported" % s


def test_ctypes_prototypes_cover_header(built_lib):
    import avsi_amd
    assert sorted(avsi_amd._lib.PROTOTYPES) == _declared_symbols()


def test_abi_version_and_status_strings(built_lib):
    assert built_lib.avsi_abi_version() == 12
    assert built_lib.avsi_status_string(0) == b"ok"
    assert b"unsupported" in built_lib.avsi_status_string(-2)
    import ctypes
    built_lib.avsi_blstm_rec_bwd_kernel_name.restype = ctypes.c_char_p
    if not os.environ.get("AVSI_BWD_PP") and not os.environ.get("AVSI_BWD_KH"):
        assert built_lib.avsi_blstm_rec_bwd_kernel_name(8192) == b"blstm_rec_bwd_pp_kernel"      # 4096 < Bp <= 8192
        assert built_lib.avsi_blstm_rec_bwd_kernel_name(4096) == b"blstm_rec_bwd_kh_kernel"


def test_table_size_query_is_host_only(built_lib):
    assert built_lib.avsi_frontend_table_floats(384, 512) > 0
    assert built_lib.avsi_frontend_table_floats(384, 1024) == 0     # unsupported geometry


def test_compute_entry_points_refuse_to_run_without_gpu():
    import torch
    import avsi_amd
    from avsi_amd import audio_processing as ap
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(avsi_amd._lib.AvsiError):
        ap.frontend(torch.zeros(1, 4800), want_spec=True)
