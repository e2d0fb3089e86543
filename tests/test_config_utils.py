"""config_utils against golden fixtures produced by the reference's own parser
(tests/golden/make_config_golden.py).  No GPU."""
import contextlib
import io
import json
import os

import pytest

import avsi_amd  # noqa: F401
from avsi_amd import config_utils as cu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "config_golden.json")))
REF_CFG = "/root/reference/scripts/config"


def _text(name, case):
    if "text" in case:
        return case["text"]
    path = os.path.join(REF_CFG, name.split(":", 1)[1])
    if not os.path.isfile(path):
        pytest.skip("reference tree not mounted")
    return open(path).read()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_matches_reference_parser(name, tmp_path):
    case = GOLD[name]
    path = tmp_path / "c.config"
    path.write_text(_text(name, case))
    if "load_error" in case:
        with pytest.raises(Exception) as ei:
            cu.load_configfile(str(path))
        assert type(ei.value).__name__ == case["load_error"][0]
        assert [str(a).replace(str(path), "<file>") for a in ei.value.args] == case["load_error"][1]
        return
    cfg = cu.load_configfile(str(path))
    assert cfg == case["load"]
    assert {k: type(v).__name__ for k, v in cfg.items()} == {k: type(v).__name__ for k, v in case["load"].items()}
    assert list(cfg) == list(case["load"])
    err = io.StringIO()
    if "check_error" in case:
        with pytest.raises(Exception) as ei, contextlib.redirect_stderr(err):
            cu.check_trainconfiguration(dict(cfg))
        assert type(ei.value).__name__ == case["check_error"][0]
        assert [str(a) for a in ei.value.args] == case["check_error"][1]
    else:
        with contextlib.redirect_stderr(err):
            got = cu.check_trainconfiguration(dict(cfg))
        assert got == case["check"]
        assert list(got) == list(case["check"])          # insertion order of the defaults too
    assert err.getvalue() == case["stderr"]


def test_missing_file_raises_value_error():
    with pytest.raises(ValueError):
        cu.load_configfile("/nonexistent/file.config")


def test_num_asr_labels_incremented_on_every_call():
    """Reference quirk (config_utils.py:91): the blank label is added each time."""
    cfg = dict(root_folder="/r", exp_folder="/e", model="m", net_dim=[1], audio_feat_mean="/m", audio_feat_std="/s")
    with contextlib.redirect_stderr(io.StringIO()):
        cu.check_trainconfiguration(cfg)
        assert cfg["num_asr_labels"] == 34
        cu.check_trainconfiguration(cfg)
        assert cfg["num_asr_labels"] == 35
