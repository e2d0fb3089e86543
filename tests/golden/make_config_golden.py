#!/usr/bin/env python3
"""Generate tests/golden/config_golden.json by running the REFERENCE parser
(/root/reference/av_speech_inpainting/config_utils.py -- the only module of the reference that
executes in the build container) on its own config files and on crafted edge cases.

Only inputs and expected outputs are stored: the crafted case texts are written here; the four
reference config files are read where they lie and only their parsed RESULT is stored (the test
re-reads them from /root/reference when it is mounted).  Run in the build container:
    python tests/golden/make_config_golden.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

REF = "/root/reference/av_speech_inpainting"
REF_CFG = "/root/reference/scripts/config"
HERE = os.path.dirname(os.path.abspath(__file__))

CRAFTED = {
    "minimal": "root_folder = /data/x\nexp_folder = /exp/y\nmodel = a-blstm\nnet_dim = [250, 250, 250]\n"
               "audio_feat_mean = /m.npy\naudio_feat_std = /s.npy\n",
    "numbers": "a = 1\nb = 1.5\nc = 1e-3\nd = -7\ne = abc\nf = a1\n",
    "paths_and_lists": "p = /a/b1/c\nq = [1, 2, 3]\nr = ['x', 'y']\ns=[ ]\nt   =   5\n",
    "comments_blank": "# hi\n\n### more\nx = 3\n   \ny = z\n",
    "bad_space": "x = a b\n",
    "bad_list": "x = [1, 2\n",
    "bad_mixed": "x = 3abc\n",
    "bad_syntax": "just words\n",
    "missing_root": "exp_folder = /e\nmodel = m\nnet_dim = [1]\naudio_feat_mean = /m\naudio_feat_std = /s\n",
    "missing_std": "root_folder = /r\nexp_folder = /e\nmodel = m\nnet_dim = [1]\naudio_feat_mean = /m\n",
    "momentum_dlr": "root_folder = /r\nexp_folder = /e\nmodel = m\nnet_dim = [1]\naudio_feat_mean = /m\n"
                    "audio_feat_std = /s\noptimizer_type = momentum_dlr\n",
    "all_given": "root_folder = /r\nexp_folder = /e\nmodel = av-blstm\nnet_dim = [250, 250, 250]\n"
                 "audio_feat_mean = /m\naudio_feat_std = /s\ndevice = /gpu:0\nintegration_layer = 1\n"
                 "audio_feat_dim = 257\nvideo_feat_dim = 136\naudio_len = 48000\nnum_asr_labels = 40\nbatch_size = 8\n"
                 "dropout_rate = 0.0\nstarter_learning_rate = 0.001\nlearning_rate = 0.001\nlr_updating_steps = 5000\n"
                 "lr_decay = 1.0\nl2 = 0.0\noptimizer_type = adam\nmax_n_epochs = 50\nn_earlystop_epochs = 5\n",
}


def run(mod, text):
    out = {"text": text}
    with tempfile.NamedTemporaryFile("w", suffix=".config", delete=False) as fh:
        fh.write(text)
        path = fh.name
    try:
        try:
            cfg = mod.load_configfile(path)
            out["load"] = cfg
        except Exception as e:            # noqa: BLE001
            out["load_error"] = [type(e).__name__, [str(a).replace(path, "<file>") for a in e.args]]
            return out
        err = io.StringIO()
        try:
            with contextlib.redirect_stderr(err):
                checked = mod.check_trainconfiguration(dict(cfg))
            out["check"] = checked
        except Exception as e:            # noqa: BLE001
            out["check_error"] = [type(e).__name__, [str(a) for a in e.args]]
        out["stderr"] = err.getvalue()
    finally:
        os.unlink(path)
    return out


def main():
    sys.path.insert(0, REF)
    import config_utils as ref
    cases = {}
    for name in sorted(os.listdir(REF_CFG)):
        if not name.endswith(".config"):
            continue
        case = run(ref, open(os.path.join(REF_CFG, name)).read())
        del case["text"]                  # the file itself stays in the reference tree
        cases["reference:" + name] = case
    for name, text in CRAFTED.items():
        cases["crafted:" + name] = run(ref, text)
    with open(os.path.join(HERE, "config_golden.json"), "w") as fh:
        json.dump({k: cases[k] for k in sorted(cases)}, fh, indent=1)   # inner dicts keep insertion order
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
