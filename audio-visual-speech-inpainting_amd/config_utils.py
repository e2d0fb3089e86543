"""Configuration files of the drivers: ``key = value`` text, same grammar, defaults, warnings and
errors as the reference ``av_speech_inpainting/config_utils.py`` (load_configfile :7-52,
check_trainconfiguration :55-129).  Behaviour is pinned by ``tests/golden/config_*.json``, produced
by running the reference parser itself (tests/golden/make_config_golden.py).

Grammar (reference :22-48): blank lines and lines starting with ``#`` are skipped; a line is
``<word> = <value>``; a value containing ``[`` is a Python literal list; any other value must not
contain a space, is a Python literal when it has a digit and no ``/`` (numbers), else a raw string
(paths, names).
"""
import ast
import os
import re
import sys

_LINE = re.compile(r'(\w+)\s*=\s*(.*)')


def _parse_value(text, lineno):
    if '[' in text:
        try:
            return ast.literal_eval(text)
        except Exception:
            raise ValueError("Wrong syntax in the configuration file at line {:d} "
                             "(may be a missing square parenthesis?)".format(lineno))
    if ' ' in text:
        raise ValueError("Wrong syntax in the configuration file at line {:d} "
                         "(may be a space in the param value?)".format(lineno))
    if re.search('[0-9]', text) and '/' not in text:
        try:
            return ast.literal_eval(text)
        except Exception:
            raise ValueError("Wrong syntax in the configuration file at line {:d} "
                             "(may be due to mixed letters and integers?)".format(lineno))
    return text


def load_configfile(cfile):
    """Read a configuration file into a dict (reference config_utils.py:7-52)."""
    if not os.path.isfile(cfile):
        raise ValueError("Cannot find configuration file ", cfile)
    config = {}
    with open(cfile, 'r') as fh:
        for lineno, raw in enumerate(fh, 1):
            line = raw.rstrip()
            if not line or line[0] == '#':
                continue
            hit = _LINE.search(line)
            if hit is None:
                raise ValueError("Wrong syntax in the configuration file at line ", lineno)
            config[hit.group(1)] = _parse_value(hit.group(2), lineno)
    return config


def _warn(msg):
    print("WARNING: " + msg, file=sys.stderr)


# (key, default, warning text) in the order the reference applies them (config_utils.py:96-127).
# The texts are the reference's own, including the places where they disagree with the value set.
_TRAIN_DEFAULTS = (
    ('batch_size', 1, "Batch size not defined in config file. Set to 1 by default"),
    ('dropout_rate', 0.0, "Dropout rate not defined in config file. Set to 1 by default"),
    ('starter_learning_rate', 0.06, "Starter learning rate not defined in config file. Set to 0.06 by default"),
    ('learning_rate', 0.06, "Learning rate not defined in config file. Set to 0.06 by default"),
    ('lr_updating_steps', 10000,
     "Updating steps of learning rate decay not defined in config file. Set to 10000 by default"),
    ('lr_decay', 0.5, "Learning rate decay not defined in config file. Set to 0.5 by default"),
    ('l2', 0.0, "L2 regularization coefficient not defined in config file. Set to 0 by default"),
    ('optimizer_type', 'adam', "Optimizer type not defined in config file. Set to 'adam' by default"),
)


def check_trainconfiguration(config):
    """Validate a training configuration and fill in defaults, in place
    (reference config_utils.py:55-129; quirks kept)."""
    if 'root_folder' not in config:
        raise ValueError("Root folder not defined")
    if 'exp_folder' not in config:
        raise ValueError("Experiment folder (exp_folder) not defined")
    config.setdefault('model_ckp', "")
    config.setdefault('model_ckp_vnet', "")
    if 'device' not in config:
        _warn("using cpu as device has not been defined in the config file")
        config['device'] = "/cpu:0"
    if 'model' not in config:
        raise ValueError("Model type (model) not defined in config file")
    if 'net_dim' not in config:
        raise ValueError("Enhancement net dimensions (enh_net_dim) not defined in config file")
    if 'integration_layer' not in config:
        config['integration_layer'] = 0
        _warn("Embedding integration layer not defined in config file. Set to 0 by default")
    if 'audio_feat_dim' not in config:
        config['audio_feat_dim'] = 257
        _warn("No. of audio input features of inpainting model not defined in config file. Set to 257 by default")
    if 'video_feat_dim' not in config:
        config['video_feat_dim'] = 136
        _warn("No. of video input features of inpainting model not defined in config file. Set to 136 by default")
    audio_len_was_missing = 'audio_len' not in config
    if audio_len_was_missing:
        config['audio_len'] = 16384
        _warn("Length of input wavs of inpainting model not defined in config file. "
              "Set to 0 by default (variable-length)")
    if 'audio_feat_mean' not in config:
        raise ValueError("File with mean of features (audio_feat_mean) not defined in config file")
    if 'audio_feat_std' not in config:
        # the reference passes file= to ValueError here, which raises TypeError instead (config_utils.py:86)
        raise TypeError("ValueError() takes no keyword arguments")
    if 'num_asr_labels' not in config:
        config['num_asr_labels'] = 33
        _warn("No. of speech recognition labels not defined in config file. Set to 33 by default")
    config['num_asr_labels'] += 1          # the "blank" label, added on every call (reference :91)
    # reference :92-94 gates the ctc_loss default on audio_len, which was just filled in: never taken
    for key, default, text in _TRAIN_DEFAULTS:
        if key not in config:
            _warn(text)
            config[key] = default
    if config['optimizer_type'] == 'momentum_dlr' and 'momentum' not in config:
        raise ValueError("momentum missing from config file")
    if 'max_n_epochs' not in config:
        _warn("max_n_epochs not defined. Set to 100 by default")
        config['max_n_epochs'] = 30
    if 'n_earlystop_epochs' not in config:
        _warn("n_earlystop_epochs not defined. Set to 3 by default")
        config['n_earlystop_epochs'] = 30
    return config
