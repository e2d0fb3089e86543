"""The command line keeps the reference's surface: every sub-command, flag, default, type, choice and
required-ness of av_speech_inpainting/speech_inpainting_main.py (golden extracted from its source text by
tests/golden/make_cli_golden.py) exists with the same meaning in this build's parser."""
import argparse
import json
import os

import pytest

import avsi_amd  # noqa: F401
from avsi_amd import speech_inpainting_main as cli

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cli_golden.json')))
TYPES = {'int': int, 'float': float, 'str': str}


def _subparsers():
    parser = cli.build_parser()
    for a in parser._actions:
        if isinstance(a, argparse._SubParsersAction):
            return a.choices
    raise AssertionError('no sub-commands')


def test_every_reference_subcommand_exists():
    assert set(GOLD['commands']) <= set(_subparsers())


@pytest.mark.parametrize('command', sorted(GOLD['commands']))
def test_flags_defaults_types_match(command):
    sub = _subparsers()[command]
    mine = {}
    for a in sub._actions:
        for f in a.option_strings:
            mine[f] = a
    for spec in GOLD['commands'][command]:
        acts = {id(mine[f]) for f in spec['flags'] if f in mine}
        assert all(f in mine for f in spec['flags']), (command, spec['flags'])
        assert len(acts) == 1, (command, spec['flags'])          # short and long form are one option
        a = mine[spec['flags'][0]]
        assert bool(a.required) == bool(spec.get('required', False)), (command, spec['flags'])
        if 'default' in spec:
            assert a.default == spec['default'], (command, spec['flags'], a.default)
        if 'type' in spec:
            assert a.type is TYPES[spec['type']], (command, spec['flags'])
        if 'choices' in spec:
            assert list(a.choices) == list(spec['choices']), (command, spec['flags'])
        if 'nargs' in spec:
            assert a.nargs == spec['nargs'], (command, spec['flags'])
        if spec.get('action') == 'store_const':
            assert a.const == spec.get('const') and a.nargs == 0, (command, spec['flags'])


def test_out_of_scope_subcommands_exit_with_message(capsys):
    with pytest.raises(SystemExit) as e:
        cli.main(['training_asr', '--config', 'x'])
    assert e.value.code == 1
    assert 'not part of the MI355X hot-path package' in capsys.readouterr().out
