"""Golden description of the reference CLI (sub-commands, flags, defaults, types, choices), extracted from
the TEXT of /root/reference/av_speech_inpainting/speech_inpainting_main.py with `ast` -- the module itself
cannot be imported here (it pulls in TensorFlow / pydub at import time), and nothing of it is executed.
Run from the repo root:  python tests/golden/make_cli_golden.py"""
import ast
import json
import os

SRC = '/root/reference/av_speech_inpainting/speech_inpainting_main.py'
tree = ast.parse(open(SRC).read())


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return ast.unparse(node)          # e.g. a type name: int, float


parsers = {}          # variable name -> sub-command
commands = {}
for node in ast.walk(tree):
    if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and getattr(node.value.func, 'attr', '') == 'add_parser':
        name = lit(node.value.args[0])
        parsers[node.targets[0].id] = name
        commands[name] = []
for node in ast.walk(tree):
    if isinstance(node, ast.Call) and getattr(node.func, 'attr', '') == 'add_argument' and isinstance(node.func.value, ast.Name):
        owner = parsers.get(node.func.value.id)
        if owner is None:
            continue
        flags = [lit(a) for a in node.args]
        kw = {k.arg: lit(k.value) for k in node.keywords if k.arg in ('required', 'default', 'type', 'nargs', 'choices', 'action', 'const')}
        commands[owner].append({'flags': flags, **kw})
out = {'source': 'av_speech_inpainting/speech_inpainting_main.py', 'commands': commands}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cli_golden.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
print({k: len(v) for k, v in commands.items()})
