"""Golden vectors for the phoneme-label helpers, produced by the REFERENCE's own
transcription2phonemes.load_dictionary / get_labels (pure Python, importable in the build container).
Run from the repo root:  python tests/golden/make_labels_golden.py   (needs /root/reference)."""
import json
import os
import sys
import tempfile
import warnings

warnings.simplefilter('ignore')
sys.path.insert(0, '/root/reference/av_speech_inpainting')
import transcription2phonemes as t2p  # noqa: E402

DICT_TEXT = "B IH N\nB L UW\nAE T\nSP\nF AY V\nAH G EH N\nS P IY CH\n\nW IH TH  Z IY R OW\n"
TRANSCRIPTIONS = ["B,IH,N,SP,B,L,UW,AE,T,F,AY,V,AH,G,EH,N", "SP,S,P,IY,CH,SP", "W,IH,TH,,Z,IY,R,OW,", ""]

with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, 'dict.txt')
    open(path, 'w').write(DICT_TEXT)
    dictionary = t2p.load_dictionary(path)
out = {'dict_text': DICT_TEXT, 'dictionary': dictionary,
       'cases': [{'transcription': t, 'labels': [int(x) for x in t2p.get_labels(t, dictionary)]} for t in TRANSCRIPTIONS]}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'labels_golden.json'), 'w') as f:
    json.dump(out, f, indent=1)
print('wrote', len(out['cases']), 'cases; dictionary of', len(dictionary))
