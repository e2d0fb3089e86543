"""Time the fused front end with different output sets (which side of the kernel costs what).  python tools/frontend_parts.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import audio_processing as ap
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wav = torch.round(torch.randn(B, 48000, device='cuda') * 3000)
mask = torch.ones(B, 250, 257, device='cuda')
mean = torch.zeros(257, device='cuda'); std = torch.ones(257, device='cuda')
x0 = torch.zeros(250, B, 272, device='cuda')
xb = torch.zeros(B, 250, 272, device='cuda')


def run(name, **kw):
    for _ in range(2):
        ap.frontend(wav, mean=mean, std=std, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ap.frontend(wav, mean=mean, std=std, **kw)
    e.record(); torch.cuda.synchronize()
    print("%-44s %.3f ms" % (name, s.elapsed_time(e) / 5), flush=True)


run("spec only", want_spec=True)
run("feat time-major + mask", masks=mask, want_feat=True, time_major=True, feat_cols=272, _feat_out=x0)
run("feat batch-major + mask", masks=mask, want_feat=True, time_major=False, feat_cols=272, _feat_out=xb)
run("feat batch-major, no mask", want_feat=True, time_major=False, feat_cols=272, _feat_out=xb)
run("spec + feat time-major + mask (model)", masks=mask, want_spec=True, want_feat=True, time_major=True, feat_cols=272, _feat_out=x0)
run("spec + feat batch-major + mask", masks=mask, want_spec=True, want_feat=True, time_major=False, feat_cols=272, _feat_out=xb)
run("logmel only", want_logmel=True)
run("stft only", want_stft=True)
