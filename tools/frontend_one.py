"""Run the fused front end a few times on a synthetic batch (for rocprofv3 --pmc).  python tools/frontend_one.py [B]"""
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
for _ in range(3):
    ap.frontend(wav, mean=mean, std=std, masks=mask, want_spec=True, want_feat=True, time_major=True, feat_cols=272, _feat_out=x0)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    ap.frontend(wav, mean=mean, std=std, masks=mask, want_spec=True, want_feat=True, time_major=True, feat_cols=272, _feat_out=x0)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print("B=%d %.3f ms  %.0f GB/s algorithmic" % (B, ms, 706000.0 * B / ms / 1e6))
