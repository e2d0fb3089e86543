"""Time the LWS phase refinement (inference.py:141-154) on a synthetic batch.  python tools/lws_time.py [B] [U]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import lws as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
U = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 0
G = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = torch.Generator(device='cuda'); g.manual_seed(0)
t = torch.arange(48000, device='cuda')[None, :].float()
f0 = 150 + 100 * torch.rand(B, 1, generator=g, device='cuda')
wav = sum(2000 / h * torch.sin(2 * 3.14159265 * h * f0 * t / 16000) for h in range(1, 9))
wav = wav * (0.6 + 0.4 * torch.sin(2 * 3.14159265 * 4 * t / 16000)) + 100 * torch.randn(B, 48000, generator=g, device='cuda')
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
p = L.lws(384, 192, fftsize=512, mode='speech', utterances_per_wave=U, waves_per_group=NW, groups_per_utterance=G)      # AVSI_LWS_KERNEL=raster|skew
print("kernel:", p.kernel, flush=True)
out = p.refine_enhanced(wav, masks, num_samples=48000)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(2):
    out = p.refine_enhanced(wav, masks, num_samples=48000)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 2
print("B=%d U=%d NW=%d G=%d: %.1f ms per batch, %.0f utterances/s" % (B, U, NW, G, ms, B / ms * 1e3), flush=True)
