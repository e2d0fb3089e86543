"""Skewed-frame LWS kernel against the frame-by-frame one on a small input.  python tools/lws_skew_probe.py [n] [B] [NW] [G] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import lws as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 4
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator(device='cuda'); g.manual_seed(0)
t = torch.arange(n, device='cuda')[None, :].float()
f0 = 150 + 100 * torch.rand(B, 1, generator=g, device='cuda')
wav = sum(2000 / h * torch.sin(2 * 3.14159265 * h * f0 * t / 16000) for h in range(1, 9)) + 100 * torch.randn(B, n, generator=g, device='cuda')
kw = dict(nofuture_iterations=1, online_iterations=1, batch_iterations=iters, batch_alpha=100, batch_beta=0.9)
pr = L.lws(384, 192, fftsize=512, kernel='raster', **kw)
S = pr.stft(wav)
S0 = S.clone()
S0[:, 8:14] = S0[:, 8:14].abs().to(torch.complex64)
print("frames", S0.shape, flush=True)
ref = pr.run_lws(S0)
torch.cuda.synchronize()
print("raster done", flush=True)
ps = L.lws(384, 192, fftsize=512, kernel='skew', waves_per_group=NW, groups_per_utterance=G, **kw)
out = ps.run_lws(S0)
torch.cuda.synchronize()
err = (out - ref).abs()
print("skew done: rel err %.3e, max %.3e of %.3e; changed vs input %.3e" % (
    float((err ** 2).sum().sqrt() / (ref.abs() ** 2).sum().sqrt()), float(err.max()), float(ref.abs().max()),
    float((out - S0).abs().max())), flush=True)
