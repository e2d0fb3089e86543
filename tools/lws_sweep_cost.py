"""Per-sweep cost of the LWS kernel on ONE wave (U = NW = G = 1, no pipeline): batch_iterations = 20, 40, .. 100 on the
synthetic signal of tools/lws_time.py; differences give the time of sweeps 0-19, 20-39, ... (their thresholds fall from
100 x to 0.005 x the mean magnitude, so later sweeps touch more bins).  python tools/lws_sweep_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import lws as L
g = torch.Generator(device='cuda'); g.manual_seed(0)
B = 1
t = torch.arange(48000, device='cuda')[None, :].float()
f0 = 150 + 100 * torch.rand(B, 1, generator=g, device='cuda')
wav = sum(2000 / h * torch.sin(2 * 3.14159265 * h * f0 * t / 16000) for h in range(1, 9))
wav = wav * (0.6 + 0.4 * torch.sin(2 * 3.14159265 * 4 * t / 16000)) + 100 * torch.randn(B, 48000, generator=g, device='cuda')
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
prev = 0.0
for n in (0, 20, 40, 60, 80, 100):
    p = L.lws(384, 192, fftsize=512, nofuture_iterations=0, online_iterations=0, batch_iterations=n,
              utterances_per_wave=1, waves_per_group=1, groups_per_utterance=1)
    p.refine_enhanced(wav, masks, num_samples=48000)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    p.refine_enhanced(wav, masks, num_samples=48000)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    if n:
        print("sweeps %3d..%3d: %.1f ms = %.1f us per frame" % (n - 20, n - 1, ms - prev, (ms - prev) / 20 / 250 * 1e3), flush=True)
    else:
        print("no sweeps: %.2f ms" % ms, flush=True)
    prev = ms
