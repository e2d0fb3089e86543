import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avsi_amd
from avsi_amd import lws as L
lib = ctypes.CDLL(os.path.join(os.path.dirname(avsi_amd.__file__), 'csrc', 'libavsi_hip.so'))
B, U, NW, G = 1, 1, 4, 26
g = torch.Generator(device='cuda'); g.manual_seed(0)
t = torch.arange(48000, device='cuda')[None, :].float()
f0 = 150 + 100 * torch.rand(B, 1, generator=g, device='cuda')
wav = sum(2000 / h * torch.sin(2 * 3.14159265 * h * f0 * t / 16000) for h in range(1, 9))
wav = wav * (0.6 + 0.4 * torch.sin(2 * 3.14159265 * 4 * t / 16000)) + 100 * torch.randn(B, 48000, generator=g, device='cuda')
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
p = L.lws(384, 192, fftsize=512, mode='speech', utterances_per_wave=U, waves_per_group=NW, groups_per_utterance=G)
for _ in range(2):
    p.refine_enhanced(wav, masks, num_samples=48000)
    torch.cuda.synchronize()
pub = np.zeros((256, 256), np.uint64); obs = np.zeros_like(pub); beg = np.zeros_like(pub)
lib.avsi_diag_lws_stamps(pub.ctypes.data_as(ctypes.c_void_p), obs.ctypes.data_as(ctypes.c_void_p), beg.ctypes.data_as(ctypes.c_void_p))
pub = pub.astype(np.int64); obs = obs.astype(np.int64); spins = beg.astype(np.int64)
t0 = pub[pub > 0].min()
for s in (1, 2, 3, 4, 5, 30, 31, 32, 33, 60, 61, 100, 101):
    # obs[s][r]: sweep s saw that sweep s-1 had finished r+1 rows; pub[s-1][r]: when sweep s-1 published r+1 rows
    rows = [2, 3, 10, 100, 200, 249]
    print("sweep %3d:" % s, " ".join("r%d pub %.0f obs %.0f (spins %d, +%.1f us)" % (r, (pub[s - 1][r] - t0) / 100, (obs[s][r] - t0) / 100, spins[s][r], (obs[s][r] - pub[s - 1][r]) / 100) for r in rows))
lat = []
for s in range(1, 102):
    for r in range(2, 250):
        if spins[s][r] > 0 and pub[s - 1][r] > 0:
            lat.append((obs[s][r] - pub[s - 1][r]) / 100)
lat = np.array(lat)
print("waited handoffs: %d, latency us: median %.2f mean %.2f p90 %.2f max %.2f" % (len(lat), np.median(lat), lat.mean(), np.percentile(lat, 90), lat.max()))
print("start of sweep s (publish of its row 1), us:", [(s, round((pub[s][0] - t0) / 100)) for s in range(0, 102, 10)])
print("end of sweep s (publish of row 250), us:", [(s, round((pub[s][249] - t0) / 100)) for s in range(0, 102, 10)])
