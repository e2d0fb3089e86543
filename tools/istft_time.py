import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import avsi_amd
from avsi_amd import audio_processing as ap
B = 4096
wav = torch.round(torch.randn(B, 48000, device='cuda') * 3000)
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
fe = ap.frontend(wav, want_spec=True, want_stft=True)
pred, stft = fe['spec'], fe['stft']
mean, std = torch.zeros(257, device='cuda'), torch.ones(257, device='cuda')
for _ in range(2): out = ap.enhanced_from_prediction(pred, mean, std, stft, masks, num_samples=48000)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = ap.enhanced_from_prediction(pred, mean, std, stft, masks, num_samples=48000)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
bytes_ = B * (250 * 257 * 4 * 2 + 250 * 257 * 8 + 48000 * 4)
print('B=%d istft+phase: %.2f ms  %.0f GB/s (pred+mask+stft in, wav out)' % (B, dt * 1e3, bytes_ / dt / 1e9))
