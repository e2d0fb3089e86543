"""Time enhanced_sources at B utterances: python tools/istft_time.py [B]   (mode 3: phase from the waveform; mode 2: from a stored STFT)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import audio_processing as ap
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wav = torch.round(torch.randn(B, 48000, device='cuda') * 3000)
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
fe = ap.frontend(wav, want_spec=True, want_stft=True)
pred, stft = fe['spec'], fe['stft']
mean, std = torch.zeros(257, device='cuda'), torch.ones(257, device='cuda')


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


dt3 = timeit(lambda: ap.enhanced_from_prediction_wav(pred, mean, std, wav, masks, num_samples=48000))
dt2 = timeit(lambda: ap.enhanced_from_prediction(pred, mean, std, stft, masks, num_samples=48000))
b3 = B * (250 * 257 * 4 * 2 + 48000 * 4 * 2)
b2 = B * (250 * 257 * 4 * 2 + 250 * 257 * 8 + 48000 * 4)
print('B=%d from the waveform (mode 3): %.3f ms  %.0f GB/s of %.2f MB/utt' % (B, dt3 * 1e3, b3 / dt3 / 1e9, b3 / B / 1e6))
print('B=%d from a stored STFT (mode 2): %.3f ms  %.0f GB/s of %.2f MB/utt' % (B, dt2 * 1e3, b2 / dt2 / 1e9, b2 / B / 1e6))
