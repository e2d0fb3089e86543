import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import lws as L, _lib
lib = ctypes.CDLL(os.path.join(os.path.dirname(avsi_amd.__file__), 'csrc', 'libavsi_hip.so'))
def run(B, U, NW, G):
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    t = torch.arange(48000, device='cuda')[None, :].float()
    f0 = 150 + 100 * torch.rand(B, 1, generator=g, device='cuda')
    wav = sum(2000 / h * torch.sin(2 * 3.14159265 * h * f0 * t / 16000) for h in range(1, 9))
    wav = wav * (0.6 + 0.4 * torch.sin(2 * 3.14159265 * 4 * t / 16000)) + 100 * torch.randn(B, 48000, generator=g, device='cuda')
    masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
    p = L.lws(384, 192, fftsize=512, mode='speech', utterances_per_wave=U, waves_per_group=NW, groups_per_utterance=G)
    p.refine_enhanced(wav, masks, num_samples=48000)
    torch.cuda.synchronize()
    acc = (ctypes.c_ulonglong * 8)()
    lib.avsi_diag_lws_acc(acc)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    p.refine_enhanced(wav, masks, num_samples=48000)
    e.record(); torch.cuda.synchronize()
    lib.avsi_diag_lws_acc(acc)
    n = max(acc[4], 1)
    print("B=%d %dx%dx%d: %.1f ms; frames %d; per frame (us): wait %.2f  phase1 %.2f  phase2 %.2f  tail %.2f ; prologue per sweep %.1f us"
          % (B, U, NW, G, s.elapsed_time(e), acc[4], acc[0] / n / 100, acc[1] / n / 100, acc[2] / n / 100, acc[3] / n / 100,
             acc[5] / 100 / (102 * ((B + max(U, 1) - 1) // max(U, 1)))), flush=True)
for a in ((1, 1, 1, 1), (1, 1, 4, 1), (1, 1, 4, 26), (32, 1, 8, 8), (32, 4, 4, 16), (256, 4, 4, 4), (1024, 4, 4, 1)):
    run(*a)
