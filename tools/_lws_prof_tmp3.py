import os, sys, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import avsi_amd
from avsi_amd import lws as L
lib = ctypes.CDLL(os.path.join(os.path.dirname(avsi_amd.__file__), 'csrc', 'libavsi_hip.so'))
def run(B, U, NW, G):
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    t = torch.arange(48000, device='cuda')[None, :].float()
    wav = 2000 * torch.sin(2 * 3.14159265 * 200 * t / 16000).repeat(B, 1) + 100 * torch.randn(B, 48000, generator=g, device='cuda')
    masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
    p = L.lws(384, 192, fftsize=512, mode='speech', utterances_per_wave=U, waves_per_group=NW, groups_per_utterance=G)
    p.refine_enhanced(wav, masks, num_samples=48000)
    torch.cuda.synchronize()
    hw = np.zeros((4096, 2), np.uint32)
    lib.avsi_diag_lws_hw(hw.ctypes.data_as(ctypes.c_void_p))
    nwg = ((B + U - 1) // U) * G
    cus = collections.Counter()
    simds = collections.Counter()
    for w in range(nwg):
        for v in range(NW):
            xcc, h = int(hw[w * NW + v][0]) & 0xF, int(hw[w * NW + v][1])
            cu = (xcc, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF)
            if v == 0: cus[cu] += 1
            simds[cu + ((h >> 4) & 3,)] += 1
    print("B=%d %dx%dx%d: %d workgroups on %d distinct CUs (max %d per CU); waves per SIMD: max %d, SIMDs used %d of %d waves"
          % (B, U, NW, G, nwg, len(cus), max(cus.values()), max(simds.values()), len(simds), nwg * NW))
    print("   first WGs (xcc, se, sh, cu):", [ (int(hw[w * NW][0]) & 0xF, (int(hw[w * NW][1]) >> 13) & 7, (int(hw[w*NW][1]) >> 12) & 1, (int(hw[w * NW][1]) >> 8) & 0xF) for w in range(min(nwg, 12))])
for a in ((1, 1, 4, 26), (8, 1, 4, 26), (32, 1, 8, 8), (100, 4, 4, 10)):
    run(*a)
