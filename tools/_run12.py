import sys, os
sys.path.insert(0, os.getcwd())
import torch, avsi_amd
from avsi_amd import ops
T = 250
whp = torch.randn(2 * 262144, device='cuda') * 0.05
for Bp in (160, 192, 224, 256):
    xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.3
    hout = torch.empty(T, Bp, 512, device='cuda')
    resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
    for save in (False, True):
        out = []
        for sp in (16, -16, -32):
            for _ in range(3): ops.blstm_rec_fwd(xproj, whp, hout, resv if save else None, split=sp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.blstm_rec_fwd(xproj, whp, hout, resv if save else None, split=sp)
            e1.record(); torch.cuda.synchronize()
            out.append("split %3d: %.3f" % (sp, e0.elapsed_time(e1) / 10))
        print("Bp=%d reserve=%d  %s" % (Bp, save, "  ".join(out)), flush=True)
    ops.coop_check()
