"""Time the BPTT recurrent kernel of one layer at small batches for every cooperative split.
python tools/rec_coop_bwd_time.py [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for Bp in [int(x) for x in sys.argv[2:]] or (32, 64, 128, 256, 512):
    dh = torch.randn(T, Bp, 512, device='cuda')
    resv = torch.rand(T, Bp, 2, 5, 256, device='cuda') * 0.9 + 0.05
    whbt = torch.randn(2 * 262144, device='cuda') * 0.05
    dz = torch.empty(T, Bp, 2048, device='cuda')
    line = []
    for sp in (0, 4, 8, 16, 32):
        for _ in range(2):
            ops.blstm_rec_bwd(dh, resv, whbt, dz, split=sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.blstm_rec_bwd(dh, resv, whbt, dz, split=sp)
        e1.record()
        torch.cuda.synchronize()
        line.append("%d: %.3f" % (sp, e0.elapsed_time(e1) / 5))
    ops.coop_check()
    print("Bp=%d ms/layer  " % Bp + "  ".join(line), flush=True)
