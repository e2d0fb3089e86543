"""Time the recurrent forward kernel of one layer at mid batches: batch-stationary (0), cooperative reduction-split
(4 / 8) and column-split (-16 / -32 utterances per group) forms, with the largest difference to the first.
python tools/rec_cs_time.py [T] [Bp ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
sizes = [int(x) for x in sys.argv[2:]] or [256, 512, 1024, 2048]
for Bp in sizes:
    xproj = torch.randn(T, Bp, 2048, device='cuda')
    xother = torch.randn(T, Bp, 2048, device='cuda')
    whp = torch.randn(2 * 262144, device='cuda') * 0.05
    resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
    ref = None
    for save in (False, True):
        line = []
        for sp in (0, 4, 8, -32, -16):
            hout = torch.zeros(T, Bp, 512, device='cuda')
            for _ in range(2):     # other inputs into the same buffers first: stale cache lines would show
                ops.blstm_rec_fwd(xother, whp, hout, resv if save else None, split=sp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.blstm_rec_fwd(xproj, whp, hout, resv if save else None, split=sp)
            e1.record()
            torch.cuda.synchronize()
            ops.blstm_rec_fwd(xother, whp, hout, resv if save else None, split=sp)     # stale cache lines would show
            ops.blstm_rec_fwd(xproj, whp, hout, resv if save else None, split=sp)
            torch.cuda.synchronize()
            ops.coop_check()
            if ref is None:
                ref = hout.clone()
            line.append("%d: %.3f (%.1e)" % (sp, e0.elapsed_time(e1) / 5, float((hout - ref).abs().max())))
        print("Bp=%d %s ms/layer  " % (Bp, 'save' if save else 'infer') + "  ".join(line), flush=True)
