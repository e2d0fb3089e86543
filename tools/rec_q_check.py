"""The quarter-block forward recurrence (rows_per_wg = 66) against the ping-pong kernel (64): results, then launch times.
python tools/rec_q_check.py [Bp] [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
g = torch.Generator(device='cuda'); g.manual_seed(1)
for bp, t in ((192, 7), (100 * 64, 3)):
    xproj = torch.randn(t, bp, 2048, device='cuda', generator=g) * 0.5
    whp = torch.randn(2 * 262144, device='cuda', generator=g) * 0.05
    outs = {}
    for rpw in (64, 66):
        hout = torch.full((t, bp, 512), 7.0, device='cuda')
        resv = torch.full((t, bp, 2, 5, 256), 7.0, device='cuda')
        ops.blstm_rec_fwd(xproj, whp, hout, resv, rows_per_wg=rpw, split=0)
        h2 = torch.full((t, bp, 512), 7.0, device='cuda')
        ops.blstm_rec_fwd(xproj, whp, h2, None, rows_per_wg=rpw, split=0)
        torch.cuda.synchronize()
        outs[rpw] = (hout, resv, h2)
    for name, i in (('h (with reserve)', 0), ('reserve', 1), ('h (no reserve)', 2)):
        d = (outs[64][i] - outs[66][i]).abs().max().item()
        print("Bp=%d T=%d %s: max |pp - q| = %.3g (max |pp| %.3g)" % (bp, t, name, d, outs[64][i].abs().max().item()), flush=True)
xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.3
whp = torch.randn(2 * 262144, device='cuda') * 0.05
hout = torch.empty(T, Bp, 512, device='cuda')
resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
for rpw in (64, 66, 64, 66):
    for name, r in (('no reserve', None), ('with reserve', resv)):
        for _ in range(2):
            ops.blstm_rec_fwd(xproj, whp, hout, r, rows_per_wg=rpw, split=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.blstm_rec_fwd(xproj, whp, hout, r, rows_per_wg=rpw, split=0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        print("Bp=%d T=%d rows_per_wg=%d %s: %.2f ms  %.1f TFLOP/s (padded)" % (Bp, T, rpw, name, ms, 2.0 * 256 * 1024 * 2 * T * Bp / ms / 1e9), flush=True)
