"""Time the batch-stationary forward recurrence of one layer, without and with the BPTT reserve: python tools/rec_fwd_time.py [Bp] [T] [rows_per_wg]
(rows_per_wg 0 = the policy's kernel, 64 = ping-pong, 65 = both 32-row tiles in ONE MFMA phase on the same Wh fragments, 32 = one tile)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
RPW = int(sys.argv[3]) if len(sys.argv) > 3 else 0
xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.3
whp = torch.randn(2 * 262144, device='cuda') * 0.05
hout = torch.empty(T, Bp, 512, device='cuda')
resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
for name, r in (('no reserve', None), ('with reserve', resv)):
    for _ in range(2):
        ops.blstm_rec_fwd(xproj, whp, hout, r, rows_per_wg=RPW, split=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.blstm_rec_fwd(xproj, whp, hout, r, rows_per_wg=RPW, split=0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("Bp=%d T=%d rows_per_wg=%d %s: %.2f ms  %.1f TFLOP/s" % (Bp, T, RPW, name, ms, 2.0 * 256 * 1024 * 2 * T * Bp / ms / 1e9), flush=True)
