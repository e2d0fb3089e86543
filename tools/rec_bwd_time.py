"""Time the batch-stationary BPTT kernel of one layer.  python tools/rec_bwd_time.py [Bp] [T]   (AVSI_BWD_KH=0: whole-tile kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
dh = torch.randn(T, Bp, 512, device='cuda')
resv = torch.rand(T, Bp, 2, 5, 256, device='cuda') * 0.9 + 0.05
whbt = torch.randn(2 * 262144, device='cuda') * 0.05
dz = torch.empty(T, Bp, 2048, device='cuda')
for _ in range(2):
    ops.blstm_rec_bwd(dh, resv, whbt, dz, split=0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.blstm_rec_bwd(dh, resv, whbt, dz, split=0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("Bp=%d T=%d: %.2f ms  %.1f TFLOP/s (KH=%s)" % (Bp, T, ms, 2.0 * 256 * 1024 * 2 * T * Bp / ms / 1e9, os.environ.get('AVSI_BWD_KH', '1')), flush=True)
