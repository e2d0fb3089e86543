"""Is the recurrent forward kernel slower on buffers no cache holds?  Rotates hout and / or xproj over enough buffers to
exceed the 256 MB memory-side cache.  python tools/rec_cold_time.py [Bp] [split]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = 250
whp = torch.randn(2 * 262144, device='cuda') * 0.05
nx = max(2, int(600e6 / (T * Bp * 2048 * 4)) + 1)
nh = max(2, int(600e6 / (T * Bp * 512 * 4)) + 1)
xs = [torch.randn(T, Bp, 2048, device='cuda') for _ in range(nx)]
hs = [torch.zeros(T, Bp, 512, device='cuda') for _ in range(nh)]
for rot_x, rot_h in ((False, False), (True, False), (False, True), (True, True)):
    for i in range(3):
        ops.blstm_rec_fwd(xs[i % nx if rot_x else 0], whp, hs[i % nh if rot_h else 0], None, split=sp)
    n = 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.blstm_rec_fwd(xs[i % nx if rot_x else 0], whp, hs[i % nh if rot_h else 0], None, split=sp)
    e1.record(); torch.cuda.synchronize()
    print("Bp=%d split %d  xproj %s, hout %s: %.3f ms" % (Bp, sp, "rotating" if rot_x else "fixed", "rotating" if rot_h else "fixed", e0.elapsed_time(e1) / n), flush=True)
ops.coop_check()
