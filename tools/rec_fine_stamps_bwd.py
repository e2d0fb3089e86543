"""Where a step of the half-tile cooperative BPTT kernel goes (32 utterances: 16 unit slices x 2 row halves per tile and direction):
wall-clock stamps (10 ns) of eight points of steps 64 .. 71, waves 0 and 1 of the first 32 workgroups (the diagnostic
instantiation of blstm_rec_bwd_coop_fine_kernel<2, true>).  python tools/rec_fine_stamps_bwd.py [Bp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops, _lib
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 250
dh = torch.randn(T, Bp, 512, device='cuda')
resv = torch.rand(T, Bp, 2, 5, 256, device='cuda') * 0.9 + 0.05
whbt = torch.randn(2 * 262144, device='cuda') * 0.05
dz = torch.empty(T, Bp, 2048, device='cuda')
for _ in range(3):
    ops.blstm_rec_bwd(dh, resv, whbt, dz, split=32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.blstm_rec_bwd(dh, resv, whbt, dz, split=32)
e1.record(); torch.cuda.synchronize()
print("Bp=%d production kernel: %.3f ms per layer, %.2f us per step" % (Bp, e0.elapsed_time(e1) / 10, e0.elapsed_time(e1) / 10 / T * 1e3))
st = torch.zeros(32 * 8 * 2 * 8, dtype=torch.int64, device='cuda')
_lib.lib().avsi_diag_cs_stamps(_lib.ptr(st))
ops.blstm_rec_bwd(dh, resv, whbt, dz, split=32)
torch.cuda.synchronize()
_lib.lib().avsi_diag_cs_stamps(None)
s = st.cpu().view(32, 8, 2, 8).double() * 0.01      # us
names = ["top", "counter seen (tid 0)", "barrier passed", "dz fragments landed", "MFMAs, park, barrier", "cell done, stores issued",
         "stores acknowledged", "barrier, counter incremented"]
live = [b for b in range(32) if float(s[b].abs().sum()) > 0]
print("workgroups with stamps:", live)
for b in live[:4]:
    print("block %d, wave 0, steps 64..67 (us since step 64 top):" % b)
    for k in range(4):
        print("   ", ["%.2f" % float(s[b, k, 0, ph] - s[b, 0, 0, 0]) for ph in range(8)])
sl = s[live]
step = (sl[:, 1:, 0, 0] - sl[:, :-1, 0, 0])
print("stamped build: step (wave 0, top to top) mean %.2f us" % step.mean())
prev = torch.zeros_like(sl[:, :, 0, 0])
for ph in range(1, 8):
    d = sl[:, :, 0, ph] - sl[:, :, 0, 0]
    print("   %-30s %6.2f   segment %5.2f us = %4.1f %%" % (names[ph], d.mean(), (d - prev).mean(), 100 * (d - prev).mean() / step.mean()))
    prev = d
