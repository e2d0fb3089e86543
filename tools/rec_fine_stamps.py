"""Where a step of the 32-way cooperative forward kernel goes at the reference's batch sizes: wall-clock stamps (10 ns) of eight
points of steps 64 .. 71, waves 0 and 1 of the first 32 workgroups (the diagnostic instantiation of
blstm_rec_fwd_coop_fine_kernel).  python tools/rec_fine_stamps.py [Bp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops, _lib
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 250
xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.3
whp = torch.randn(2 * 262144, device='cuda') * 0.05
hout = torch.zeros(T, Bp, 512, device='cuda')
for _ in range(3):
    ops.blstm_rec_fwd(xproj, whp, hout, None, split=32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.blstm_rec_fwd(xproj, whp, hout, None, split=32)
e1.record(); torch.cuda.synchronize()
print("Bp=%d production kernel: %.3f ms per layer, %.2f us per step" % (Bp, e0.elapsed_time(e1) / 10, e0.elapsed_time(e1) / 10 / T * 1e3))
st = torch.zeros(32 * 8 * 2 * 8, dtype=torch.int64, device='cuda')
_lib.lib().avsi_diag_cs_stamps(_lib.ptr(st))
ops.blstm_rec_fwd(xproj, whp, hout, None, split=32)
torch.cuda.synchronize()
_lib.lib().avsi_diag_cs_stamps(None)
s = st.cpu().view(32, 8, 2, 8).double() * 0.01      # us
names = ["top", "counter seen (tid 0)", "barrier passed", "h fragments landed", "MFMAs, park, barrier", "cell done, stores issued",
         "stores acknowledged", "barrier, counter incremented"]
step = (s[:, 1:, 0, 0] - s[:, :-1, 0, 0])
print("stamped build: step time (wave 0, top to top) mean %.2f us  min %.2f  max %.2f" % (step.mean(), step.min(), step.max()))
for w in (0, 1):
    print("wave %d: mean time since the top of the step [us], and the share of the step each segment takes" % w)
    prev = torch.zeros_like(s[:, :, w, 0])
    for ph in range(1, 8):
        if w == 1 and ph in (1, 7):
            continue
        d = s[:, :, w, ph] - s[:, :, w, 0]
        print("   %-30s %6.2f  (min %.2f max %.2f)   segment %5.2f us = %4.1f %%" % (names[ph], d.mean(), d.min(), d.max(),
              (d - prev).mean(), 100 * (d - prev).mean() / step.mean()))
        prev = d
print("block 0, wave 0, steps 64..67 (us since step 64 top):")
for k in range(4):
    print("   ", ["%.2f" % float(s[0, k, 0, ph] - s[0, 0, 0, 0]) for ph in range(8)])
