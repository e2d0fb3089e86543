"""Where a step of the column-split recurrent kernel goes: wall-clock stamps (10 ns) of eight phases, steps 64 .. 71,
waves 0 and 1 of the first 32 workgroups.  python tools/rec_cs_stamps.py [Bp] [rows_per_group]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import ops, _lib
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = 250
xproj = torch.randn(T, Bp, 2048, device='cuda')
whp = torch.randn(2 * 262144, device='cuda') * 0.05
hout = torch.zeros(T, Bp, 512, device='cuda')
for _ in range(2):
    ops.blstm_rec_fwd(xproj, whp, hout, None, split=-rows)
st = torch.zeros(32 * 8 * 2 * 8, dtype=torch.int64, device='cuda')
_lib.lib().avsi_diag_cs_stamps(_lib.ptr(st))
ops.blstm_rec_fwd(xproj, whp, hout, None, split=-rows)
torch.cuda.synchronize()
_lib.lib().avsi_diag_cs_stamps(None)
s = st.cpu().view(32, 8, 2, 8).double() * 0.01      # us
names = ["top", "polled", "barrier C", "h loaded", "barrier A", "cells done", "barrier B", "published"]
step = (s[:, 1:, 0, 0] - s[:, :-1, 0, 0])
print("Bp=%d rows=%d: step time (wave 0, top to top) mean %.2f us  min %.2f  max %.2f" % (Bp, rows, step.mean(), step.min(), step.max()))
for w in (0, 1):
    print("wave %d: mean time since the top of the step [us]" % w)
    for ph in range(1, 8):
        if w == 1 and ph in (1, 7):
            continue
        d = s[:, :, w, ph] - s[:, :, w, 0]
        print("   %-10s %6.2f  (min %.2f max %.2f)" % (names[ph], d.mean(), d.min(), d.max()))
print("block 0, wave 0, steps 64..67 (us since step 64 top):")
for k in range(4):
    print("   ", ["%.2f" % float(s[0, k, 0, ph] - s[0, 0, 0, 0]) for ph in range(8)])
