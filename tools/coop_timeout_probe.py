"""Diagnostic: what a cooperative recurrent launch does while most of the chip is parked (tests/test_step_guard_gpu.py)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import avsi_amd  # noqa: E402,F401
from avsi_amd import ops  # noqa: E402

T, Bp = 10, 32
xproj = torch.randn(T, Bp, 2048, device='cuda') * 0.1
whp = torch.randn(2 * 262144, device='cuda') * 0.05
hout = torch.zeros(T, Bp, 512, device='cuda')
ops.blstm_rec_fwd(xproj, whp, hout, None)
torch.cuda.synchronize()
print('split', ops.coop_split(Bp), 'status', [int(w[0]) for w in ops._COOP_WS.values()])

for parked in (int(a) for a in sys.argv[1:] or ['240']):
    release = torch.zeros(1, dtype=torch.int32, device='cuda')
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.cuda.stream(side):
        ops.occupy_cus(parked, release, max_ms=8000)
    x = torch.ones(1 << 20, device='cuda')
    y = (x * 2).sum()
    float(y)
    t1 = time.time()
    ops.blstm_rec_fwd(xproj, whp, hout, None)
    torch.cuda.current_stream().synchronize()
    t2 = time.time()
    ops.blstm_rec_fwd(xproj, whp, hout, None)
    torch.cuda.current_stream().synchronize()
    t3 = time.time()
    still = not side.query()
    st = [int(w[:1].item()) for w in ops._COOP_WS.values()]
    release.fill_(1)
    torch.cuda.synchronize()
    print('parked %d: small torch kernels %.3f s, cooperative launch %.3f s, the next one %.3f s, parked kernel still there %s, status %s'
          % (parked, t1 - t0, t2 - t1, t3 - t2, still, st), flush=True)
    for ws in ops._COOP_WS.values():
        ws.zero_()
