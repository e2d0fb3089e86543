"""The 257-bin projection of a batch as one product (five 64-column tiles) and as 256 + 1 bins (wide tile + narrow tail):
time of each and that they agree bit for bit.  python tools/proj_time.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
Bp, F, K, ldp = B, 257, 512, 260
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn((T * Bp, K), device='cuda', generator=g)
w = torch.randn((K, ldp), device='cuda', generator=g) * 0.05
b = torch.randn((ldp,), device='cuda', generator=g)
rs = (torch.rand((T * Bp,), device='cuda', generator=g) > 0.1).float()
rm = (Bp, T, B)


def one(out):
    ops.gemm(x, w, out=out, n=F, bias=b, row_scale=rs, row_map=rm)


def two(out):
    ops.gemm(x, w, out=out, n=256, bias=b, row_scale=rs, row_map=rm)
    ops.gemm(x, w[:, 256:F], out=out[:, 256:], n=F - 256, bias=b[256:F], row_scale=rs, row_map=rm)


def t(fn, out, n=10):
    for _ in range(3):
        fn(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


o1 = torch.full((B * T, F), -7.0, device='cuda')
o2 = torch.full((B * T, F), -9.0, device='cuda')
t1, t2 = t(one, o1), t(two, o2)
fl = 2.0 * T * Bp * K * F
print('one product %.3f ms (%.1f TFLOP/s)   256 + 1: %.3f ms (%.1f TFLOP/s)   equal: %s   max diff %.3g'
      % (t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, bool(torch.equal(o1, o2)), float((o1 - o2).abs().max())))
