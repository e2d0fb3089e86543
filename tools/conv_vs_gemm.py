"""Implicit-GEMM convolution against the plain GEMM of the same shape (is it the gather or the tile?):
python tools/conv_vs_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops


def t(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ops._CONV_SPLITK = False
B = 512
for name, k, c0, c1, cout, H in (('e3', 5, 32, 0, 64, 32), ('d3', 3, 64, 128, 64, 16), ('e4', 3, 64, 0, 128, 16), ('d2', 3, 128, 128, 128, 8),
                                 ('d4', 3, 32, 64, 32, 32)):
    W = H
    M, K = B * H * W, k * k * (c0 + c1)
    s0 = torch.randn(M, c0, device='cuda')
    s1 = torch.randn(B * (H // 2) * (W // 2), c1, device='cuda') if c1 else None
    filt = torch.randn(K, cout, device='cuda') * 0.05
    bias = torch.randn(cout, device='cuda')
    out = torch.empty(M, cout, device='cuda')
    a = torch.randn(M, K, device='cuda')
    ms_c = t(lambda: ops.conv2d(s0, c0, s1, c1, B, H, W, k, filt, bias, out, cout))
    ms_g = t(lambda: ops.gemm(a, filt, out=out, bias=bias))
    fl = 2.0 * M * K * cout
    print('%s M=%d K=%d N=%d: implicit conv %.1f us (%.1f TFLOP/s), plain GEMM %.1f us (%.1f TFLOP/s)'
          % (name, M, K, cout, ms_c * 1e3, fl / ms_c / 1e9, ms_g * 1e3, fl / ms_g / 1e9), flush=True)
    del a
