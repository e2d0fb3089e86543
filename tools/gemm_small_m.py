"""Layer-GEMM shapes at small batches (M = 250 * Bp rows, N = 2048): time per launch with the 128- and 256-wide N tile.
python tools/gemm_small_m.py [Bp ...]   (AVSI_GEMM_WIDE_MIN = workgroups from which the 128 x 256 tile is taken)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import ops
for Bp in [int(x) for x in sys.argv[1:]] or (32, 64, 96, 128, 160, 256):
    M = 250 * Bp
    line = []
    for K in (272, 512):
        a = torch.randn(M, K, device='cuda'); b = torch.randn(K, 2048, device='cuda'); bias = torch.randn(2048, device='cuda')
        out = torch.empty(M, 2048, device='cuda')
        for _ in range(3):
            ops.gemm(a, b, out=out, bias=bias)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, out=out, bias=bias)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line.append("K=%d %.1f us (%.0f TFLOP/s)" % (K, us, 2.0 * M * K * 2048 / us / 1e6))
    print("Bp=%d M=%d  " % (Bp, M) + "  ".join(line), flush=True)
