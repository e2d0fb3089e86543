"""Time the exploratory split-bf16 GEMM against the fp32 one on the layer shapes.  python tools/gemm_bf16x3_time.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048000
for K in (272, 512):
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, 2048, device='cuda') * 0.05; bias = torch.randn(2048, device='cuda')
    out = torch.empty(M, 2048, device='cuda')
    bp = ops.pack_bf16x3_b(b)
    for name, fn in (("f32", lambda: ops.gemm(a, b, out=out, bias=bias)), ("bf16x3", lambda: ops.gemm_bf16x3(a, bp, out, K, bias=bias))):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("M=%d K=%d %-7s %.2f ms  %.0f TFLOP/s (fp32-equivalent)" % (M, K, name, ms, 2.0 * M * K * 2048 / ms / 1e9), flush=True)
