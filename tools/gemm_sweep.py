"""Time avsi_gemm_f32 over shapes (diagnostic).  python tools/gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import ops


def t(M, N, K, ta=False, tb=False, reps=5):
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, -(-N // 4) * 4), device='cuda')
    fill = os.environ.get('AVSI_SWEEP_FILL')  # 'zero' / 'const': operand-data dependence of the MFMA rate
    if fill == 'zero':
        a.zero_(); b.zero_()
    elif fill == 'const':
        a.fill_(1.5); b.fill_(0.75)
    out = torch.empty(M, N, device='cuda')
    ops.gemm(a, b, out=out, trans_a=ta, trans_b=tb, m=M, n=N, k=K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(a, b, out=out, trans_a=ta, trans_b=tb, m=M, n=N, k=K)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print("M=%d N=%d K=%d ta=%d tb=%d: %.3f ms  %.1f TFLOP/s" % (M, N, K, ta, tb, ms, 2.0 * M * N * K / ms / 1e9))


if __name__ == "__main__":
    for K in (256, 264, 512, 1024, 4096):
        t(524288, 2048, K)
    t(524288, 257, 512)
    t(524288, 256, 512)
    t(8192, 8192, 8192)
    t(512, 2048, 524288, ta=True)
    t(524288, 512, 2048, tb=True)
