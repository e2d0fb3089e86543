"""Where does the layer-0 product (K = 272) lose against the K = 512 layers?  Times C[M, 2048] = A[M, K] . B + bias at the
benchmark's M = 2,048,000 rows for K = 272 / 512 / 1024, as the model calls it (k_zero promise for the padding).  Run twice:
plain, and with AVSI_GEMM_DIAG=1 (the wide tile stores nothing: what the output stores cost).
python tools/gemm_layer0_gap.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd  # noqa: F401
from avsi_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048000
N = 2048
out = torch.empty(M, N, device='cuda')
bias = torch.randn(N, device='cuda')
for K, kz in ((272, ((257, 272),)), (512, ((250, 256), (506, 512))), (1024, None)):
    a = torch.randn(M, K, device='cuda')
    b = torch.randn(K, N, device='cuda') * 0.05
    if kz:
        for lo, hi in kz:
            b[lo:hi] = 0
    for _ in range(2):
        ops.gemm(a, b, out=out, bias=bias, k_zero=kz)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.gemm(a, b, out=out, bias=bias, k_zero=kz)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    kreal = K - sum(hi - lo for lo, hi in (kz or ()))
    print("AVSI_GEMM_DIAG=%s M=%d N=%d K=%d (%d real): %.2f ms  %.1f TFLOP/s on the real k, %.1f on the padded; output %.1f GB"
          % (os.environ.get('AVSI_GEMM_DIAG', '0'), M, N, K, kreal, ms, 2.0 * M * N * kreal / ms / 1e9, 2.0 * M * N * K / ms / 1e9,
             M * N * 4 / 1e9), flush=True)
    del a, b
