"""Run one GEMM shape a few times (for rocprofv3 --pmc).  python tools/gemm_one.py M N K [ta tb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avsi_amd
from avsi_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
ta = len(sys.argv) > 4 and sys.argv[4] == '1'
tb = len(sys.argv) > 5 and sys.argv[5] == '1'
a = torch.randn((K, M) if ta else (M, K), device='cuda')
b = torch.randn((N, K) if tb else (K, N), device='cuda')
out = torch.empty(M, N, device='cuda')
for _ in range(3):
    ops.gemm(a, b, out=out, trans_a=ta, trans_b=tb, m=M, n=N, k=K)
torch.cuda.synchronize()
