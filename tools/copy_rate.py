"""Streaming-copy yardstick: python tools/copy_rate.py  (AVSI_DIAG_COPY_NT=1: non-temporal)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda', 0)
for mib in (512, 2048, 8192):
    print(mib, 'MiB: float4 %.0f GB/s, torch %.0f GB/s' % (bench.device_copy_rate(torch, dev, mib=mib), bench.device_copy_rate(torch, dev, mib=mib, kernel='torch')), flush=True)
