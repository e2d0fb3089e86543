"""U-Net inference at the reference's batch size, eager vs captured HIP graph: python tools/unet_graph_bench.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 16384
cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
           lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav = torch.round(torch.randn(B, N, device='cuda') * 3000)
masks = torch.ones(B, 128, 128, device='cuda'); masks[:, 40:52] = 0
seq = np.full(B, 128)
m = models.UNetFConvModel(seq, wav, masks, torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2, 0.0, cfg, is_training=False)
def run(tag):
    for _ in range(5):
        m.feed(seq, wav, masks); _ = m.loss_func
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        m.feed(seq, wav, masks); _ = m.loss_func
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("B=%d %s: %.3f ms/step  %.0f clips/s" % (B, tag, dt * 1e3, B / dt))
run("eager")
m.capture_graph()
run("graph")
