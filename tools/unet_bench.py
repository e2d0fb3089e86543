"""Time the U-Net inpainter (configs[4]): python tools/unet_bench.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 16384
cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
           learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav = torch.round(torch.randn(B, N, device='cuda') * 3000)
masks = torch.ones(B, 128, 128, device='cuda'); masks[:, 40:52] = 0
seq = np.full(B, 128)
for train in (False, True):
    m = models.UNetFConvModel(seq, wav, masks, torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2, 0.0, cfg, is_training=train)
    def step():
        m.feed(sequence_lengths=seq, target_sources=wav, masks=masks)
        l = m.loss_func
        if train: m.train_op
        return l
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    flops = 0.52e9 * B * (3 if train else 1)
    print("B=%d %s: %.2f ms/step  %.0f clips/s  ~%.1f TFLOP/s" % (B, "train" if train else "infer", dt * 1e3, B / dt, flops / dt / 1e12))
