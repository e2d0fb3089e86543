"""U-Net training steps for profiling: python tools/unet_train_one.py [B] [steps]  (default 10 steps: the first creates every buffer --
its zero fills and index kernels are a tenth of the table then; the printed ms is the steady state of the last steps - 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = 16384
cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
           lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav = torch.round(torch.randn(B, N, device='cuda') * 3000)
masks = torch.ones(B, 128, 128, device='cuda'); masks[:, 40:52] = 0
seq = np.full(B, 128)
m = models.UNetFConvModel(seq, wav, masks, torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2, 0.0, cfg, is_training=True)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
import time
for i in range(steps):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.feed(sequence_lengths=seq, target_sources=wav, masks=masks)
    _ = m.loss_func
    m.train_op
torch.cuda.synchronize()
if steps > 2:
    print('B=%d unet training: %.3f ms per step' % (B, (time.perf_counter() - t0) / (steps - 2) * 1e3), flush=True)
