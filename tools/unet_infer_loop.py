"""U-Net inference steps only (for rocprofv3 --kernel-trace --stats).  python tools/unet_infer_loop.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
N = 16384
cfg = dict(audio_feat_dim=128, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam', starter_learning_rate=1e-3,
           learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav = torch.round(torch.randn(B, N, device='cuda') * 3000)
masks = torch.ones(B, 128, 128, device='cuda'); masks[:, 40:52] = 0
seq = np.full(B, 128)
m = models.UNetFConvModel(seq, wav, masks, torch.zeros(128, device='cuda') + 6, torch.ones(128, device='cuda') * 2, 0.0, cfg, is_training=False)
for _ in range(steps):
    m.feed(sequence_lengths=seq, target_sources=wav, masks=masks)
    l = m.loss_func
torch.cuda.synchronize()
print(float(l))
