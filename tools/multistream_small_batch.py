"""Independent small batches on several streams (serving pattern): python tools/multistream_small_batch.py [streams]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models
from avsi_amd import audio_processing as ap
B, NS = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ.setdefault('AVSI_COOP_CUS', str(256 // NS))     # the streams share the chip: no launch may need more than its part
cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=48000, net_dim=[250, 250, 250], optimizer_type='adam',
           starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav = torch.round(torch.randn(B, 48000, device='cuda') * 3000)
masks = torch.ones(B, 250, 257, device='cuda'); masks[:, 100:133] = 0
spec = ap.frontend(wav, want_spec=True)['spec']
mean, std = spec.mean(dim=(0, 1)), spec.std(dim=(0, 1), unbiased=False)
seq = np.full(B, 250)
variables = models.BLSTMVariables(models.ParamLayout(257), seed=1)
streams = [torch.cuda.Stream() for _ in range(NS)]
ms = []
for s in streams:
    with torch.cuda.stream(s):
        ms.append(models.StackedBLSTMModel(seq, wav, masks, mean, std, 0.0, cfg, input='a', is_training=False, variables=variables))
def round_():
    outs = []
    for s, m in zip(streams, ms):
        with torch.cuda.stream(s):
            m.feed(seq, wav, masks)
            outs.append(m.prediction)
    return outs
for _ in range(2): round_()
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 5
for _ in range(R): outs = round_()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('streams %d: %.2f ms per round, %.0f utt/s aggregate' % (NS, dt / R * 1e3, NS * B * R / dt))
ref = outs[0]
print('all equal:', all(torch.equal(o, ref) for o in outs))
