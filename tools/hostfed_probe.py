"""Why does the double-buffered host-fed step of bench.py not overlap its upload?  python tools/hostfed_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models, ops, audio_processing as ap
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda', 0)
wav_d, masks_d = bench.synth_batch(torch, B, 1234, dev)
spec = ap.frontend(wav_d[:256], want_spec=True)['spec']
mean, std = spec.mean(dim=(0, 1)), spec.std(dim=(0, 1), unbiased=False)
cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=48000, net_dim=[250] * 3, optimizer_type='adam',
           starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
wav_h = torch.empty(wav_d.shape, dtype=torch.float32, pin_memory=True).copy_(wav_d)
masks_h = torch.empty(masks_d.shape, dtype=torch.float32, pin_memory=True).copy_(masks_d)
sets = [(wav_d, masks_d), (torch.empty_like(wav_d), torch.empty_like(masks_d))]
seq = np.full(B, 250)
m = models.StackedBLSTMModel(seq, wav_d, masks_d, mean, std, 0.0, cfg, input='a', seed=7, is_training=False)
NPRE = int(os.environ.get('PROBE_PRE_STREAMS', '0'))
pre = [torch.cuda.Stream(device=dev) for _ in range(NPRE)]          # what a longer program has created before
for st in pre:
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
copy = torch.cuda.Stream(device=dev, priority=-1 if os.environ.get('PROBE_HIGH', '0') == '1' else 0)
print("pre-created streams %d, copy stream priority %s" % (NPRE, copy.priority), flush=True)
main = torch.cuda.current_stream(dev)
a = torch.randn(8192, 8192, device=dev)

def compute_model(cur):
    m.feed(sequence_lengths=seq, target_sources=cur[0], masks=cur[1])
    _ = m.prediction
    return m.loss_func

def compute_blas(cur):
    x = a
    for _ in range(18):
        x = x @ a

def run(compute, steps, pieces=0, upload=True):
    ready = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        cur, nxt = sets[i % 2], sets[(i + 1) % 2]
        if ready is not None:
            main.wait_event(ready)
        if upload:
            copy.wait_stream(main)
            with torch.cuda.stream(copy):
                for dst, src in ((nxt[0], wav_h), (nxt[1], masks_h)):
                    if pieces:
                        d, s = dst.view(-1), src.view(-1)
                        for o in range(0, d.numel(), pieces):
                            d[o:o + pieces].copy_(s[o:o + pieces], non_blocking=True)
                    else:
                        dst.copy_(src, non_blocking=True)
                ready = torch.cuda.Event(); ready.record(copy)
        compute(cur)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps

for name, comp in (("model", compute_model),):
    run(comp, 2)
    print("%s: no upload %.1f ms, upload %.1f ms, upload in 64 MiB pieces %.1f ms" % (
        name, run(comp, 4, upload=False), run(comp, 4), run(comp, 4, pieces=1 << 24)), flush=True)
