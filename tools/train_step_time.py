"""AV training step at a small batch under different stream arrangements (HIP multiplexes streams onto hardware queues:
which streams share one decides what overlaps).  python tools/train_step_time.py [B]
env: MAIN_HIGH=1 run the step on a high-priority stream; PRE_STREAMS=n create n streams first (what a reader thread etc. do)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import avsi_amd
from avsi_amd import models, ops, audio_processing as ap
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
pre = [torch.cuda.Stream(device=dev, priority=-1 if os.environ.get('PRE_HIGH') == '1' else 0) for _ in range(int(os.environ.get('PRE_STREAMS', '0')))]
for st in pre:
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
wav, masks, video = bench.av_batch(torch, B, 1, dev)
spec = ap.frontend(wav, want_spec=True)['spec']
mean, std = spec.mean(dim=(0, 1)), spec.std(dim=(0, 1), unbiased=False)
cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=48000, net_dim=[250] * 3, optimizer_type='adam',
           starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
seq = np.full(B, 250)
if os.environ.get('DUMMY_THREAD'):
    # what a reader thread does to the launching thread: short bursts of interpreter work between blocking calls
    import threading
    busy_us = int(os.environ.get('DUMMY_THREAD'))
    def dummy():
        while True:
            time.sleep(0.003)
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e6 < busy_us:
                sum(range(50))
    threading.Thread(target=dummy, daemon=True).start()
if os.environ.get('SWITCH_US'):
    sys.setswitchinterval(int(os.environ['SWITCH_US']) * 1e-6)
main = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get('MAIN_HIGH') == '1' else torch.cuda.current_stream(dev)
with torch.cuda.stream(main):
    m = models.StackedBLSTMModel(seq, wav, masks, mean, std, 0.0, cfg, video_features=video, input='av', seed=7, is_training=True)
    def step():
        m.feed(sequence_lengths=seq, target_sources=wav, masks=masks, video_features=video)
        loss = m.loss_func
        m.train_op
        return loss
    torch.set_num_threads(1)
    ms = bench.time_steps(torch, step, 40, 8)
    ops.coop_check(dev)
print("dummy thread %s us, switch interval %s us: " % (os.environ.get('DUMMY_THREAD', '-'), os.environ.get('SWITCH_US', 'default')), end="")
print("B=%d MAIN_HIGH=%s PRE_STREAMS=%s PRE_HIGH=%s GPU_MAX_HW_QUEUES=%s: %.2f ms per step" % (
    B, os.environ.get('MAIN_HIGH', '0'), os.environ.get('PRE_STREAMS', '0'), os.environ.get('PRE_HIGH', '0'),
    os.environ.get('GPU_MAX_HW_QUEUES', 'default'), ms), flush=True)
