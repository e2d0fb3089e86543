"""Worker of tests/test_dp_gpu.py: one data-parallel rank (launched by torch.distributed.run) trains the AV model
for a few steps on ITS half of a fixed batch; rank 0 stores the resulting variables."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_inputs(B, N=2880):
    rng = np.random.default_rng(123)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        masks[b, 3 + b % 4: 7 + b % 4] = 0
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    mean = rng.normal(0, 1, size=257).astype(np.float32)
    std = (1.0 + rng.random(257)).astype(np.float32)
    return wav, masks, video, mean, std, T


def config(N, B):
    return dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
                starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)


def run(rank, world, steps, B_global=4, N=2880):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    wav, masks, video, mean, std, T = make_inputs(B_global, N)
    per = B_global // world
    sl = slice(rank * per, (rank + 1) * per)
    m = models.StackedBLSTMModel(np.full(per, T), wav[sl], masks[sl], mean, std, 0.0, config(wav.shape[1], per),
                                 video_features=video[sl], input='av', seed=7)
    from avsi_amd import ops
    losses = []
    for _ in range(steps):
        while True:
            m.feed(sequence_lengths=np.full(per, T), target_sources=wav[sl], masks=masks[sl], video_features=video[sl])
            loss = float(m.loss_func)
            m.train_op
            # what training.train does with the step guard: ranks that share ONE GPU can starve each other's 32-way
            # cooperative groups (every rank wants all of XCD 0 and 1; the bounded waits resolve the stand-off after 2 s).
            # The guard words are summed over the ranks, so every rank repeats the void step together.
            if float(m.step_guard[1]) == 0.0:
                break
            ops.coop_fall_back()
            m.variables.rewind_step()
        losses.append(loss)
    return m.variables.flat.cpu().numpy(), losses


def run_variant(rank, world, steps, B_global=4, N=2880, gaps='uneven'):
    """The external-embedding variant (objective loss_hole = sum|err|(1-m) / sum(1-m)) on a batch whose gap frames are
    spread UNEVENLY over the ranks' shards: utterance b has a gap of 2 + 3 b frames.  Returns the variables, the per-step
    local loss_hole and the per-step loss_hole_global."""
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    from avsi_amd.model_variants import StackedBLSTMEmbeddingModel
    wav, masks, video, mean, std, T = make_inputs(B_global, N)
    masks[:] = 1
    for b in range(B_global):
        if gaps == 'zero' and b < B_global // 2:
            continue                # the first half of the batch (rank 0's shard at world 2) has no gap element at all
        masks[b, 1: 3 + 3 * b] = 0
    emb = np.random.default_rng(9).normal(size=(B_global, 512)).astype(np.float32)
    per = B_global // world
    sl = slice(rank * per, (rank + 1) * per)
    cfg = dict(config(N, per), integration_layer=1)
    m = StackedBLSTMEmbeddingModel(np.full(per, T), wav[sl], masks[sl], mean, std, 0.0, cfg, video_features=video[sl],
                                   embeddings=emb[sl], input='av', seed=7)
    local, glob = [], []
    for _ in range(steps):
        while True:
            m.feed(sequence_lengths=np.full(per, T), target_sources=wav[sl], masks=masks[sl], video_features=video[sl])
            m.feed_embeddings(emb[sl])
            hole, hole_g = float(m.loss_hole), float(m.loss_hole_global)
            m.train_op
            if float(m.step_guard[1]) == 0.0:
                break
            ops.coop_fall_back()
            m.variables.rewind_step()
        local.append(hole)
        glob.append(hole_g)
    return m.variables.flat.cpu().numpy(), local, glob


if __name__ == '__main__':
    out = sys.argv[1]
    if len(sys.argv) > 5 and sys.argv[5] == 'emb':
        from avsi_amd import parallel
        rank, world = parallel.init()
        flat, local, glob = run_variant(rank, world, int(sys.argv[4]), int(sys.argv[2]), int(sys.argv[3]),
                                        gaps=sys.argv[6] if len(sys.argv) > 6 else 'uneven')
        np.save(os.path.join(out, 'flat_rank%d.npy' % rank), flat)
        np.save(os.path.join(out, 'loss_rank%d.npy' % rank), np.array([local, glob]))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0)
    from avsi_amd import parallel
    rank, world = parallel.init()
    import torch.distributed as dist
    calls = {'sync': 0, 'async': 0}
    plain_all_reduce = dist.all_reduce

    def counted(t, *a, **kw):
        calls['async' if kw.get('async_op') else 'sync'] += 1
        return plain_all_reduce(t, *a, **kw)
    dist.all_reduce = counted
    flat, losses = run(rank, world, steps=int(sys.argv[4]) if len(sys.argv) > 4 else 3,
                       B_global=int(sys.argv[2]) if len(sys.argv) > 2 else 4, N=int(sys.argv[3]) if len(sys.argv) > 3 else 2880)
    np.save(os.path.join(out, 'flat_rank%d.npy' % rank), flat)
    np.save(os.path.join(out, 'loss_rank%d.npy' % rank), np.array(losses))
    from avsi_amd import ops
    print('RANK %d BACKEND %s SHARES_GPU %s COOP_CUS %d ALL_REDUCE async %d sync %d' % (
        rank, dist.get_backend(), parallel.collectives_share_the_gpu(), ops.coop_cu_budget(), calls['async'], calls['sync']),
        flush=True)
    dist.barrier()
    dist.destroy_process_group()
