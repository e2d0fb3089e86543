"""The device-side step guard and the fall-back from a cooperative-kernel timeout.

No reference counterpart: the reference's only failure rule is the host-side NaN / Inf abort AFTER the update
(training_emb.py:244-249).  Here (a) the fused Adam reads two guard words on the device -- "a loss is not finite on some
rank", "a cooperative recurrent launch timed out on some rank" -- and leaves variables and slots alone when either is
set, so neither a checkpoint nor a retry ever sees a void update; (b) a timeout costs the trainer one repetition of the
skipped batches on the batch-stationary kernels, in the same process, instead of the run."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_guarded_adam_obeys_the_skip_words():
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(0)
    n = 100003
    p0 = torch.randn(n, generator=g, device='cuda')
    grad = torch.randn(n, generator=g, device='cuda')
    m0, v0 = torch.randn(n, generator=g, device='cuda') * 0.1, torch.rand(n, generator=g, device='cuda') * 0.1

    def run(skip):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        ops.adam_tf(p, grad, m, v, 3, 1e-3, skip=None if skip is None else torch.tensor(skip, dtype=torch.float32, device='cuda'))
        return p, m, v
    plain = run(None)
    assert not torch.equal(plain[0], p0)
    for clean in ([0.0, 0.0], [0.0], [-0.0, 0.0]):
        assert all(torch.equal(a, b) for a, b in zip(run(clean), plain))
    for void in ([float('nan'), 0.0], [0.0, 1.0], [0.0, 3.0], [float('inf'), 0.0], [1e-30]):
        p, m, v = run(void)
        assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0), void


def test_step_guard_words():
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    out = torch.full((2,), 7.0, device='cuda')
    for loss, want in ((1.5, 0.0), (0.0, 0.0), (float('nan'), None), (float('inf'), None), (-float('inf'), None)):
        ops.step_guard(torch.tensor([loss], device='cuda'), out)
        got = out.cpu().numpy()
        assert (np.isnan(got[0]) if want is None else got[0] == want), loss
        assert got[1] in (0.0, 1.0)
    ops.step_guard(None, out)
    assert out[0].item() == 0.0


def _model(B, T=10, seed=3, nan=False):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    N = 192 * T
    g = torch.Generator(device='cuda')
    g.manual_seed(11)
    wav = torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000)
    masks = torch.ones(B, T, 257, device='cuda')
    masks[:, 3:6] = 0
    video = torch.randn(B, T, 136, generator=g, device='cuda')
    mean, std = torch.full((257,), 6.0, device='cuda'), torch.full((257,), 2.0, device='cuda')
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
               starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    m = models.StackedBLSTMModel(np.full(B, T), wav, masks, mean, std, 0.0, cfg, video_features=video, input='av', seed=seed)
    return m, dict(sequence_lengths=np.full(B, T), target_sources=wav, masks=masks, video_features=video)


@pytest.mark.parametrize("optimizer", ["adam", "momentum", "sgd"])
def test_a_non_finite_loss_never_reaches_the_variables(optimizer):
    m, feed = _model(8)
    m.optimizer_choice = optimizer
    m.feed(**feed)
    m.train_op                                  # a good step first: slots exist, variables moved
    assert m.step_guard.cpu().tolist() == [0.0, 0.0]
    v = m.variables
    before = [t.clone() for t in (v.flat, v.packed) + ((v.adam_m,) if v.adam_m is not None else ())
              + ((v.adam_v,) if v.adam_v is not None else ())]
    bad = dict(feed, masks=feed['masks'] * float('nan'))
    m.feed(**bad)
    assert not np.isfinite(float(m.loss_func))
    m.train_op
    guard = m.step_guard.cpu().numpy()
    assert np.isnan(guard[0]) and guard[1] == 0.0
    after = (v.flat, v.packed) + ((v.adam_m,) if v.adam_m is not None else ()) + ((v.adam_v,) if v.adam_v is not None else ())
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    m.feed(**feed)                              # and the model goes on from the last good state
    m.train_op
    assert m.step_guard.cpu().tolist() == [0.0, 0.0] and not torch.equal(v.flat, before[0])


def _park(cus, ms, priority=-1):
    """Park `cus` whole compute units (160 KB of LDS each) on a stream of ANOTHER priority level than the stream the model
    runs on: HIP maps its streams onto a few hardware queues per priority level, and a parked kernel that shares the
    model's queue would simply hold the model's kernels back in order (tools/coop_timeout_probe.py)."""
    from avsi_amd import ops
    release = torch.zeros(1, dtype=torch.int32, device='cuda')
    side = torch.cuda.Stream(priority=priority)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.occupy_cus(cus, release, max_ms=ms)
    return release, side


def test_a_cooperative_timeout_voids_the_step_and_the_fall_back_repeats_it(monkeypatch):
    """16 of the 256 CUs are parked with all their LDS taken (the stand-in for another resident of the GPU): a 32-way
    cooperative group needs the 32 CUs of ITS XCD, cannot become resident, the bounded wait gives up after 2 s and the
    guard voids the update.  After ops.coop_fall_back() the same model repeats the step on the 8-way kernels WHILE THE
    NEIGHBOUR IS STILL THERE; a second fall-back leads to the batch-stationary kernels.  The variables in the end equal
    a run that never used the cooperative kernels (AVSI_REC_COOP=0), to rounding."""
    from avsi_amd import ops
    for name in ('AVSI_REC_COOP', 'AVSI_COOP_CUS', 'AVSI_COOP_SPLIT_FWD', 'AVSI_COOP_SPLIT_BWD', 'AVSI_COOP_TIMEOUT_MS'):
        monkeypatch.delenv(name, raising=False)
    ops.coop_fall_back_reset()
    ops.set_coop_cu_budget(None)
    monkeypatch.setattr(ops, 'COOP_POLL_RAISES', False)         # the guard decides, as in the trainer
    B = 32
    assert ops.coop_split(B) == 64 and ops.coop_split(B, backward=True) == 32      # forward: the 32-way kernel on 16-row halves
    m, feed = _model(B)
    m.feed(**feed)
    m.train_op                                  # step 1: 32-way cooperative kernels, nothing in their way
    assert m.step_guard.cpu().tolist() == [0.0, 0.0]
    v1 = m.variables.flat.clone()
    m1, s1 = m.variables.adam_m.clone(), m.variables.global_step

    release, side = _park(16, 40000)
    try:
        m.feed(**feed)
        m.train_op                              # step 2: the forward launch of layer 0 gives up after 2 s; every launch
        guard = m.step_guard.cpu().numpy()      # behind it leaves at once (sticky status word)
        assert guard[1] == 1.0
        assert torch.equal(m.variables.flat, v1) and torch.equal(m.variables.adam_m, m1)       # untouched
        assert m.variables.global_step == s1 + 1                                               # the host counted it ...
        with pytest.raises(ops.CoopTimeout):
            ops.coop_check()
        ops.coop_fall_back()
        m.variables.rewind_step()                                                              # ... and takes it back
        assert ops.coop_level() == 1 and ops.coop_fallbacks() == 1 and not ops.coop_disabled()
        assert ops.coop_split(B) == 8 and ops.coop_split(B, backward=True) == 8
        m.feed(**feed)                                                                         # step 2 again
        m.train_op
        assert m.step_guard.cpu().tolist() == [0.0, 0.0]
        assert not side.query()                                  # the neighbour was there all the time
        ops.coop_fall_back()                                     # a second conflict would end here
        assert ops.coop_level() == 2 and ops.coop_disabled() and ops.coop_split(B) == 0 and ops.coop_split(B, backward=True) == 0
        m.feed(**feed)                                                                         # step 3
        m.train_op
        assert m.step_guard.cpu().tolist() == [0.0, 0.0]
        ops.coop_check()
        assert m.variables.global_step == 3
        got = m.variables.flat.clone()
    finally:
        release.fill_(1)
        torch.cuda.synchronize()
        ops.coop_fall_back_reset()

    monkeypatch.setenv('AVSI_REC_COOP', '0')
    ref, _ = _model(B)
    for _ in range(3):
        ref.feed(**feed)
        ref.train_op
    assert (ref.variables.flat - v1).abs().max().item() > 1e-3          # two further steps moved the weights
    assert (got - ref.variables.flat).abs().max().item() < 2e-5


def test_trainer_recovers_from_a_cooperative_timeout(tmp_path, monkeypatch, capsys):
    """training.train() end to end: from the third batch of the first epoch on, 16 CUs are parked for 10 s.  The run must
    go on (one fall-back, logged once), take exactly as many optimiser steps as there are batches, and end with the
    variables of a run on the batch-stationary kernels, to rounding."""
    import avsi_amd  # noqa: F401
    from avsi_amd import ops, training
    from test_drivers_gpu import _make_dataset
    for name in ('AVSI_REC_COOP', 'AVSI_COOP_CUS', 'AVSI_COOP_SPLIT_FWD', 'AVSI_COOP_SPLIT_BWD', 'AVSI_COOP_TIMEOUT_MS',
                 'AVSI_TRAIN_STREAM'):
        monkeypatch.delenv(name, raising=False)
    ops.coop_fall_back_reset()
    ops.set_coop_cu_budget(None)
    monkeypatch.setenv('AVSI_SHUFFLE_SEED', '5')
    data = str(tmp_path / "tfrecords")
    _make_dataset(os.path.join(data, "training-set"), 20, 0)
    _make_dataset(os.path.join(data, "validation-set"), 4, 1)
    np.save(str(tmp_path / "mean.npy"), np.full(257, 6.0))
    np.save(str(tmp_path / "std.npy"), np.full(257, 2.0))

    def config(name):
        cfg = tmp_path / (name + ".config")
        cfg.write_text("\n".join([
            "model = av-blstm", "audio_feat_dim = 257", "video_feat_dim = 136", "audio_len = 3840", "batch_size = 4",
            "net_dim = [250, 250, 250]", "dropout_rate = 0.0", "max_n_epochs = 2", "n_earlystop_epochs = 5",
            "optimizer_type = adam", "starter_learning_rate = 0.001", "lr_decay = 1.0", "lr_updating_steps = 10000",
            "l2 = 0.0", "root_folder = %s" % data, "exp_folder = %s" % (tmp_path / "logs" / name), "device = /gpu:0",
            "audio_feat_mean = %s" % (tmp_path / "mean.npy"), "audio_feat_std = %s" % (tmp_path / "std.npy"), ""]))
        return str(cfg)

    plain = training.unpack_batch
    calls = {'n': 0, 'parked': None}

    def parking(*a, **kw):
        calls['n'] += 1
        if calls['n'] == 3:
            # the trainer launches on a high-priority stream: the neighbour sits on a normal one; it leaves by itself
            calls['parked'] = _park(16, 10000, priority=0)
        return plain(*a, **kw)
    monkeypatch.setattr(training, 'unpack_batch', parking)
    try:
        model = training.train(config("guarded"))
        text = capsys.readouterr()
        assert model.coop_fallbacks in (1, 2) and ops.coop_fallbacks() == model.coop_fallbacks      # (2: the 8-way kernels starved too)
        assert text.err.count('falling back to the cooperative kernels that tolerate neighbours') == 1
        assert model.global_step == 10                    # 5 batches x 2 epochs: no step lost, none counted twice
        assert '+---- Done training: epoch limit reached ----+' in text.out
        got = model.variables.flat.clone()
    finally:
        ops.coop_fall_back_reset()
        if calls['parked'] is not None:
            calls['parked'][0].fill_(1)
        torch.cuda.synchronize()
    monkeypatch.setattr(training, 'unpack_batch', plain)
    monkeypatch.setenv('AVSI_REC_COOP', '0')
    ref = training.train(config("stationary"))
    assert ref.coop_fallbacks == 0 and ref.global_step == 10
    # same batches in the same order (AVSI_SHUFFLE_SEED): the two runs differ by the kernels they ran on only
    assert (got - ref.variables.flat).abs().max().item() < 1e-4
