"""The cooperative recurrent kernels next to other residents of the chip.  Every workgroup of a cooperative launch
must be resident at once (its peers spin on it); under data parallelism RCCL kernels of the bucketed gradient
all-reduce run CONCURRENTLY with the BPTT of the layers below and hold compute units.  A persistent kernel that parks
whole CUs on a second stream stands in for them here (one GPU, no RCCL): with the CU budget that
``parallel.collectives_share_the_gpu()`` selects (256 - 32), training at 512 and 1024 utterances -- chip-filling
cooperative grids -- finishes without a bounded-spin failure and with bit-identical results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(B, seed):
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    N, T = 1920, 10
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    wav = torch.round(torch.randn(B, N, generator=g, device='cuda') * 3000)
    masks = torch.ones(B, T, 257, device='cuda')
    masks[:, 3:6] = 0
    video = torch.randn(B, T, 136, generator=g, device='cuda')
    mean, std = torch.full((257,), 6.0, device='cuda'), torch.full((257,), 2.0, device='cuda')
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
               starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    m = models.StackedBLSTMModel(np.full(B, T), wav, masks, mean, std, 0.0, cfg, video_features=video, input='av', seed=3)
    return m, dict(sequence_lengths=np.full(B, T), target_sources=wav, masks=masks, video_features=video)


def _two_steps(m, feed):
    from avsi_amd import ops
    losses = []
    for _ in range(2):
        m.feed(**feed)
        losses.append(float(m.loss_func))
        m.train_op
    ops.coop_check()
    return losses, m.variables.flat.clone(), m.prediction.clone()


@pytest.mark.parametrize("B,occupied", [(512, 32), (1024, 32), (512, 16)])
def test_training_beside_parked_compute_units(B, occupied):
    from avsi_amd import ops
    assert ops.coop_split(B) and ops.coop_split(B, backward=True)      # these sizes take the cooperative kernels
    ops.set_coop_cu_budget(None)
    m, feed = _model(B, 1)
    ref_losses, ref_vars, ref_pred = _two_steps(m, feed)

    release = torch.zeros(1, dtype=torch.int32, device='cuda')
    side = torch.cuda.Stream()
    ops.set_coop_cu_budget(256 - ops.COOP_CU_RESERVE)
    splits = {back: ops.coop_split(B, back) for back in (False, True)}      # what the budgeted run uses
    try:
        m2, feed2 = _model(B, 1)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            ops.occupy_cus(occupied, release, max_ms=20000)
        losses, vars_, pred = _two_steps(m2, feed2)       # runs while `occupied` CUs are taken
        parked_during = not side.query()                   # the parked kernel is still there: the overlap was real
    finally:
        release.fill_(1)
        torch.cuda.synchronize()
        ops.set_coop_cu_budget(None)
    assert parked_during
    same_kernels = all(ops.coop_split(B, back) == split for back, split in splits.items())
    if same_kernels:
        # the budget only cut the batch into more launches of the same kernels: nothing may change
        assert losses == ref_losses
        assert torch.equal(vars_, ref_vars) and torch.equal(pred, ref_pred)
    else:
        # the budget also moved the policy to a coarser split (another reduction order): equal up to rounding
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
        assert (vars_ - ref_vars).abs().max().item() < 1e-5 and (pred - ref_pred).abs().max().item() < 1e-4


def test_one_tile_that_does_not_fit_is_refused():
    import avsi_amd
    from avsi_amd import ops
    ops.set_coop_cu_budget(8)
    try:
        T, Bp = 4, 32
        xproj = torch.zeros(T, Bp, 2048, device='cuda')
        whp = torch.zeros(2 * 262144, device='cuda')
        hout = torch.zeros(T, Bp, 512, device='cuda')
        with pytest.raises(avsi_amd._lib.AvsiError):
            ops.blstm_rec_fwd(xproj, whp, hout, None, split=8)      # 16 workgroups per tile > 8 CUs
    finally:
        ops.set_coop_cu_budget(None)
