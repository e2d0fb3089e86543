"""Training half of the hot path on the GPU: BPTT kernel + split-K weight-gradient GEMMs + column
sums + TF-Adam, against the CPU oracle (manual BPTT in float64, itself pinned to torch.autograd in
tests/test_oracle_blstm.py)."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as OF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import avsi_amd
    from avsi_amd import models, ops, blstm_layout
    return models, ops, blstm_layout


def _config(**kw):
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=48000, net_dim=[250, 250, 250],
               optimizer_type='adam', starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0,
               batch_size=8, l2=0.0)
    cfg.update(kw)
    return cfg


def _inputs(B, N, seed):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    gap = max(1, T // 4)
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, T - gap)
        masks[b, s:s + gap] = 0
    spec = OF.get_spectrogram(OF.get_stft(wav, window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(spec))
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    return wav, masks, mean.astype(np.float32), std.astype(np.float32), video, T


def _rand_biases(params, seed):
    rng = np.random.default_rng(seed)
    for layer in params['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = rng.normal(0, 0.1, size=layer[d]['bias'].shape).astype(np.float32)
    params['proj']['biases'] = rng.normal(0, 0.1, size=params['proj']['biases'].shape).astype(np.float32)
    return params


def _flat_grads(layout, grads):
    return layout.flatten_oracle_params({'layers': grads['layers'], 'proj': grads['proj']}).astype(np.float64)


def test_helper_kernels(mods):
    models, ops, bl = mods
    rng = np.random.default_rng(0)
    # column sums
    x = rng.normal(size=(5000, 300)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    out = torch.empty(257, device='cuda')
    ops.colsum(xt, out, n=257)
    np.testing.assert_allclose(out.cpu().numpy(), x[:, :257].astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)
    # row re-layout with scale and zero fill
    B, T, C, Bp, Cp = 3, 5, 7, 32, 8
    src = rng.normal(size=(B, T, C)).astype(np.float32)
    scale = rng.uniform(size=(T, Bp)).astype(np.float32)
    dst = torch.full((T, Bp, Cp), 9.0, device='cuda')
    ops.relayout_rows(torch.from_numpy(src).cuda(), dst, B, T, C, Cp, (T * C, C), (Cp, Bp * Cp),
                      row_scale=torch.from_numpy(scale).cuda(), scale_strides=(1, Bp))
    got = dst.cpu().numpy()
    np.testing.assert_allclose(got[:, :B, :C], src.transpose(1, 0, 2) * scale[:, :B, None], rtol=1e-6)
    assert np.all(got[:, :B, C:] == 0) and np.all(got[:, B:] == 9.0)
    # split-K transposed GEMM (weight-gradient shape)
    A = rng.normal(size=(3000, 264)).astype(np.float32)
    Z = rng.normal(size=(3000, 512)).astype(np.float32)
    o = torch.empty(264, 512, device='cuda')
    ops.gemm_splitk(torch.from_numpy(A).cuda(), torch.from_numpy(Z).cuda(), o, trans_a=True, splits=7)
    ref = A.astype(np.float64).T @ Z
    assert np.abs(o.cpu().numpy() - ref).max() < 2e-3


def test_adam_kernel_matches_tf_form(mods):
    models, ops, bl = mods
    rng = np.random.default_rng(1)
    n = 10007
    p0 = rng.normal(size=n)
    p, m, v = p0.copy(), np.zeros(n), np.zeros(n)
    tp = torch.from_numpy(p0.astype(np.float32)).cuda()
    tm, tv = torch.zeros_like(tp), torch.zeros_like(tp)
    for step in range(1, 6):
        g = rng.normal(size=n) * 10.0 ** rng.integers(-4, 1)
        O.adam_tf_step(p, g, m, v, step, lr=1e-3)
        ops.adam_tf(tp, torch.from_numpy(g.astype(np.float32)).cuda(), tm, tv, step, 1e-3)
    np.testing.assert_allclose(tp.cpu().numpy(), p, rtol=0, atol=2e-6)
    np.testing.assert_allclose(tm.cpu().numpy(), m, rtol=1e-5, atol=1e-7)   # fp32 slots vs float64 oracle


@pytest.mark.parametrize("momentum", [None, 0.9])
def test_sgd_momentum_kernel_matches_tf_form(mods, momentum):
    """avsi_sgd_momentum_f32 against the numpy restatement of tf.train.GradientDescentOptimizer / MomentumOptimizer
    (models.py:170-176) over five steps with a rate that changes between steps (the decayed rate multiplies the WHOLE
    accumulator in TF's form), a gradient scale, an l2 term, an odd length, and a void step in the middle."""
    models, ops, bl = mods
    rng = np.random.default_rng(2)
    n = 10007
    p0 = rng.normal(size=n)
    p, acc = p0.copy(), np.zeros(n)
    tp = torch.from_numpy(p0.astype(np.float32)).cuda()
    ta = torch.zeros_like(tp) if momentum is not None else None
    ok, void = torch.zeros(2, device='cuda'), torch.tensor([0.0, 1.0], device='cuda')
    for step in range(5):
        g = rng.normal(size=n) * 10.0 ** rng.integers(-3, 1)
        lr = O.exponential_decay(0.05, step, 2, 0.5)
        tg = torch.from_numpy(g.astype(np.float32)).cuda()
        if step == 2:
            before = tp.clone()
            ops.sgd_momentum(tp, tg, ta, lr, momentum=momentum or 0.0, grad_scale=0.5, l2=1e-3, skip=void)
            assert torch.equal(tp, before)                      # guarded: nothing moved
            continue
        gg = 0.5 * g + 1e-3 * p
        if momentum is None:
            O.sgd_tf_step(p, gg, lr)
        else:
            O.momentum_tf_step(p, gg, acc, lr, momentum)
        ops.sgd_momentum(tp, tg, ta, lr, momentum=momentum or 0.0, grad_scale=0.5, l2=1e-3, skip=ok)
    np.testing.assert_allclose(tp.cpu().numpy(), p, rtol=0, atol=3e-6)
    if momentum is not None:
        np.testing.assert_allclose(ta.cpu().numpy(), acc, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("optimizer", ["sgd", "momentum"])
def test_three_sgd_and_momentum_steps_track_oracle(mods, optimizer):
    """The reference's other optimizer_type choices through train_op (models.py:165-176): staircase-decayed rate
    (rate 1.0 -- the gradients of a mean over B*T*F are small --, lr_updating_steps = 2, decay 0.5: the third step runs at half
    the rate), three steps against the oracle's forward / backward and the numpy update.  Not scale-free like Adam: the variables track the oracle to the gradients' accuracy."""
    models, ops, bl = mods
    B, N = 4, 2880
    wav, masks, mean, std, video, T = _inputs(B, N, 51)
    p = _rand_biases(O.init_params(11, 257), 12)
    seq_len = np.full(B, T)
    cfg = _config(audio_len=N, optimizer_type=optimizer, starter_learning_rate=1.0, lr_updating_steps=2, lr_decay=0.5)
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, cfg, input='a')
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    p64 = O.cast_params(p, np.float64)
    init = m.layout.flatten_oracle_params(p64).astype(np.float64)
    flat = [v for _, v in O.flatten_params(p64)]
    accs = [np.zeros_like(v) for v in flat]
    rates = []
    for step in range(3):
        fwd = O.model_forward(wav, masks, mean, std, seq_len, p64, keep=True)
        g = O.model_backward(fwd, masks.astype(np.float64), seq_len)
        lr = O.exponential_decay(1.0, step, 2, 0.5)
        for (_, gv), pv, av in zip(O.flatten_params(g), flat, accs):
            if optimizer == 'sgd':
                O.sgd_tf_step(pv, gv, lr)
            else:
                O.momentum_tf_step(pv, gv, av, lr, 0.9)
        m.feed(sequence_lengths=seq_len, target_sources=wav, masks=masks)
        rates.append(m.learning_rate)
        assert m.train_op is None and m.global_step == step + 1
    assert rates == [1.0, 1.0, 0.5]
    ref_flat = m.layout.flatten_oracle_params(p64).astype(np.float64)
    got_flat = m.variables.flat.cpu().numpy().astype(np.float64)
    moved = np.abs(ref_flat - init).max()
    assert moved > 1e-3
    assert np.abs(got_flat - ref_flat).max() < 2e-3 * moved + 1e-7
    if optimizer == 'momentum':
        acc_ref = m.layout.flatten_oracle_params(_as_params(p64, accs)).astype(np.float64)
        acc_got = m.variables.adam_m.cpu().numpy().astype(np.float64)
        assert np.abs(acc_got - acc_ref).max() < 2e-3 * np.abs(acc_ref).max()


def _as_params(like, arrays):
    """The oracle's parameter structure with `arrays` (in O.flatten_params order) in place of the values."""
    import copy
    out = copy.deepcopy(like)
    for (_, dst), src in zip(O.flatten_params(out), arrays):
        dst[...] = src
    return out


@pytest.mark.parametrize("input_type,B,N,Dv", [('a', 5, 3840, 136), ('av', 3, 2880, 136), ('a', 34, 1920, 136),
                                              ('av', 3, 2880, 15)])
def test_gradients_match_oracle(mods, input_type, B, N, Dv):
    """Dv = 15 makes the network input 272 wide: no padded column to spare, so layer 0 takes its bias gradient
    from a column sum instead of the constant-1 column of the weight-gradient GEMM (ParamLayout.ones_col)."""
    models, ops, bl = mods
    wav, masks, mean, std, video, T = _inputs(B, N, 30 + B)
    video = video[:, :, :Dv]
    D = {'a': 257, 'av': 257 + Dv}[input_type]
    p = _rand_biases(O.init_params(9, D), 10)
    seq_len = np.full(B, T)
    seq_len[0] = T - 2
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(audio_len=N, video_feat_dim=Dv),
                                 video_features=video, input=input_type)
    assert (m.layout.ones_col[0] < 0) == (D % 16 == 0) and m.layout.ones_col[1] == 250
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    got = m.gradients.cpu().numpy().astype(np.float64)
    fwd = O.model_forward(wav, masks, mean, std, seq_len, p, video=video, input_type=input_type, keep=True)
    ref = _flat_grads(m.layout, O.model_backward(fwd, masks.astype(np.float64), seq_len))
    assert got.shape == ref.shape
    # per-variable relative error (gradients span orders of magnitude across variables)
    for name, shape, off in m.layout.ref_entries:
        n = int(np.prod(shape))
        g, r = got[off:off + n], ref[off:off + n]
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 2e-3 * scale + 1e-9, (name, np.abs(g - r).max(), scale)
        assert np.sqrt(np.mean((g - r) ** 2)) <= 2e-4 * scale + 1e-10, name


def test_three_adam_steps_track_oracle(mods):
    models, ops, bl = mods
    B, N = 4, 2880
    wav, masks, mean, std, video, T = _inputs(B, N, 50)
    p = _rand_biases(O.init_params(11, 257), 12)
    seq_len = np.full(B, T)
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(audio_len=N), input='a')
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    p64 = O.cast_params(p, np.float64)
    flat = [v for _, v in O.flatten_params(p64)]
    ms = [np.zeros_like(v) for v in flat]
    vs = [np.zeros_like(v) for v in flat]
    losses_ref, losses_got = [], []
    for step in range(1, 4):
        fwd = O.model_forward(wav, masks, mean, std, seq_len, p64, keep=True)
        g = O.model_backward(fwd, masks.astype(np.float64), seq_len)
        losses_ref.append(fwd['loss'])
        for (_, gv), pv, mv, vv in zip(O.flatten_params(g), flat, ms, vs):
            O.adam_tf_step(pv, gv, mv, vv, step, lr=1e-3)
        m.feed(sequence_lengths=seq_len, target_sources=wav, masks=masks)
        losses_got.append(float(m.loss))
        assert m.train_op is None
        assert m.global_step == step
    np.testing.assert_allclose(losses_got, losses_ref, rtol=5e-4)
    assert losses_ref[2] < losses_ref[0]
    ref_flat = m.layout.flatten_oracle_params(p64).astype(np.float64)
    got_flat = m.variables.flat.cpu().numpy().astype(np.float64)
    # Adam normalises the step to ~lr, so parameters agree to a small fraction of 3 * lr
    assert np.abs(got_flat - ref_flat).max() < 5e-4
    assert np.sqrt(np.mean((got_flat - ref_flat) ** 2)) < 2e-5


@pytest.mark.parametrize("net_dim", [(96, 250, 40), (256, 17), (8,)])
def test_unequal_layer_widths_match_oracle(mods, net_dim):
    """The reference takes any num_units per layer (models.py:95-99,107: one LSTMCell per entry of net_dim, the next layer reads
    2 x the width below).  Every layer is packed to 256 units of its own (ParamLayout.Hs); prediction, loss and every variable's
    gradient against the oracle, which builds the same stack from the same net_dim."""
    models, ops, bl = mods
    B, N = 5, 2880
    wav, masks, mean, std, video, T = _inputs(B, N, 77)
    p = _rand_biases(O.init_params(13, 257, net_dim=net_dim), 14)
    seq_len = np.full(B, T)
    seq_len[1] = T - 3
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(audio_len=N, net_dim=list(net_dim)), input='a')
    assert m.layout.Hs == tuple(net_dim) and m.layout.H == net_dim[-1]
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    fwd = O.model_forward(wav, masks, mean, std, seq_len, p, input_type='a', keep=True)
    np.testing.assert_allclose(m.prediction.cpu().numpy(), fwd['prediction'], rtol=0, atol=2e-4)
    assert m.rnn_outputs.shape == (B, T, 2 * net_dim[-1])
    got = m.gradients.cpu().numpy().astype(np.float64)
    ref = _flat_grads(m.layout, O.model_backward(fwd, masks.astype(np.float64), seq_len))
    assert got.shape == ref.shape
    for name, shape, off in m.layout.ref_entries:
        n = int(np.prod(shape))
        g, r = got[off:off + n], ref[off:off + n]
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 2e-3 * scale + 1e-9, (name, np.abs(g - r).max(), scale)
    # one Adam step moves only real weights: the packed padding stays exactly zero
    m.train_op
    torch.cuda.synchronize()
    packed = m.variables.packed.cpu().numpy()
    keep = np.zeros(packed.size, dtype=bool)
    keep[np.flatnonzero(m.layout.pack_index < m.layout.ref_size)] = True
    assert not packed[~keep].any()


def test_loss_from_the_waveform_gives_the_same_step(monkeypatch):
    """AVSI_LOSS_FROM_WAV=1 (opt-in): the front end stores the masked features only and the loss recomputes the normalised
    target from the waveform inside the front-end kernel.  Same prediction bits (the features are the same), same losses to
    summation order, gradients equal to 1e-6 of their scale (a sign may flip where p and t agree to the last bit), and
    `target_spec_norm` is still there when asked for."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import models
    rng = np.random.default_rng(11)
    B, N = 6, 9600
    T = N // 192
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    masks = np.ones((B, T, 257), dtype=np.float32)
    masks[:, 10:14] = 0
    mean = rng.normal(5, 1, 257).astype(np.float32)
    std = (1 + rng.random(257)).astype(np.float32)
    cfg = dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
               starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('AVSI_LOSS_FROM_WAV', flag)
        m = models.StackedBLSTMModel(np.full(B, T), wav, masks, mean, std, 0.0, cfg, input='a', seed=3)
        assert m._loss_from_wav() is (flag == '1')
        pred = m.prediction.clone()
        losses = torch.stack([m.loss_func, m.loss_hole, m.loss_valid]).clone()
        assert ('target_spec_norm' in m._cache) is (flag == '0')
        grads = m.gradients.clone()
        tgt = m.target_spec_norm.clone()
        res.append((pred, losses, grads, tgt))
    assert torch.equal(res[0][0], res[1][0])
    # (the target on demand comes from the generic instantiation of the kernel: last-bit differences against the model's own)
    np.testing.assert_allclose(res[1][3].cpu().numpy(), res[0][3].cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(res[1][1].cpu().numpy(), res[0][1].cpu().numpy(), rtol=2e-5)
    scale = float(res[0][2].abs().max())
    assert float((res[0][2] - res[1][2]).abs().max()) < 1e-5 * scale
