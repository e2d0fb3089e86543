"""Stacked-BLSTM forward of the gfx950 path (hoisted MFMA GEMM + recurrent kernel + projection
+ L1 loss, all through the C ABI) against the CPU oracle.  Tolerances: the path computes in
fp32; BASELINE.json asks <= 1e-3 RMS on the reconstructed log-mel, we hold the normalised
log-spectrum prediction itself to 1e-4 RMS vs the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as OF

pytestmark = pytest.mark.gpu


def _rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - b) ** 2)))


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


def _inputs(B, N, seed, gap=33):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    gap = min(gap, max(1, T // 3))
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, max(1, T - gap))
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


@pytest.mark.parametrize("rows_per_wg", [32, 64])
def test_recurrent_kernel_single_layer(mods, rows_per_wg):
    """One layer, both directions, vs the oracle's explicit per-step loop."""
    models, ops, bl = mods
    H, D, T, B, Bp = 250, 40, 13, 37, 64
    lay = bl.ParamLayout(D, (H,), 257)
    p = _rand_biases(O.init_params(3, D, (H,), 257), 4)
    flat = lay.flatten_oracle_params(p)
    packed = torch.from_numpy(np.concatenate([flat, [0]]).astype(np.float32)[lay.pack_index]).cuda()
    rng = np.random.default_rng(5)
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    xp = torch.zeros(T, Bp, lay.kp[0], device='cuda')
    xp[:, :B, :D] = torch.from_numpy(x).cuda().transpose(0, 1)
    xproj = ops.gemm(xp.view(T * Bp, -1), lay.packed_view(packed, 'wx0'), bias=lay.packed_view(packed, 'b0'))
    hout = torch.empty(T, Bp, 512, device='cuda')
    resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
    ops.blstm_rec_fwd(xproj.view(T, Bp, 2048), lay.packed_view(packed, 'wh0'), hout, resv, rows_per_wg)
    got = hout.cpu().numpy()
    p64 = O.cast_params(p, np.float64)
    fw, cf = O.lstm_direction(x.astype(np.float64), p64['layers'][0]['fw']['kernel'], p64['layers'][0]['fw']['bias'],
                              False, True)
    bw, cb = O.lstm_direction(x.astype(np.float64), p64['layers'][0]['bw']['kernel'], p64['layers'][0]['bw']['bias'],
                              True, True)
    np.testing.assert_allclose(got[:, :B, :H].transpose(1, 0, 2), fw, atol=2e-5)
    np.testing.assert_allclose(got[:, :B, 256:256 + H].transpose(1, 0, 2), bw, atol=2e-5)
    assert np.all(got[:, :, H:256] == 0) and np.all(got[:, :, 256 + H:] == 0)   # padded units stay 0
    # reserve: activated gates and cell state at every step
    r = resv.cpu().numpy()
    for d, cache in ((0, cf), (1, cb)):
        for (t, i, j, f, o, c_new, _, _) in cache:
            for gi, ref in enumerate((i, j, f, o, c_new)):
                np.testing.assert_allclose(r[t, :B, d, gi, :H], ref, atol=3e-5)


@pytest.mark.parametrize("rows_per_wg,B", [(64, 70), (32, 37)])
def test_recurrent_kernel_inference_form_full_length(mods, rows_per_wg, B):
    """The kernel the benchmark times -- blstm_rec_fwd_pp_kernel<false>: 64 utterances per workgroup, NO reserve
    (inference), all T = 250 steps -- against the oracle's per-step loop (reference models.py:106-115).  B = 70
    gives one full and one partial 64-row workgroup per direction."""
    models, ops, bl = mods
    H, D, T = 250, 257, 250
    Bp = -(-B // 32) * 32
    lay = bl.ParamLayout(D, (H,), 257)
    p = _rand_biases(O.init_params(13, D, (H,), 257), 14)
    flat = lay.flatten_oracle_params(p)
    packed = torch.from_numpy(np.concatenate([flat, [0]]).astype(np.float32)[lay.pack_index]).cuda()
    rng = np.random.default_rng(15)
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    xp = torch.zeros(T, Bp, lay.kp[0], device='cuda')
    xp[:, :B, :D] = torch.from_numpy(x).cuda().transpose(0, 1)
    xproj = ops.gemm(xp.view(T * Bp, -1), lay.packed_view(packed, 'wx0'), bias=lay.packed_view(packed, 'b0'))
    hout = torch.full((T, Bp, 512), 9.0, device='cuda')
    ops.blstm_rec_fwd(xproj.view(T, Bp, 2048), lay.packed_view(packed, 'wh0'), hout, None, rows_per_wg)
    got = hout.cpu().numpy()
    p64 = O.cast_params(p, np.float64)
    fw = O.lstm_direction(x.astype(np.float64), p64['layers'][0]['fw']['kernel'], p64['layers'][0]['fw']['bias'], False)
    bw = O.lstm_direction(x.astype(np.float64), p64['layers'][0]['bw']['kernel'], p64['layers'][0]['bw']['bias'], True)
    np.testing.assert_allclose(got[:, :B, :H].transpose(1, 0, 2), fw, atol=3e-5)
    np.testing.assert_allclose(got[:, :B, 256:256 + H].transpose(1, 0, 2), bw, atol=3e-5)
    assert np.all(got[:, :, H:256] == 0) and np.all(got[:, :, 256 + H:] == 0)


def test_model_forward_bench_kernels_match_oracle(mods):
    """Whole model through the kernels of the B = 8192 benchmark step at a size the oracle finishes in seconds:
    rows_per_wg = 64 forces blstm_rec_fwd_pp_kernel<false> (is_training=False: no reserve), and B = 64, T = 250
    gives the layer GEMMs M = 16000 rows -> 125 row blocks x 8 = 1000 >= 512 tiles -> the 128 x 256 tile."""
    models, ops, bl = mods
    B, N = 64, 48000
    wav, masks, mean, std, video, T = _inputs(B, N, 77)
    p = _rand_biases(O.init_params(17, 257), 18)
    seq_len = np.full(B, T)
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(audio_len=N, rows_per_wg=64), input='a',
                                 is_training=False)
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    ref = O.model_forward(wav, masks, mean, std, seq_len, p)
    pred = m.prediction.cpu().numpy()
    assert m._cache['reserve'] == [None, None, None]
    assert _rms(pred, ref['prediction']) < 1e-4
    assert np.abs(pred - ref['prediction']).max() < 2e-3
    lm_ref = OF.logmel_of_prediction(ref['prediction'], mean, std)
    lm_got = OF.logmel_of_prediction(pred.astype(np.float64), mean, std)
    assert _rms(lm_got, lm_ref) < 1e-3
    assert float(m.loss_func) == pytest.approx(ref['loss_func'], rel=2e-4)


@pytest.mark.parametrize("input_type,B,N", [('a', 33, 48000), ('av', 3, 9600), ('v', 2, 9600), ('a', 33, 3840)])
def test_model_forward_matches_oracle(mods, input_type, B, N):
    models, ops, bl = mods
    wav, masks, mean, std, video, T = _inputs(B, N, 10 + B)
    D = {'a': 257, 'v': 136, 'av': 393}[input_type]
    p = _rand_biases(O.init_params(7, D), 8)
    seq_len = np.full(B, T)
    seq_len[-1] = max(1, T - 3)              # ragged: last utterance shorter
    cfg = _config(audio_len=N)
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, cfg, video_features=video, input=input_type)
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    ref = O.model_forward(wav, masks, mean, std, seq_len, p, video=video, input_type=input_type)
    pred = m.prediction.cpu().numpy()
    assert pred.shape == ref['prediction'].shape
    assert _rms(m.target_spec_norm.cpu().numpy(), ref['target_spec_norm']) < 1e-4
    assert _rms(m.net_inputs.cpu().numpy(), ref['net_inputs']) < 1e-4
    assert _rms(pred, ref['prediction']) < 1e-4
    assert np.abs(pred - ref['prediction']).max() < 2e-3
    assert np.all(pred[-1, seq_len[-1]:] == 0)                   # sequence_mask
    # reconstructed log-mel metric of BASELINE.json (<= 1e-3 RMS)
    lm_ref = OF.logmel_of_prediction(ref['prediction'], mean, std)
    lm_got = OF.logmel_of_prediction(pred.astype(np.float64), mean, std)
    assert _rms(lm_got, lm_ref) < 1e-3
    # losses
    assert float(m.loss_func) == pytest.approx(ref['loss_func'], rel=2e-4)
    assert float(m.loss_hole) == pytest.approx(ref['loss_hole'], rel=2e-4)
    assert float(m.loss_valid) == pytest.approx(ref['loss_valid'], rel=2e-4)
    assert float(m.loss) == pytest.approx(ref['loss'], rel=2e-4)


def test_l1_loss_kernel_and_gradient(mods):
    models, ops, bl = mods
    rng = np.random.default_rng(11)
    for n_shape in [(3, 50, 257), (1, 1, 5), (2, 7, 258)]:
        t = rng.normal(size=n_shape).astype(np.float32)
        p = rng.normal(size=n_shape).astype(np.float32)
        p.flat[0] = t.flat[0]                                       # sign(0) = 0
        m = (rng.uniform(size=n_shape) > 0.3).astype(np.float32)
        out3, g = ops.l1_loss(torch.from_numpy(t).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(m).cuda(),
                              want_grad=True)
        ref = O.losses(t.astype(np.float64), p.astype(np.float64), m.astype(np.float64))
        got = out3.cpu().numpy()
        assert got[0] == pytest.approx(ref['loss_func'], rel=1e-5)
        assert got[1] == pytest.approx(ref['loss_hole'], rel=1e-5)
        assert got[2] == pytest.approx(ref['loss_valid'], rel=1e-5)
        np.testing.assert_allclose(g.cpu().numpy(), np.sign(p - t) / t.size, rtol=1e-6)


def test_feed_reuses_model_and_variables(mods):
    models, ops, bl = mods
    wav, masks, mean, std, video, T = _inputs(2, 3840, 20)
    cfg = _config(audio_len=3840)
    m = models.StackedBLSTMModel(np.full(2, T), wav, masks, mean, std, 0.0, cfg, input='a')
    p1 = m.prediction.clone()
    wav2, masks2, _, _, _, _ = _inputs(2, 3840, 21)
    m.feed(sequence_lengths=np.full(2, T), target_sources=wav2, masks=masks2)
    p2 = m.prediction
    assert not torch.allclose(p1, p2)
    m.feed(sequence_lengths=np.full(2, T), target_sources=wav, masks=masks)
    assert torch.equal(m.prediction, p1)                            # deterministic, bitwise


@pytest.mark.parametrize("Bp,split,save", [(32, 8, False), (64, 8, True), (512, 8, False), (1024, 4, True), (96, 4, False),
                                           (1056, 4, False), (544, 8, True),    # these two: more than one resident-sized launch
                                           (32, 16, False), (256, 16, True), (288, 16, False),   # finer splits: 16 / 8 units per
                                           (64, 32, True), (160, 32, False),                      # workgroup (288, 160: two launches)
                                           (32, 64, False), (64, 64, True), (160, 64, False),     # 32-way on 16-row halves (160: three launches)
                                           # column-split kernel (avsi_blstm_rec_fwd_cs_f32): 16 / 32 utterances per group
                                           (32, -32, False), (64, -16, True), (512, -16, False), (1024, -32, True),
                                           (1088, -32, False), (544, -16, True),                  # these two: two launches
                                           (3104, -32, False)])                                    # four (the policy's range ends at 3584)
def test_cooperative_small_batch_kernel_matches_batch_stationary(Bp, split, save, monkeypatch):
    """avsi_blstm_rec_fwd_coop_f32 (weights resident in registers, h exchanged through hout with a
    per-step counter) against avsi_blstm_rec_fwd_f32 on the same operands: same maths, different
    summation order of the 256-long reduction."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 37
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp + abs(split))
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for sp in (0, split):
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda') if save else None
        ops.blstm_rec_fwd(xproj, whp, hout, resv, split=sp)
        outs.append((hout, resv))
    ops.coop_check()
    np.testing.assert_allclose(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy(), rtol=0, atol=2e-5)
    if save:
        np.testing.assert_allclose(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), rtol=0, atol=5e-5)
    for name in ('AVSI_REC_CS', 'AVSI_COOP_CUS', 'AVSI_REC_COOP'):      # the default policy, whatever the caller's switches
        monkeypatch.delenv(name, raising=False)
    assert ops.coop_split(32) == 64 and ops.coop_split(128) == 32 and ops.coop_split(32, backward=True) == 32 and ops.coop_split(256) == -16
    assert ops.coop_split(256, backward=True) == 16 and ops.coop_split(512, backward=True) == 8
    assert ops.coop_split(512) == -16 and ops.coop_split(1024) == -32 and ops.coop_split(2048) == -32 and ops.coop_split(4096) == 0


@pytest.mark.parametrize("Bp,save", [(544, False), (640, True), (768, False), (1088, True), (1536, False), (2112, False),
                                     (4160, False), (4672, True)])          # these two: 4096 batch-stationary + remainder
def test_forward_recurrence_in_pieces_matches_batch_stationary(Bp, save, monkeypatch):
    """The automatic choice cuts these batches into a large part and a remainder on the kernel of its size
    (ops.rec_fwd_parts, avsi_blstm_rec_fwd_{cs,coop}_rows_f32): same result as the batch-stationary kernel, every
    utterance written exactly once, counters and status word left at zero."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    for name in ('AVSI_REC_CS', 'AVSI_COOP_CUS', 'AVSI_REC_COOP', 'AVSI_COOP_SPLIT_FWD'):
        monkeypatch.delenv(name, raising=False)
    assert len(ops.rec_fwd_parts(Bp)) > 1
    T = 19
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for auto in (False, True):
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda') if save else None
        if auto:
            ops.blstm_rec_fwd(xproj, whp, hout, resv)
        else:
            ops.blstm_rec_fwd(xproj, whp, hout, resv, split=0)
        outs.append((hout, resv))
    ops.coop_check()
    np.testing.assert_allclose(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy(), rtol=0, atol=2e-5)
    if save:
        np.testing.assert_allclose(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy(), rtol=0, atol=5e-5)
    from avsi_amd import _lib
    ws = ops._COOP_WS[(torch.cuda.current_device(), _lib.stream_ptr().value)]
    assert int(ws[: 64 * (1 + 4 * Bp // 32)].abs().sum()) == 0           # status word and every counter back at zero


def test_batch_stationary_kernel_choice_above_4096():
    """avsi_blstm_rec_fwd_f32 with rows_per_wg = 0: beyond 4096 utterances (where 32-row workgroups would need a second
    round of the chip) the 64-row ping-pong kernel runs -- same result as forcing either form; at 4096 the 32-row one."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T, Bp = 5, 4160
    g = torch.Generator(device='cuda')
    g.manual_seed(11)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = {}
    for rows in (0, 32, 64):
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        ops.blstm_rec_fwd(xproj, whp, hout, None, rows_per_wg=rows, split=0)
        outs[rows] = hout
    assert torch.equal(outs[0], outs[64])                    # the automatic choice IS the 64-row kernel here
    np.testing.assert_allclose(outs[0].cpu().numpy(), outs[32].cpu().numpy(), rtol=0, atol=2e-5)
    Bp = 4096
    xproj = xproj[:, :Bp].contiguous()
    for rows in (0, 32):
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        ops.blstm_rec_fwd(xproj, whp, hout, None, rows_per_wg=rows, split=0)
        outs[rows] = hout
    assert torch.equal(outs[0], outs[32])


@pytest.mark.parametrize("Bp,split", [(32, 8), (256, 8), (1024, 4), (64, 4), (1088, 4), (3104, 4), (32, 16), (256, 16), (288, 16), (32, 32), (128, 32), (160, 32)])
def test_cooperative_bptt_matches_batch_stationary(Bp, split):
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 29
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp * 3 + split)
    dh = torch.randn(T, Bp, 512, generator=g, device='cuda')
    resv = torch.rand(T, Bp, 2, 5, 256, generator=g, device='cuda') * 0.9 + 0.05      # activated gates in (0, 1)
    resv[:, :, :, 1] = resv[:, :, :, 1] * 2 - 1                                        # j in (-1, 1)
    resv[:, :, :, 4] = torch.randn(T, Bp, 2, 256, generator=g, device='cuda')          # c
    whbt = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for sp in (0, split):
        dz = torch.full((T, Bp, 2048), 7.0, device='cuda')
        ops.blstm_rec_bwd(dh, resv, whbt, dz, split=sp)
        outs.append(dz.cpu().numpy())
    ops.coop_check()
    scale = np.abs(outs[0]).max()
    np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=2e-6 * scale)


def test_summaries_hold_the_tensorboard_tensors(mods):
    models, ops, bl = mods
    wav, masks, mean, std, video, T = _inputs(3, 3840, 5)
    m = models.StackedBLSTMModel(np.full(3, T), wav, masks, mean, std, 0.0, _config(audio_len=3840), input='a', is_training=False)
    s = m.summaries
    assert set(s) == {'Target_spectrogram', 'Enhanced_spectrogram', 'Mask', 'Target_audio', 'Enhanced_audio'}
    assert tuple(s['Target_spectrogram'].shape) == (3, 257, T, 1) and tuple(s['Mask'].shape) == (3, 257, T, 1)
    np.testing.assert_array_equal(s['Mask'][0, :, :, 0].cpu().numpy()[::-1], masks[0].T)       # flipped upside down
    assert abs(float(s['Enhanced_audio'].abs().amax(dim=1)[0]) - 1.0) < 1e-6 and tuple(s['Target_audio'].shape) == (3, 3840)
