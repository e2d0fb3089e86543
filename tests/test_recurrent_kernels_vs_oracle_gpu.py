"""Every recurrent kernel family against the CPU oracle DIRECTLY (oracle/blstm.py: the explicit per-step loop of
reference models.py:106-115 in float64, and its manual BPTT), not through another HIP kernel: the batch-stationary
kernels, the reduction-split cooperative kernels (4, 8, 16, 32 workgroups per tile, 32 per 16-row half-tile), the column-split kernel (16 / 32
utterances per group), forward and BPTT -- the kernels `ops.coop_split` / `ops.rec_fwd_parts` choose between 32 and 8192
utterances -- and the whole model at batch sizes that take the column-split kernel and the pieces path."""
import numpy as np
import pytest
import torch

from oracle import blstm as O
from oracle import frontend as OF

pytestmark = pytest.mark.gpu
H = 250


@pytest.fixture(scope="module")
def mods():
    import avsi_amd  # noqa: F401
    from avsi_amd import models, ops, blstm_layout
    return models, ops, blstm_layout


def _rand_biases(params, seed):
    rng = np.random.default_rng(seed)
    for layer in params['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = rng.normal(0, 0.1, size=layer[d]['bias'].shape).astype(np.float32)
    params['proj']['biases'] = rng.normal(0, 0.1, size=params['proj']['biases'].shape).astype(np.float32)
    return params


def _one_layer(bl, ops, B, T, D, seed):
    """Seeded one-layer problem: oracle parameters, packed device parameters, input and its hoisted projection."""
    Bp = -(-B // 32) * 32
    lay = bl.ParamLayout(D, (H,), 257)
    p = _rand_biases(O.init_params(seed, D, (H,), 257), seed + 1)
    flat = lay.flatten_oracle_params(p)
    packed = torch.from_numpy(np.concatenate([flat, [0]]).astype(np.float32)[lay.pack_index]).cuda()
    x = np.random.default_rng(seed + 2).normal(size=(B, T, D)).astype(np.float32)
    xp = torch.zeros(T, Bp, lay.kp[0], device='cuda')
    xp[:, :B, :D] = torch.from_numpy(x).cuda().transpose(0, 1)
    xproj = ops.gemm(xp.view(T * Bp, -1), lay.packed_view(packed, 'wx0'), bias=lay.packed_view(packed, 'b0')).view(T, Bp, 2048)
    return lay, O.cast_params(p, np.float64), packed, x.astype(np.float64), xproj, Bp


def _packed_gate_columns(dz_dir):
    """dz of one direction in the kernels' packed gate order (column 128 w + 32 gate + u = gate `gate` of hidden unit
    32 w + u) -> [.., 4, 256] indexed (gate, unit)."""
    lead = dz_dir.shape[:-1]
    return dz_dir.reshape(*lead, 8, 4, 32).swapaxes(-3, -2).reshape(*lead, 4, 256)


# (utterances, split): 0 = batch-stationary (rows_per_wg picks the 32-row or the 64-row ping-pong kernel), > 0 = workgroups per
# tile of the reduction-split cooperative kernels, < 0 = column-split kernel with that many utterances per group
FWD_KINDS = [(37, 0, 32), (70, 0, 64), (37, 4, 0), (70, 8, 0), (37, 16, 0), (70, 32, 0), (37, -16, 0), (70, -32, 0),
             (37, 64, 0), (70, 64, 0),        # 64: the 32-way kernel on 16-row halves (round 5)
             (70, 0, 65), (70, 0, 66), (130, 0, 66)]   # rows_per_wg 65 / 66: both tiles in one MFMA phase / the quarter-product
                                                       # kernel (round 6: the measured alternatives to the ping-pong kernel)


@pytest.mark.parametrize("B,split,rows_per_wg", FWD_KINDS)
@pytest.mark.parametrize("save", [False, True])
def test_forward_kernel_families_match_oracle(mods, B, split, rows_per_wg, save):
    models, ops, bl = mods
    T, D = 13, 40
    lay, p64, packed, x, xproj, Bp = _one_layer(bl, ops, B, T, D, 100 + B + abs(split))
    hout = torch.full((T, Bp, 512), 9.0, device='cuda')
    resv = torch.full((T, Bp, 2, 5, 256), 9.0, device='cuda') if save else None
    ops.blstm_rec_fwd(xproj, lay.packed_view(packed, 'wh0'), hout, resv, rows_per_wg, split=split)
    ops.coop_check()
    got = hout.cpu().numpy()
    for d, (name, rev) in enumerate((('fw', False), ('bw', True))):
        ref, cache = O.lstm_direction(x, p64['layers'][0][name]['kernel'], p64['layers'][0][name]['bias'], rev, True)
        np.testing.assert_allclose(got[:, :B, 256 * d:256 * d + H].transpose(1, 0, 2), ref, rtol=0, atol=2e-5)
        assert np.all(got[:, :, 256 * d + H:256 * (d + 1)] == 0)            # padded units stay exactly 0
        if save:
            r = resv.cpu().numpy()
            for (t, i, j, f, o, c_new, _, _) in cache:
                for gi, want in enumerate((i, j, f, o, c_new)):
                    np.testing.assert_allclose(r[t, :B, d, gi, :H], want, rtol=0, atol=3e-5)


@pytest.mark.parametrize("B,split,T", [(37, 0, 11), (70, 0, 11), (37, 4, 11), (70, 8, 11), (37, 16, 11), (70, 32, 11),
                                       (37, 'pp', 11), (70, 'pp', 11), (130, 'pp', 2), (33, 'pp', 1)])
def test_bptt_kernel_families_match_oracle(mods, B, split, T, monkeypatch):
    """dz of one layer, both directions: blstm_rec_bwd_kh_kernel (split 0: the batch-stationary kernel of batches up to
    4096 and beyond 8192 utterances), blstm_rec_bwd_pp_kernel ('pp': the ping-pong kernel of 4096 < batch <= 8192, forced
    here with AVSI_BWD_PP=1 -- 37 utterances: one workgroup, both 32-row tiles; 70: the second workgroup has one tile only;
    one- and two-step recurrences) and the cooperative BPTT kernels, against the float64 manual BPTT of the oracle.  The
    reserve comes from the forward kernel of the same family, as in a training step."""
    models, ops, bl = mods
    monkeypatch.delenv('AVSI_BWD_KH', raising=False)
    monkeypatch.setenv('AVSI_BWD_PP', '1' if split == 'pp' else '0')
    split = 0 if split == 'pp' else split
    D = 40
    lay, p64, packed, x, xproj, Bp = _one_layer(bl, ops, B, T, D, 300 + B + split)
    hout = torch.empty(T, Bp, 512, device='cuda')
    resv = torch.empty(T, Bp, 2, 5, 256, device='cuda')
    ops.blstm_rec_fwd(xproj, lay.packed_view(packed, 'wh0'), hout, resv, split=split)
    dh = np.random.default_rng(7).normal(size=(B, T, 2, H))
    dh_dev = torch.zeros(T, Bp, 512, device='cuda')
    dh_dev[:, :B, :H] = torch.from_numpy(dh[:, :, 0].astype(np.float32)).cuda().transpose(0, 1)
    dh_dev[:, :B, 256:256 + H] = torch.from_numpy(dh[:, :, 1].astype(np.float32)).cuda().transpose(0, 1)
    dz = torch.full((T, Bp, 2048), 9.0, device='cuda')
    ops.blstm_rec_bwd(dh_dev, resv, lay.packed_view(packed, 'whb0'), dz, split=split)
    ops.coop_check()
    got = dz.cpu().numpy()
    for d, (name, rev) in enumerate((('fw', False), ('bw', True))):
        k = p64['layers'][0][name]['kernel']
        _, cache = O.lstm_direction(x, k, p64['layers'][0][name]['bias'], rev, True)
        _, _, _, ref = O._lstm_direction_bwd(x, k, cache, dh[:, :, d].astype(np.float64), return_dz=True)     # [B, T, 4H]
        g = _packed_gate_columns(got[:, :B, 1024 * d:1024 * (d + 1)]).transpose(1, 0, 2, 3)               # [B, T, 4, 256]
        scale = np.abs(ref).max()
        np.testing.assert_allclose(g[..., :H], ref.reshape(B, T, 4, H), rtol=0, atol=3e-6 * scale + 1e-7)
        assert np.all(g[..., H:] == 0)


def _model_inputs(B, N, seed):
    rng = np.random.default_rng(seed)
    wav = np.clip(np.round(rng.normal(0, 3000, size=(B, N))), -32768, 32767).astype(np.float32)
    T = -(-N // 192)
    masks = np.ones((B, T, 257), dtype=np.float32)
    for b in range(B):
        s = rng.integers(0, T - 5)
        masks[b, s:s + 5] = 0
    spec = OF.get_spectrogram(OF.get_stft(wav[:64], window_size=24, step_size=12), log=True)
    mean, std = OF.feature_stats(list(spec))
    return wav, masks, mean.astype(np.float32), std.astype(np.float32), T


def _config(N, B):
    return dict(audio_feat_dim=257, video_feat_dim=136, audio_len=N, net_dim=[250, 250, 250], optimizer_type='adam',
                starter_learning_rate=1e-3, lr_updating_steps=10000, lr_decay=1.0, batch_size=B, l2=0.0)


@pytest.mark.parametrize("B", [300, 1030])
def test_model_forward_at_column_split_sizes_matches_oracle(mods, B, monkeypatch):
    """The whole model where the policy takes the column-split kernel (B = 300 -> Bp = 320: by 16) and the pieces path
    (B = 1030 -> Bp = 1056 = 1024 column-split by 32 + 32 on the half-row 32-way kernel): what `also.infer_b1024` of the bench times."""
    models, ops, bl = mods
    for name in ('AVSI_REC_CS', 'AVSI_COOP_CUS', 'AVSI_REC_COOP', 'AVSI_COOP_SPLIT_FWD', 'AVSI_REC_PARTS'):
        monkeypatch.delenv(name, raising=False)
    N = 3840
    Bp = -(-B // 32) * 32
    parts = ops.rec_fwd_parts(Bp)
    if B == 300:
        assert parts == [(0, 320, -16)]
    else:
        assert parts == [(0, 1024, -32), (1024, 32, 64)]         # the remainder on the half-row 32-way kernel (round 5)
    wav, masks, mean, std, T = _model_inputs(B, N, 900 + B)
    p = _rand_biases(O.init_params(21, 257), 22)
    seq_len = np.full(B, T)
    seq_len[-1] = T - 3
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(N, B), input='a', is_training=False)
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    ref = O.model_forward(wav, masks, mean, std, seq_len, p)
    pred = m.prediction.cpu().numpy()
    ops.coop_check()
    assert float(np.sqrt(np.mean((pred - ref['prediction']) ** 2))) < 1e-4
    assert np.abs(pred - ref['prediction']).max() < 2e-3
    assert float(m.loss_func) == pytest.approx(ref['loss_func'], rel=2e-4)


@pytest.mark.parametrize("coop", ['0', '1'])
def test_gradients_match_oracle_batch_stationary_and_cooperative(mods, coop, monkeypatch):
    """Model gradients against the float64 BPTT with AVSI_REC_COOP=0 -- every recurrence on the batch-stationary kernels
    (blstm_rec_fwd with reserve, blstm_rec_bwd_kh_kernel: the kernels of a training step at 8192 utterances) -- and with
    the default policy (cooperative kernels at this size)."""
    models, ops, bl = mods
    monkeypatch.setenv('AVSI_REC_COOP', coop)
    monkeypatch.delenv('AVSI_BWD_KH', raising=False)
    B, N = 40, 1920
    assert (ops.coop_split(64, backward=True) == 0) == (coop == '0')
    wav, masks, mean, std, T = _model_inputs(B, N, 77)
    rng = np.random.default_rng(3)
    video = rng.normal(size=(B, T, 136)).astype(np.float32)
    p = _rand_biases(O.init_params(9, 393), 10)
    seq_len = np.full(B, T)
    seq_len[0] = T - 2
    m = models.StackedBLSTMModel(seq_len, wav, masks, mean, std, 0.0, _config(N, B), video_features=video, input='av')
    m.variables.load_flat(m.layout.flatten_oracle_params(p))
    got = m.gradients.cpu().numpy().astype(np.float64)
    ops.coop_check()
    fwd = O.model_forward(wav, masks, mean, std, seq_len, p, video=video, input_type='av', keep=True)
    grads = O.model_backward(fwd, masks.astype(np.float64), seq_len)
    ref = m.layout.flatten_oracle_params({'layers': grads['layers'], 'proj': grads['proj']}).astype(np.float64)
    for name, shape, off in m.layout.ref_entries:
        n = int(np.prod(shape))
        g, r = got[off:off + n], ref[off:off + n]
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 2e-3 * scale + 1e-9, (name, np.abs(g - r).max(), scale)
        assert np.sqrt(np.mean((g - r) ** 2)) <= 2e-4 * scale + 1e-10, name
