"""Host logic: the reference <-> packed parameter layouts are consistent (CPU, numpy only)."""
import numpy as np

import avsi_amd  # noqa: F401
from avsi_amd.blstm_layout import GP, HP, ParamLayout, packed_gate_col
from oracle import blstm as O


def _packed(layout, params):
    flat = layout.flatten_oracle_params(params)
    ext = np.concatenate([flat, np.zeros(1, np.float32)])
    return flat, ext[layout.pack_index]


def test_sizes_match_reference_param_counts():
    assert ParamLayout(257).ref_size == 4148757
    assert ParamLayout(393).ref_size == 4420757
    assert ParamLayout(257).kp == [272, 512, 512]
    assert ParamLayout(136).kp == [144, 512, 512]


def test_roundtrip_flat_oracle_params():
    lay = ParamLayout(257)
    p = O.init_params(1, 257)
    flat = lay.flatten_oracle_params(p)
    q = lay.unflatten_to_oracle_params(flat)
    for a, b in zip(O.flatten_params(p), O.flatten_params(q)):
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1], b[1])


def test_pack_index_covers_every_parameter_and_pads_with_zero():
    lay = ParamLayout(393, (250, 250))
    flat = np.arange(1, lay.ref_size + 1, dtype=np.float64)
    packed = np.concatenate([flat, [0.0]])[lay.pack_index]
    # forward layout: every parameter once, plus the recurrent kernels a second time (BPTT order)
    rec = sum(250 * 1000 for _ in range(2 * 2))
    assert np.count_nonzero(packed) == lay.ref_size + rec
    for li in range(2):
        a = np.sort(lay.packed_view(packed, 'wh%d' % li)[lay.packed_view(packed, 'wh%d' % li) != 0])
        b = np.sort(lay.packed_view(packed, 'whb%d' % li)[lay.packed_view(packed, 'whb%d' % li) != 0])
        np.testing.assert_array_equal(a, b)


def test_grad_index_is_a_bijection_onto_parameter_slots():
    lay = ParamLayout(257)
    assert lay.grad_index.shape == (lay.ref_size,)
    assert len(np.unique(lay.grad_index)) == lay.ref_size
    assert lay.grad_index.max() < lay.gpacked_size
    # spot check: d/d kernel[row, col] of layer 1 bw, recurrent part -> dwh1[1][k][packed col]
    H, D = 250, 500
    name = 'cell_1/bw/kernel'
    off = [o for n, _, o in lay.ref_entries if n == name][0]
    k, g, u = 17, 2, 133
    pos = lay.grad_index[off + (D + k) * 1000 + g * H + u]
    dwh_off, _ = lay.gpacked['dwh1']
    assert pos == dwh_off + (1 * HP + k) * GP + (u // 32) * 128 + g * 32 + u % 32
    # input part of layer 1: reference row 260 (bw half, unit 10) is padded row 256 + 10
    pos = lay.grad_index[off + 260 * 1000 + g * H + u]
    dwx_off, _ = lay.gpacked['dwx1']
    assert pos == dwx_off + (256 + 10) * 2 * GP + packed_gate_col(1, g, u)


def test_bptt_fragment_order():
    """whb[d][w][q][lane][s] = Wh[unit' = 32w + (lane&31)][packed col 8q + 4(lane>>5) + s]."""
    H = 250
    lay = ParamLayout(257)
    p = O.init_params(5, 257)
    _, packed = _packed(lay, p)
    whb = lay.packed_view(packed, 'whb2').reshape(2, 8, 128, 64, 4)
    K = p['layers'][2]['fw']['kernel']
    for (w, q, lane, s) in [(0, 0, 0, 0), (3, 77, 45, 3), (7, 127, 63, 3), (5, 20, 25, 1)]:
        col, up = 8 * q + 4 * (lane >> 5) + s, 32 * w + (lane & 31)
        g, cu = (col % 128) // 32, (col // 128) * 32 + col % 32
        want = K[500 + up, g * H + cu] if (up < H and cu < H) else 0.0
        assert whb[0, w, q, lane, s] == want


def test_packed_input_projection_equals_reference_product():
    H = 250
    lay = ParamLayout(257)
    p = O.init_params(2, 257)
    for layer in p['layers']:
        for d in ('fw', 'bw'):
            layer[d]['bias'] = np.random.default_rng(3).normal(size=1000).astype(np.float32)
    flat, packed = _packed(lay, p)
    rng = np.random.default_rng(4)
    for li, D in enumerate((257, 500, 500)):
        x = rng.normal(size=(5, D))
        xp = np.zeros((5, lay.kp[li]))
        if li == 0:
            xp[:, :D] = x
        else:
            xp[:, :H], xp[:, HP:HP + H] = x[:, :H], x[:, H:]
        wx = lay.packed_view(packed, 'wx%d' % li).astype(np.float64)
        b = lay.packed_view(packed, 'b%d' % li).astype(np.float64)
        z = xp @ wx + b
        for d, dname in enumerate(('fw', 'bw')):
            K = p['layers'][li][dname]['kernel'].astype(np.float64)
            ref = x @ K[:D] + p['layers'][li][dname]['bias']
            u = np.arange(H)
            for g in range(4):
                np.testing.assert_allclose(z[:, packed_gate_col(d, g, u)], ref[:, g * H:(g + 1) * H], atol=1e-12)
        # padded hidden units: zero weights and zero bias
        for d in range(2):
            for g in range(4):
                assert np.all(z[:, packed_gate_col(d, g, np.arange(H, HP))] == 0)


def test_recurrent_fragment_order():
    """whp[d][w][q][g][lane][s] = Wh[k = 8q + 4(lane>>5) + s][unit 32w + (lane&31)][gate g]."""
    H = 250
    lay = ParamLayout(257)
    p = O.init_params(5, 257)
    _, packed = _packed(lay, p)
    wh = lay.packed_view(packed, 'wh1').reshape(2, 8, 32, 4, 64, 4)
    K = p['layers'][1]['bw']['kernel']
    for (w, q, g, lane, s) in [(0, 0, 0, 0, 0), (3, 17, 2, 45, 3), (7, 31, 3, 63, 3), (7, 31, 1, 25, 1)]:
        k, u = 8 * q + 4 * (lane >> 5) + s, 32 * w + (lane & 31)
        want = K[500 + k, g * H + u] if (k < H and u < H) else 0.0
        assert wh[1, w, q, g, lane, s] == want


def test_projection_padding():
    lay = ParamLayout(257)
    p = O.init_params(6, 257)
    _, packed = _packed(lay, p)
    pw = lay.packed_view(packed, 'pw')
    assert pw.shape == (512, 260)
    np.testing.assert_array_equal(pw[:250, :257], p['proj']['weights'][:250])
    np.testing.assert_array_equal(pw[256:506, :257], p['proj']['weights'][250:])
    assert np.all(pw[250:256] == 0) and np.all(pw[506:] == 0) and np.all(pw[:, 257:] == 0)


def test_side_input_and_mlp_layouts():
    """The variants' layouts: side rows of the TF kernel are packed apart ('we'), the recurrent rows move
    behind them, the speaker-embedding MLP gets dense entries; every parameter has exactly one packed copy
    (two for the recurrent kernels) and one gradient slot."""
    E, W = 24, 40
    for side_layer in (0, 1, 2):
        lay = ParamLayout(257, (250, 250, 250), 257, side=(side_layer, E), mlp=W, mlp_in_pitch=272)
        base = ParamLayout(257)
        assert lay.ref_size == base.ref_size + 2 * E * 1000 + (2 * 257 * W + W) + 2 * (W * W + W)
        flat = np.arange(1, lay.ref_size + 1, dtype=np.float64)
        packed = np.concatenate([flat, [0.0]])[lay.pack_index]
        rec = 3 * 2 * 250 * 1000
        assert np.count_nonzero(packed) == lay.ref_size + rec
        # side rows: kernel rows [D, D + E) of the side layer, both directions, land in 'we'
        D = 257 if side_layer == 0 else 500
        we = lay.packed_view(packed, 'we')
        kf = lay.ref_view(flat, 'cell_%d/fw/kernel' % side_layer)
        kb = lay.ref_view(flat, 'cell_%d/bw/kernel' % side_layer)
        assert kf.shape == (D + E + 250, 1000)
        g, u = 3, 77
        assert we[5, packed_gate_col(0, g, u)] == kf[D + 5, g * 250 + u]
        assert we[E - 1, packed_gate_col(1, g, u)] == kb[D + E - 1, g * 250 + u]
        assert np.all(we[E:] == 0)
        # recurrent rows start behind the side rows: compare against the no-side layout's packed Wh
        p0 = O.init_params(3, 257)
        fl0 = base.flatten_oracle_params(p0)
        wh0 = np.concatenate([fl0, [0.0]])[base.pack_index]
        k_with_side = np.insert(p0['layers'][side_layer]['fw']['kernel'], [D] * E, 7.0, axis=0)
        p1 = {'layers': [dict(l) for l in p0['layers']], 'proj': p0['proj'],
              'mlp': {'weights_1': np.ones((514, W)), 'biases_1': np.ones(W), 'weights_2': np.ones((W, W)), 'biases_2': np.ones(W),
                      'weights_3': np.ones((W, W)), 'biases_3': np.ones(W)}}
        p1['layers'][side_layer] = {'fw': {'kernel': k_with_side, 'bias': p0['layers'][side_layer]['fw']['bias']},
                                    'bw': {'kernel': np.insert(p0['layers'][side_layer]['bw']['kernel'], [D] * E, 7.0, axis=0),
                                           'bias': p0['layers'][side_layer]['bw']['bias']}}
        pk1 = np.concatenate([lay.flatten_oracle_params(p1), [0.0]])[lay.pack_index]
        for name in ('wh%d' % side_layer, 'whb%d' % side_layer, 'wx%d' % side_layer):
            np.testing.assert_array_equal(lay.packed_view(pk1, name), base.packed_view(wh0, name))
        # MLP halves: weights_1 rows [0, F) -> mw1a, [F, 2F) -> mw1b, padded rows zero
        w1 = lay.ref_view(flat, 'speaker_embedding/weights_1')
        assert np.array_equal(lay.packed_view(packed, 'mw1a')[:257], w1[:257]) and np.all(lay.packed_view(packed, 'mw1a')[257:] == 0)
        assert np.array_equal(lay.packed_view(packed, 'mw1b')[:257], w1[257:])
        # gradient slots: a bijection
        assert len(np.unique(lay.grad_index)) == lay.ref_size and lay.grad_index.max() < lay.gpacked_size
        assert lay.signature() == [257, 250, 3, 257, side_layer, E, -W]


def test_asr_head_shares_the_packed_projection():
    """Multi-task CTC models: the asr columns start at F rounded up to 4 inside the same packed matrix."""
    lay = ParamLayout(257, asr=34)
    assert lay.ref_size == ParamLayout(257).ref_size + 500 * 34 + 34
    assert (lay.asr_col, lay.ldp) == (260, 296)
    assert lay.packed['pw'][1] == (2 * HP, 296) and lay.gpacked['dpw'][1] == (2 * HP, 296)
    flat = np.arange(1, lay.ref_size + 1, dtype=np.float64)
    pw = lay.packed_view(np.concatenate([flat, [0.0]])[lay.pack_index], 'pw')
    wa = lay.ref_view(flat, 'asr/weights')
    wi = lay.ref_view(flat, 'logits/weights')
    np.testing.assert_array_equal(pw[:250, 260:294], wa[:250])
    np.testing.assert_array_equal(pw[HP:HP + 250, 260:294], wa[250:])
    np.testing.assert_array_equal(pw[:250, :257], wi[:250])
    assert not pw[:, 257:260].any() and not pw[:, 294:].any() and not pw[250:HP].any()
    pb = lay.packed_view(np.concatenate([flat, [0.0]])[lay.pack_index], 'pb')
    np.testing.assert_array_equal(pb[260:294], lay.ref_view(flat, 'asr/biases'))
    # gradient index: a bijection onto distinct slots, asr entries inside dpw / dpb
    assert len(np.unique(lay.grad_index)) == lay.ref_size
    off, shape = lay.gpacked['dpw']
    gi = lay.ref_view(lay.grad_index, 'asr/weights')
    assert gi[0, 0] == off + 260 and gi[250, 3] == off + HP * 296 + 263
    assert lay.signature()[-1] == -100034
    with np.testing.assert_raises(ValueError):
        ParamLayout(257, asr=1)


def test_net_dim_support_is_decided_by_the_c_abi():
    """The reference takes any num_units per layer (models.py:95-99,107); the recurrent kernels any widths of 1 .. 256 units
    (each padded to 256).  The rule lives in the C ABI (avsi_blstm_net_supported) and the host layer reports ITS status."""
    import ctypes
    import pytest
    from avsi_amd import _lib
    L = _lib.lib()
    for dims, want in (([250, 250, 250], _lib.AVSI_OK), ([256], _lib.AVSI_OK), ([7, 7], _lib.AVSI_OK), ([250, 128], _lib.AVSI_OK),
                       ([257, 257], _lib.AVSI_ERR_UNSUPPORTED), ([250, 300], _lib.AVSI_ERR_UNSUPPORTED), ([0], _lib.AVSI_ERR_INVALID_ARG)):
        assert L.avsi_blstm_net_supported((ctypes.c_int * len(dims))(*dims), len(dims)) == want, dims
    assert ParamLayout(257, (64, 64)).H == 64
    for bad in ((250, 257, 250), (300, 300)):
        with pytest.raises(_lib.AvsiError, match="unsupported shape"):
            ParamLayout(257, bad)


def test_unequal_layer_widths_pack_layer_by_layer():
    """net_dim = (96, 250, 40): every layer's kernel is (d + H_l, 4 H_l) with d = 2 H_{l-1} (models.py:95-99,107); each is padded
    to 256 units of its own, the ones column of a layer's input sits behind the units of the layer BELOW, the projection reads
    the TOP layer's 2 x 40."""
    dims = (96, 250, 40)
    lay = ParamLayout(257, dims)
    assert (lay.H, lay.Hs, lay.in_dims) == (40, dims, [257, 192, 500])
    assert lay.ones_col == [257, 96, 250] and lay.ones_col_top == 40
    p = O.init_params(5, 257, net_dim=dims)
    flat = lay.flatten_oracle_params(p)
    assert flat.size == lay.ref_size
    packed = np.concatenate([flat, [0.0]])[lay.pack_index]
    for li, H in enumerate(dims):
        kf, kb = p['layers'][li]['fw']['kernel'], p['layers'][li]['bw']['kernel']
        D = lay.in_dims[li]
        assert kf.shape == (D + H, 4 * H)
        wx = lay.packed_view(packed, 'wx%d' % li)
        rm = lay.input_row_map(li)
        g, u = 2, H - 1
        for k in np.flatnonzero(rm >= 0)[[0, 1, -1]]:
            assert wx[k, packed_gate_col(0, g, u)] == kf[rm[k], g * H + u]
            assert wx[k, packed_gate_col(1, g, u)] == kb[rm[k], g * H + u]
        # the constant-1 column of the layer's input: zero weights forward, the bias gradient's row in the gradient layout
        if lay.ones_col[li] >= 0:
            assert not wx[lay.ones_col[li]].any()
            off, shape = lay.gpacked['dwx%d' % li]
            gb = lay.ref_view(lay.grad_index, 'cell_%d/fw/bias' % li)
            assert gb[g * H + u] == off + lay.ones_col[li] * shape[1] + packed_gate_col(0, g, u)
        if H < HP:
            assert not wx[:, packed_gate_col(0, g, H)].any()        # a unit past the layer's width: no weights at all
        # recurrent fragments: count of non-zeros = 2 directions x 4 gates x H x H in both orders
        assert np.count_nonzero(lay.packed_view(packed, 'wh%d' % li)) == 2 * 4 * H * H
        assert np.count_nonzero(lay.packed_view(packed, 'whb%d' % li)) == 2 * 4 * H * H
    pw = lay.packed_view(packed, 'pw')
    np.testing.assert_array_equal(pw[:40, :257], p['proj']['weights'][:40])
    np.testing.assert_array_equal(pw[HP:HP + 40, :257], p['proj']['weights'][40:])
    assert not pw[40:HP].any() and not pw[HP + 40:].any()
    assert len(np.unique(lay.grad_index)) == lay.ref_size and lay.grad_index.max() < lay.gpacked_size
    assert lay.signature() == [257, 40, 3, 257, -200000, 96, 250, 40]
    assert ParamLayout(257, (250, 250, 250)).signature() == [257, 250, 3, 257]
