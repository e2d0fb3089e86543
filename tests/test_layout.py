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
