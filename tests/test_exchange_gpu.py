"""Host-visible contract of the cooperative recurrent kernels' exchange machinery (C ABI level): the sticky status word,
the residency query of the column-split kernel, the stream head-start helper."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_status_word_is_sticky_and_counters_are_left_at_zero():
    """The caller zeroes the counter workspace once (include/avsi_hip.h); every launch puts its step counters back to zero
    itself, so calls follow each other without a memset.  Word 0 is a sticky status no launch clears -- and a launch that
    finds it set does NOTHING (its results would be void anyway: the launch that set it may have left step counters
    behind, and waiting out the time bound on them again would turn one timeout into a dozen)."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib, ops
    L = _lib.lib()
    T, Bp = 12, 64
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    ref = torch.empty(T, Bp, 512, device='cuda')
    ops.blstm_rec_fwd(xproj, whp, ref, None, split=0)
    need = max(L.avsi_blstm_rec_fwd_coop_workspace_bytes(Bp), L.avsi_blstm_rec_fwd_cs_workspace_bytes(Bp))
    ws = torch.zeros(need // 4, dtype=torch.int32, device='cuda')
    families = ((L.avsi_blstm_rec_fwd_coop_f32, 32), (L.avsi_blstm_rec_fwd_cs_f32, 16), (L.avsi_blstm_rec_fwd_coop_f32, 16),
                (L.avsi_blstm_rec_fwd_cs_f32, 32), (L.avsi_blstm_rec_fwd_coop_f32, 8), (L.avsi_blstm_rec_fwd_coop_f32, 4))
    # the same workspace through every forward kernel family, back to back
    for entry, split in families:
        hout = torch.zeros(T, Bp, 512, device='cuda')
        rc = entry(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), None, T, Bp, split, 0, _lib.ptr(ws), need, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert int(ws.abs().max()) == 0                         # status clean, every counter back at zero
        np.testing.assert_allclose(hout.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)
    ws[0] = 7                                                   # a failure recorded earlier
    for entry, split in families:
        hout = torch.zeros(T, Bp, 512, device='cuda')
        rc = entry(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), None, T, Bp, split, 0, _lib.ptr(ws), need, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert int(ws[0]) == 7                                  # sticky: the call did not touch it
        assert int(ws[1:].abs().max()) == 0                     # ... nor any counter
        assert float(hout.abs().max()) == 0.0                   # ... nor the output: a launch behind a failure is void


def test_column_split_residency_query_and_workspace():
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib
    L = _lib.lib()
    for rows in (16, 32):
        for reserve in (0, 1):
            per = L.avsi_blstm_rec_fwd_cs_groups_per_launch(rows, reserve, 0)
            assert per in (32, 64)                              # 8 workgroups per group, one or two of them per CU
            assert L.avsi_blstm_rec_fwd_cs_groups_per_launch(rows, reserve, 128) == per // 2
    assert L.avsi_blstm_rec_fwd_cs_groups_per_launch(24, 0, 0) == 0
    # one 256-byte line for the status word and one per (16-utterance tile, direction) counter
    assert L.avsi_blstm_rec_fwd_cs_workspace_bytes(1024) == 256 * (1 + 2 * 64)


def test_stream_delay_holds_a_stream_back():
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib, ops
    L = _lib.lib()
    assert L.avsi_stream_delay_us(-1, None) != 0 and L.avsi_stream_delay_us(1001, None) != 0
    assert L.avsi_stream_delay_us(0, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.stream_delay(500)
    e1.record()
    torch.cuda.synchronize()
    assert 0.45 <= e0.elapsed_time(e1) <= 5.0                  # ms: at least the half millisecond asked for


def test_reported_failure_resets_the_workspaces():
    """ops.coop_check raises once for a recorded failure and leaves clean workspaces behind (the kernels only put
    their counters back to zero on a clean end), so the next launch works again."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import _lib, ops
    T, Bp = 6, 32
    xproj = torch.randn(T, Bp, 2048, device='cuda')
    whp = torch.randn(2 * 262144, device='cuda') * 0.05
    hout = torch.zeros(T, Bp, 512, device='cuda')
    ops.blstm_rec_fwd(xproj, whp, hout, None, split=32)
    ops.coop_check()
    key = (torch.cuda.current_device(), _lib.stream_ptr().value)
    ws = ops._COOP_WS[key]
    ws[0] = 1                      # what a workgroup that gave up waiting would have written
    ws[64] = 999                   # ... and a counter it left behind
    with pytest.raises(_lib.AvsiError):
        ops.coop_check()
    assert int(ws.abs().max()) == 0
    ref = torch.zeros(T, Bp, 512, device='cuda')
    ops.blstm_rec_fwd(xproj, whp, ref, None, split=0)
    ops.blstm_rec_fwd(xproj, whp, hout, None, split=32)
    ops.coop_check()
    np.testing.assert_allclose(hout.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("Bp,split,save", [(32, 32, False), (96, 32, True), (64, 16, False), (288, 16, True)])
def test_exchange_layout_gives_the_same_result(Bp, split, save, monkeypatch):
    """The fine forward kernels with the exchange copy of h behind the counters (whole-line stores, contiguous fragment
    loads) against the same kernels exchanging through hout: identical summation order, so identical bits -- in hout
    and in the reserve."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 19
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp + split)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for exchange in (True, False):
        monkeypatch.setattr(ops, '_COOP_EXCHANGE', exchange)
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda') if save else None
        for _ in range(2):          # twice into the same buffers: a stale cached line of the first run would show
            ops.blstm_rec_fwd(xproj if _ else xproj * 0.5, whp, hout, resv, split=split)
        outs.append((hout, resv))
    ops.coop_check()
    assert torch.equal(outs[0][0], outs[1][0])
    if save:
        assert torch.equal(outs[0][1], outs[1][1])
    ref = torch.zeros(T, Bp, 512, device='cuda')
    ops.blstm_rec_fwd(xproj, whp, ref, None, split=0)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("Bp,save", [(32, False), (64, True), (96, False)])
def test_half_row_kernel_equals_the_32_way_kernel(Bp, save):
    """Split code 64 (the 32-way kernel on two independent 16-row halves per tile, v_mfma_f32_16x16x4_f32): the same reduction
    order per (row, unit) as the 32-row kernel -- eight waves' partial sums added in wave order -- but another MFMA shape, so
    equal to 2e-6, not to the bit; 96 utterances run as two launches (two tiles, then one).  Twice into the same buffers."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 21
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for split in (64, 32):
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda') if save else None
        for k in range(2):
            ops.blstm_rec_fwd(xproj if k else xproj * 0.5, whp, hout, resv, split=split)
        outs.append((hout, resv))
    ops.coop_check()
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=0, atol=2e-6)
    if save:
        np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=0, atol=5e-6)


@pytest.mark.parametrize("Bp,split,save", [(32, 32, False), (96, 32, True), (64, 16, False), (288, 16, True)])
def test_publication_without_the_store_acknowledgement_is_bit_identical(Bp, split, save, monkeypatch):
    """AVSI_COOP_NOACK=1 (opt-in, round 5): the exchange copy is preset to a poison pattern, members publish without waiting for
    the acknowledgement of their h store, a reader that meets a poisoned word loads again at device scope.  Same arithmetic,
    same order: identical bits, also into buffers that hold another input's results (no stale line, no stale poison)."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 23
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp + split + 1)
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for noack in ('0', '1'):
        monkeypatch.setenv('AVSI_COOP_NOACK', noack)
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda') if save else None
        for k in range(3):
            ops.blstm_rec_fwd(xproj if k == 2 else xproj * (0.25 + 0.25 * k), whp, hout, resv, split=split)
        outs.append((hout, resv))
    ops.coop_check()
    assert torch.equal(outs[0][0], outs[1][0])
    if save:
        assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("Bp,split", [(32, 32), (160, 32), (64, 16), (288, 16)])
def test_exchange_layout_gives_the_same_bptt_result(Bp, split, monkeypatch):
    """The fine BPTT kernels with dz exchanged through the copy in exchange layout against the same kernels exchanging
    through dz itself: identical bits, and both equal to the batch-stationary kernel within rounding."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 17
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp * 3 + split)
    dh = torch.randn(T, Bp, 512, generator=g, device='cuda')
    resv = torch.rand(T, Bp, 2, 5, 256, generator=g, device='cuda') * 0.9 + 0.05
    whbt = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    outs = []
    for exchange in (True, False):
        monkeypatch.setattr(ops, '_COOP_EXCHANGE', exchange)
        dz = torch.full((T, Bp, 2048), 7.0, device='cuda')
        for k in range(2):          # twice into the same buffers: a stale cached line of the first run would show
            ops.blstm_rec_bwd(dh if k else dh * 0.5, resv, whbt, dz, split=split)
        outs.append(dz)
    ops.coop_check()
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(T, Bp, 2048, device='cuda')
    ops.blstm_rec_bwd(dh, resv, whbt, ref, split=0)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-6 * scale + 1e-7)


@pytest.mark.parametrize("Bp,fwd_split,bwd_split", [(32, 32, 32), (128, 32, 32), (256, 16, 16), (288, 8, 8), (544, 4, 4),
                                                   (64, -16, 8), (1056, -32, 4)])
def test_coherent_switch_is_bit_identical_on_every_split_kind(Bp, fwd_split, bwd_split, monkeypatch):
    """AVSI_COOP_COHERENT=1 takes every cooperative exchange back to device-scope (`sc0 sc1`) loads -- the triage switch
    for the cacheable-load invariants of DESIGN 4.3a.  Same arithmetic in the same order: results must be BIT-identical to
    the default on every kind of split (exchange-layout fine kernels, 4- / 8-way BPTT with plain loads of dz, column-split)."""
    import torch
    import avsi_amd  # noqa: F401
    from avsi_amd import ops
    T = 23
    g = torch.Generator(device='cuda')
    g.manual_seed(Bp + 5 * abs(fwd_split))
    xproj = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    whp = torch.randn(2 * 262144, generator=g, device='cuda') * 0.05
    dh = torch.randn(T, Bp, 512, generator=g, device='cuda')
    junk = torch.randn(T, Bp, 2048, generator=g, device='cuda')
    results = {}
    for coherent in ('0', '1'):
        monkeypatch.setenv('AVSI_COOP_COHERENT', coherent)
        hout = torch.full((T, Bp, 512), 7.0, device='cuda')
        resv = torch.full((T, Bp, 2, 5, 256), 7.0, device='cuda')
        dz = torch.full((T, Bp, 2048), 7.0, device='cuda')
        # other data through the same buffers first: a stale line would show in the second result
        ops.blstm_rec_fwd(junk, whp, hout, resv, split=fwd_split)
        ops.blstm_rec_bwd(dh * 3.0, resv, whp, dz, split=bwd_split)
        ops.blstm_rec_fwd(xproj, whp, hout, resv, split=fwd_split)
        ops.blstm_rec_bwd(dh, resv, whp, dz, split=bwd_split)
        ops.coop_check()
        results[coherent] = (hout.clone(), resv.clone(), dz.clone())
    for a, b in zip(results['0'], results['1']):
        assert torch.equal(a, b)
    # and both agree with the batch-stationary kernels to rounding
    hout0 = torch.empty(T, Bp, 512, device='cuda')
    resv0 = torch.empty(T, Bp, 2, 5, 256, device='cuda')
    dz0 = torch.empty(T, Bp, 2048, device='cuda')
    ops.blstm_rec_fwd(xproj, whp, hout0, resv0, split=0)
    ops.blstm_rec_bwd(dh, resv0, whp, dz0, split=0)
    np.testing.assert_allclose(results['1'][0].cpu().numpy(), hout0.cpu().numpy(), rtol=0, atol=2e-5)
    scale = float(dz0.abs().max())
    np.testing.assert_allclose(results['1'][2].cpu().numpy(), dz0.cpu().numpy(), rtol=0, atol=5e-6 * scale)
