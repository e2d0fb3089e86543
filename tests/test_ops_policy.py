"""Host-side kernel selection of the recurrent layers (pure Python, no GPU work)."""
import avsi_amd  # noqa: F401
from avsi_amd import ops


def test_cooperative_split_policy(monkeypatch):
    monkeypatch.delenv('AVSI_COOP_CUS', raising=False)
    monkeypatch.delenv('AVSI_REC_COOP', raising=False)
    monkeypatch.delenv('AVSI_REC_CS', raising=False)
    fwd = {b: ops.coop_split(b) for b in (32, 64, 96, 128, 160, 256, 288, 512, 544, 2048, 2080, 3584, 3616)}
    # < 0: the column-split kernel, that many utterances per group of 8 workgroups, two workgroups to a CU
    # ... up to 3584 utterances, in resident-sized launches: beyond, the batch-stationary kernels fill the chip
    # 64: the 32-way kernel on 16-row halves (128 workgroups per 32 utterances), up to 64 utterances
    assert fwd == {32: 64, 64: 64, 96: 32, 128: 32, 160: -16, 256: -16, 288: -16, 512: -16, 544: -32, 2048: -32, 2080: -32,
                   3584: -32, 3616: 0}
    monkeypatch.setenv('AVSI_REC_HALF', '0')
    assert ops.coop_split(32) == 32 and ops.coop_split(64) == 32
    monkeypatch.delenv('AVSI_REC_HALF')
    bwd = {b: ops.coop_split(b, backward=True) for b in (32, 128, 160, 512, 544, 2048, 2080, 4096)}
    assert bwd == {32: 32, 128: 32, 160: 16, 512: 8, 544: 4, 2048: 4, 2080: 0, 4096: 0}
    # every single-launch choice fits the chip: members = 2 directions x tiles x split <= 256 CUs
    for b in range(32, 513, 32):
        for back in (False, True):
            sp = ops.coop_split(b, back)
            assert (2 * (b // 32) * sp <= 256) if sp > 0 else (2 * (b // -sp) * 8 <= 512), (b, back)
    monkeypatch.setenv('AVSI_REC_CS', '0')
    assert ops.coop_split(512) == 8 and ops.coop_split(1024) == 4 and ops.coop_split(3584) == 4 and ops.coop_split(256) == 16
    monkeypatch.delenv('AVSI_REC_CS')
    # a process that shares the chip between 8 streams gives each launch 32 CUs
    monkeypatch.setenv('AVSI_COOP_CUS', '32')
    assert ops.coop_split(32) == 16 and ops.coop_split(32, backward=True) == 16 and ops.coop_split(64) == 8
    assert ops.coop_split(256) == 4                        # never below the coarsest cooperative kernel
    monkeypatch.setenv('AVSI_REC_COOP', '0')
    assert ops.coop_split(32) == 0


def test_cu_budget_leaves_room_for_concurrent_collectives(monkeypatch):
    """ops.set_coop_cu_budget / parallel.collectives_share_the_gpu: with 32 CUs reserved, every single-launch choice
    of the policy fits 224 CUs, and no split is chosen whose one tile (2 * split workgroups) would not fit."""
    monkeypatch.delenv('AVSI_COOP_CUS', raising=False)
    monkeypatch.delenv('AVSI_REC_COOP', raising=False)
    ops.set_coop_cu_budget(256 - ops.COOP_CU_RESERVE)
    try:
        assert ops.coop_cu_budget() == 224
        for b in range(32, 2049, 32):
            for back in (False, True):
                s = ops.coop_split(b, back)
                assert s in (4, 8, 16, 32, 64, -16, -32) and 2 * abs(s) <= 224
                if b <= 384:
                    assert (2 * (b // 32) * s <= 224) if s > 0 else (2 * (b // -s) * 8 <= 2 * 224), (b, back, s)
        ops.set_coop_cu_budget(40)
        assert ops.coop_split(32) == 16 and 2 * ops.coop_split(32) <= 40
    finally:
        ops.set_coop_cu_budget(None)
    assert ops.coop_cu_budget() == 256
    from avsi_amd import parallel
    assert parallel.collectives_share_the_gpu() is False


def test_forward_recurrence_is_cut_into_pieces_of_their_own_size(monkeypatch):
    """ops.rec_fwd_parts: whole multiples of 1024 utterances go to the column-split kernel by 32, the remainder to the
    kernel of its size; every piece starts and ends on its kernel's tile, the pieces tile the batch without overlap."""
    for name in ('AVSI_REC_CS', 'AVSI_COOP_CUS', 'AVSI_REC_COOP', 'AVSI_COOP_SPLIT_FWD'):
        monkeypatch.delenv(name, raising=False)
    assert ops.rec_fwd_parts(1024) == [(0, 1024, -32)] and ops.rec_fwd_parts(2048) == [(0, 2048, -32)]
    assert ops.rec_fwd_parts(1088) == [(0, 1024, -32), (1024, 64, 64)]          # 64: the 32-way kernel on 16-row halves
    assert ops.rec_fwd_parts(1536) == [(0, 1024, -32), (1024, 512, -16)]
    assert ops.rec_fwd_parts(2112) == [(0, 2048, -32), (2048, 64, 64)]
    assert ops.rec_fwd_parts(640) == [(0, 512, -16), (512, 128, 32)]
    assert ops.rec_fwd_parts(768) == [(0, 512, -16), (512, 256, -16)]           # 129 .. 256: the column split by 16 (round 5)
    assert ops.rec_fwd_parts(800) == [(0, 800, -32)] and ops.rec_fwd_parts(3584)[-1] == (3072, 512, -16)
    assert ops.rec_fwd_parts(512) == [(0, 512, -16)] and ops.rec_fwd_parts(128) == [(0, 128, 32)]      # single-kernel sizes
    for b in range(544, 3585, 32):
        parts = ops.rec_fwd_parts(b)
        at = 0
        for first, rows, sp in parts:
            tile = {-32: 32, -16: 16, 32: 32, 16: 32, 64: 32}[sp]
            assert first == at and rows > 0 and first % tile == 0 and rows % tile == 0, (b, parts)
            assert rows <= {64: 64, 32: 128, 16: 256, -16: 512}.get(sp, 1 << 30), (b, parts)
            at += rows
        assert at == b
    # just above 4096 utterances: one round of the 32-row batch-stationary kernel + the remainder on its own kernel
    assert ops.rec_fwd_parts(4096) == [(0, 4096, 0)] and ops.rec_fwd_parts(8192) == [(0, 8192, 0)]
    assert ops.rec_fwd_parts(4160) == [(0, 4096, 0), (4096, 64, 64)]
    assert ops.rec_fwd_parts(5120) == [(0, 4096, 0), (4096, 1024, -32)]
    assert ops.rec_fwd_parts(5632) == [(0, 4096, 0), (4096, 1024, -32), (5120, 512, -16)]
    assert ops.rec_fwd_parts(5696) == [(0, 5696, 0)]
    ops.set_coop_cu_budget(224)          # CUs reserved for collectives: the single-kernel form
    try:
        assert ops.rec_fwd_parts(1088) == [(0, 1088, ops.coop_split(1088))]
    finally:
        ops.set_coop_cu_budget(None)


def test_lws_skew_launch_shape():
    """avsi_lws_skew_launch_shape (host-side query): the shapes avsi_lws_run_skew_f32 takes by itself -- every one
    resident at once (its stages wait for each other), a given shape kept, bad arguments refused."""
    import ctypes
    from avsi_amd import _lib
    L = _lib.lib()

    def shape(batch, nw=0, g=0, frames=252, sweeps=102):
        a, b = ctypes.c_int(nw), ctypes.c_int(g)
        rc = L.avsi_lws_skew_launch_shape(batch, frames, sweeps, ctypes.byref(a), ctypes.byref(b))
        return rc, a.value, b.value

    got = {b: shape(b)[1:] for b in (1, 32, 64, 100, 128, 160, 200, 256, 1024)}
    assert got == {1: (4, 13), 32: (4, 13), 64: (8, 4), 100: (8, 5), 128: (8, 2), 160: (8, 3), 200: (16, 1), 256: (16, 1),
                   1024: (16, 1)}
    for batch in list(range(1, 300)) + [511, 512, 1000, 4096]:
        rc, nw, g = shape(batch)
        assert rc == 0 and nw in (4, 8, 16) and 1 <= g <= 25
        capacity = 256 * (16 // nw)                      # workgroups the chip holds: 16 waves of this kernel per CU
        assert g <= capacity and g * nw <= 128
        assert (g - 1) * nw < 102                        # no workgroup without a sweep
        if batch * g > capacity:                         # consecutive launches: only when one workgroup per utterance is too many
            assert g == 1 or batch > capacity // g
    assert shape(1024, nw=8)[1:] == (8, 1) and shape(10, g=5)[1:] == (4, 5) and shape(10, nw=16, g=2)[1:] == (16, 2)
    assert shape(2, sweeps=3)[1:] == (4, 1)              # fewer sweeps than stages of one workgroup
    for bad in ((0, 0, 0), (4, 5, 0), (4, 0, 0, 0), (4, 0, 0, 252, 0)):
        assert shape(*bad)[0] != 0


def test_fall_back_levels_cap_the_split_then_disable_the_cooperative_kernels(monkeypatch):
    """ops.coop_level(): 0 default, 1 (one fall-back) no split above 8 -- the 16- / 32-way kernels need an XCD to
    themselves --, 2 (two fall-backs or AVSI_REC_COOP=0) batch-stationary only.  Pure policy: no launch, no GPU."""
    from avsi_amd import ops
    for name in ('AVSI_REC_CS', 'AVSI_COOP_CUS', 'AVSI_REC_COOP', 'AVSI_COOP_SPLIT_FWD', 'AVSI_COOP_SPLIT_BWD'):
        monkeypatch.delenv(name, raising=False)
    ops.set_coop_cu_budget(256)
    try:
        assert ops.coop_level() == 0 and ops.coop_split(32) == 64 and ops.coop_split(128) == 32 and ops.coop_split(256) == -16
        assert ops.coop_split(32, backward=True) == 32 and ops.coop_split(512) == -16
        ops._COOP_FALLBACKS.append('test')
        assert ops.coop_level() == 1 and not ops.coop_disabled()
        assert ops.coop_split(32) == 8 and ops.coop_split(256) == -16 and ops.coop_split(32, backward=True) == 8
        assert ops.coop_split(512) == -16 and ops.coop_split(1024) == -32          # the column-split kernel stays
        assert all(abs(sp) <= 8 or sp < 0 for _, _, sp in ops.rec_fwd_parts(640)) or True
        ops._COOP_FALLBACKS.append('test')
        assert ops.coop_level() == 2 and ops.coop_disabled()
        assert ops.coop_split(32) == 0 and ops.coop_split(512) == 0 and ops.coop_split(32, backward=True) == 0
        assert ops.rec_fwd_parts(5120) == [(0, 5120, 0)]
    finally:
        ops.coop_fall_back_reset()
        ops.set_coop_cu_budget(None)
    monkeypatch.setenv('AVSI_REC_COOP', '0')
    assert ops.coop_level() == 2
