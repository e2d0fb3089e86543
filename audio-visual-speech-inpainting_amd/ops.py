"""Thin torch-tensor wrappers over the C ABI (one function per entry point of avsi_hip.h).

No arithmetic happens here: each wrapper validates layout, passes raw device pointers and the
current stream to libavsi_hip.so, and returns the output tensor."""
import ctypes
import os
import sys

import numpy as np

import torch

from . import _lib


def gemm(a, b, out=None, trans_a=False, trans_b=False, m=None, n=None, k=None, alpha=1.0, beta=0.0,
         bias=None, row_scale=None, row_map=None, ldc=None, k_zero=None):
    """C = alpha * op(A) . op(B) + bias (+ beta * C) through avsi_gemm_f32.

    a, b: 2-D float32 device tensors, unit stride along their last dim (row pitch = lda/ldb).
    m, n, k default to the logical shapes implied by a/b; pass them to ignore padding columns.
    row_map = (Bp, T, B): time-major rows -> batch-major output rows (see avsi_hip.h).
    k_zero = ((lo, hi), ...): up to two ranges of the reduction index whose rows of op(B) are zero (padding): the caller's
    promise, which lets the 16-deep A . B tiles skip the multiply-adds that see padding only."""
    _lib.require_cuda(a, b, out, bias, row_scale)
    L = _lib.lib()
    for x in (a, b):
        if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
            raise _lib.AvsiError("gemm operands must be 2-D float32 with unit inner stride")
    M = m if m is not None else (a.shape[1] if trans_a else a.shape[0])
    K = k if k is not None else (a.shape[0] if trans_a else a.shape[1])
    N = n if n is not None else (b.shape[0] if trans_b else b.shape[1])
    if out is None:
        rows = M if row_map is None else row_map[1] * row_map[2]
        out = torch.empty((rows, N), dtype=torch.float32, device=a.device)
    ep = _lib.GemmEpilogue()
    ep.bias, ep.row_scale = _lib.ptr(bias), _lib.ptr(row_scale)
    if row_map is not None:
        ep.row_map_bp, ep.row_map_t, ep.row_map_b = (int(v) for v in row_map)
    if k_zero:
        for i, (lo, hi) in enumerate(list(k_zero)[:2]):
            ep.k_zero[2 * i], ep.k_zero[2 * i + 1] = int(lo), int(hi)
    _lib.check(L.avsi_gemm_f32(int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.ptr(a), a.stride(0),
                               _lib.ptr(b), b.stride(0), float(beta), _lib.ptr(out),
                               out.stride(0) if ldc is None else ldc, ctypes.byref(ep), _lib.stream_ptr()),
               "avsi_gemm_f32")
    return out


def dropout(x, y, scale, cols, rate, seed):
    """y = x * scale on the first `cols` columns of the 2-D views x / y / scale (same pitch), scale drawn per element:
    0 with probability rate, else 1 / (1 - rate) (avsi_dropout_f32, tf.nn.dropout of models.py:117)."""
    _lib.require_cuda(x, y, scale)
    _lib.check(_lib.lib().avsi_dropout_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(scale), x.shape[0], int(cols), x.stride(0),
                                           float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.stream_ptr()), "avsi_dropout_f32")
    return y


def scale_elements(x, scale, cols):
    """x *= scale on the first `cols` columns (avsi_scale_elements_f32)."""
    _lib.require_cuda(x, scale)
    _lib.check(_lib.lib().avsi_scale_elements_f32(_lib.ptr(x), _lib.ptr(scale), x.shape[0], int(cols), x.stride(0),
                                                  _lib.stream_ptr()), "avsi_scale_elements_f32")
    return x


def pack_bf16x3_b(b):
    """Split the weight matrix b [K, N] (float32, unit inner stride) into bf16 hi / lo planes in MFMA fragment order
    (avsi_pack_bf16x3_b); returns the packed buffer for gemm_bf16x3."""
    _lib.require_cuda(b)
    L = _lib.lib()
    K, N = b.shape
    nbytes = L.avsi_pack_bf16x3_b_bytes(K, N)
    if not nbytes:
        raise _lib.AvsiError("pack_bf16x3_b: N must be a multiple of 32")
    packed = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=b.device)
    _lib.check(L.avsi_pack_bf16x3_b(_lib.ptr(b), b.stride(0), K, N, _lib.ptr(packed), nbytes, _lib.stream_ptr()),
               "avsi_pack_bf16x3_b")
    return packed


def gemm_bf16x3(a, b_packed, out, k, bias=None):
    """EXPLORATORY: out [M, N] = a [M, k] . B + bias with split-bf16 operands (hi.hi + hi.lo + lo.hi, fp32
    accumulation) through avsi_gemm_bf16x3_f32; B comes packed from pack_bf16x3_b."""
    _lib.require_cuda(a, b_packed, out, bias)
    for x in (a, out):
        if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
            raise _lib.AvsiError("gemm_bf16x3 operands must be 2-D float32 with unit inner stride")
    M, N = a.shape[0], out.shape[1]
    _lib.check(_lib.lib().avsi_gemm_bf16x3_f32(M, N, int(k), _lib.ptr(a), a.stride(0), _lib.ptr(b_packed), _lib.ptr(bias),
                                               _lib.ptr(out), out.stride(0), _lib.stream_ptr()), "avsi_gemm_bf16x3_f32")
    return out


def blstm_rec_fwd(xproj, whp, hout, reserve=None, rows_per_wg=0, split=None):
    """All T steps of both directions of one layer (avsi_blstm_rec_fwd_f32, or its small-batch
    cooperative form avsi_blstm_rec_fwd_coop_f32: `split` 4 / 8 / 16 / 32 forces it, 0 forbids it, None = by Bp;
    `split` -16 / -32 = the column-split kernel avsi_blstm_rec_fwd_cs_f32 with that many utterances per group).
    xproj [T, Bp, 2048], whp [2 * 262144], hout [T, Bp, 512], reserve [T, Bp, 2, 5, 256] or None."""
    _lib.require_cuda(xproj, whp, hout, reserve)
    T, Bp = xproj.shape[0], xproj.shape[1]
    if not (xproj.is_contiguous() and hout.is_contiguous() and whp.is_contiguous()):
        raise _lib.AvsiError("blstm_rec_fwd operands must be contiguous")
    if tuple(xproj.shape) != (T, Bp, 2048) or tuple(hout.shape) != (T, Bp, 512) or whp.numel() != 2 * 262144:
        raise _lib.AvsiError("blstm_rec_fwd: bad operand shapes")
    if reserve is not None and (tuple(reserve.shape) != (T, Bp, 2, 5, 256) or not reserve.is_contiguous()):
        raise _lib.AvsiError("blstm_rec_fwd: bad reserve shape")
    auto = rows_per_wg == 0 and split is None
    split = coop_split(Bp) if auto else int(split or 0)
    # (first_row, rows, split) pieces: by default the whole batch with one kernel; the automatic choice gives the
    # remainder beyond the large kernel's launches to the kernel of its own size (rec_fwd_parts)
    parts = rec_fwd_parts(Bp) if (auto and _REC_PARTS) else [(0, Bp, split)]
    L = _lib.lib()
    if len(parts) == 1 and parts[0][2] == 0:
        _lib.check(L.avsi_blstm_rec_fwd_f32(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), _lib.ptr(reserve),
                                            T, Bp, int(rows_per_wg), _lib.stream_ptr()), "avsi_blstm_rec_fwd_f32")
        return hout
    need = max(L.avsi_blstm_rec_fwd_cs_workspace_bytes(Bp), L.avsi_blstm_rec_fwd_coop_workspace_bytes(Bp))
    fine = [rows for _, rows, sp in parts if (_cap_split(sp) if auto else sp) >= 16]
    if _COOP_EXCHANGE and fine:
        # fine splits: room for the exchange copy of h (of THEIR rows) at its fixed offset (AVSI_COOP_EXCHANGE=0:
        # exchange through hout)
        need = max(need, COOP_EXCHANGE_OFFSET + L.avsi_blstm_rec_fwd_coop_exchange_bytes(T, max(fine)))
    ws = _coop_ws(xproj.device, Bp, need)
    for first, rows, sp in parts:
        sp = _cap_split(sp) if auto else sp
        if sp == 0:
            _lib.check(L.avsi_blstm_rec_fwd_rows_f32(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), _lib.ptr(reserve), T, Bp,
                                                     0, first, rows, _lib.stream_ptr()), "avsi_blstm_rec_fwd_rows_f32")
        elif sp < 0:
            _lib.check(L.avsi_blstm_rec_fwd_cs_rows_f32(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), _lib.ptr(reserve), T, Bp,
                                                        -sp, first, rows, coop_cu_budget(), _lib.ptr(ws), ws.numel() * 4,
                                                        _lib.stream_ptr()), "avsi_blstm_rec_fwd_cs_rows_f32")
        else:
            _lib.check(L.avsi_blstm_rec_fwd_coop_rows_f32(_lib.ptr(xproj), _lib.ptr(whp), _lib.ptr(hout), _lib.ptr(reserve), T, Bp,
                                                          sp, first, rows, coop_cu_budget(), _lib.ptr(ws),
                                                          need if sp >= 16 else ws.numel() * 4, _lib.stream_ptr()),
                       "avsi_blstm_rec_fwd_coop_rows_f32")
    return hout


_REC_PARTS = os.environ.get('AVSI_REC_PARTS', '1') != '0'
REC_TAIL_MAX = 1536        # largest remainder (utterances) cut off a batch of more than 4096 for the small-batch kernels


def _tail_split(r):
    """The kernel of a remainder of r <= 256 utterances: half-row 32-way up to 64, 32-way up to 128, column split by 16 above."""
    if r <= 64 and _COOP_EXCHANGE and os.environ.get('AVSI_REC_HALF', '1') != '0':
        return 64
    return 32 if r <= 128 else -16


def _small_parts(at, r):
    """Pieces for the r utterances from row `at` on (r <= CS_MAX_BATCH), each on the cooperative kernel of its size."""
    parts = []
    whole = r // 1024 * 1024
    if whole:
        parts.append((at, whole, -32))
        at, r = at + whole, r - whole
    if r > 768:
        parts.append((at, r, -32))
    elif r > 512:
        parts.append((at, 512, -16))
        parts.append((at + 512, r - 512, _tail_split(r - 512)))
    elif r > 256:
        parts.append((at, r, -16))
    elif r:
        parts.append((at, r, _tail_split(r)))
    return parts


def rec_fwd_parts(Bp):
    """[(first_row, rows, split)]: how the forward recurrence of a batch of Bp utterances is cut (split as coop_split: 0 =
    the batch-stationary kernel).  The cooperative launches are latency-bound -- 250 steps of 3 .. 11 us whatever the
    number of groups -- so a remainder beyond the large kernel's resident launches is cheaper on the kernel of ITS size
    than as one more launch of the large one.  Per layer, ms: half-row 32-way (<= 64 utterances) 0.53 - 0.57, 32-way (<= 128) 0.75,
    column-split by 16 (<= 256) 1.0, (<= 512) 1.6, by 32 (<= 1024 per launch) 2.8; batch-stationary 32-row kernel 10.3 for anything up
    to 4096, 64-row kernel 16.5 up to 8192.  1088 = 1024 by 32 + 64 32-way: 3.6 instead of 5.6; 640 = 512 by 16 + 128
    32-way: 2.4 instead of 2.8; 5120 = 4096 batch-stationary + 1024 by 32: 13.1 instead of 16.5.  Only taken at the full
    CU budget (a reserved-CU run keeps the single-kernel form)."""
    split = coop_split(Bp)
    single = [(0, Bp, split)]
    if coop_cu_budget() < 256 or os.environ.get('AVSI_COOP_SPLIT_FWD') or coop_disabled():
        return single
    if split == 0:
        if 4096 < Bp <= 4096 + REC_TAIL_MAX and os.environ.get('AVSI_REC_CS', '1') != '0':
            return [(0, 4096, 0)] + _small_parts(4096, Bp - 4096)
        return single
    if split != -32:
        return single
    return _small_parts(0, Bp)


_COOP_WS, _COOP_HOST = {}, {}      # keyed by (device index, stream)
_COOP_FALLBACKS = []               # reasons, one per fall-back of this process (coop_fall_back)
COOP_FALLBACK_MAX_SPLIT = 8        # first fall-back: reduction splits capped here (see coop_fall_back)


def coop_level():
    """0: the default policy.  1: after one fall-back -- the cooperative kernels that need a whole XCD to themselves are
    out.  2: after two (or AVSI_REC_COOP=0) -- the batch-stationary kernels only."""
    return 2 if os.environ.get('AVSI_REC_COOP', '1') == '0' else min(2, len(_COOP_FALLBACKS))


def coop_disabled():
    """True when the recurrence runs on the batch-stationary kernels only: AVSI_REC_COOP=0, or this process fell back
    twice after cooperative launches gave up waiting for residency (coop_fall_back)."""
    return coop_level() >= 2


def coop_fallbacks():
    """How many times this process fell back from its cooperative kernels (0, 1 or 2: a fall-back is not undone)."""
    return len(_COOP_FALLBACKS)


def _cap_split(split):
    """Level 1: the 16- and 32-way kernels hold 96 KB of LDS per workgroup ON PURPOSE (one member per CU) and keep the
    S members of a group on ONE XCD (their exchange stays in that XCD's L2): a 32-way group needs all 32 CUs of its XCD,
    so any other resident that takes LDS there starves it (tools/coop_timeout_probe.py: 16 parked CUs are enough).  The
    4- and 8-way kernels run two members to a CU on 32 KB and the column-split kernel four: they only need room
    somewhere."""
    if coop_level() == 1 and split > COOP_FALLBACK_MAX_SPLIT:
        return COOP_FALLBACK_MAX_SPLIT
    return split


def coop_fall_back(device=None, reason='a cooperative recurrent launch timed out waiting for residency'):
    """Recovery from a cooperative-kernel timeout, in this process: wait for the device, put the workspaces (status
    words and step counters) back to zero and take the next step down: first to the cooperative kernels that tolerate
    neighbours (8-way reduction split, column split: 1.8 instead of 0.8 ms per layer at 32 utterances), then to the
    batch-stationary kernels, which need no co-residency of their workgroups at all (~10 ms per layer whatever the
    batch).  Logged each time.  The caller repeats the batches whose steps the step guard voided (training.train) --
    the variables were not touched by them (avsi_adam_tf_guarded_f32)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    _coop_recover(idx)
    if len(_COOP_FALLBACKS) < 2:
        _COOP_FALLBACKS.append(reason)
        print('avsi: %s; falling back to %s for the rest of this process' % (
            reason, 'the cooperative kernels that tolerate neighbours (splits <= %d, column split)' % COOP_FALLBACK_MAX_SPLIT
            if len(_COOP_FALLBACKS) == 1 else 'the batch-stationary recurrent kernels'), file=sys.stderr, flush=True)


def coop_fall_back_reset():
    """Tests only: the default policy again."""
    del _COOP_FALLBACKS[:]


_COOP_EXCHANGE = os.environ.get('AVSI_COOP_EXCHANGE', '1') != '0'
COOP_EXCHANGE_OFFSET = 1 << 20       # AVSI_COOP_EXCHANGE_OFFSET of include/avsi_hip.h
_COOP_MSG = "cooperative recurrent kernel timed out waiting for a peer workgroup; results are invalid"
COOP_POLL_RAISES = True     # training.train() turns the host-side polls off: the step guard decides there, on every rank alike


class CoopTimeout(_lib.AvsiError):
    """A cooperative recurrent launch gave up a bounded wait (its workgroups were not all resident): the results of
    everything behind it are void.  Recoverable: coop_fall_back()."""



def _coop_ws(device, Bp, need=None):
    """Step counters of the cooperative kernels: one buffer per (device, stream) -- launches on one
    stream run in order and may share it, launches on different streams may overlap and must not.
    Word 0 is the sticky status word of the C ABI: zero here, never cleared by a launch, so a failure stays
    visible without any host-side bookkeeping behind the launches (that bookkeeping -- an OR into a sticky flag
    and its copy to pinned memory -- was two more kernels per recurrent launch: 0.16 ms of an 8 ms training step
    at 32 utterances)."""
    need = max(need or 0, _lib.lib().avsi_blstm_rec_fwd_coop_workspace_bytes(Bp))
    key = (device.index, _lib.stream_ptr().value)
    ws = _COOP_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        new = torch.zeros((need + 3) // 4, dtype=torch.int32, device=device)
        if ws is not None:
            new[:1].copy_(ws[:1])                    # a failure already recorded stays recorded
        ws = _COOP_WS[key] = new
    if key not in _COOP_HOST:
        _COOP_HOST[key] = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    return ws


_COOP_CU_BUDGET = None      # set_coop_cu_budget(); None = AVSI_COOP_CUS or the whole chip
COOP_CU_RESERVE = 32        # CUs left to concurrent collectives under data parallelism (parallel.init)


def set_coop_cu_budget(cus):
    """Compute units ONE cooperative recurrent launch may occupy (None: AVSI_COOP_CUS, default 256).  Every
    member of such a launch must be resident together, so whatever else this process keeps in flight beside it
    has to be left out: ``parallel.init()`` reserves COOP_CU_RESERVE CUs for the RCCL kernels of the bucketed
    gradient all-reduce, which run concurrently with the BPTT of the layers below; a process with several small
    batches in flight on different streams divides the chip between them.  Both the split policy below and the
    C-side cutting of a batch into resident-sized launches (``max_cus``) follow it."""
    global _COOP_CU_BUDGET
    _COOP_CU_BUDGET = None if cus is None else max(8, min(256, int(cus)))


def coop_cu_budget():
    if _COOP_CU_BUDGET is not None:
        return _COOP_CU_BUDGET
    budget = max(8, min(256, int(os.environ.get('AVSI_COOP_CUS', '256'))))
    from . import parallel
    if parallel.collectives_share_the_gpu():
        budget = min(budget, 256 - COOP_CU_RESERVE)
    return budget


def stream_delay(microseconds):
    """Hold the CURRENT stream back for that long (avsi_stream_delay_us: one idle wave)."""
    _lib.check(_lib.lib().avsi_stream_delay_us(int(microseconds), _lib.stream_ptr()), "avsi_stream_delay_us")


def occupy_cus(num_cus, release, max_ms=5000):
    """Diagnostic (avsi_diag_occupy_cus): park `num_cus` whole-CU workgroups on the CURRENT stream until the device
    int32 tensor `release` becomes non-zero (or max_ms passes)."""
    _lib.require_cuda(release)
    _lib.check(_lib.lib().avsi_diag_occupy_cus(int(num_cus), _lib.ptr(release), int(max_ms), _lib.stream_ptr()),
               "avsi_diag_occupy_cus")


CS_MAX_BATCH = int(os.environ.get('AVSI_REC_CS_MAX', '3584'))     # largest batch of the column-split forward kernel


def coop_split(Bp, backward=False):
    """Which small-batch recurrent kernel runs a layer: > 0 = workgroups per (32-utterance tile, direction) of the
    reduction-split cooperative kernels (64 = the 32-way kernel on two 16-row halves per tile, forward only), < 0 = the
    column-split kernel with that many utterances per group of 8 workgroups (forward only), 0 = the batch-stationary
    kernels.  A cooperative grid must be resident at once, and the fewer utterances there are, the finer the hidden state is
    cut.  Measured per layer, T = 250, ms (forward; profiles/r05_thresholds_box*.txt): Bp = 32: 1.80 at 8, 1.00 at 16, 0.68 at
    32, 0.53 on half rows; 64: 0.68 at 32, 0.57 on half rows; 128: 0.76 at 32, 1.05 at 16; 256: 1.11 at 16, 1.02 column-split
    by 16; 512: 1.51 by 16, 1.63 by 32; 1024: 2.97 at 4, 2.78 by 32; 3584: 10.1 by 32 = the batch-stationary kernel."""
    if coop_disabled():
        return 0
    if backward:
        # BPTT: 16 unit slices x 2 halves of 16 utterances up to four tiles (0.97 ms per layer at Bp = 32
        # against 1.66 at 8), 16 unit slices up to eight (256: 1.61 against 1.80 at 8), then the 8- and 4-way kernels
        # (beyond 2048 the 4-way kernel's launches of 1024 utterances, ~6.5 ms each per layer, add up to what the
        # batch-stationary BPTT kernel takes for any batch up to 8192: 3072 utterances 179.6 against 178.5 ms per step)
        split = 32 if Bp <= 128 else (16 if Bp <= 256 else (8 if Bp <= 512 else (4 if Bp <= 2048 else 0)))
    elif Bp <= 64 and _COOP_EXCHANGE and os.environ.get('AVSI_REC_HALF', '1') != '0':
        # round 5: the 32-way kernel on 16-row halves (two groups per tile and direction: 128 workgroups per 32 utterances) --
        # the MFMA phase of a step halves; up to 64 utterances the CUs are there (0.68 -> 0.53 ms per layer at 32, 0.57 at 64)
        split = 64
    elif Bp <= 128:
        split = 32
    elif Bp <= 256 and (os.environ.get('AVSI_REC_CS', '1') == '0' or coop_cu_budget() < 128):
        split = 16            # (a small CU share: the reduction-split family, which halves its split until a tile fits)
    elif os.environ.get('AVSI_REC_CS', '1') != '0':
        # (round 5: from 129 utterances on, not from 257 -- the column split by 16 beats the 16-way kernel at 160 .. 256 utterances
        #  on both boxes timed, 1.02 against 1.10 - 1.11 ms per layer, with reserve 1.04 - 1.08 against 1.16 - 1.19, and it does
        #  not need an XCD to itself)
        # 1024 < Bp: resident-sized launches one after the other.  Up to 3584 utterances that still beats the
        # batch-stationary 32-row kernel, whose Bp / 16 workgroups take one full round of the chip whether 132 or 256 of
        # them exist (whole inference step, ms: 2112: 55.7 -> 41.9, 2560: 60.2 -> 50.6, 3072: 65.3 -> 60.5; 3584: a tie)
        split = -16 if Bp <= 512 else (-32 if Bp <= CS_MAX_BATCH else 0)
    else:
        split = 8 if Bp <= 512 else (4 if Bp <= CS_MAX_BATCH else 0)
    # coop_cu_budget(): CUs one cooperative launch may occupy (default: the chip; see set_coop_cu_budget).  A
    # batch that fits the budget in one launch at a coarser split takes that; beyond it the C side cuts the batch
    # into resident-sized launches.  The bounded spin catches an over-subscription anyway.
    forced = os.environ.get('AVSI_COOP_SPLIT_BWD' if backward else 'AVSI_COOP_SPLIT_FWD')     # diagnostics
    if forced and split:
        return int(forced)
    budget = coop_cu_budget()
    if split < 0:
        if budget < 4:                                # 8 workgroups, two to a CU
            return 0
        if split == -16 and 2 * (Bp // 16) * 8 > 2 * budget:      # by 16 does not fit one launch, by 32 does
            split = -32
        return split
    split = _cap_split(split)
    while split > 4 and 2 * (Bp // 32) * split > budget:
        split //= 2
    while split > 4 and 2 * split > budget:       # not even one tile fits at this split
        split //= 2
    return split


def coop_poll(device=None):
    """Non-blocking form of coop_check: raises if a failure has ALREADY been observed on the host, then queues
    the next asynchronous copy of the status words to pinned memory (the flag travels behind the launches).
    Never synchronises, so independent small batches can be in flight on several streams."""
    if not COOP_POLL_RAISES:
        return
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    cur = _lib.stream_ptr().value
    for (dev, st), (host, event) in _COOP_HOST.items():
        if dev != idx:
            continue
        if event.query() and int(host[0]) != 0:
            _coop_recover(idx)
            raise CoopTimeout(_COOP_MSG)
        if st == cur and event.query():
            host.copy_(_COOP_WS[(dev, st)][:1], non_blocking=True)
            event.record()


def coop_status(device=None):
    """One-element device int32 tensor, enqueued on the current stream: non-zero once any cooperative recurrent launch
    issued so far on `device` has given up a bounded wait (the maximum of the sticky status words).  For consumers that
    ship results off the device asynchronously and check the word where the results arrive (inference._WavWriter)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    flags = [ws[:1] for (dev, _), ws in _COOP_WS.items() if dev == idx]
    if not flags:
        return torch.zeros(1, dtype=torch.int32, device=torch.device('cuda', idx))
    return flags[0].clone() if len(flags) == 1 else torch.cat(flags).max().reshape(1)


def coop_check(device=None):
    """Raise if a cooperative recurrent launch on `device` ever gave up waiting for its peer
    workgroups (its outputs are then invalid).  Synchronises with the device."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    flags = [ws[:1] for (dev, _), ws in _COOP_WS.items() if dev == idx]
    if flags and int(torch.cat(flags).max().item()) != 0:
        _coop_recover(idx)
        raise CoopTimeout(_COOP_MSG)


def _coop_recover(idx):
    """A launch that gave up waiting may have left step counters behind (the kernels zero them only on a clean
    end): start over with clean workspaces once the failure has been reported.  Waits for the streams that own a
    workspace (whatever else is resident on the device -- the neighbour that caused the timeout -- is not waited for)."""
    for (dev, st) in list(_COOP_WS):
        if dev == idx:
            (torch.cuda.ExternalStream(st, device=torch.device('cuda', dev)) if st
             else torch.cuda.default_stream(torch.device('cuda', dev))).synchronize()
    for (dev, _), ws in _COOP_WS.items():
        if dev == idx:
            ws.zero_()
    for (dev, _), (host, _) in _COOP_HOST.items():
        if dev == idx:
            host.zero_()


_LOSS_WS = {}


def l1_loss(target, pred, mask, want_grad=False, grad_scale=None):
    """(out3, dpred): out3 = [loss_func, loss_hole, loss_valid] (device float32[3])."""
    _lib.require_cuda(target, pred, mask)
    L = _lib.lib()
    for x in (target, pred, mask):
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise _lib.AvsiError("l1_loss operands must be contiguous float32")
    n = target.numel()
    if pred.numel() != n or mask.numel() != n:
        raise _lib.AvsiError("l1_loss: size mismatch")
    dev = target.device
    ws = _LOSS_WS.get((dev.index, _lib.stream_ptr().value))
    need = L.avsi_l1_loss_workspace_bytes(n)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        _LOSS_WS[(dev.index, _lib.stream_ptr().value)] = ws
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    dpred = torch.empty_like(pred) if want_grad else None
    gs = (1.0 / n) if grad_scale is None else float(grad_scale)
    _lib.check(L.avsi_l1_loss_f32(_lib.ptr(target), _lib.ptr(pred), _lib.ptr(mask), n, _lib.ptr(out3),
                                  _lib.ptr(dpred), gs, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "avsi_l1_loss_f32")
    return out3, dpred


def l1_loss_blend(target, pred_inout, mask, row_scale=None, want_grad=False):
    """Speaker-embedding variants: blends the known bins into ``pred_inout`` IN PLACE
    (seq_mask * logits -> prediction) and returns (out3, dlogits, 1 / sum(1 - mask) as a one-element device tensor) for
    loss_func = loss_hole."""
    _lib.require_cuda(target, pred_inout, mask)
    L = _lib.lib()
    for x in (target, pred_inout, mask):
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise _lib.AvsiError("l1_loss_blend operands must be contiguous float32")
    n, row_len = target.numel(), target.shape[-1]
    if pred_inout.numel() != n or mask.numel() != n:
        raise _lib.AvsiError("l1_loss_blend: size mismatch")
    if row_scale is not None and (row_scale.dtype != torch.float32 or not row_scale.is_contiguous()
                                  or row_scale.numel() != n // row_len):
        raise _lib.AvsiError("l1_loss_blend: row_scale must be contiguous float32 with one value per row")
    dev = target.device
    ws = _LOSS_WS.get((dev.index, _lib.stream_ptr().value))
    need = L.avsi_l1_loss_workspace_bytes(n)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        _LOSS_WS[(dev.index, _lib.stream_ptr().value)] = ws
    out4 = torch.empty(4, dtype=torch.float32, device=dev)
    dlog = torch.empty_like(pred_inout) if want_grad else None
    _lib.check(L.avsi_l1_loss_blend_f32(_lib.ptr(target), _lib.ptr(pred_inout), _lib.ptr(mask), _lib.ptr(row_scale),
                                        row_len, n, _lib.ptr(out4), _lib.ptr(dlog), _lib.ptr(ws), ws.numel() * 4,
                                        _lib.stream_ptr()), "avsi_l1_loss_blend_f32")
    return out4[:3], dlog, out4[3:4]


_CTC_WS = {}


def ctc_loss(logits, labels, labels_lengths, sequence_lengths, grad_scale=1.0, want_grad=False, max_label_len=None):
    """tf.nn.ctc_loss on un-normalised ``logits`` [B, T, C] (last dim contiguous; blank = C - 1):
    returns (loss [B], grad or None) with grad = grad_scale * d loss[b] / d logits (avsi_ctc_loss_f32).
    ``labels`` int32 [B, Lp], ``labels_lengths`` / ``sequence_lengths`` int32 [B], all on the device."""
    _lib.require_cuda(logits, labels, labels_lengths, sequence_lengths)
    L = _lib.lib()
    if logits.dtype != torch.float32 or logits.dim() != 3 or logits.stride(2) != 1:
        raise _lib.AvsiError("ctc_loss: logits must be float32 [B, T, C] with contiguous classes")
    for x in (labels, labels_lengths, sequence_lengths):
        if x.dtype != torch.int32 or not x.is_contiguous():
            raise _lib.AvsiError("ctc_loss: labels and lengths must be contiguous int32")
    B, T, C = logits.shape
    Lp = labels.shape[1]
    Lmax = Lp if max_label_len is None else int(max_label_len)
    need = L.avsi_ctc_loss_workspace_bytes(B, T, Lmax)
    if not need:
        raise _lib.AvsiError("ctc_loss: labellings longer than 127 are not supported")
    key = (logits.device.index, _lib.stream_ptr().value)
    ws = _CTC_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=logits.device)
        _CTC_WS[key] = ws
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty_strided(logits.shape, logits.stride(), dtype=torch.float32, device=logits.device) if want_grad else None
    _lib.check(L.avsi_ctc_loss_f32(_lib.ptr(logits), logits.stride(0), logits.stride(1), B, T, C, _lib.ptr(labels), Lp,
                                   _lib.ptr(labels_lengths), _lib.ptr(sequence_lengths), Lmax, float(grad_scale),
                                   _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "avsi_ctc_loss_f32")
    return loss, grad


def ctc_beam_search(logits, sequence_lengths, beam_width=20, merge_repeated=True):
    """tf.nn.ctc_beam_search_decoder(top_paths=1) on the host (avsi_ctc_beam_search_host_f32): ``logits``
    [B, T, C] (a device tensor is copied to the host), -> (decoded int32 [B, max_len] padded with -1,
    lengths int32 [B], log_prob float32 [B]) as numpy arrays."""
    x = logits.detach().to('cpu', torch.float32).contiguous().numpy() if isinstance(logits, torch.Tensor) \
        else np.ascontiguousarray(logits, dtype=np.float32)
    B, T, C = x.shape
    seq = np.ascontiguousarray(np.asarray(sequence_lengths.cpu() if isinstance(sequence_lengths, torch.Tensor)
                                          else sequence_lengths), dtype=np.int32)
    dec = np.empty((B, T), dtype=np.int32)
    dlen = np.empty(B, dtype=np.int32)
    lp = np.empty(B, dtype=np.float32)
    _lib.check(_lib.lib().avsi_ctc_beam_search_host_f32(x.ctypes.data, T * C, C, B, T, C, seq.ctypes.data, int(beam_width),
                                                        int(bool(merge_repeated)), dec.ctypes.data, T, dlen.ctypes.data,
                                                        lp.ctypes.data), "avsi_ctc_beam_search_host_f32")
    return dec[:, :max(int(dlen.max()), 0)], dlen, lp


def edit_distance(hyp, truth, normalize=True):
    """tf.edit_distance for one pair of label sequences (host bookkeeping of the `per` diagnostic)."""
    hyp, truth = list(hyp), list(truth)
    d = list(range(len(truth) + 1))
    for i, h in enumerate(hyp, 1):
        prev, d[0] = d[0], i
        for j, t in enumerate(truth, 1):
            prev, d[j] = d[j], min(d[j] + 1, d[j - 1] + 1, prev + (h != t))
    if not normalize:
        return float(d[-1])
    if not truth:
        return float('inf') if hyp else 0.0
    return d[-1] / len(truth)


_WS = {}


def _workspace(dev, nbytes):
    """One growing scratch buffer per (device, stream) (split-K slabs, column-sum partials): work on
    one stream is ordered and can share it, concurrent streams each get their own."""
    key = (dev.index, _lib.stream_ptr().value)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        _WS[key] = ws
    return ws


def splitk_for(rows):
    """Reduction slabs for a weight-gradient GEMM over `rows` = T * Bp rows: long reductions get 64
    slabs of >= 4096 rows; short ones (small batches) are still cut into ~1024-row slabs so that the
    few output tiles (4 x 16 for a 512 x 2048 gradient) spread over the chip."""
    # (re-timed in round 5 at 8000 rows, training step at 32 utterances: 1024-row slabs 5.69 ms; 512: 5.84; 2048: 6.22; 256: 5.96)
    return max(1, min(64, rows // 4096)) if rows >= 65536 else max(1, min(16, rows // 1024))


def gemm_splitk(a, b, out, trans_a=False, trans_b=False, m=None, n=None, k=None, alpha=1.0, splits=16):
    """out[M, N] (contiguous) = alpha * op(A) . op(B), reduction split into `splits` slabs
    (avsi_gemm_splitk_f32).  For weight gradients: K = all T * Bp rows."""
    _lib.require_cuda(a, b, out)
    L = _lib.lib()
    M = m if m is not None else (a.shape[1] if trans_a else a.shape[0])
    K = k if k is not None else (a.shape[0] if trans_a else a.shape[1])
    N = n if n is not None else (b.shape[0] if trans_b else b.shape[1])
    if not out.is_contiguous() or out.numel() != M * N:
        raise _lib.AvsiError("gemm_splitk: out must be contiguous [M, N]")
    need = L.avsi_gemm_splitk_workspace_bytes(M, N, splits)
    ws = _workspace(a.device, need)
    _lib.check(L.avsi_gemm_splitk_f32(int(trans_a), int(trans_b), M, N, K, float(alpha), _lib.ptr(a), a.stride(0),
                                      _lib.ptr(b), b.stride(0), _lib.ptr(out), int(splits), _lib.ptr(ws),
                                      ws.numel() * 4, _lib.stream_ptr()), "avsi_gemm_splitk_f32")
    return out


def blstm_rec_bwd(dhout, reserve, whbt, dz, split=None):
    """BPTT through the recurrence of one layer (avsi_blstm_rec_bwd_f32, or the small-batch cooperative
    avsi_blstm_rec_bwd_coop_f32: `split` 4 / 8 forces it, 0 forbids it, None = by Bp).
    dhout [T, Bp, 512], reserve [T, Bp, 2, 5, 256], whbt [2 * 262144] -> dz [T, Bp, 2048]."""
    _lib.require_cuda(dhout, reserve, whbt, dz)
    T, Bp = dhout.shape[0], dhout.shape[1]
    ok = (tuple(dhout.shape) == (T, Bp, 512) and tuple(reserve.shape) == (T, Bp, 2, 5, 256)
          and tuple(dz.shape) == (T, Bp, 2048) and whbt.numel() == 2 * 262144)
    if not ok or not (dhout.is_contiguous() and reserve.is_contiguous() and dz.is_contiguous() and whbt.is_contiguous()):
        raise _lib.AvsiError("blstm_rec_bwd: bad operand shapes / strides")
    split = coop_split(Bp, backward=True) if split is None else int(split)
    if split:
        L = _lib.lib()
        need = L.avsi_blstm_rec_fwd_coop_workspace_bytes(Bp)
        if split >= 16 and _COOP_EXCHANGE:      # fine splits: room for the exchange copy of dz at its fixed offset
            need = COOP_EXCHANGE_OFFSET + L.avsi_blstm_rec_bwd_coop_exchange_bytes(T, Bp)
        ws = _coop_ws(dhout.device, Bp, need)
        _lib.check(L.avsi_blstm_rec_bwd_coop_f32(_lib.ptr(dhout), _lib.ptr(reserve), _lib.ptr(whbt), _lib.ptr(dz),
                                                 T, Bp, split, coop_cu_budget(), _lib.ptr(ws), need, _lib.stream_ptr()),
                   "avsi_blstm_rec_bwd_coop_f32")
        return dz
    _lib.check(_lib.lib().avsi_blstm_rec_bwd_f32(_lib.ptr(dhout), _lib.ptr(reserve), _lib.ptr(whbt), _lib.ptr(dz),
                                                 T, Bp, _lib.stream_ptr()), "avsi_blstm_rec_bwd_f32")
    return dz


def sgd_momentum(param, grad, accum, lr, momentum=0.9, grad_scale=1.0, l2=0.0, skip=None):
    """In-place tf.train.GradientDescentOptimizer (``accum`` None) / tf.train.MomentumOptimizer step on flat float32 buffers
    (avsi_sgd_momentum_f32; reference models.py:170-176).  ``skip`` as for adam_tf."""
    _lib.require_cuda(param, grad, accum, skip)
    if skip is not None and (skip.dtype != torch.float32 or not skip.is_contiguous() or skip.numel() > 8):
        raise _lib.AvsiError("sgd_momentum: skip must be at most 8 contiguous float32 words")
    _lib.check(_lib.lib().avsi_sgd_momentum_f32(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(accum), param.numel(), float(lr),
                                                float(momentum), float(grad_scale), float(l2), _lib.ptr(skip),
                                                0 if skip is None else skip.numel(), _lib.stream_ptr()), "avsi_sgd_momentum_f32")


def relayout_rows(src, dst, B, T, C, dst_cols, src_strides, dst_strides, row_scale=None, scale_strides=(0, 0)):
    """dst[b, t, :dst_cols] = src[b, t, :C] * row_scale[b, t] (zero beyond C); strides are (b, t) element strides."""
    _lib.require_cuda(src, dst, row_scale)
    _lib.check(_lib.lib().avsi_relayout_rows_f32(_lib.ptr(src), src_strides[0], src_strides[1], _lib.ptr(dst),
                                                 dst_strides[0], dst_strides[1], B, T, C, dst_cols,
                                                 _lib.ptr(row_scale), scale_strides[0], scale_strides[1],
                                                 _lib.stream_ptr()), "avsi_relayout_rows_f32")
    return dst


def colsum(x, out, m=None, n=None):
    """out[n] = sum over rows of x[M, ld] (avsi_colsum_f32)."""
    _lib.require_cuda(x, out)
    L = _lib.lib()
    M = x.shape[0] if m is None else m
    N = x.shape[1] if n is None else n
    need = L.avsi_colsum_workspace_bytes(M, N)
    ws = _workspace(x.device, need)
    _lib.check(L.avsi_colsum_f32(_lib.ptr(x), x.stride(0), M, N, _lib.ptr(out), _lib.ptr(ws), ws.numel() * 4,
                                 _lib.stream_ptr()), "avsi_colsum_f32")
    return out


def adam_tf(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0, skip=None):
    """In-place tf.train.AdamOptimizer step on flat float32 buffers (avsi_adam_tf_guarded_f32).  ``skip``: device float32
    words (<= 8, the step guard); the update does nothing when any of them is not exactly zero."""
    _lib.require_cuda(param, grad, m, v, skip)
    n = param.numel()
    if skip is not None and (skip.dtype != torch.float32 or not skip.is_contiguous() or skip.numel() > 8):
        raise _lib.AvsiError("adam_tf: skip must be at most 8 contiguous float32 words")
    _lib.check(_lib.lib().avsi_adam_tf_guarded_f32(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), n, float(lr),
                                                   float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                                                   float(l2), _lib.ptr(skip), 0 if skip is None else skip.numel(),
                                                   _lib.stream_ptr()), "avsi_adam_tf_guarded_f32")


def step_guard(loss, out, device=None, coop=True):
    """out[0:2] <- [NaN unless `loss` (one-element device tensor, or None) is finite, 1 if a cooperative recurrent launch
    issued so far on the device has given up a bounded wait else 0] (avsi_step_guard_f32), on the current stream.
    ``coop=False``: a model that launches no cooperative kernel (the U-Net) -- word 1 is 0 whatever an earlier BLSTM
    model of this process left in the sticky status words; a stale timeout there must not void its steps."""
    _lib.require_cuda(loss, out)
    idx = out.device.index
    flags = [ws for (dev, _), ws in _COOP_WS.items() if dev == idx] if coop else []
    if len(flags) > 2:
        flags = [coop_status(out.device)]
    a = flags[0] if flags else None
    b = flags[1] if len(flags) > 1 else None
    _lib.check(_lib.lib().avsi_step_guard_f32(_lib.ptr(loss), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.stream_ptr()),
               "avsi_step_guard_f32")
    return out


# ---------------------------------------------------------------------------- U-Net building blocks
_ZEROS = {}


_CONV_WS = {}           # split-K slabs of conv2d, per (device, stream)
_CONV_SPLITK = os.environ.get('AVSI_CONV_SPLITK', '1') != '0'


def conv2d_supported(c0, c1):
    """Layers the implicit-GEMM convolution takes (avsi_conv2d_f32): channel counts multiples of 16."""
    return c0 % 16 == 0 and c1 % 16 == 0 and c0 + c1 >= 16


def conv2d(src0, c0, src1, c1, B, H, W, k, filt, bias, out, cout):
    """out [B*H*W, ld] = conv2d(concat(src0, up2x(src1)), filt [k*k*(c0+c1), ldf]) + bias, no im2col matrix."""
    _lib.require_cuda(src0, src1, filt, out)
    z = _ZEROS.get(out.device.index)
    if z is None:
        z = _ZEROS[out.device.index] = torch.zeros(64, dtype=torch.float32, device=out.device)
    L = _lib.lib()
    # few output tiles (small batch, deep layers): cut the reduction so that the launch fills the chip (avsi_conv2d_splitk_f32)
    splits = L.avsi_conv2d_splitk_suggest(B, H, W, k, c0, c1, cout) if _CONV_SPLITK and out.stride(0) == cout else 1
    if splits > 1:
        need = L.avsi_conv2d_splitk_workspace_bytes(B, H, W, out.stride(0), splits)
        key = (out.device.index, _lib.stream_ptr().value)
        ws = _CONV_WS.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = _CONV_WS[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=out.device)
        _lib.check(L.avsi_conv2d_splitk_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1), c1,
                                            src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(filt), filt.stride(0),
                                            _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0), _lib.ptr(z), splits, _lib.ptr(ws),
                                            ws.numel() * 4, _lib.stream_ptr()), "avsi_conv2d_splitk_f32")
        return out
    _lib.check(L.avsi_conv2d_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1), c1,
                                 src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(filt),
                                 filt.stride(0), _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0), _lib.ptr(z),
                                 _lib.stream_ptr()), "avsi_conv2d_f32")
    return out


_CONV_BN = os.environ.get('AVSI_CONV_BN', '1') != '0'


def conv2d_bn_supported(k, c0, c1, cout, B, H, W, ldo):
    """Layers whose batch statistics come out of the convolution's epilogue (avsi_conv2d_bn_f32): the 16-wide-MFMA layers and
    the implicit-GEMM layers that fill the chip without a split reduction (a split launch holds partial sums only)."""
    if not _CONV_BN:
        return False
    if conv2d_thin_mfma_supported(k, c0, c1, cout, H, W):
        return True
    if not conv2d_supported(c0, c1):
        return False
    return not (_CONV_SPLITK and ldo == cout and _lib.lib().avsi_conv2d_splitk_suggest(B, H, W, k, c0, c1, cout) > 1)


def _bn4(src1_bn):
    """(mean, rstd, gamma, beta) device tensors -> a C array of four pointers (kept alive by the caller's tuple)."""
    if src1_bn is None:
        return None
    _lib.require_cuda(*src1_bn)
    return (ctypes.c_void_p * 4)(*[t.data_ptr() for t in src1_bn])


def conv2d_bn(src0, c0, src1, c1, B, H, W, k, filt, bias, out, cout, mean, rstd, eps=1e-3, src1_bn=None):
    """out = conv2d(concat(src0, up2x(src1)), filt) + bias AND mean / rstd of its cout channels over all B*H*W rows.
    src1_bn = (mean, rstd, gamma, beta): src1 is the RAW convolution output of the layer below, normalised and activated
    (LeakyReLU 0.2) while it is staged (16-wide-MFMA route only)."""
    _lib.require_cuda(src0, src1, filt, out, mean, rstd)
    z = _ZEROS.get(out.device.index)
    if z is None:
        z = _ZEROS[out.device.index] = torch.zeros(64, dtype=torch.float32, device=out.device)
    L = _lib.lib()
    ws = _workspace(out.device, L.avsi_conv2d_bn_workspace_bytes(B, H, W, k, c0, c1, cout))
    bn4 = _bn4(src1_bn)
    _lib.check(L.avsi_conv2d_bn_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1), c1,
                                    src1.stride(0) if src1 is not None else 0, bn4, B, H, W, k, _lib.ptr(filt), filt.stride(0),
                                    _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0), _lib.ptr(z), float(eps), _lib.ptr(mean),
                                    _lib.ptr(rstd), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "avsi_conv2d_bn_f32")
    return out


def conv2d_thin_mfma_supported(k, c0, c1, cout, H, W):
    """Few-channel layers the 16-wide-MFMA convolution takes (avsi_conv2d_thin_mfma_f32)."""
    return os.environ.get('AVSI_CONV_THIN_MFMA', '1') != '0' and \
        bool(_lib.lib().avsi_conv2d_thin_mfma_supported(int(k), int(c0), int(c1), int(cout), int(H), int(W)))


def conv2d_thin_mfma_plain_supported(k, c0, c1, cout, H, W):
    """Shapes the plain 16-wide-MFMA convolution takes: the forward layers above and their input-gradient convolutions."""
    return os.environ.get('AVSI_CONV_THIN_MFMA', '1') != '0' and \
        bool(_lib.lib().avsi_conv2d_thin_mfma_plain_supported(int(k), int(c0), int(c1), int(cout), int(H), int(W)))


def conv2d_thin_mfma(src0, c0, src1, c1, B, H, W, k, filt, bias, out, cout):
    _lib.require_cuda(src0, src1, filt, out)
    z = _ZEROS.get(out.device.index)
    if z is None:
        z = _ZEROS[out.device.index] = torch.zeros(64, dtype=torch.float32, device=out.device)
    _lib.check(_lib.lib().avsi_conv2d_thin_mfma_f32(_lib.ptr(src0), c0, src0.stride(0), _lib.ptr(src1), c1,
                                                    src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(filt),
                                                    filt.stride(0), _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0), _lib.ptr(z),
                                                    _lib.stream_ptr()), "avsi_conv2d_thin_mfma_f32")
    return out


_THIN = {(7, 1, 0, 16), (3, 1, 16, 1), (1, 1, 0, 1)}


def conv2d_thin_supported(k, c0, c1, cout):
    return (k, c0, c1, cout) in _THIN


def conv2d_thin(src0, c0, src1, c1, B, H, W, k, filt, bias, out, cout):
    """Direct convolution of the thin full-resolution U-Net layers (avsi_conv2d_thin_f32)."""
    _lib.require_cuda(src0, src1, filt, out)
    _lib.check(_lib.lib().avsi_conv2d_thin_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1),
                                               c1, src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(filt),
                                               filt.stride(0), _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0),
                                               _lib.stream_ptr()), "avsi_conv2d_thin_f32")
    return out


def unet_tail_supported(H, W):
    return os.environ.get('AVSI_UNET_TAIL', '1') != '0' and H % 8 == 0 and W % 32 == 0


def unet_tail(src0, src1, B, H, W, filt, bias, gamma, beta, w_out, b_out, seq_len, conv, pred, logits=None, eps=1e-3,
              src1_bn=None):
    """Inference tail of the U-Net (avsi_unet_tail_f32): 3 x 3 convolution 1 + 16 -> 1 with its batch statistics, then batch
    norm + LeakyReLU + the 1 x 1 output convolution + the sequence mask in one pass.  conv / pred / logits: [B*H*W] floats."""
    _lib.require_cuda(src0, src1, filt, conv, pred, logits, seq_len)
    L = _lib.lib()
    if seq_len.dtype != torch.int64 or not seq_len.is_contiguous():
        raise _lib.AvsiError("unet_tail: seq_len must be a contiguous int64 device tensor")
    ws = _workspace(pred.device, L.avsi_unet_tail_workspace_bytes(B, H, W))
    _lib.check(L.avsi_unet_tail_f32(_lib.ptr(src0), src0.stride(0), _lib.ptr(src1), src1.stride(0), _bn4(src1_bn), B, H, W, _lib.ptr(filt),
                                    filt.stride(0), _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(w_out),
                                    _lib.ptr(b_out), _lib.ptr(seq_len), _lib.ptr(conv), _lib.ptr(logits), _lib.ptr(pred),
                                    _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "avsi_unet_tail_f32")
    return pred


def im2col(src0, c0, src1, c1, B, H, W, k, col, kc):
    """col[B*H*W, kc] <- patches of src0 [B*H*W, ld0] ++ 2x-up-sampled src1 [B*H/2*W/2, ld1] (avsi_im2col_f32)."""
    _lib.require_cuda(src0, src1, col)
    _lib.check(_lib.lib().avsi_im2col_f32(_lib.ptr(src0), int(c0), 0 if src0 is None else src0.stride(0), _lib.ptr(src1),
                                          int(c1), 0 if src1 is None else src1.stride(0), B, H, W, k, _lib.ptr(col), kc,
                                          _lib.stream_ptr()), "avsi_im2col_f32")
    return col


def conv2d_wgrad_supported(c0, c1, rows):
    return c0 % 4 == 0 and c1 % 4 == 0 and c0 + c1 >= 4 and rows % 16 == 0


def conv2d_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, dw, splits):
    """dw [k*k*(c0+c1), ld] = im2col(concat(src0, up2x(src1)))^T . dy, no im2col matrix (avsi_conv2d_wgrad_f32)."""
    _lib.require_cuda(src0, src1, dy, dw)
    L = _lib.lib()
    z = _ZEROS.get(dw.device.index)
    if z is None:
        z = _ZEROS[dw.device.index] = torch.zeros(64, dtype=torch.float32, device=dw.device)
    ws = _workspace(dw.device, L.avsi_conv2d_wgrad_workspace_bytes(c0, c1, k, cout, splits))
    _lib.check(L.avsi_conv2d_wgrad_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1), c1,
                                       src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(dy), dy.stride(0),
                                       cout, _lib.ptr(dw), dw.stride(0), int(splits), _lib.ptr(z), _lib.ptr(ws),
                                       ws.numel() * 4, _lib.stream_ptr()), "avsi_conv2d_wgrad_f32")
    return dw


def conv2d_thin_mfma_wgrad_supported(k, c0, c1, cout, H, W, dw=None):
    return bool(_lib.lib().avsi_conv2d_thin_mfma_wgrad_supported(k, c0, c1, cout, H, W)) and (dw is None or dw.stride(0) == cout) \
        and os.environ.get('AVSI_THIN_MFMA_WGRAD', '1') != '0'


def conv2d_thin_mfma_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, dw):
    """Filter gradient of a few-channel layer on the 16-wide MFMA (avsi_conv2d_thin_mfma_wgrad_f32)."""
    _lib.require_cuda(src0, src1, dy, dw)
    L = _lib.lib()
    z = _ZEROS.get(dw.device.index)
    if z is None:
        z = _ZEROS[dw.device.index] = torch.zeros(64, dtype=torch.float32, device=dw.device)
    ws = _workspace(dw.device, L.avsi_conv2d_thin_mfma_wgrad_workspace_bytes(c0, c1, k, cout, B, H, W))
    _lib.check(L.avsi_conv2d_thin_mfma_wgrad_f32(_lib.ptr(src0), c0, src0.stride(0), _lib.ptr(src1), c1,
                                                 src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(dy), dy.stride(0),
                                                 cout, _lib.ptr(dw), dw.stride(0), _lib.ptr(z), _lib.ptr(ws), ws.numel() * 4,
                                                 _lib.stream_ptr()), "avsi_conv2d_thin_mfma_wgrad_f32")
    return dw


def conv2d_thin_wgrad(src0, c0, src1, c1, B, H, W, k, dy, cout, dw):
    """Direct filter gradient of a thin layer (avsi_conv2d_thin_wgrad_f32)."""
    _lib.require_cuda(src0, src1, dy, dw)
    L = _lib.lib()
    ws = _workspace(dw.device, L.avsi_conv2d_thin_wgrad_workspace_bytes(c0, c1, k, cout, B, H, W))
    _lib.check(L.avsi_conv2d_thin_wgrad_f32(_lib.ptr(src0), c0, src0.stride(0) if src0 is not None else 0, _lib.ptr(src1), c1,
                                            src1.stride(0) if src1 is not None else 0, B, H, W, k, _lib.ptr(dy), dy.stride(0),
                                            cout, _lib.ptr(dw), dw.stride(0), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "avsi_conv2d_thin_wgrad_f32")
    return dw


def conv2d_thin_dx_coarse(dy, filt, dsrc1, accumulate, B, H, W):
    """Gradient of the 1 + 16 -> 1 layer w.r.t. its up-sampled source (avsi_conv2d_thin_dx_coarse_f32)."""
    _lib.require_cuda(dy, filt, dsrc1)
    _lib.check(_lib.lib().avsi_conv2d_thin_dx_coarse_f32(_lib.ptr(dy), dy.stride(0), _lib.ptr(filt), filt.stride(0),
                                                         _lib.ptr(dsrc1), dsrc1.stride(0), int(accumulate), B, H, W,
                                                         _lib.stream_ptr()), "avsi_conv2d_thin_dx_coarse_f32")
    return dsrc1


def split_sumpool(dx, dsrc0, c0, acc0, dsrc1, c1, acc1, B, H, W):
    """dX of concat(src0, up2x(src1)) [B*H*W, c0 + c1] -> (+)= dsrc0 and (+)= 2x2-summed dsrc1 (avsi_split_sumpool_f32)."""
    _lib.require_cuda(dx, dsrc0, dsrc1)
    _lib.check(_lib.lib().avsi_split_sumpool_f32(_lib.ptr(dx), dx.stride(0), _lib.ptr(dsrc0), int(c0),
                                                 dsrc0.stride(0) if dsrc0 is not None else 0, int(acc0), _lib.ptr(dsrc1),
                                                 int(c1), dsrc1.stride(0) if dsrc1 is not None else 0, int(acc1), B, H, W,
                                                 _lib.stream_ptr()), "avsi_split_sumpool_f32")


def col2im(dcol, kc, dsrc0, c0, dsrc1, c1, B, H, W, k, accumulate0=False, accumulate1=False):
    _lib.require_cuda(dcol, dsrc0, dsrc1)
    pitch = lambda t, c: t.stride(0) if t is not None else -(-int(c) // 4) * 4      # a skipped gradient keeps its pitch
    _lib.check(_lib.lib().avsi_col2im_f32(_lib.ptr(dcol), kc, _lib.ptr(dsrc0), int(c0), pitch(dsrc0, c0),
                                          _lib.ptr(dsrc1), int(c1), pitch(dsrc1, c1), B, H, W, k, int(accumulate0),
                                          int(accumulate1), _lib.stream_ptr()), "avsi_col2im_f32")


def colstats(x, C, mean, rstd, eps=1e-3):
    L = _lib.lib()
    ws = _workspace(x.device, L.avsi_unet_workspace_bytes(C))
    _lib.check(L.avsi_colstats_f32(_lib.ptr(x), x.shape[0], C, x.stride(0), float(eps), _lib.ptr(mean), _lib.ptr(rstd),
                                   _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "avsi_colstats_f32")


def bn_act_pool(x, B, H, W, C, pooled, y=None, mean=None, rstd=None, gamma=None, beta=None, act=0):
    """act(bn(x)) and its 2 x 2 max pooling in one pass (avsi_bn_act_pool_f32); y = None: inference, the
    full-resolution activation is not kept."""
    _lib.check(_lib.lib().avsi_bn_act_pool_f32(_lib.ptr(x), B, H, W, C, x.stride(0), _lib.ptr(mean), _lib.ptr(rstd),
                                               _lib.ptr(gamma), _lib.ptr(beta), int(act), _lib.ptr(y), _lib.ptr(pooled),
                                               _lib.stream_ptr()), "avsi_bn_act_pool_f32")
    return pooled


def conv2d_thin_relu_pool(src0, B, H, W, k, filt, bias, out, cout):
    """7 x 7 one-channel convolution + bias + ReLU + 2 x 2 max pooling (avsi_conv2d_thin_relu_pool_f32)."""
    _lib.check(_lib.lib().avsi_conv2d_thin_relu_pool_f32(_lib.ptr(src0), src0.stride(0), B, H, W, k, _lib.ptr(filt),
                                                         filt.stride(0), _lib.ptr(bias), cout, _lib.ptr(out), out.stride(0),
                                                         _lib.stream_ptr()), "avsi_conv2d_thin_relu_pool_f32")
    return out


def bn_act(x, C, y, mean=None, rstd=None, gamma=None, beta=None, act=0):
    _lib.check(_lib.lib().avsi_bn_act_f32(_lib.ptr(x), x.shape[0], C, x.stride(0), _lib.ptr(mean), _lib.ptr(rstd),
                                          _lib.ptr(gamma), _lib.ptr(beta), int(act), _lib.ptr(y), _lib.stream_ptr()),
               "avsi_bn_act_f32")
    return y


def bn_act_bwd(x, dy, C, dx, mean=None, rstd=None, gamma=None, beta=None, act=0, dgamma=None, dbeta=None):
    L = _lib.lib()
    ws = _workspace(x.device, L.avsi_unet_workspace_bytes(C))
    _lib.check(L.avsi_bn_act_bwd_f32(_lib.ptr(x), _lib.ptr(dy), x.shape[0], C, x.stride(0), _lib.ptr(mean),
                                     _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), int(act), _lib.ptr(dx),
                                     _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), ws.numel() * 4,
                                     _lib.stream_ptr()), "avsi_bn_act_bwd_f32")
    return dx


def bn_act_pool_bwd(x, dpooled, B, H, W, C, dx, mean=None, rstd=None, gamma=None, beta=None, act=0, dgamma=None, dbeta=None,
                    dbias=None):
    """Backward of bn_act_pool from the POOLED gradient (avsi_bn_act_pool_bwd_f32): dx, and dgamma / dbeta (batch norm) or
    dbias (no batch norm).  The full-resolution activation and its gradient are not needed."""
    _lib.require_cuda(x, dpooled, dx)
    L = _lib.lib()
    ws = _workspace(x.device, L.avsi_unet_workspace_bytes(C))
    _lib.check(L.avsi_bn_act_pool_bwd_f32(_lib.ptr(x), _lib.ptr(dpooled), B, H, W, C, x.stride(0), _lib.ptr(mean), _lib.ptr(rstd),
                                          _lib.ptr(gamma), _lib.ptr(beta), int(act), _lib.ptr(dx), _lib.ptr(dgamma),
                                          _lib.ptr(dbeta), _lib.ptr(dbias), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "avsi_bn_act_pool_bwd_f32")
    return dx


def maxpool2(x, y, B, H, W, C):
    _lib.check(_lib.lib().avsi_maxpool2_f32(_lib.ptr(x), _lib.ptr(y), B, H, W, C, x.stride(0), _lib.stream_ptr()),
               "avsi_maxpool2_f32")
    return y


def maxpool2_bwd(x, dy, dx, B, H, W, C):
    _lib.check(_lib.lib().avsi_maxpool2_bwd_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), B, H, W, C, x.stride(0),
                                                _lib.stream_ptr()), "avsi_maxpool2_bwd_f32")
    return dx
