"""Inference driver on MI355X: TFRecords in, inpainted ``enhanced/<prefix>.wav`` files out.

``infer`` keeps the signature, directory contract and console output of the reference
(``av_speech_inpainting/inference.py:20-170``): the model directory holds ``config.txt``,
``audio_features_{mean,std}.npy`` and the ``sinet`` checkpoint; each utterance is written to
``<audio_path>/<sample_path>/enhanced/<out_file_prefix>.wav`` as 16 kHz int16 truncated to
``seq_len * 192`` samples.

Phase: ``oracle_phase=True`` uses the target phase everywhere; otherwise the phase of the masked
target STFT (zero inside gaps), whose gap frames are then refined by LWS phase reconstruction exactly
as the reference does with the ``lws`` package (inference.py:119,141-154) -- here ``avsi_amd.lws``,
the gfx950 implementation of the published algorithm (see that module: unpinned against the package).
"""
import ctypes
import os
import sys
import threading
from glob import glob

import numpy as np

from . import _lib, ops, parallel
from . import lws as lws_mod
from .config_utils import check_trainconfiguration, load_configfile
from .dataset_reader import DataManager, OutOfRangeError
from .training import EMBEDDING_SIZE, build_model, unpack_batch, uses_embeddings


class _WavWriter(object):
    """Turns finished batches into ``<audio_path>/<sample>/enhanced/<prefix>.wav`` files (16 kHz int16, ``seq_len * 192``
    samples: the reference's loop, inference.py:159-162) WITHOUT holding up the launch thread: a batch is submitted as
    a device tensor; its waveforms are copied into a pinned host buffer on a stream of their own (high priority: see
    dataset_reader._Uploader), and a few host threads wait for that copy and write the files natively, outside the
    interpreter lock (avsi_wav_write_batch_int16_host: directory creation, float -> int16, header, write).
    The reference writes per utterance on the thread that also feeds the GPU; at batch 1024 that thread spent more time in
    ``wavfile.write`` (50 us a file) and in the pageable device-to-host copy (33 ms per 196 MB) than the GPU on the batch.
    A batch whose cooperative-kernel / LWS status word is non-zero when its copy arrives is NOT written and fails the run
    (``close`` raises): outputs of a kernel that gave up a bounded wait are invalid."""

    def __init__(self, audio_path, prefix, threads, device):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        self.torch = torch
        self.audio_path, self.prefix = audio_path, prefix
        self.device = device
        self.threads = max(1, threads)
        self.pool = ThreadPoolExecutor(max_workers=self.threads, thread_name_prefix='avsi-wav')
        self.stream = torch.cuda.Stream(device=device, priority=-1)
        self.slots = []                 # [host buffer, status words, event, pending futures]
        self.nslots = None              # decided by the first batch's size
        self.next = 0
        self.errors = []
        self.lock = threading.Lock()

    def _slot(self, shape):
        torch = self.torch
        # three page-locked buffers in rotation for large batches; small ones (32 utterances: 6 MB) get up to sixteen
        # within the same ~64 MB, so that the launch thread, which submits the batches of a whole LWS / model group in
        # a row, does not wait for the files of the batch three submissions back
        nbytes = 4
        for d in shape:
            nbytes *= int(d)
        if self.nslots is None:
            self.nslots = max(3, min(16, (64 << 20) // max(1, nbytes)))
        if len(self.slots) < self.nslots:
            self.slots.append([None, torch.zeros(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event(), []])
        slot = self.slots[self.next % self.nslots]
        self.next += 1
        for f in slot[3]:               # the files of the batch that used this buffer a rotation ago are on disk
            f.result()
        slot[3] = []
        if slot[0] is None or tuple(slot[0].shape[1:]) != tuple(shape[1:]) or slot[0].shape[0] < shape[0]:
            # (allocated page-locked, not `.pin_memory()` of a pageable tensor: that is a second allocation plus a copy of
            #  pages nobody has touched yet -- 14 ms per 6 MB slot, a quarter of a 4096-utterance run at batch 32)
            slot[0] = torch.empty(tuple(shape), dtype=torch.float32, pin_memory=True)
        return slot

    def _write(self, slot, lo, hi, paths, counts):
        try:
            slot[2].synchronize()                              # the waveforms (and the status words) have arrived
            if int(slot[1][0]) != 0 or int(slot[1][1]) != 0:
                raise _lib.AvsiError("a cooperative recurrent kernel or an LWS pipeline stage gave up a bounded wait: the outputs of "
                                     "this batch are invalid and were not written")
            host = slot[0]
            arr = (ctypes.c_char_p * (hi - lo))(*paths[lo:hi])
            n = np.ascontiguousarray(counts[lo:hi], dtype=np.int32)
            rc = _lib.lib().avsi_wav_write_batch_int16_host(arr, host.data_ptr() + lo * host.stride(0) * 4, host.stride(0),
                                                           n.ctypes.data, hi - lo, 16000, 1)
            if rc != _lib.AVSI_OK:
                raise IOError("could not write the enhanced wavs of %s ..." % paths[lo].decode())
        except Exception as e:          # surfaces in close(): a failed write must not pass silently
            with self.lock:
                self.errors.append(e)

    def submit(self, wavs, sample_dirs, seq_lens, status=()):
        """wavs: device tensor [B, n] (results of work already enqueued on the current stream); ``status``: device int
        tensors whose first word must be zero for the batch to be valid (cooperative kernels, LWS)."""
        torch = self.torch
        B = int(wavs.shape[0])
        slot = self._slot(wavs.shape)
        cur = torch.cuda.current_stream(wavs.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            slot[0][:B].copy_(wavs, non_blocking=True)
            slot[1].zero_()
            for i, st in enumerate(status[:2]):
                slot[1][i:i + 1].copy_(st.reshape(-1)[:1], non_blocking=True)
                st.record_stream(self.stream)      # a copy taken on the launch stream: its block must outlive this read
            slot[2].record(self.stream)
        wavs.record_stream(self.stream)
        paths = [os.path.join(self.audio_path, d.decode(), 'enhanced', self.prefix + '.wav').encode() for d in sample_dirs]
        counts = np.minimum(np.asarray(seq_lens, dtype=np.int64) * 192, int(wavs.shape[1]))
        step = -(-B // self.threads)
        slot[3] = [self.pool.submit(self._write, slot, lo, min(lo + step, B), paths, counts) for lo in range(0, B, step)]

    def close(self):
        for slot in self.slots:
            for f in slot[3]:
                f.result()
        self.pool.shutdown()
        if self.errors:
            raise self.errors[0]


def _adjacent(parts):
    """True when `parts` are consecutive row ranges, from row 0 on, of one contiguous tensor (views taken with [a:b])."""
    base = parts[0]._base
    if base is None or not base.is_contiguous() or any(p._base is not base or not p.is_contiguous() for p in parts):
        return False
    at = base.data_ptr()
    for p in parts:
        if p.data_ptr() != at or tuple(p.shape[1:]) != tuple(base.shape[1:]):
            return False
        at += p.numel() * p.element_size()
    return True


def infer(model_path, data_path_test, audio_path, out_file_prefix, norm=True, oracle_phase=False, batch_size=1):
    from time import perf_counter
    stamps = [('start', perf_counter())] if os.environ.get('AVSI_INFER_TIMING') else None

    def stamp(name):
        if stamps is not None:
            stamps.append((name, perf_counter()))
    config = check_trainconfiguration(load_configfile(os.path.join(model_path, 'config.txt')))
    rank, world = parallel.init()

    dm = DataManager(num_audio_samples=config['audio_len'], audio_feat_size=config['audio_feat_dim'],
                     video_feat_size=config['video_feat_dim'], buffer_size=4000, mode='fixed',
                     embedding_size=EMBEDDING_SIZE if uses_embeddings(config) else None)
    test_files = sorted(glob(os.path.join(data_path_test, '*.tfrecord')))
    import torch
    _, test_it = dm.get_iterator(dm.get_dataset(test_files, shuffle=False), batch_size=batch_size, n_epochs=1,
                                 drop_remainder=False, shard=(rank, world),
                                 device=torch.device('cuda', torch.cuda.current_device()))

    if norm:
        audio_feat_mean = np.load(os.path.join(model_path, 'audio_features_mean.npy'))
        audio_feat_std = np.load(os.path.join(model_path, 'audio_features_std.npy'))
    else:
        audio_feat_mean = np.zeros(config['audio_feat_dim'])
        audio_feat_std = np.ones(config['audio_feat_dim'])

    stamp('reader')
    print('Building speech inpainting inference model:')
    model = build_model(config, audio_feat_mean, audio_feat_std, is_training=False)
    print('Model building done.')
    print('done.\n')
    print('Restore weigths:')
    try:
        model.variables.restore(os.path.join(model_path, 'sinet'))
    except ValueError as e:
        print(str(e))
        sys.exit(2)
    print('done.\n')
    stamp('model + checkpoint')

    # LWS module initialization (reference inference.py:119)
    lws_processor = lws_mod.lws(384, 192, fftsize=512, mode='speech')

    total_samples = 0
    loss_list = []
    device = torch.device('cuda', torch.cuda.current_device())
    writer = _WavWriter(audio_path, out_file_prefix, int(os.environ.get('AVSI_WAV_THREADS', '6')), device)
    # The LWS sweeps are a pipeline of ~100 stages per utterance: batches are collected until LWS_GROUP utterances wait for their
    # phase, refined in one launch and written in the order they came; the per-batch lines below are printed when their files
    # are queued.  1024 since round 6: the two-utterances-per-wave kernel fills the chip at 512 pairs (33 us per utterance at
    # 1024, 58 at 256, 150 at 32)
    lws_group = int(os.environ.get('AVSI_LWS_GROUP', '1024'))
    pending = []                    # (enhanced, masks, paths, lengths) of batches whose phase is still to be refined

    def written(paths, lengths):
        nonlocal total_samples
        total_samples += len(lengths)
        print('Written {:d} enhanced wavs. Total samples written so far {:d}.'.format(len(lengths), total_samples))

    def flush():
        if not pending:
            return
        if len(pending) == 1:
            wavs, masks = pending[0][0], pending[0][1]
        elif _adjacent([p[0] for p in pending]) and _adjacent([p[1] for p in pending]):
            # the reader batches of ONE model step, in order: rows of the same two tensors -- no copy
            n = sum(p[0].shape[0] for p in pending)
            wavs = pending[0][0]._base[:n] if pending[0][0]._base is not None else pending[0][0]
            masks = pending[0][1]._base[:n] if pending[0][1]._base is not None else pending[0][1]
        else:
            wavs = torch.cat([p[0] for p in pending])
            masks = torch.cat([p[1] for p in pending])
        # Reconstruct phase with LWS algorithm (reference inference.py:141-154), all collected batches on the device;
        # nothing here waits for the GPU: the status words travel with the waveforms and are checked before a file is written
        out = lws_processor.refine_enhanced(wavs, masks, num_samples=wavs.shape[1], check=False)
        status = (ops.coop_status(device), lws_processor.status_word())
        at = 0
        for _, _, paths, lengths in pending:
            writer.submit(out[at:at + len(lengths)], paths, lengths, status)
            at += len(lengths)
            written(paths, lengths)
        pending.clear()

    # Utterances are independent in inference, and `batch_size` only says how many of them the reference hands to one
    # sess.run (inference.py:121-131): the model step of SEVERAL reader batches is one launch sequence here -- up to
    # AVSI_INFER_COALESCE utterances (default 1024 = the LWS group; at 32 utterances a step is latency-bound, 67 us per
    # utterance, at 1024 it is 20 us).  What the caller sees per batch stays per batch: the 'Written ...' lines, the files,
    # and the loss of every reader batch in the mean over batches (taken on that batch's rows of the group's prediction:
    # for the plain models loss = mean |prediction - target| over the batch, models.py:144-158).  The variants, whose
    # losses are ratios of sums or carry further heads, run batch by batch as before; AVSI_INFER_COALESCE=0 does for all.
    coalesce = int(os.environ.get('AVSI_INFER_COALESCE', '1024'))
    plain = type(model).__name__ == 'StackedBLSTMModel' and not getattr(model, 'blend', False)
    if not plain or coalesce < 2 * batch_size:
        coalesce = 0

    spent = [0.0, 0.0, 0.0, 0.0]          # AVSI_INFER_TIMING: host seconds in concatenation, model launches, losses, hand-over

    def lap(i, t0):
        if stamps is not None:
            spent[i] += perf_counter() - t0
        return perf_counter()

    def run(group):
        """One model step over the reader batches of `group` [(feed, paths, lengths)], then per reader batch: loss, phase, files."""
        t0 = perf_counter()
        if len(group) == 1:
            feed = group[0][0]
        else:
            feed = {}
            for k, v0 in group[0][0].items():
                vals = [g[0][k] for g in group]
                feed[k] = None if v0 is None else (np.concatenate(vals) if isinstance(v0, np.ndarray) else torch.cat(vals))
        t0 = lap(0, t0)
        model.feed(**feed)
        enhanced = model.enhanced_sources_oracle_phase if oracle_phase else model.enhanced_sources
        t0 = lap(1, t0)
        if len(group) == 1:
            loss = model.loss                       # read back once, after the loop: no host round trip per batch
            loss_list.append(loss.detach().clone() if hasattr(loss, 'detach') else float(loss))
        else:
            tgt, pred = model.target_spec_norm, model.prediction
            msk = model.masks[:, :tgt.shape[1]]
            reg = model.regularization * model.reg_loss if model.regularization else None
            at = 0
            for _, _, lengths in group:
                sl = slice(at, at + len(lengths))
                at += len(lengths)
                parts = [tgt[sl], pred[sl], msk[sl]]
                parts = [x.contiguous() if x.storage_offset() % 4 == 0 else x.clone() for x in parts]     # 16-byte loads
                out3, _ = ops.l1_loss(*parts)
                loss_list.append((out3[0] + reg if reg is not None else out3[0]).detach().clone())
        t0 = lap(2, t0)
        masks = None if oracle_phase else model.masks
        status = (ops.coop_status(device),)

        def deliver():
            """Phase refinement and files of this group.  Called AFTER the next group's model step has been enqueued: handing
            32 reader batches to the writer waits for page-locked slots, i.e. for files of earlier batches to be on disk,
            and the GPU would otherwise sit idle behind this thread (150 - 250 ms of a 4096-utterance run)."""
            t1 = perf_counter()
            if oracle_phase:
                at = 0
                for _, paths, lengths in group:
                    writer.submit(enhanced[at:at + len(lengths)], paths, lengths, status)
                    at += len(lengths)
                    written(paths, lengths)
                lap(3, t1)
                return
            if pending and (pending[0][0].shape[1:] != enhanced.shape[1:] or pending[0][1].shape[1:] != masks.shape[1:]):
                flush()                             # a batch of another length: it cannot share the launch
            # (enhanced and the fed masks are tensors of THIS step: the next step, already enqueued, has its own)
            alone = enhanced.shape[0] >= lws_group and not pending
            at = 0
            for _, paths, lengths in group:
                sl = slice(at, at + len(lengths))
                at += len(lengths)
                if alone or len(group) > 1:         # rows of the group's own tensors: no copy needed
                    pending.append((enhanced[sl], masks[sl], paths, lengths))
                else:                               # a single reader batch: its mask tensor is the reader's upload
                    pending.append((enhanced[sl].clone(), masks[sl].clone(), paths, lengths))
            if sum(len(p[3]) for p in pending) >= lws_group:
                flush()
            lap(3, t1)
        return deliver

    print('Starting inference on dataset: {:s}'.format(data_path_test))
    stamp('lws + writer')
    waited = 0.0
    group, held, started = [], 0, False
    deferred = []                       # deliver() of the group whose model step was enqueued last

    def step(batches):
        d = run(batches)
        while deferred:
            deferred.pop(0)()
        deferred.append(d)

    while True:
        try:
            if stamps is not None:
                t_w = perf_counter()
            nxt = test_it.get_next()
            if stamps is not None:
                waited += perf_counter() - t_w
                if stamps[-1][0] == 'lws + writer':
                    stamp('first batch read')
            feed, test_sample_path = unpack_batch(nxt, uses_embeddings(config))
            test_length = feed['sequence_lengths']
        except OutOfRangeError:
            if group:
                step(group)
            while deferred:
                deferred.pop(0)()
            flush()
            print('done.')
            break
        if group and any(getattr(feed[k], 'shape', (0,))[1:] != getattr(group[0][0][k], 'shape', (0,))[1:] for k in feed):
            step(group)                         # a batch of another shape cannot share the step
            group, held = [], 0
        group.append((feed, test_sample_path, test_length))
        held += len(test_length)
        # (the first group is a quarter of the others: the GPU starts after 256 utterances have been read, not 1024)
        if held + len(test_length) > max(coalesce if started else min(coalesce, max(256, batch_size)), 1):
            step(group)                         # the next batch of this size would not fit any more
            group, held, started = [], 0, True
    stamp('loop (launches; %.1f ms of it waiting for the reader, %.1f concatenating, %.1f launching the model steps, %.1f the losses, '
          '%.1f handing over to LWS / the writer)' % ((waited * 1e3,) + tuple(1e3 * x for x in spent)))
    writer.close()          # every file is on disk (or its error raised) before the summary line
    stamp('GPU drained, files written')
    ops.coop_check()
    lws_processor.check()
    loss_list = [float(x) for x in loss_list]

    # np.mean over the batches of ALL ranks (reference inference.py:170), whatever share of them each rank had
    tot, cnt = parallel.all_reduce_sum_scalars([float(np.sum(loss_list)), float(len(loss_list))])
    mean_loss = tot / cnt if cnt else 0.0
    print('Loss hole: {:.5}'.format(mean_loss))
    if stamps is not None:
        stamp('checks + loss')
        print('infer() timing, ms: ' + '; '.join('%s %.1f' % (n, (t - stamps[i][1]) * 1e3) for i, (n, t) in enumerate(stamps[1:])),
              file=sys.stderr, flush=True)
    return mean_loss
