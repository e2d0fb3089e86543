"""TFRecord dataset reader of the drivers (host side, no TensorFlow).

``DataManager`` keeps the interface of the reference classes (``dataset_reader.py:12-99`` and
``dataset_reader_emb.py``): ``get_dataset(file_list, shuffle, seed)`` then
``get_iterator(dataset, batch_size, n_epochs, drop_remainder)``; ``iterator.get_next()`` returns the
next batch as numpy arrays and raises ``OutOfRangeError`` at the end (``tf.errors.OutOfRangeError``
in the reference loops, training_emb.py:273), ``iterator.initializer()`` rewinds it.

Batch layout ('fixed' mode, dataset_reader.py:77-79): ``(sequence_length int32[B],
labels_length int32[B], target_audio_wav int32[B, N] (tf.to_int32 truncation),
sample_path [B] bytes, labels f32[B, L], video_features f32[B, T, Dv], mask f32[B, T, F])``;
with ``embedding_size`` the f32[B, E] ``embedding`` context feature is inserted after the audio
(dataset_reader_emb.py:63-81).  'var' mode (dataset_reader.py:82-99, read side only in the reference: its writer cannot run,
SURVEY App. B11): everything but the two lengths is a feature list, batches are zero-padded to the longest record
(``padded_batch``), the audio stays float32 and the sample path is a row of character codes; parsed in Python, host arrays.

Like the reference's tf.data pipeline, parsing is native and runs ahead of the consumer: records go
through the shuffle buffer as raw payloads, a batch is parsed by ``avsi_sequence_example_decode_fixed_host``
straight into its arrays (ctypes drops the GIL), files are read and checksummed a bounded distance
ahead on a small thread pool (in file order: seeded runs and the round-robin sharding of batches over
ranks stay deterministic), and a background thread keeps ``prefetch`` batches ready while the GPU works
on the current one.
``read_data_format_fixed`` (pure Python, one record) stays as the readable definition of the schema;
tests hold the native path to it.
"""
import ctypes
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, tfrecord_io

_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        # a record is ~0.2 ms of open + read + checksum + parse on one core: sixteen threads keep a GPU fed at batches of a
        # thousand (AVSI_READER_THREADS overrides; a GPU box grants a process about that many cores per GPU)
        n = int(os.environ.get('AVSI_READER_THREADS', '0')) or min(16, os.cpu_count() or 1)
        _POOL = ThreadPoolExecutor(max_workers=max(1, n), thread_name_prefix='avsi-parse')
    return _POOL


class OutOfRangeError(Exception):
    """End of the dataset (the role of tf.errors.OutOfRangeError)."""


class RecordFile(str):
    """Path of a .tfrecord file that holds exactly one record (the reference's datasets).  It stands in for the
    record's payload in the stream: the native batch decoder reads, checks and parses such files itself, off the
    interpreter lock and several at a time (avsi_tfrecord_file_decode_fixed_host); everything else gets the bytes
    through ``payload_bytes``."""
    __slots__ = ()


def payload_bytes(item):
    """The serialized record behind an element of ``Dataset.payloads()``."""
    if isinstance(item, RecordFile):
        return next(iter(tfrecord_io.read_records(item)))
    return item


class Dataset(object):
    def __init__(self, files, shuffle, seed, buffer_size, parse):
        self.files = list(files)
        self.shuffle = shuffle
        self.seed = seed
        self.buffer_size = buffer_size
        self._parse = parse
        self._epoch = 0
        # one record per file (checked on the first file; a later file with more records fails loudly in the decoder):
        # the stream then carries paths, and the files are read where they are decoded
        self.record_files = False
        if self.files and os.environ.get('AVSI_READER_FILES', '1') != '0':
            shape = (ctypes.c_int64 * 5)()
            rc = _lib.lib().avsi_tfrecord_file_shape_host(os.fsencode(self.files[0]), 0, shape)
            self.record_files = rc == _lib.AVSI_OK

    def parse(self, item):
        return self._parse(payload_bytes(item))

    def payloads(self):
        """One pass: serialized records (or ``RecordFile`` paths, see there) in file order, passed through a tf.data-style
        shuffle buffer."""
        def read_file(path):
            return list(tfrecord_io.read_records(path))

        def raw():
            if self.record_files:
                for path in self.files:
                    yield RecordFile(path)
                return
            # the reference's datasets are one file per sample: files are read (and their checksums verified)
            # on the pool a bounded distance ahead, and handed on strictly in file order
            pending, ahead = [], 16
            files = iter(self.files)
            while True:
                while len(pending) < ahead:
                    path = next(files, None)
                    if path is None:
                        break
                    pending.append(_pool().submit(read_file, path))
                if not pending:
                    return
                for payload in pending.pop(0).result():
                    yield payload
        if not self.shuffle:
            for payload in raw():
                yield payload
            return
        rng = np.random.default_rng(None if self.seed is None else self.seed + self._epoch)
        self._epoch += 1
        buf = []
        for payload in raw():
            if len(buf) < self.buffer_size:
                buf.append(payload)
                continue
            j = int(rng.integers(len(buf)))
            out, buf[j] = buf[j], payload
            yield out
        while buf:
            j = int(rng.integers(len(buf)))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield buf.pop()

    def examples(self):
        """The same pass, every record parsed on its own (pure Python)."""
        for payload in self.payloads():
            yield self.parse(payload)


class InterleavedDataset(Dataset):
    """``tf.data.Dataset.list_files(patterns, shuffle, seed).interleave(TFRecordDataset, cycle_length, block_length)`` (reference
    dataset_reader.py:36-45, get_dataset_group): `cycle_length` files are open at a time and give `block_length` consecutive
    records each in turn; a file that runs out is replaced by the next one of the list.  The file list is re-shuffled per
    pass like list_files does (seeded: deterministic); no record-level shuffle buffer."""

    def __init__(self, patterns, cycle_length, block_length, shuffle, seed, parse):
        from glob import glob
        files = []
        for pat in ([patterns] if isinstance(patterns, str) else list(patterns)):
            files.extend(sorted(glob(pat)) if any(ch in pat for ch in '*?[') else [pat])
        Dataset.__init__(self, files, False, seed, 0, parse)
        self.record_files = False
        self.cycle_length, self.block_length, self.shuffle_files = max(1, int(cycle_length)), max(1, int(block_length)), shuffle

    def payloads(self):
        files = list(self.files)
        if self.shuffle_files:
            rng = np.random.default_rng(None if self.seed is None else self.seed + self._epoch)
            rng.shuffle(files)
        self._epoch += 1
        waiting = iter(files)
        cycle = []
        for _ in range(self.cycle_length):
            path = next(waiting, None)
            if path is None:
                break
            cycle.append(iter(tfrecord_io.read_records(path)))
        at = 0
        while cycle:
            at %= len(cycle)
            taken = 0
            while taken < self.block_length:
                rec = next(cycle[at], None)
                if rec is None:
                    break
                taken += 1
                yield rec
            if taken < self.block_length:           # this file is done: the next one of the list takes its place in the cycle
                path = next(waiting, None)
                if path is None:
                    cycle.pop(at)
                    continue
                cycle[at] = iter(tfrecord_io.read_records(path))
                if taken:
                    at += 1
                continue
            at += 1


class Batch(tuple):
    """The batch tuple of the module docstring (numpy arrays).  With ``get_iterator(..., device=...)`` the large
    arrays have also been copied to the GPU by the prefetch thread: ``device_arrays`` maps positions of the tuple
    to device tensors, ``ready`` is the event a consuming stream has to wait for (see ``to_device``)."""
    device_arrays = None
    ready = None
    def gap_count(self):
        """Zero elements of the mask field as a DEVICE scalar (None without an uploaded mask), counted on the current
        stream behind the upload -- the trainer's loss weights (training.gap_elements): a scan of the host array cost 2 ms
        per 32 records, and kernels on the upload stream itself slow the step down (see _Uploader)."""
        t = self.to_device(len(self) - 1)
        return None if t is None else (t == 0).sum()

    def to_device(self, index):
        """Device tensor of field ``index`` (or None if it was not uploaded), ordered after the upload on the
        current stream -- no host synchronisation."""
        import torch
        if not self.device_arrays or index not in self.device_arrays:
            return None
        t = self.device_arrays[index]
        cur = torch.cuda.current_stream(t.device)
        cur.wait_event(self.ready)
        t.record_stream(cur)
        return t


class _Uploader(object):
    """Copies the bulky fields of a batch (audio, video, mask, embedding) to the GPU from the prefetch thread, on a
    stream of its own, one event per batch.  The reference's feed_dict crossing is a synchronous pageable copy
    on the training thread; here the records are parsed straight into PINNED arenas (DataManager.decode_batch) and the
    copies are asynchronous at the link's rate -- 600 MB per batch of 1024 utterances: 11 ms, where the pageable copy took
    32 ms -- and the training stream waits for the event."""

    def __init__(self, device, fields, count_gaps=False):
        import torch
        self.torch = torch
        self.count_gaps = bool(count_gaps)
        self.fields = tuple(fields)         # positions of the batch tuple to upload (negative = from the end)
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.stream = None
        self.parts = 1                      # consumer batches per uploaded batch (BatchIterator's macro reads)

    def __call__(self, batch, arena=None):
        torch = self.torch
        if self.stream is None:
            torch.cuda.set_device(self.device)
            # (per batch the CONSUMER sees: a macro read of `parts` small batches is still small-batch work)
            nbytes = sum(a.nbytes for a in batch if isinstance(a, np.ndarray) and a.dtype != object) // max(1, self.parts)
            # HIP maps its streams onto four hardware queues per priority level, and a normal stream created after the
            # model's side streams can land on the queue of the launch stream, where the upload and its event sit in order
            # between the step's kernels (tools/hostfed_probe.py: a 3.7 GB upload then adds its full 67 ms to a 140 ms step
            # instead of hiding under it); queues of another priority level are not shared.  But work on a high-priority
            # queue holds the cooperative recurrent kernels of a small-batch training step back, copies included (8.2
            # instead of 6.9 ms per step of 32 utterances).  So: high priority for batches whose upload is worth hiding
            # (tens of MB: the steps behind them are long), normal priority for small ones.  AVSI_UPLOAD_PRIORITY overrides.
            prio = os.environ.get('AVSI_UPLOAD_PRIORITY')
            prio = int(prio) if prio is not None else (-1 if nbytes >= (64 << 20) else 0)
            self.stream = torch.cuda.Stream(device=self.device, priority=prio)
        device_arrays = {}
        host = list(batch)
        pinned = (arena or {}).get('_pinned', {})
        with torch.cuda.stream(self.stream):
            for i in sorted(f % len(batch) for f in self.fields):
                a = batch[i]
                if isinstance(a, np.ndarray) and a.dtype != object and a.size:
                    t = pinned.get(a.ctypes.data)           # the pinned tensor this array is a view of, if any
                    if t is not None:
                        device_arrays[i] = t[:a.shape[0]].to(self.device, non_blocking=True)
                        # the host view points into an arena the reader recycles two batches later: a consumer that read
                        # it then would silently get ANOTHER batch's data -- the device copy is the field (to_device)
                        host[i] = None
                    else:
                        device_arrays[i] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        out = Batch(host)
        out.device_arrays, out.ready = device_arrays, ready
        if arena is not None:
            arena['_event'] = out.ready             # the arena's next user waits for these copies (BatchIterator._make)
        return out


class _Prefetcher(object):
    """Runs a generator on a background thread, `depth` items ahead of the consumer."""
    _END = object()

    def __init__(self, gen, depth):
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._gen = gen
        self._thread = None         # started by the first next(): a pass that is rewound before use costs nothing

    def _start(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, args=(self._gen,), daemon=True, name='avsi-prefetch')
            self._thread.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, gen):
        try:
            for item in gen:
                if not self._put(item):
                    return
            self._put(self._END)
        except BaseException as e:          # handed to the consumer, raised by its next()
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        self._start()
        item = self._q.get()
        if item is self._END:
            self._q.put(item)               # stay exhausted
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(self._END)
            raise item
        return item

    def close(self):
        """Stop the pass and WAIT for its thread: the next pass reuses the same host arenas and batch counter, which
        a still-running thread would write into."""
        self._stop.set()
        t = self._thread
        if t is not None and t is not threading.current_thread():
            t.join()


class BatchIterator(object):
    def __init__(self, dataset, batch_size, n_epochs, drop_remainder, shard=(0, 1), decode_batch=None, prefetch=2,
                 device=None, upload_fields=(2, -2, -1), even_rounds=False, count_gaps=False):
        self.dataset = dataset
        # data-parallel training does one collective per step: every rank must see the same number of batches, so
        # batches are dealt in whole rounds of `world` and a last incomplete round (and a short last batch) is dropped
        self.even_rounds = bool(even_rounds) and shard[1] > 1
        self.batch_size = int(batch_size)
        self.n_epochs = n_epochs
        self.drop_remainder = drop_remainder
        self.shard = shard
        self.decode_batch = decode_batch    # list of payloads -> batch tuple (native); None = parse + stack in Python
        # records per macro read of small batches (see _batches); AVSI_READER_MACRO=<records> overrides, 0 = one batch per read
        target = int(os.environ.get('AVSI_READER_MACRO', '128'))
        self.macro = max(1, target // self.batch_size) if self.batch_size > 0 else 1
        self.prefetch = int(prefetch) * (self.macro if device is not None else 1)      # the thread stays a whole read ahead
        self.upload = _Uploader(device, upload_fields, count_gaps) if device is not None else None
        if self.upload is not None and shard[1] == 1 and not self.even_rounds:
            self.upload.parts = self.macro
        # with a device the consumer works on the uploaded copies, so the host arrays of the bulky fields (audio, video,
        # mask, embedding) are recycled and the batch tuple holds None in their places -- use Batch.to_device()
        # ... two arenas when they are page-locked: one is parsed into while the other's copies run (its next user waits for
        # their event); pinning is the expensive part of a short run (~0.3 s per GB), so no more of them than needed
        self._arenas = [{'_pin': True} for _ in range(2)] if device is not None else None
        self._made = 0
        self._gen = None
        self.initializer()

    def initializer(self):
        """Rewind (the reference runs sess.run(iterator.initializer) per epoch)."""
        if isinstance(self._gen, _Prefetcher):
            self._gen.close()
        gen = self._batches()
        self._gen = _Prefetcher(gen, self.prefetch) if self.prefetch > 0 else gen
        return self

    def _payloads(self):
        epoch = 0
        while self.n_epochs is None or epoch < self.n_epochs:
            for p in self.dataset.payloads():
                yield p
            epoch += 1

    def _make(self, payloads):
        if self.decode_batch is None:
            batch = getattr(self, 'collate', _collate)([self.dataset.parse(p) for p in payloads])
        elif self._arenas is not None:
            arena = self._arenas[self._made % len(self._arenas)]
            self._made += 1
            if arena.get('_event') is not None:
                arena['_event'].synchronize()       # its previous contents are on the device
            batch = self.decode_batch(payloads, arena)
            return self.upload(batch, arena) if self.upload is not None else batch
        else:
            batch = self.decode_batch(payloads)
        return self.upload(batch) if self.upload is not None else batch

    def _split(self, big, sizes):
        """The consumer's batches out of one macro read: slices of the same host arrays and of the same uploaded device
        tensors (views: nothing is copied), all behind the one upload event."""
        at = 0
        for n in sizes:
            sub = Batch(None if a is None else a[at:at + n] for a in big)
            if getattr(big, 'device_arrays', None):
                sub.device_arrays = {i: t[at:at + n] for i, t in big.device_arrays.items()}
                sub.ready = big.ready
            at += n
            yield sub

    def _batches(self):
        rank, world = self.shard
        # Small batches on one rank: `macro` consecutive batches are read, parsed and uploaded as ONE (one native call per
        # reader thread, one arena, three copies, one event) and handed out as views.  At 32 records per batch the per-batch
        # work of this thread -- pool hand-overs, three asynchronous copies, an event, all under the interpreter lock the
        # launch thread also wants -- was what infer() waited for (1 ms per batch of a 2.3 ms step)
        macro = self.macro if (world == 1 and not self.even_rounds and self.decode_batch is not None
                               and self._arenas is not None) else 1
        batch, index, mine = [], 0, None
        group = []
        for p in self._payloads():
            batch.append(p)
            if len(batch) == self.batch_size:
                if macro > 1:
                    group.append(batch)
                    batch = []
                    if len(group) == macro:
                        for sub in self._split(self._make([q for b in group for q in b]), [len(b) for b in group]):
                            yield sub
                        group = []
                    continue
                if index % world == rank:
                    mine = batch
                    if not self.even_rounds:
                        yield self._make(mine)
                index += 1
                batch = []
                if self.even_rounds and index % world == 0:     # a whole round of `world` batches exists: hand out ours
                    yield self._make(mine)
                    mine = None
        if macro > 1:
            if batch and not self.drop_remainder:
                group.append(batch)
            if group:
                for sub in self._split(self._make([q for b in group for q in b]), [len(b) for b in group]):
                    yield sub
            return
        if batch and not self.drop_remainder and not self.even_rounds and index % world == rank:
            yield self._make(batch)

    def get_next(self):
        try:
            return next(self._gen)
        except StopIteration:
            raise OutOfRangeError("End of sequence")

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    def __del__(self):
        if isinstance(self._gen, _Prefetcher):
            self._gen.close()


def _collate(batch):
    cols = list(zip(*batch))
    out = []
    for col in cols:
        first = col[0]
        if isinstance(first, bytes):
            out.append(np.array(col, dtype=object))
        else:
            out.append(np.stack(col))
    return tuple(out)


def _collate_padded(batch):
    """tf.data padded_batch with zero padding: every field becomes [B, longest, ...]."""
    out = []
    for col in zip(*batch):
        first = np.asarray(col[0])
        if first.ndim == 0:
            out.append(np.stack([np.asarray(c) for c in col]))
            continue
        shape = tuple(max(np.asarray(c).shape[d] for c in col) for d in range(first.ndim))
        arr = np.zeros((len(col),) + shape, dtype=first.dtype)
        for i, c in enumerate(col):
            c = np.asarray(c)
            arr[(i,) + tuple(slice(0, n) for n in c.shape)] = c
        out.append(arr)
    return tuple(out)


class DataManager:
    """Utilities to read TFRecords"""

    def __init__(self, num_audio_samples=48000, audio_feat_size=257, video_feat_size=136, buffer_size=1000,
                 mode='fixed', embedding_size=None):
        if mode not in ('fixed', 'var'):
            raise ValueError("TFRecord schema must be 'fixed' or 'var'")
        if mode == 'var' and embedding_size:
            raise ValueError("the 'var' schema has no embedding feature (dataset_reader_emb.py reads 'fixed' records only)")
        self.num_audio_samples = num_audio_samples
        self.audio_feat_size = audio_feat_size
        self.video_feat_size = video_feat_size
        self.embedding_size = embedding_size
        self.buffer_size = buffer_size
        self.mode = mode

    def get_dataset(self, file_list, shuffle=True, seed=None):
        if self.mode == 'var':
            ds = Dataset(file_list, shuffle, seed, self.buffer_size, self.read_data_format_var)
            ds.record_files = False         # (the one-record-per-file fast path parses 'fixed' records natively)
            return ds
        return Dataset(file_list, shuffle, seed, self.buffer_size, self.read_data_format_fixed)

    def get_dataset_group(self, file_list=(), cycle_length=100, block_length=16, shuffle=True, seed=None):
        """Reference dataset_reader.py:36-45: records of `cycle_length` files interleaved in blocks of `block_length`."""
        return InterleavedDataset(file_list, cycle_length, block_length, shuffle, seed,
                                  self.read_data_format_var if self.mode == 'var' else self.read_data_format_fixed)

    def get_iterator(self, dataset, batch_size=16, n_epochs=None, drop_remainder=False, shard=(0, 1), native=True,
                     prefetch=2, device=None, even_rounds=False, count_gaps=False):
        """`shard=(rank, world)` deals whole batches round-robin to data-parallel ranks.  `native=False`
        parses with the pure-Python decoder, `prefetch=0` parses on the caller's thread, `device` (e.g. 'cuda')
        also uploads audio / video / mask from the prefetch thread (see `Batch`), `even_rounds` gives every rank the same
        number of (full) batches -- what a training loop with one collective per step needs."""
        if self.mode == 'var':
            # dataset.padded_batch(batch_size, padded_shapes=([], [], [None], [None], [None], [None, None], [None, None])) of the
            # reference (dataset_reader.py:49-55): every field padded with zeros to the longest of the batch; parsed in Python
            # (48,000 one-float features per record: a format for small corpora), host arrays only
            it = BatchIterator(dataset, batch_size, n_epochs, drop_remainder, shard, decode_batch=None, prefetch=prefetch,
                               device=None, even_rounds=even_rounds)
            it.collate = _collate_padded
            return it, it
        it = BatchIterator(dataset, batch_size, n_epochs, drop_remainder, shard,
                           decode_batch=self.decode_batch if native else None, prefetch=prefetch, device=device,
                           upload_fields=(2, 3, -2, -1) if self.embedding_size else (2, -2, -1), even_rounds=even_rounds,
                           count_gaps=count_gaps)
        return it, it

    def decode_batch(self, payloads, arena=None):
        """Serialized records -> the batch tuple of the module docstring, parsed natively.  ``arena`` (a dict the
        caller keeps) lets consecutive calls reuse their arrays instead of mapping 19 MB of fresh pages per batch
        (half of the 0.22 ms per record); the arrays of a call are then only valid until the arena's next use."""
        L = _lib.lib()
        B = len(payloads)
        shape = (ctypes.c_int64 * 5)()
        from_files = isinstance(payloads[0], RecordFile)
        if from_files:
            paths_fs = [os.fsencode(p) for p in payloads]
            if L.avsi_tfrecord_file_shape_host(paths_fs[0], 1, shape) != _lib.AVSI_OK:
                raise IOError("%s: unreadable, truncated, corrupted (crc mismatch) or malformed record" % payloads[0])
        elif L.avsi_sequence_example_shape_host(payloads[0], len(payloads[0]), shape) != _lib.AVSI_OK:
            raise ValueError("malformed SequenceExample record")
        n_wav, n_emb, T, Tv, n_lab = (int(v) for v in shape)
        if n_wav != self.num_audio_samples:
            raise ValueError("target_audio_wav has %d samples, expected %d" % (n_wav, self.num_audio_samples))
        E = int(self.embedding_size or 0)
        if E and n_emb != E:
            raise ValueError("embedding has %d values, expected %d" % (n_emb, E))
        def alloc(name, shape, dtype):
            if arena is None:
                return np.empty(shape, dtype=dtype)
            a = arena.get(name)
            if a is None or a.dtype != dtype or a.shape[1:] != tuple(shape[1:]) or a.shape[0] < shape[0]:
                if a is not None:
                    arena.get('_pinned', {}).pop(a.ctypes.data, None)       # the outgrown page-locked block goes with its array
                if arena.get('_pin') and int(np.prod(shape)) > 0:
                    # page-locked: the uploader's copies then run at the link's rate and asynchronously (_Uploader)
                    import torch
                    t = torch.empty(tuple(int(v) for v in shape), dtype={np.dtype(np.int32): torch.int32,
                                                                         np.dtype(np.float32): torch.float32}[np.dtype(dtype)],
                                    pin_memory=True)
                    a = arena[name] = t.numpy()
                    arena.setdefault('_pinned', {})[a.ctypes.data] = t
                else:
                    a = arena[name] = np.empty(shape, dtype=dtype)
            return a[:shape[0]]
        lengths = np.empty((B, 2), dtype=np.int32)
        wav = alloc('wav', (B, n_wav), np.int32)
        emb = alloc('emb', (B, E), np.float32) if E else None
        paths = ctypes.create_string_buffer(B * 1024)
        labels = np.empty((B, n_lab), dtype=np.float32)         # small, and read on the host (CTC): never from a recycled arena
        video = alloc('video', (B, Tv, self.video_feat_size), np.float32)
        mask = alloc('mask', (B, T, self.audio_feat_size), np.float32)
        paths_addr = ctypes.addressof(paths)

        def row(a, i):
            return a.ctypes.data + i * a.strides[0] if a is not None and a.size else 0

        def one(i):
            if from_files:       # open + read + checksums + parse in one native call: no interpreter lock held
                return L.avsi_tfrecord_file_decode_fixed_host(
                    paths_fs[i], 1, n_wav, self.audio_feat_size, self.video_feat_size, E, T, Tv, n_lab,
                    row(lengths, i), row(wav, i), row(emb, i), paths_addr + i * 1024, 1024, row(labels, i), row(video, i),
                    row(mask, i))
            return L.avsi_sequence_example_decode_fixed_host(
                payloads[i], len(payloads[i]), n_wav, self.audio_feat_size, self.video_feat_size, E, T, Tv, n_lab,
                row(lengths, i), row(wav, i), row(emb, i), paths_addr + i * 1024, 1024, row(labels, i), row(video, i),
                row(mask, i))
        # records already in memory: serial on purpose (a record is ~0.2 ms of memcpy, which threads only contend on);
        # files: on the pool, one contiguous SLICE of the batch per thread and native call (the calls release the interpreter
        # lock; a task per record cost ~30 us of hand-over under the lock, which capped 1024-record batches at 14 k records/s)
        if from_files and B > 1:
            threads = min(_pool()._max_workers, max(1, B // 4))
            step = -(-B // threads)
            arr = (ctypes.c_char_p * B)(*paths_fs)
            arr_addr = ctypes.addressof(arr)
            codes_a = np.zeros(B, dtype=np.int32)

            def some(lo):
                n = min(step, B - lo)
                L.avsi_tfrecord_files_decode_fixed_host(
                    arr_addr + lo * ctypes.sizeof(ctypes.c_char_p), n, 1, n_wav, self.audio_feat_size, self.video_feat_size, E, T,
                    Tv, n_lab, row(lengths, lo), row(wav, lo), row(emb, lo), paths_addr + lo * 1024, 1024, row(labels, lo),
                    row(video, lo), row(mask, lo), codes_a.ctypes.data + 4 * lo)
            list(_pool().map(some, range(0, B, step)))
            codes = codes_a.tolist()
        else:
            codes = [one(i) for i in range(B)]
        for i, rc in enumerate(codes):
            if from_files and rc == _lib.AVSI_ERR_INVALID_ARG:
                raise IOError("%s: unreadable, truncated, corrupted (crc mismatch) or malformed record" % payloads[i])
            if rc == _lib.AVSI_ERR_UNSUPPORTED:
                raise ValueError("record %d of the batch does not have the sizes of the first one / of the DataManager "
                                 "configuration (feature sizes, frame or label counts)%s" % (
                                     i, " -- or its file holds more than one record (AVSI_READER_FILES=0 reads such datasets)"
                                     if from_files else ""))
            if rc != _lib.AVSI_OK:
                raise ValueError("malformed SequenceExample record (or a feature of the 'fixed' schema is missing)")
        sample_paths = np.array([ctypes.string_at(paths_addr + i * 1024) for i in range(B)], dtype=object)
        out = [lengths[:, 0].copy(), lengths[:, 1].copy(), wav]
        if E:
            out.append(emb)
        return tuple(out + [sample_paths, labels, video, mask])

    def read_data_format_var(self, sample):
        """Parse one serialized 'var' SequenceExample (reference dataset_reader.py:82-99): (sequence_length int32, labels_length
        int32, target_audio_wav f32 [n] -- no int truncation in this mode --, sample_path int64 [chars], labels f32 [l],
        video_features f32 [t, Dv], mask f32 [t, F])."""
        ctx, seq = tfrecord_io.decode_sequence_example(sample)

        def scalars(name, dtype):
            steps = seq.get(name, [])
            return np.array([s[0] for s in steps], dtype=dtype) if steps else np.zeros(0, dtype=dtype)

        def vectors(name, width):
            steps = seq.get(name, [])
            if not steps:
                return np.zeros((0, width), dtype=np.float32)
            a = np.stack(steps).astype(np.float32)
            if a.shape[1] != width:
                raise ValueError("feature sizes of the record do not match the DataManager configuration")
            return a
        return (np.int32(ctx['sequence_length'][0]), np.int32(ctx['labels_length'][0]), scalars('target_audio_wav', np.float32),
                scalars('sample_path', np.int64), scalars('labels', np.float32), vectors('video_features', self.video_feat_size),
                vectors('mask', self.audio_feat_size))

    def read_data_format_fixed(self, sample):
        """Parse one serialized SequenceExample (reference dataset_reader.py:62-79)."""
        ctx, seq = tfrecord_io.decode_sequence_example(sample)
        wav = ctx['target_audio_wav']
        if wav.shape[0] != self.num_audio_samples:
            raise ValueError("target_audio_wav has %d samples, expected %d" % (wav.shape[0], self.num_audio_samples))
        video = np.stack(seq['video_features']).astype(np.float32) if seq.get('video_features') else \
            np.zeros((0, self.video_feat_size), dtype=np.float32)
        mask = np.stack(seq['mask']).astype(np.float32)
        if mask.shape[1] != self.audio_feat_size or (video.size and video.shape[1] != self.video_feat_size):
            raise ValueError("feature sizes of the record do not match the DataManager configuration")
        labels = np.array([s[0] for s in seq.get('labels', [])], dtype=np.float32)
        fields = [np.int32(ctx['sequence_length'][0]), np.int32(ctx['labels_length'][0]),
                  wav.astype(np.int32)]                     # tf.to_int32: truncation toward zero
        if self.embedding_size:
            emb = ctx['embedding']
            if emb.shape[0] != self.embedding_size:
                raise ValueError("embedding has %d values, expected %d" % (emb.shape[0], self.embedding_size))
            fields.append(emb.astype(np.float32))
        fields += [ctx['sample_path'][0], labels, video, mask]
        return tuple(fields)
