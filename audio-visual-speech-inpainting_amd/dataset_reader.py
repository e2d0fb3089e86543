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
(dataset_reader_emb.py:63-81).  Only the 'fixed' schema is supported.
"""
import numpy as np

from . import tfrecord_io


class OutOfRangeError(Exception):
    """End of the dataset (the role of tf.errors.OutOfRangeError)."""


class Dataset(object):
    def __init__(self, files, shuffle, seed, buffer_size, parse):
        self.files = list(files)
        self.shuffle = shuffle
        self.seed = seed
        self.buffer_size = buffer_size
        self.parse = parse
        self._epoch = 0

    def examples(self):
        """One pass: records in file order, passed through a tf.data-style shuffle buffer."""
        def raw():
            for path in self.files:
                for payload in tfrecord_io.read_records(path):
                    yield payload
        if not self.shuffle:
            for payload in raw():
                yield self.parse(payload)
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
            yield self.parse(out)
        while buf:
            j = int(rng.integers(len(buf)))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield self.parse(buf.pop())


class BatchIterator(object):
    def __init__(self, dataset, batch_size, n_epochs, drop_remainder, shard=(0, 1)):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.n_epochs = n_epochs
        self.drop_remainder = drop_remainder
        self.shard = shard
        self._gen = None
        self.initializer()

    def initializer(self):
        """Rewind (the reference runs sess.run(iterator.initializer) per epoch)."""
        self._gen = self._batches()
        return self

    def _examples(self):
        epoch = 0
        while self.n_epochs is None or epoch < self.n_epochs:
            for ex in self.dataset.examples():
                yield ex
            epoch += 1

    def _batches(self):
        rank, world = self.shard
        batch, index = [], 0
        for ex in self._examples():
            batch.append(ex)
            if len(batch) == self.batch_size:
                if index % world == rank:
                    yield _collate(batch)
                index += 1
                batch = []
        if batch and not self.drop_remainder and index % world == rank:
            yield _collate(batch)

    def get_next(self):
        try:
            return next(self._gen)
        except StopIteration:
            raise OutOfRangeError("End of sequence")

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)


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


class DataManager:
    """Utilities to read TFRecords"""

    def __init__(self, num_audio_samples=48000, audio_feat_size=257, video_feat_size=136, buffer_size=1000,
                 mode='fixed', embedding_size=None):
        if mode != 'fixed':
            raise ValueError("only the 'fixed' TFRecord schema is supported (the reference's 'var' writer is broken)")
        self.num_audio_samples = num_audio_samples
        self.audio_feat_size = audio_feat_size
        self.video_feat_size = video_feat_size
        self.embedding_size = embedding_size
        self.buffer_size = buffer_size
        self.mode = mode

    def get_dataset(self, file_list, shuffle=True, seed=None):
        return Dataset(file_list, shuffle, seed, self.buffer_size, self.read_data_format_fixed)

    def get_iterator(self, dataset, batch_size=16, n_epochs=None, drop_remainder=False, shard=(0, 1)):
        """`shard=(rank, world)` deals whole batches round-robin to data-parallel ranks."""
        it = BatchIterator(dataset, batch_size, n_epochs, drop_remainder, shard)
        return it, it

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
