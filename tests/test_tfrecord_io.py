"""TFRecord framing + SequenceExample codec + DataManager (host logic, no GPU).  The codec is
cross-checked against google.protobuf (an independent encoder/decoder) when it is importable."""
import os
import struct

import numpy as np
import pytest

import avsi_amd  # noqa: F401
from avsi_amd import dataset_reader as dr
from avsi_amd import tfrecord_io as tio


def _sample(seed, T=12, N=2304, with_emb=False):
    rng = np.random.default_rng(seed)
    wav = np.round(rng.normal(0, 3000, N)).astype(np.float32) + 0.75       # fractional part: to_int32 truncates
    mask = np.ones((T, 257), dtype=np.float32)
    mask[3:6] = 0
    video = rng.normal(size=(T, 136)).astype(np.float32)
    labels = np.pad(rng.integers(1, 30, 7), (0, 43)).astype(np.float32)
    emb = rng.normal(size=512).astype(np.float32) if with_emb else None
    rec = tio.serialize_sample_fixed(T, 7, wav, video, mask, labels, "s%02d_clip" % seed, embedding=emb)
    return rec, (T, 7, wav, video, mask, labels, ("s%02d_clip" % seed).encode(), emb)


def test_crc32c_known_answers():
    assert tio._crc32c(b"123456789") == 0xE3069283
    assert tio._crc32c(b"") == 0
    assert tio._crc32c(bytes(32)) == 0x8A9136AA
    assert tio.masked_crc32c(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xFFFFFFFF)


def test_framing_roundtrip_and_corruption(tmp_path):
    path = str(tmp_path / "a.tfrecord")
    payloads = [b"hello", b"", bytes(range(256)) * 10]
    tio.write_records(path, payloads)
    assert list(tio.read_records(path)) == payloads
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack("<Q", raw[:8])[0] == 5
    raw[14] ^= 0xFF                                         # flip a payload byte of the first record
    open(path, "wb").write(raw)
    with pytest.raises(IOError):
        list(tio.read_records(path))
    assert len(list(tio.read_records(path, verify=False))) == 3


def test_sequence_example_roundtrip():
    rec, (T, L, wav, video, mask, labels, name, _) = _sample(1)
    ctx, seq = tio.decode_sequence_example(rec)
    assert ctx['sequence_length'].tolist() == [T] and ctx['labels_length'].tolist() == [L]
    np.testing.assert_array_equal(ctx['target_audio_wav'], wav)
    assert ctx['sample_path'] == [name]
    np.testing.assert_array_equal(np.stack(seq['mask']), mask)
    np.testing.assert_array_equal(np.stack(seq['video_features']), video)
    np.testing.assert_array_equal(np.array([s[0] for s in seq['labels']]), labels)


def test_negative_int64_and_unpacked_lists():
    buf = tio.encode_sequence_example({'x': np.array([-3, 5, 2 ** 40], dtype=np.int64)}, {})
    ctx, _ = tio.decode_sequence_example(buf)
    assert ctx['x'].tolist() == [-3, 5, 2 ** 40]
    # un-packed repeated float (field 1, wire type 5) must also be read
    fl = b''.join(tio._varint((1 << 3) | 5) + struct.pack('<f', v) for v in (1.5, -2.0))
    feat = tio._ld(2, fl)
    np.testing.assert_array_equal(tio.decode_feature(feat), [1.5, -2.0])


def test_wire_compatible_with_google_protobuf():
    """Encode the same SequenceExample with google.protobuf's generic descriptor machinery."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="ex.proto", package="t", syntax="proto3")

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=1, type_name=None, packed=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
        if type_name:
            f.type_name = type_name
        return f
    T = descriptor_pb2.FieldDescriptorProto
    m = msg("BytesList"); field(m, "value", 1, T.TYPE_BYTES, 3)
    m = msg("FloatList"); field(m, "value", 1, T.TYPE_FLOAT, 3)
    m = msg("Int64List"); field(m, "value", 1, T.TYPE_INT64, 3)
    m = msg("Feature")
    field(m, "bytes_list", 1, T.TYPE_MESSAGE, 1, ".t.BytesList")
    field(m, "float_list", 2, T.TYPE_MESSAGE, 1, ".t.FloatList")
    field(m, "int64_list", 3, T.TYPE_MESSAGE, 1, ".t.Int64List")
    m = msg("FEntry"); field(m, "key", 1, T.TYPE_STRING); field(m, "value", 2, T.TYPE_MESSAGE, 1, ".t.Feature")
    m = msg("Features"); field(m, "feature", 1, T.TYPE_MESSAGE, 3, ".t.FEntry")
    m = msg("FeatureList"); field(m, "feature", 1, T.TYPE_MESSAGE, 3, ".t.Feature")
    m = msg("FLEntry"); field(m, "key", 1, T.TYPE_STRING); field(m, "value", 2, T.TYPE_MESSAGE, 1, ".t.FeatureList")
    m = msg("FeatureLists"); field(m, "feature_list", 1, T.TYPE_MESSAGE, 3, ".t.FLEntry")
    m = msg("SequenceExample")
    field(m, "context", 1, T.TYPE_MESSAGE, 1, ".t.Features")
    field(m, "feature_lists", 2, T.TYPE_MESSAGE, 1, ".t.FeatureLists")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    SE = message_factory.GetMessageClass(pool.FindMessageTypeByName("t.SequenceExample"))

    rec, (Tn, L, wav, video, mask, labels, name, _) = _sample(2, T=4, N=64)
    ex = SE()
    ex.ParseFromString(rec)                                  # our bytes parse with the stock decoder
    ctx = {e.key: e.value for e in ex.context.feature}
    assert list(ctx['sequence_length'].int64_list.value) == [Tn]
    np.testing.assert_array_equal(np.array(ctx['target_audio_wav'].float_list.value, dtype=np.float32), wav)
    assert list(ctx['sample_path'].bytes_list.value) == [name]
    fl = {e.key: e.value for e in ex.feature_lists.feature_list}
    assert len(fl['mask'].feature) == Tn
    np.testing.assert_array_equal(np.array(fl['mask'].feature[1].float_list.value, dtype=np.float32), mask[1])
    # and the stock encoder's bytes parse with our decoder
    ctx2, seq2 = tio.decode_sequence_example(ex.SerializeToString())
    np.testing.assert_array_equal(ctx2['target_audio_wav'], wav)
    np.testing.assert_array_equal(np.stack(seq2['video_features']), video)


def _write_dataset(tmp_path, n, with_emb=False):
    files, truth = [], []
    for i in range(n):
        rec, t = _sample(10 + i, with_emb=with_emb)
        path = str(tmp_path / ("data_%05d.tfrecord" % (i + 1)))
        tio.write_records(path, [rec])
        files.append(path)
        truth.append(t)
    return files, truth


def test_datamanager_batches_fixed_schema(tmp_path):
    files, truth = _write_dataset(tmp_path, 5)
    dm = dr.DataManager(num_audio_samples=2304, audio_feat_size=257, video_feat_size=136, buffer_size=4, mode='fixed')
    ds = dm.get_dataset(files, shuffle=False)
    _, it = dm.get_iterator(ds, batch_size=2, n_epochs=1)
    batches = []
    while True:
        try:
            batches.append(it.get_next())
        except dr.OutOfRangeError:
            break
    assert [len(b[0]) for b in batches] == [2, 2, 1]
    seq_len, lab_len, wav, path, labels, video, mask = batches[0]
    assert seq_len.dtype == np.int32 and wav.dtype == np.int32 and mask.dtype == np.float32
    assert wav.shape == (2, 2304) and video.shape == (2, 12, 136) and mask.shape == (2, 12, 257)
    np.testing.assert_array_equal(wav[1], np.trunc(truth[1][2]).astype(np.int32))     # tf.to_int32 truncation
    assert path[0] == truth[0][6]
    np.testing.assert_array_equal(labels[0], truth[0][5])
    # initializer rewinds; drop_remainder drops the short batch
    it.initializer()
    assert len(it.get_next()[0]) == 2
    _, it2 = dm.get_iterator(ds, batch_size=2, n_epochs=1, drop_remainder=True)
    assert sum(1 for _ in it2) == 2


def test_shuffle_is_a_permutation_and_seeded(tmp_path):
    files, truth = _write_dataset(tmp_path, 7)
    dm = dr.DataManager(2304, 257, 136, buffer_size=3)
    names = lambda it: [p for b in it for p in b[3]]
    _, a = dm.get_iterator(dm.get_dataset(files, shuffle=True, seed=5), batch_size=3, n_epochs=1)
    _, b = dm.get_iterator(dm.get_dataset(files, shuffle=True, seed=5), batch_size=3, n_epochs=1)
    na, nb = names(a), names(b)
    assert na == nb and sorted(na) == sorted(t[6] for t in truth) and na != [t[6] for t in truth]


def test_embedding_reader_and_rank_sharding(tmp_path):
    files, truth = _write_dataset(tmp_path, 6, with_emb=True)
    dm = dr.DataManager(2304, 257, 136, buffer_size=2, embedding_size=512)
    ds = dm.get_dataset(files, shuffle=False)
    _, it0 = dm.get_iterator(ds, batch_size=2, n_epochs=1, shard=(0, 2))
    _, it1 = dm.get_iterator(ds, batch_size=2, n_epochs=1, shard=(1, 2))
    b0, b1 = list(it0), list(it1)
    assert len(b0) == 2 and len(b1) == 1
    assert len(b0[0]) == 8 and b0[0][3].shape == (2, 512)
    got = [p for b in (b0[0], b1[0], b0[1]) for p in b[4]]
    assert got == [t[6] for t in truth]


@pytest.mark.parametrize("with_emb", [False, True])
def test_native_decoder_equals_python_parser(tmp_path, with_emb):
    """avsi_sequence_example_decode_fixed_host (what the iterator uses) against read_data_format_fixed, the
    pure-Python statement of the schema: same dtypes, shapes and bits, batch by batch."""
    files, _ = _write_dataset(tmp_path, 5, with_emb=with_emb)
    dm = dr.DataManager(2304, 257, 136, buffer_size=4, embedding_size=512 if with_emb else None)
    ds = dm.get_dataset(files, shuffle=False)
    _, fast = dm.get_iterator(ds, batch_size=2, n_epochs=2)
    _, slow = dm.get_iterator(ds, batch_size=2, n_epochs=2, native=False, prefetch=0)
    nb = 0
    for a, b in zip(fast, slow):
        nb += 1
        assert len(a) == len(b) == (8 if with_emb else 7)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape
            assert (x == y).all()
    assert nb == 5                                          # 10 examples over two epochs, batches of 2


def test_native_decoder_handles_unpacked_lists_missing_video_and_bad_records():
    """Wire-format corners: unpacked float lists (legal protobuf), a sample without video features,
    ragged batches and truncated records."""
    rec, t = _sample(3)
    dm = dr.DataManager(2304, 257, 136)
    # re-encode the mask frames with UNPACKED floats (field 1, wire type 5 per value)
    def unpacked_feature(row):
        body = b''.join(tio._varint((1 << 3) | 5) + struct.pack('<f', v) for v in row)
        return tio._ld(2, body)
    ctx, lists = tio.decode_sequence_example(rec)
    fl = tio._encode_map({'mask': lists['mask']}, lambda steps: b''.join(tio._ld(1, unpacked_feature(s)) for s in steps))
    fl += tio._encode_map({'labels': lists['labels']}, lambda steps: b''.join(tio._ld(1, tio.encode_feature(s)) for s in steps))
    novideo = tio._ld(1, tio._encode_map({k: (v if not isinstance(v, list) else v[0]) for k, v in ctx.items()},
                                         tio.encode_feature)) + tio._ld(2, fl)
    out = dm.decode_batch([novideo])
    ref = dm.read_data_format_fixed(novideo)
    assert out[5].shape == (1, 0, 136) and ref[5].shape == (0, 136)
    np.testing.assert_array_equal(out[6][0], t[4])
    np.testing.assert_array_equal(out[2][0], np.trunc(t[2]).astype(np.int32))
    # ragged batch: second record has another number of frames
    other, _ = _sample(4, T=9)
    with pytest.raises(ValueError, match="sizes of the first"):
        dm.decode_batch([rec, other])
    with pytest.raises(ValueError, match="malformed"):
        dm.decode_batch([rec[:len(rec) // 2]])
    with pytest.raises(ValueError, match="expected 100"):
        dr.DataManager(100, 257, 136).decode_batch([rec])
    with pytest.raises(ValueError):
        dr.DataManager(2304, 200, 136).decode_batch([rec])          # mask width mismatch
    with pytest.raises(ValueError):
        dr.DataManager(2304, 257, 136, embedding_size=512).decode_batch([rec])   # no embedding in the record


def test_prefetcher_propagates_errors_and_rewinds(tmp_path):
    files, truth = _write_dataset(tmp_path, 6)
    dm = dr.DataManager(2304, 257, 136, buffer_size=2)
    ds = dm.get_dataset(files, shuffle=False)
    _, it = dm.get_iterator(ds, batch_size=2, n_epochs=1, prefetch=3)
    first = it.get_next()
    it.initializer()                                        # rewind while the background thread is ahead
    again = it.get_next()
    assert (first[2] == again[2]).all() and list(first[3]) == list(again[3])
    assert sum(1 for _ in it) == 2
    with pytest.raises(dr.OutOfRangeError):
        it.get_next()
    with pytest.raises(dr.OutOfRangeError):                 # stays exhausted
        it.get_next()
    # a corrupted file surfaces as the reader's IOError on the consumer's thread
    raw = bytearray(open(files[3], "rb").read())
    raw[100] ^= 0xFF
    open(files[3], "wb").write(bytes(raw))
    _, bad = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=2, n_epochs=1)
    assert len(bad.get_next()[0]) == 2
    with pytest.raises(IOError):
        bad.get_next()
        bad.get_next()


def test_native_decoder_survives_corrupted_records():
    """The decoder parses untrusted bytes: truncations, bit flips and random splices of a valid record must
    end in a clean status (a ValueError here), never in a crash, a hang or an out-of-bounds write."""
    rec, _ = _sample(5, with_emb=True)
    dm = dr.DataManager(2304, 257, 136, embedding_size=512)
    ref = dm.decode_batch([rec])
    rng = np.random.default_rng(0)
    outcomes = {'ok': 0, 'rejected': 0}
    for trial in range(600):
        buf = bytearray(rec)
        kind = trial % 4
        if kind == 0:
            buf = buf[:rng.integers(0, len(buf))]
        elif kind == 1:
            for _ in range(rng.integers(1, 6)):
                buf[rng.integers(0, len(buf))] ^= 1 << rng.integers(0, 8)
        elif kind == 2:                                      # corrupt a length / tag byte near the start of a field
            pos = rng.integers(0, min(len(buf), 4000))
            buf[pos] = rng.integers(0, 256)
        else:                                                # splice a random chunk somewhere
            a, b = sorted(rng.integers(0, len(buf), size=2))
            buf[a:a] = bytes(rng.integers(0, 256, size=rng.integers(1, 64), dtype=np.uint8))
            del buf[b:b + rng.integers(0, 64)]
        try:
            out = dm.decode_batch([bytes(buf)])
            outcomes['ok'] += 1
            assert out[2].shape == ref[2].shape and out[7].shape[2] == 257
        except ValueError:
            outcomes['rejected'] += 1
    assert outcomes['rejected'] > 300 and outcomes['ok'] + outcomes['rejected'] == 600
    np.testing.assert_array_equal(dm.decode_batch([rec])[2], ref[2])     # and the decoder still works afterwards


def test_even_rounds_give_every_rank_the_same_number_of_batches(tmp_path):
    """7 samples, batches of 2, two ranks: 3 full batches + 1 short one.  A training loop does one collective per
    step, so with even_rounds each rank gets exactly one batch (round 0) and the rest of the epoch is dropped;
    without it rank 0 would get two and rank 1 one -- and rank 0 would wait for ever."""
    files, truth = _write_dataset(tmp_path, 7)
    dm = dr.DataManager(2304, 257, 136, buffer_size=2)
    ds = dm.get_dataset(files, shuffle=False)
    got = []
    for rank in (0, 1):
        _, it = dm.get_iterator(ds, batch_size=2, n_epochs=1, shard=(rank, 2), even_rounds=True)
        got.append([list(b[3]) for b in it])
    assert [len(g) for g in got] == [1, 1]
    assert got[0][0] == [truth[0][6], truth[1][6]] and got[1][0] == [truth[2][6], truth[3][6]]
    _, plain = dm.get_iterator(ds, batch_size=2, n_epochs=1, shard=(0, 2))
    assert [len(b[0]) for b in plain] == [2, 2]            # batches 0 and 2 (the short one, index 3, is rank 1's)
    _, single = dm.get_iterator(ds, batch_size=2, n_epochs=1, even_rounds=True)
    assert [len(b[0]) for b in single] == [2, 2, 2, 1]     # one process: nothing to even out


def test_even_rounds_with_eight_readers(tmp_path):
    """BASELINE configs[3]'s world size: 8 readers over 21 batches of 1 = two whole rounds of 8 + 5 left over.  Every rank
    gets exactly two batches, rank r the batches r and 8 + r; the 5 left over are dropped on every rank alike."""
    files, truth = _write_dataset(tmp_path, 21)
    dm = dr.DataManager(2304, 257, 136, buffer_size=2)
    ds = dm.get_dataset(files, shuffle=False)
    seen = []
    for rank in range(8):
        _, it = dm.get_iterator(ds, batch_size=1, n_epochs=1, shard=(rank, 8), even_rounds=True)
        got = [list(b[3]) for b in it]
        assert got == [[truth[rank][6]], [truth[8 + rank][6]]], rank
        seen += [g[0] for g in got]
    assert len(set(seen)) == 16


def test_whole_file_reader_matches_the_framed_path(tmp_path, monkeypatch):
    """One-record files are read, checked and parsed natively on parallel threads (avsi_tfrecord_file_decode_fixed_host):
    same batches, same shuffle order as the Python framing loop + in-memory decoder (AVSI_READER_FILES=0), with and
    without the embedding feature; a dataset whose FIRST file holds several records takes the framed path by itself,
    and one that turns out to hold several records later fails with a message that says so."""
    files, _ = _write_dataset(tmp_path, 11, with_emb=True)
    dm = dr.DataManager(2304, 257, 136, buffer_size=5, embedding_size=512)

    def batches(flag):
        monkeypatch.setenv('AVSI_READER_FILES', flag)
        ds = dm.get_dataset(files, shuffle=True, seed=3)
        assert ds.record_files == (flag == '1')
        _, it = dm.get_iterator(ds, batch_size=4, n_epochs=2, prefetch=0)
        return [tuple(np.array(f) for f in b) for b in it]

    native, framed = batches('1'), batches('0')
    assert [len(b[0]) for b in native] == [len(b[0]) for b in framed] and len(native) == 6
    for a, b in zip(native, framed):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape
            assert (x == y).all()
    monkeypatch.setenv('AVSI_READER_FILES', '1')
    # several records in the first file: the framed path, by itself
    multi = str(tmp_path / "multi_00000.tfrecord")
    tio.write_records(multi, [_sample(50, with_emb=True)[0], _sample(51, with_emb=True)[0]])
    ds = dm.get_dataset([multi] + files[:2], shuffle=False)
    assert not ds.record_files
    _, it = dm.get_iterator(ds, batch_size=4, n_epochs=1, prefetch=0)
    assert len(next(it)[0]) == 4
    # several records in a LATER file: a clear error
    ds = dm.get_dataset(files[:2] + [multi], shuffle=False)
    assert ds.record_files
    _, it = dm.get_iterator(ds, batch_size=3, n_epochs=1, prefetch=0)
    with pytest.raises(ValueError, match="more than one record"):
        next(it)


def test_var_schema_round_trip_and_padded_batches(tmp_path):
    """The 'var' TFRecord schema (reference dataset_reader.py:82-99; get_iterator :49-55): lengths in the context, audio /
    sample path / labels / video / mask as feature lists; batches padded with zeros to the longest record, the audio stays
    float32, the sample path is a row of character codes.  The reference's writer for this mode cannot run:
    the records come from serialize_sample_var, which writes what that reader parses."""
    import avsi_amd  # noqa: F401
    from avsi_amd import tfrecord_io as tio
    from avsi_amd.dataset_reader import DataManager, OutOfRangeError
    rng = np.random.default_rng(5)
    samples = []
    for i, (n, t) in enumerate([(700, 4), (960, 5), (500, 3), (640, 6), (320, 2)]):
        wav = (rng.normal(size=n) * 1000).astype(np.float32) + 0.25          # (fractions survive: no tf.to_int32 in this mode)
        video = rng.normal(size=(t, 6)).astype(np.float32)
        mask = np.ones((t, 9), np.float32)
        mask[1:2] = 0
        labels = np.arange(2 + i, dtype=np.float32)
        samples.append((t, 2 + i, wav, video, mask, labels, "dir/clip_%d" % i))
    files = []
    for k in range(2):                  # two files, three and two records
        part = samples[3 * k: 3 * k + 3]
        f = str(tmp_path / ("part%d.tfrecord" % k))
        tio.write_records(f, [tio.serialize_sample_var(t, ll, w, v, m, lab, path) for t, ll, w, v, m, lab, path in part])
        files.append(f)
    # one record, field by field
    ctx, seq = tio.decode_sequence_example(next(iter(tio.read_records(files[0]))))
    assert sorted(ctx) == ['labels_length', 'sequence_length'] and int(ctx['sequence_length'][0]) == 4
    assert sorted(seq) == ['labels', 'mask', 'sample_path', 'target_audio_wav', 'video_features']
    assert len(seq['target_audio_wav']) == 700 and len(seq['sample_path']) == len("dir/clip_0")
    dm = DataManager(num_audio_samples=0, audio_feat_size=9, video_feat_size=6, mode='var')
    one = dm.read_data_format_var(next(iter(tio.read_records(files[0]))))
    np.testing.assert_array_equal(one[2], samples[0][2])
    assert one[2].dtype == np.float32 and one[3].dtype == np.int64 and bytes(one[3].tolist()).decode() == "dir/clip_0"
    for drop, sizes in ((False, [2, 2, 1]), (True, [2, 2])):
        _, it = dm.get_iterator(dm.get_dataset(files, shuffle=False), batch_size=2, n_epochs=1, drop_remainder=drop)
        got = []
        while True:
            try:
                got.append(it.get_next())
            except OutOfRangeError:
                break
        assert [len(b[0]) for b in got] == sizes
        at = 0
        for b in got:
            part = samples[at: at + len(b[0])]
            at += len(b[0])
            n_max, t_max = max(len(p[2]) for p in part), max(p[0] for p in part)
            lab_max, path_max = max(len(p[5]) for p in part), max(len(p[6]) for p in part)
            assert b[2].shape == (len(part), n_max) and b[3].shape == (len(part), path_max) and b[4].shape == (len(part), lab_max)
            assert b[5].shape == (len(part), t_max, 6) and b[6].shape == (len(part), t_max, 9)
            for i, (t, ll, w, v, m, lab, path) in enumerate(part):
                assert b[0][i] == t and b[1][i] == ll
                np.testing.assert_array_equal(b[2][i, :len(w)], w)
                assert not b[2][i, len(w):].any() and not b[6][i, t:].any() and not b[5][i, t:].any()
                np.testing.assert_array_equal(b[5][i, :t], v)
                np.testing.assert_array_equal(b[6][i, :t], m)
                np.testing.assert_array_equal(b[4][i, :len(lab)], lab)
                assert bytes(b[3][i, :len(path)].tolist()).decode() == path
    with pytest.raises(ValueError):
        DataManager(mode='grouped')
    with pytest.raises(ValueError):
        DataManager(mode='var', embedding_size=512)


def test_get_dataset_group_interleaves_files_in_blocks(tmp_path):
    """DataManager.get_dataset_group (reference dataset_reader.py:36-45: list_files + interleave(cycle_length, block_length)):
    three files of 5, 2 and 4 records, two files open at a time, blocks of two records -- the order tf.data's interleave gives."""
    import avsi_amd  # noqa: F401
    from avsi_amd import tfrecord_io as tio
    from avsi_amd.dataset_reader import DataManager
    T, N = 3, 64
    sizes = {'a': 5, 'b': 2, 'c': 4}
    for name, n in sizes.items():
        recs = [tio.serialize_sample_fixed(T, 1, np.full(N, 100 * (ord(name) - 96) + i, np.float32), np.zeros((T, 4), np.float32),
                                           np.ones((T, 5), np.float32), np.zeros(1), "%s%d" % (name, i)) for i in range(n)]
        tio.write_records(str(tmp_path / ("%s.tfrecord" % name)), recs)
    dm = DataManager(num_audio_samples=N, audio_feat_size=5, video_feat_size=4)
    ds = dm.get_dataset_group(str(tmp_path / "*.tfrecord"), cycle_length=2, block_length=2, shuffle=False)
    order = [ex[3].decode() for ex in ds.examples()]
    assert order == ['a0', 'a1', 'b0', 'b1', 'a2', 'a3', 'c0', 'c1', 'a4', 'c2', 'c3']
    _, it = dm.get_iterator(ds, batch_size=4, n_epochs=1, prefetch=0)
    batches = list(it)
    assert [len(b[0]) for b in batches] == [4, 4, 3] and [p.decode() for p in batches[0][3]] == order[:4]
    shuffled = dm.get_dataset_group([str(tmp_path / "*.tfrecord")], cycle_length=2, block_length=2, shuffle=True, seed=3)
    first, second = [ex[3] for ex in shuffled.examples()], [ex[3] for ex in shuffled.examples()]
    assert sorted(first) == sorted(second) == sorted(p.encode() for p in order) and len(first) == 11
