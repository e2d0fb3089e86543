"""TFRecord files and tf.train.SequenceExample records without TensorFlow.

Wire formats (what the reference's writer ``tfrecord_utils.serialize_sample_fixed`` :19-41 and
reader ``dataset_reader.DataManager.read_data_format_fixed`` :62-79 exchange through
``tf.python_io.TFRecordWriter`` / ``tf.data.TFRecordDataset``):

* TFRecord framing: ``u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)``,
  little endian, ``masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8 (mod 2^32)``;
* payload: protobuf ``tensorflow.SequenceExample {Features context = 1; FeatureLists feature_lists = 2}``
  with ``Features{map<string,Feature> feature = 1}``, ``Feature{BytesList = 1 | FloatList = 2 |
  Int64List = 3}``, ``FloatList/Int64List{repeated value = 1 [packed]}``, ``BytesList{repeated bytes
  value = 1}``, ``FeatureLists{map<string,FeatureList> feature_list = 1}``,
  ``FeatureList{repeated Feature feature = 1}``.

CRC-32C comes from the native library (``avsi_crc32c``, a host routine).
"""
import ctypes
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def _crc32c(data):
    from . import _lib
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, bytes) else data
    return _lib.lib().avsi_crc32c(buf, len(data), 0)


def masked_crc32c(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------------------- framing
def write_records(path, payloads):
    with open(path, 'wb') as fh:
        for p in payloads:
            head = struct.pack('<Q', len(p))
            fh.write(head)
            fh.write(struct.pack('<I', masked_crc32c(head)))
            fh.write(p)
            fh.write(struct.pack('<I', masked_crc32c(p)))


def read_records(path, verify=True):
    """Yield the payload of every record of a TFRecord file."""
    with open(path, 'rb') as fh:
        while True:
            head = fh.read(8)
            if not head:
                return
            if len(head) < 8:
                raise IOError("%s: truncated record header" % path)
            (n,) = struct.unpack('<Q', head)
            (hcrc,) = struct.unpack('<I', fh.read(4))
            if verify and hcrc != masked_crc32c(head):
                raise IOError("%s: corrupted record length (crc mismatch)" % path)
            data = fh.read(n)
            tail = fh.read(4)
            if len(data) < n or len(tail) < 4:
                raise IOError("%s: truncated record" % path)
            if verify and struct.unpack('<I', tail)[0] != masked_crc32c(data):
                raise IOError("%s: corrupted record payload (crc mismatch)" % path)
            yield data


# ---------------------------------------------------------------------------- protobuf wire
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    """length-delimited field"""
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _read_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Iterate (field number, wire type, value) over a serialized message."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


# ---------------------------------------------------------------------------- Feature
def encode_feature(value):
    """numpy float array -> FloatList, int array / int -> Int64List, bytes / list of bytes -> BytesList."""
    if isinstance(value, (bytes, str)):
        value = [value]
    if isinstance(value, (list, tuple)) and value and isinstance(value[0], (bytes, str)):
        body = b''.join(_ld(1, v.encode() if isinstance(v, str) else v) for v in value)
        return _ld(1, body)
    arr = np.asarray(value)
    if arr.dtype.kind in 'iub':
        body = b''.join(_varint(int(v)) for v in arr.reshape(-1))
        return _ld(3, _ld(1, body))
    return _ld(2, _ld(1, np.ascontiguousarray(arr, dtype='<f4').tobytes()))


def decode_feature(buf):
    for field, wt, val in _fields(buf):
        inner = bytes(val)
        if field == 1:                                   # BytesList
            return [bytes(v) for f, _, v in _fields(inner) if f == 1]
        if field == 2:                                   # FloatList (packed or not)
            chunks = []
            for f, w, v in _fields(inner):
                if f == 1:
                    chunks.append(np.frombuffer(bytes(v), dtype='<f4'))
            return np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.float32)
        if field == 3:                                   # Int64List
            out = []
            for f, w, v in _fields(inner):
                if f != 1:
                    continue
                if w == 0:
                    out.append(v)
                else:
                    b, p = bytes(v), 0
                    while p < len(b):
                        x, p = _read_varint(b, p)
                        out.append(x)
            arr = np.array(out, dtype=np.uint64).astype(np.int64)
            return arr
    return np.zeros(0, dtype=np.float32)                 # empty Feature


def _encode_map(entries, encode_value):
    out = b''
    for key in entries:
        out += _ld(1, _ld(1, key.encode()) + _ld(2, encode_value(entries[key])))
    return out


def _decode_map(buf, decode_value):
    out = {}
    for field, _, entry in _fields(buf):
        if field != 1:
            continue
        key, val = None, b''
        for f, _, v in _fields(bytes(entry)):
            if f == 1:
                key = bytes(v).decode()
            elif f == 2:
                val = bytes(v)
        out[key] = decode_value(val)
    return out


def encode_sequence_example(context, feature_lists):
    """context: {name: value}; feature_lists: {name: [value per step]} -> serialized SequenceExample."""
    ctx = _encode_map(context, encode_feature)
    fls = _encode_map(feature_lists, lambda steps: b''.join(_ld(1, encode_feature(s)) for s in steps))
    return _ld(1, ctx) + _ld(2, fls)


def decode_sequence_example(buf):
    """-> (context {name: array | [bytes]}, feature_lists {name: [array per step]})."""
    context, lists = {}, {}
    for field, _, val in _fields(buf):
        if field == 1:
            context = _decode_map(bytes(val), decode_feature)
        elif field == 2:
            lists = _decode_map(bytes(val),
                                lambda b: [decode_feature(bytes(v)) for f, _, v in _fields(b) if f == 1])
    return context, lists


# ---------------------------------------------------------------------------- the reference's schema
def serialize_sample_fixed(seq_len, lab_len, target_audio_wav, video_features, mask, labels, sample_path,
                           embedding=None):
    """Same record as the reference's serialize_sample_fixed (tfrecord_utils.py:19-41; with
    `embedding`, tfrecord_emb_utils.py:19-42)."""
    context = {
        'sequence_length': np.array([seq_len], dtype=np.int64),
        'labels_length': np.array([lab_len], dtype=np.int64),
        'target_audio_wav': np.asarray(target_audio_wav, dtype=np.float32),
        'sample_path': sample_path.encode() if isinstance(sample_path, str) else sample_path,
    }
    if embedding is not None:
        context['embedding'] = np.asarray(embedding, dtype=np.float32)
    lists = {
        'mask': [np.asarray(m, dtype=np.float32) for m in mask],
        'video_features': [np.asarray(v, dtype=np.float32) for v in video_features],
        'labels': [np.asarray([lab], dtype=np.float32) for lab in labels],
    }
    return encode_sequence_example(context, lists)


def serialize_sample_var(seq_len, lab_len, target_audio_wav, video_features, mask, labels, sample_path):
    """The 'var' record of the reference (tfrecord_utils.py:43-64) AS ITS READER EXPECTS IT (dataset_reader.py:82-99): lengths
    in the context, everything else as feature lists -- ``target_audio_wav`` one float per step, ``sample_path`` one int64
    character code per step, ``labels`` one float per step, ``video_features`` / ``mask`` one vector per frame.  The reference's
    own writer for this mode cannot run (it fills ``fl_target`` / ``fl_mix_audio_path``, names that do not exist: SURVEY App.
    B11); this is the record it was meant to produce, so that datasets of that shape can be read."""
    path = sample_path.decode() if isinstance(sample_path, bytes) else sample_path
    context = {
        'sequence_length': np.array([seq_len], dtype=np.int64),
        'labels_length': np.array([lab_len], dtype=np.int64),
    }
    lists = {
        'target_audio_wav': [np.asarray([v], dtype=np.float32) for v in np.asarray(target_audio_wav).reshape(-1)],
        'video_features': [np.asarray(v, dtype=np.float32) for v in video_features],
        'mask': [np.asarray(m, dtype=np.float32) for m in mask],
        'labels': [np.asarray([lab], dtype=np.float32) for lab in labels],
        'sample_path': [np.asarray([ord(ch)], dtype=np.int64) for ch in path],
    }
    return encode_sequence_example(context, lists)
