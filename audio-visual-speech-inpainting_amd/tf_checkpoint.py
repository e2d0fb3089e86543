"""TensorFlow checkpoints (tensor bundles) without TensorFlow: reader, writer, variable mapping.

The reference saves and restores its models with ``tf.train.Saver`` (training.py:114,334-340,
inference.py:108-109): ``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` (+ a ``checkpoint``
state file).  This module reads and writes that format so a ``sinet`` checkpoint trained with the
reference drops into this build and vice versa.

Formats (TensorFlow 1.x, ``tensorflow/core/util/tensor_bundle`` and ``tensorflow/core/lib/io/table``,
none of it present here -- restated from the published formats; **parity unpinned**: there is no
TensorFlow in this image and the reference ships no checkpoint, so the only checks are
self-consistency, the protobuf runtime's view of the same messages and hand-assembled files):

* ``.index`` is a LevelDB-style sorted table.  Data blocks hold prefix-compressed entries
  ``varint shared | varint non_shared | varint value_len | key suffix | value`` followed by the
  restart offsets (u32 each) and their count (u32); every block is followed by a 5-byte trailer
  ``compression type (0 = none, 1 = snappy) | u32 masked crc32c(block + type)``.  The index block maps
  a separator key per data block to its ``BlockHandle`` (varint offset, varint size).  The file ends
  with a 48-byte footer: metaindex handle, index handle, zero padding to 40 bytes, magic
  ``0xdb4775248b80fb57`` (little endian).
* key ``""`` -> ``BundleHeaderProto {num_shards = 1; endianness = 2; version = 3 {producer = 1}}``;
  every other key is a variable name -> ``BundleEntryProto {dtype = 1; shape = 2 {dim = 2 {size = 1}};
  shard_id = 3; offset = 4; size = 5; crc32c = 6 (fixed32, masked crc32c of the tensor bytes)}``.
* ``.data-SSSSS-of-NNNNN`` holds the raw little-endian tensor bytes at ``offset``.

Variable names follow SURVEY App. C (``<scope>/cudnn_lstm/stack_bidirectional_rnn/cell_<l>/
bidirectional_rnn/{fw,bw}/cudnn_compatible_lstm_cell/{kernel,bias}``, ``<scope>/logits/{weights,
biases}``, ``<scope>/Variable`` = global step, Adam slots ``…/Adam`` and ``…/Adam_1``); import
matches on the scope-independent suffix.
"""
import os
import re
import struct

import numpy as np

from .tfrecord_io import _crc32c, _fields, _ld, _read_varint, _varint, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
_FOOTER_LEN = 48
_BLOCK_TRAILER = 5

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BOOL = 1, 2, 3, 9, 10
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_DOUBLE: np.dtype('<f8'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8'),
           DT_BOOL: np.dtype('bool')}
_DT_OF = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
          np.dtype('int64'): DT_INT64, np.dtype('bool'): DT_BOOL}


class CheckpointError(ValueError):
    """Not a readable tensor bundle (the reference's drivers turn this into exit code 2)."""


# ------------------------------------------------------------------------------- sorted table
def _block_handle(offset, size):
    return _varint(offset) + _varint(size)


def _parse_block(block):
    """Entries of one table block (restart array stripped), in order."""
    if len(block) < 4:
        raise CheckpointError("table block too short")
    (n_restarts,) = struct.unpack_from('<I', block, len(block) - 4)
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise CheckpointError("bad restart count in table block")
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointError("corrupt table entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(buf, offset, size, verify):
    if offset + size + _BLOCK_TRAILER > len(buf):
        raise CheckpointError("table block outside the file")
    body = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        (crc,) = struct.unpack_from('<I', buf, offset + size + 1)
        if crc != masked_crc32c(bytes(buf[offset:offset + size + 1])):
            raise CheckpointError("table block checksum mismatch")
    if ctype != 0:
        raise CheckpointError("compressed table blocks (type %d) are not supported; tensor bundles are written "
                              "uncompressed" % ctype)
    return body


def read_table(path, verify=True):
    """All (key, value) pairs of a sorted-table file, in key order."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if len(buf) < _FOOTER_LEN:
        raise CheckpointError("%s: too short for a table footer" % path)
    footer = buf[-_FOOTER_LEN:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError("%s: bad table magic (not a checkpoint index)" % path)
    pos = 0
    _, pos = _read_varint(footer, pos)      # metaindex handle (unused)
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    out = []
    for _, handle in _parse_block(_read_block(buf, ioff, isize, verify)):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        out.extend(_parse_block(_read_block(buf, off, size, verify)))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.interval = restart_interval
        self.reset()

    def reset(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b''
        self.n = 0

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.count += 1
        self.n += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path, items, block_size=262144):
    """Write (key, value) pairs (keys strictly increasing) as an uncompressed sorted table."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack('<I', masked_crc32c(block + b'\x00')))
        return _block_handle(off, len(block))

    data, index = _BlockBuilder(), _BlockBuilder(restart_interval=1)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        data.add(key, value)
        prev = key
        if data.size() >= block_size:
            index.add(data.last_key, emit(data.finish()))
            data.reset()
    if data.n:
        index.add(data.last_key, emit(data.finish()))
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as fh:
        fh.write(bytes(out))


# ------------------------------------------------------------------------------- bundle protos
def _encode_header(num_shards=1):
    return _varint(1 << 3) + _varint(num_shards) + _ld(3, _varint(1 << 3) + _varint(1))


def _decode_header(buf):
    num_shards, endianness = 0, 0
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 0:
            num_shards = v
        elif f == 2 and wt == 0:
            endianness = v
    return num_shards, endianness


def _encode_entry(dtype, shape, shard_id, offset, size, crc):
    dims = b''.join(_ld(2, _varint(1 << 3) + _varint(int(d))) for d in shape)
    out = _varint(1 << 3) + _varint(dtype) + _ld(2, dims)
    if shard_id:
        out += _varint(3 << 3) + _varint(shard_id)
    if offset:
        out += _varint(4 << 3) + _varint(offset)
    out += _varint(5 << 3) + _varint(size)
    out += _varint((6 << 3) | 5) + struct.pack('<I', crc)
    return out


def _decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for f, wt, v in _fields(buf):
        if f == 1 and wt == 0:
            e['dtype'] = v
        elif f == 2 and wt == 2:
            for f2, wt2, v2 in _fields(v):
                if f2 == 2 and wt2 == 2:
                    size = 0
                    for f3, wt3, v3 in _fields(v2):
                        if f3 == 1 and wt3 == 0:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e['shape'].append(size)
                elif f2 == 3 and wt2 == 0 and v2:
                    raise CheckpointError("tensor of unknown rank in checkpoint")
        elif f == 3 and wt == 0:
            e['shard_id'] = v
        elif f == 4 and wt == 0:
            e['offset'] = v
        elif f == 5 and wt == 0:
            e['size'] = v
        elif f == 6 and wt == 5:
            (e['crc32c'],) = struct.unpack('<I', v)
        elif f == 7:
            e['sliced'] = True
    return e


def _data_path(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


# ------------------------------------------------------------------------------- bundles
def is_bundle(prefix):
    return os.path.isfile(prefix + '.index')


def list_variables(prefix):
    """[(name, shape, numpy dtype)] -- the counterpart of tf.train.list_variables."""
    out = []
    for key, value in read_table(prefix + '.index'):
        if key == b'':
            continue
        e = _decode_entry(value)
        out.append((key.decode('utf-8'), tuple(e['shape']), _DTYPES.get(e['dtype'])))
    return out


def read_bundle(prefix, verify=True):
    """{variable name: numpy array} of the checkpoint ``prefix`` (e.g. ``…/netmodel/sinet``)."""
    if not is_bundle(prefix):
        raise CheckpointError("%s.index not found" % prefix)
    items = read_table(prefix + '.index', verify=verify)
    if not items or items[0][0] != b'':
        raise CheckpointError("%s: missing bundle header" % prefix)
    num_shards, endianness = _decode_header(items[0][1])
    if endianness != 0:
        raise CheckpointError("%s: big-endian bundles are not supported" % prefix)
    shards, out = {}, {}
    for key, value in items[1:]:
        e = _decode_entry(value)
        name = key.decode('utf-8')
        if e['sliced']:
            raise CheckpointError("%s: partitioned variable %s is not supported" % (prefix, name))
        if e['dtype'] not in _DTYPES:
            raise CheckpointError("%s: variable %s has unsupported dtype %d" % (prefix, name, e['dtype']))
        if e['shard_id'] not in shards:
            p = _data_path(prefix, e['shard_id'], max(num_shards, 1))
            try:
                shards[e['shard_id']] = np.memmap(p, dtype=np.uint8, mode='r') if os.path.getsize(p) else np.zeros(0, np.uint8)
            except OSError as err:
                raise CheckpointError("%s: %s" % (prefix, err))
        data = shards[e['shard_id']]
        dt = _DTYPES[e['dtype']]
        n = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if e['size'] != n * dt.itemsize or e['offset'] + e['size'] > data.size:
            raise CheckpointError("%s: variable %s: size / shape mismatch" % (prefix, name))
        raw = bytes(data[e['offset']:e['offset'] + e['size']])
        if verify and e['crc32c'] is not None and e['crc32c'] != masked_crc32c(raw):
            raise CheckpointError("%s: variable %s: data checksum mismatch" % (prefix, name))
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    return out


def write_bundle(prefix, variables, write_state=True):
    """Write {name: array} as ``prefix.index`` + ``prefix.data-00000-of-00001`` (+ the ``checkpoint``
    state file tf.train.latest_checkpoint reads).  Returns ``prefix`` like tf.train.Saver.save."""
    names = sorted(variables, key=lambda s: s.encode('utf-8'))
    items, offset = [(b'', _encode_header(1))], 0
    with open(_data_path(prefix, 0, 1), 'wb') as fh:
        for name in names:
            a = np.asarray(variables[name])
            if a.dtype not in _DT_OF:
                raise ValueError("variable %s: dtype %s cannot be stored" % (name, a.dtype))
            raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False)).tobytes()
            fh.write(raw)
            items.append((name.encode('utf-8'), _encode_entry(_DT_OF[a.dtype], a.shape, 0, offset, len(raw), masked_crc32c(raw))))
            offset += len(raw)
    write_table(prefix + '.index', items)
    if write_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(prefix) or '.', 'checkpoint'), 'w') as fh:
            fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix


# ------------------------------------------------------------------------------- variable mapping
_CELL_RE = re.compile(r'(?:^|/)cell_(\d+)/bidirectional_rnn/(fw|bw)/[^/]+/(kernel|bias)$')
_LOGITS_RE = re.compile(r'(?:^|/)(?:logits|inpainting)/(weights|biases)$')    # multi-task models: scope 'inpainting'
_ASR_RE = re.compile(r'(?:^|/)asr/(weights|biases)$')
_STEP_RE = re.compile(r'(?:^|/)(Variable|global_step)$')
_UNET_RE = re.compile(r'(?:^|/)(w|b)(?:_(\d+))?$')
_BN_RE = re.compile(r'(?:^|/)batch_normalization(?:_(\d+))?/(gamma|beta)$')
_SPK_RE = re.compile(r'(?:^|/)speaker_embedding/((?:weights|biases)_[123])$')


def _local_name(tf_name, unet_specs=None, side_layer=0):
    """TF variable name -> this build's reference-layout entry name (None if it is not a model weight).
    The speaker-embedding models with ``integration_layer >= 1`` keep two stacks, ``blstm_1`` and
    ``blstm_2``, each numbering its cells from 0 (models.py:882-901): blstm_2's cell l is layer
    ``integration_layer + l`` here."""
    m = _CELL_RE.search(tf_name)
    if m:
        li = int(m.group(1)) + (side_layer if '/blstm_2/' in '/' + tf_name else 0)
        return 'cell_%d/%s/%s' % (li, m.group(2), m.group(3))
    m = _LOGITS_RE.search(tf_name)
    if m:
        return 'logits/' + m.group(1)
    m = _ASR_RE.search(tf_name)
    if m:
        return 'asr/' + m.group(1)
    m = _SPK_RE.search(tf_name)
    if m:
        return 'speaker_embedding/' + m.group(1)
    if unet_specs is not None:
        m = _UNET_RE.search(tf_name)
        if m:   # tf.Variable(name='w') is uniquified in creation order: w, w_1, w_2, ... (unet_layers.py:8-9,24-25)
            k = int(m.group(2) or 0)
            return '%s/%s' % (unet_specs[k][0], m.group(1)) if k < len(unet_specs) else None
        m = _BN_RE.search(tf_name)
        if m:   # tf.layers.batch_normalization scopes: batch_normalization, batch_normalization_1, ...
            bn_layers = [s[0] for s in unet_specs if s[4]]
            k = int(m.group(1) or 0)
            return '%s/bn/%s' % (bn_layers[k], m.group(2)) if k < len(bn_layers) else None
    return None


def tf_variable_names(layout, scope):
    """{local entry name: TF variable name} for a BLSTM or U-Net layout under variable scope ``scope``."""
    out = {}
    specs = getattr(layout, 'specs', None)
    if specs is not None:
        j = 0
        for k, spec in enumerate(specs):
            for v in ('w', 'b'):
                out['%s/%s' % (spec[0], v)] = '%s/%s%s' % (scope, v, '_%d' % k if k else '')
            if spec[4]:
                for v in ('gamma', 'beta'):
                    out['%s/bn/%s' % (spec[0], v)] = '%s/batch_normalization%s/%s' % (scope, '_%d' % j if j else '', v)
                j += 1
        return out
    side_layer = layout.side[0] if getattr(layout, 'side', None) else 0
    for name, _, _ in layout.ref_entries:
        m = re.match(r'cell_(\d+)/(fw|bw)/(kernel|bias)$', name)
        if m:
            li, stack = int(m.group(1)), 'cudnn_lstm'
            if side_layer:      # two stacks around the integration point (models.py:882-901)
                stack, li = ('blstm_1/cudnn_lstm', li) if li < side_layer else ('blstm_2/cudnn_lstm', li - side_layer)
            out[name] = '%s/%s/stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/cudnn_compatible_lstm_cell/%s' % (
                scope, stack, li, m.group(2), m.group(3))
        elif getattr(layout, 'asr', 0) and name.startswith('logits/'):
            out[name] = '%s/inpainting/%s' % (scope, name[7:])   # two heads: inpainting/ and asr/ (models.py:1903-1916)
        else:
            out[name] = '%s/%s' % (scope, name)      # logits/*, asr/*, speaker_embedding/*
    return out


def import_variables(bundle, layout, scope=None):
    """Map a bundle ({tf name: array}) onto ``layout``; ``scope`` restricts the match to names
    under that variable scope (the two-step model keeps two networks in one checkpoint).

    Returns (flat parameters, adam_m or None, adam_v or None, global_step).  Optimiser slots are
    taken only when every variable has its ``/Adam`` and ``/Adam_1`` slot in the canonical
    per-variable form (a CudnnLSTM-trained model keeps its slots in cuDNN's opaque layout: the
    weights import, the slots do not).  CheckpointError if a model variable is missing or has the
    wrong shape."""
    specs = getattr(layout, 'specs', None)
    side_layer = layout.side[0] if getattr(layout, 'side', None) else 0
    if scope is not None:
        bundle = {k: a for k, a in bundle.items() if ('/' + k).find('/' + scope + '/') >= 0}
    flat = np.zeros(layout.ref_size, dtype=np.float32)
    slots = {'Adam': np.zeros(layout.ref_size, dtype=np.float32), 'Adam_1': np.zeros(layout.ref_size, dtype=np.float32)}
    seen, seen_slot = set(), {'Adam': set(), 'Adam_1': set()}
    shapes = {n: tuple(s) for n, s, _ in layout.ref_entries}
    step = 0
    for tf_name, arr in bundle.items():
        slot = None
        base = tf_name
        for s in ('Adam_1', 'Adam'):
            if tf_name.endswith('/' + s):
                slot, base = s, tf_name[:-len(s) - 1]
                break
        local = _local_name(base, specs, side_layer)
        if local is None or local not in shapes:
            if slot is None and _STEP_RE.search(tf_name) and arr.ndim == 0 and arr.dtype.kind in 'iu':
                step = int(arr)
            continue
        if tuple(arr.shape) != shapes[local]:
            raise CheckpointError("variable %s has shape %s, the model expects %s" % (tf_name, tuple(arr.shape), shapes[local]))
        if slot is None:
            layout.ref_view(flat, local)[...] = arr
            seen.add(local)
        else:
            layout.ref_view(slots[slot], local)[...] = arr
            seen_slot[slot].add(local)
    missing = [n for n in shapes if n not in seen]
    if missing:
        raise CheckpointError("checkpoint lacks model variables: %s" % ', '.join(missing[:6]))
    have_slots = all(len(seen_slot[s]) == len(shapes) for s in slots)
    return flat, (slots['Adam'] if have_slots else None), (slots['Adam_1'] if have_slots else None), step


def export_variables(layout, flat, scope, adam_m=None, adam_v=None, global_step=0, beta1=0.9, beta2=0.999):
    """The inverse of import_variables: {tf name: array} as tf.train.Saver over
    the model's global variables would store it (models.py:226-231 all_vars).

    Names of the optimiser state follow how the reference creates it: ``train_op`` runs under
    ``tf.variable_scope(<model>)`` and ``tf.name_scope('optimizer')`` (models.py:163-178,224), so the step counter is
    ``<scope>/optimizer/Variable``, the Adam accumulators ``<scope>/optimizer/beta{1,2}_power``, and a slot created
    inside a variable scope gets that scope prefixed to the variable's own name (``<scope>/<scope>/.../kernel/Adam``).
    UNVERIFIED (no TensorFlow here, no checkpoint in the reference): each of these is ALSO written under the plain
    name (``<scope>/Variable``, ``<scope>/beta1_power``, ``<var>/Adam``) -- extra entries are harmless to
    ``Saver.restore``, a missing one is NotFound.  U-Net layouts also get the batch-norm moving statistics the
    reference never updates (zeros / ones: UPDATE_OPS are not run) -- they are GLOBAL_VARIABLES."""
    names = tf_variable_names(layout, scope)
    out = {}
    for local, tf_name in names.items():
        out[tf_name] = np.array(layout.ref_view(flat, local), dtype=np.float32)
        if adam_m is not None and adam_v is not None:
            for prefix in ('', scope + '/'):
                out[prefix + tf_name + '/Adam'] = np.array(layout.ref_view(adam_m, local), dtype=np.float32)
                out[prefix + tf_name + '/Adam_1'] = np.array(layout.ref_view(adam_v, local), dtype=np.float32)
        m = _BN_RE.search(tf_name)
        if m and m.group(2) == 'gamma':
            base = tf_name[:-len('gamma')]
            n = out[tf_name].shape
            out[base + 'moving_mean'] = np.zeros(n, dtype=np.float32)
            out[base + 'moving_variance'] = np.ones(n, dtype=np.float32)
    for prefix in ('%s/optimizer/' % scope, '%s/' % scope):
        out[prefix + 'Variable'] = np.array(global_step, dtype=np.int32)
        if adam_m is not None and adam_v is not None:
            out[prefix + 'beta1_power'] = np.array(beta1 ** (global_step + 1), dtype=np.float32)
            out[prefix + 'beta2_power'] = np.array(beta2 ** (global_step + 1), dtype=np.float32)
    return out
