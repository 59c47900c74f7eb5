"""TensorFlow "tensor bundle" checkpoints without TensorFlow: reader and writer for the reference's saved-model contract

    model.save_weights(os.path.join(model_fname, 'model.ckpt'), save_format='tf')        (train2D.py:235)
    model.load_weights(os.path.join(params.model_path, 'model.ckpt'))                     (Inference2D.py:34)

i.e. the pair `model.ckpt.index` + `model.ckpt.data-00000-of-00001` (the authors' pretrained models, README.md:95-97,
come in this format).  SURVEY §8f-3b.

Published format, restated here (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, LevelDB table format):

  <prefix>.index   an SSTable: sorted (key, value) pairs in prefix-compressed blocks
      block    = entries | restart offsets (uint32 each) | num_restarts (uint32)
      entry    = varint32 shared | varint32 non_shared | varint32 value_len | key[shared:] | value
      on disk  = block | 1 byte compression type (0 none, 1 snappy) | uint32 masked CRC-32C of (block + type byte)
      footer   = metaindex handle | index handle (varint64 offset, varint64 size each) | zero padding to 40 bytes |
                 magic 0xdb4775248b80fb57 (8 bytes little-endian);   the index block maps separator keys to data-block handles
      key ""   -> BundleHeaderProto {1: num_shards, 2: endianness (0 little), 3: VersionDef {1: producer}}
      key name -> BundleEntryProto  {1: dtype, 2: TensorShapeProto {2: repeated Dim {1: size}}, 3: shard_id, 4: offset,
                                     5: size, 6: fixed32 masked CRC-32C of the tensor bytes}
  <prefix>.data-00000-of-00001   the raw little-endian tensor bytes back to back, in key order
  masked crc = ((crc >> 15) | (crc << 17)) + 0xa282ead8

Object-based checkpoints (what tf.keras writes for save_format='tf') name every variable by its attribute path from the
model object, `<path>/.ATTRIBUTES/VARIABLE_VALUE`, and carry the object graph itself as a serialized
TrackableObjectGraph string tensor under `_CHECKPOINTABLE_OBJECT_GRAPH`.  For the reference's classes (Networks.py:35-254)
the paths are

    DownLayers/<i>/ConvLSTM/<j>/cell/{kernel,recurrent_kernel,bias}        UpLayers/<i>/Conv/<c>/{kernel,bias}
    DownLayers/<i>/Conv/<c>/{kernel,bias}                                   UpLayers/<i>/BN/<c>/{gamma,beta,moving_mean,
    DownLayers/<i>/BN/<c>/{gamma,beta,moving_mean,moving_variance}                            moving_variance}

(TF releases that also register `layer_with_weights-<n>` dependencies on subclassed models may name a block by that
shorter path instead; the reader accepts both spellings and checks every shape).  PINNING STATUS: no TensorFlow and no
real checkpoint exist in this environment -- the byte-level format is pinned by this module's own known-answer tests
(tests/test_tf_bundle.py: a hand-assembled bundle, CRC vectors, round trips), the key names by the published Keras
object-graph rules; tools/tf_pin.py re-checks both against a real TensorFlow wherever one is importable.
"""
import os
import re
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
HEADER_KEY = b''
OBJECT_GRAPH_KEY = b'_CHECKPOINTABLE_OBJECT_GRAPH'
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
          6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
          22: np.dtype('<u4'), 23: np.dtype('<u8')}
DT_STRING = 7
_NP2DT = {v: k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------- checksums
def crc32c(data, crc=0):
    """CRC-32C (Castagnoli); crc32c(b'123456789') == 0xE3069283.  Uses the kernel library's host routine (lu_crc32c)."""
    from lu_native import cabi, ops
    global _LIB
    if _LIB is None:
        _LIB = cabi.bind(ops.LIB_PATH)      # host routine: works without a GPU
    buf = data if isinstance(data, bytes) else bytes(data)
    return int(_LIB.lu_crc32c(buf, len(buf), crc))


_LIB = None


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------- varints / protobuf wire
def put_varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def get_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def pb_fields(buf):
    """[(field number, wire type, value)] of one protobuf message; value = int (varint / fixed) or bytes."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = get_varint(buf, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = get_varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wire)
        out.append((num, wire, v))
    return out


def pb_varint(num, v):
    return put_varint(num << 3) + put_varint(v)


def pb_bytes(num, data):
    return put_varint((num << 3) | 2) + put_varint(len(data)) + data


def pb_fixed32(num, v):
    return put_varint((num << 3) | 5) + struct.pack('<I', v)


# ------------------------------------------------------------------------------------------- snappy (read side only)
def snappy_decompress(buf):
    n, pos = get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch')
    return bytes(out)


# ------------------------------------------------------------------------------------------- SSTable
def _read_block(buf, offset, size, verify=True):
    raw, ctype = buf[offset:offset + size], buf[offset + size]
    if verify:
        stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(bytes(buf[offset:offset + size + 1])):
            raise ValueError('table block checksum mismatch at offset %d' % offset)
    if ctype == 1:
        raw = snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError('unknown block compression type %d' % ctype)
    return bytes(raw)


def _block_entries(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 * (n_restarts + 1)
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path, verify=True):
    """-> [(key bytes, value bytes)] of an SSTable file, in key order."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow / LevelDB table (bad magic)' % path)
    foot = buf[-48:]
    _, p = get_varint(foot, 0)          # metaindex offset
    _, p = get_varint(foot, p)          # metaindex size
    ioff, p = get_varint(foot, p)
    isize, p = get_varint(foot, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        off, q = get_varint(handle, 0)
        size, _ = get_varint(handle, q)
        out += _block_entries(_read_block(buf, off, size, verify))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, entries, block_size=262144):
    """entries: [(key bytes, value bytes)] sorted by key.  Uncompressed blocks, restart interval 16."""
    body, index, cur, cur_bytes = bytearray(), [], [], 0

    def emit(blk_entries):
        blk = _build_block(blk_entries)
        handle = put_varint(len(body)) + put_varint(len(blk))
        body.extend(blk + b'\x00' + struct.pack('<I', mask_crc(crc32c(blk + b'\x00'))))
        return handle

    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(cur)))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b'', emit(cur)))
    meta = emit([])
    idx = emit(index)
    foot = meta + idx
    foot += b'\x00' * (40 - len(foot)) + struct.pack('<Q', MAGIC)
    with open(path, 'wb') as fh:
        fh.write(bytes(body) + foot)


# ------------------------------------------------------------------------------------------- bundle
def _parse_entry(val):
    e = {'dtype': 0, 'shape': (), 'shard': 0, 'offset': 0, 'size': 0, 'crc': None, 'sliced': False}
    for num, _, v in pb_fields(val):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            e['shape'] = tuple(dict((n, x) for n, _, x in pb_fields(d)).get(1, 0)
                               for n2, _, d in pb_fields(v) if n2 == 2)
        elif num == 3:
            e['shard'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc'] = v
        elif num == 7:
            e['sliced'] = True
    return e


def list_bundle(prefix):
    """-> {name: entry dict} (dtype code, shape, offset, size, crc) of every tensor in the bundle."""
    out = {}
    for k, v in read_table(prefix + '.index'):
        if k == HEADER_KEY:
            hdr = dict((n, x) for n, _, x in pb_fields(v))
            if hdr.get(2, 0) != 0:
                raise ValueError('big-endian bundles are not supported')
            continue
        out[k.decode()] = _parse_entry(v)
    return out


def read_bundle(prefix, names=None, verify=True):
    """-> {name: numpy array} of the numeric tensors (string tensors such as the object graph are skipped)."""
    entries = list_bundle(prefix)
    shards = {}
    out = {}
    for name, e in entries.items():
        if (names is not None and name not in names) or e['dtype'] == DT_STRING:
            continue
        if e['sliced']:
            raise ValueError('%s: partitioned (sliced) variables are not supported' % name)
        if e['dtype'] not in DTYPES:
            raise ValueError('%s: unsupported dtype code %d' % (name, e['dtype']))
        if e['shard'] not in shards:
            n_sh = 1 + max(x['shard'] for x in entries.values())
            shards[e['shard']] = np.memmap('%s.data-%05d-of-%05d' % (prefix, e['shard'], n_sh), dtype=np.uint8, mode='r')
        raw = shards[e['shard']][e['offset']:e['offset'] + e['size']]
        if verify and e['crc'] is not None and unmask_crc(e['crc']) != crc32c(raw.tobytes()):
            raise ValueError('%s: tensor checksum mismatch' % name)
        dt = DTYPES[e['dtype']]
        if int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize != e['size']:
            raise ValueError('%s: shape %s does not match %d bytes' % (name, e['shape'], e['size']))
        out[name] = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e['shape']).copy()
    return out


def read_scalar_string(prefix, name, verify=True):
    """The bytes of a scalar DT_STRING tensor (e.g. the object graph), or None when the bundle has no such entry."""
    entries = list_bundle(prefix)
    e = entries.get(name.decode() if isinstance(name, bytes) else name)
    if e is None or e['dtype'] != DT_STRING:
        return None
    n_sh = 1 + max(x['shard'] for x in entries.values())
    with open('%s.data-%05d-of-%05d' % (prefix, e['shard'], n_sh), 'rb') as fh:
        fh.seek(e['offset'])
        raw = fh.read(e['size'])
    if verify and e['crc'] is not None and unmask_crc(e['crc']) != _string_tensor_crc(raw):
        raise ValueError('%s: string tensor checksum mismatch' % name)
    length, pos = get_varint(raw, 0)
    return raw[pos + 4:pos + 4 + length]             # varint length | uint32 checksum of the lengths | bytes


def _length_word(n):
    """A string length as the running checksum sees it: the RESTORED integer, not its varint bytes -- a uint32 while it fits
    (TensorFlow's tensor_bundle.cc keeps the checksum of pre-uint64 checkpoints valid that way), a uint64 above that."""
    return struct.pack('<I', n) if n <= 0xFFFFFFFF else struct.pack('<Q', n)


def _string_tensor_crc(raw):
    """Running checksum of a scalar string tensor as TensorFlow computes it (see _string_tensor_bytes)."""
    length, pos = get_varint(raw, 0)
    crc = crc32c(_length_word(length))
    crc = crc32c(raw[pos:pos + 4], crc)
    return crc32c(raw[pos + 4:pos + 4 + length], crc)


def parse_object_graph(blob):
    """TrackableObjectGraph bytes -> [{'children': {local_name: node_id}, 'key': checkpoint_key or None}], node 0 = root."""
    nodes = []
    for num, _, body in pb_fields(blob):
        if num != 1:
            continue
        node = {'children': {}, 'key': None}
        for n2, _, v in pb_fields(body):
            if n2 == 1:                                # ObjectReference {1: node_id, 2: local_name}
                ref = dict((a, b) for a, _, b in pb_fields(v))
                node['children'][bytes(ref.get(2, b'')).decode()] = int(ref.get(1, 0))
            elif n2 == 2:                              # SerializedTensor {1: name, 2: full_name, 3: checkpoint_key}
                att = dict((a, b) for a, _, b in pb_fields(v))
                if bytes(att.get(1, b'')) == b'VARIABLE_VALUE':
                    node['key'] = bytes(att.get(3, b'')).decode()
        nodes.append(node)
    return nodes


def resolve_through_object_graph(prefix, paths):
    """{attribute path: checkpoint key} for the variables reachable by walking the checkpoint's OWN object graph from the root
    along the attribute names (`DownLayers/0/ConvLSTM/0/cell/kernel`): whatever keys the writing TensorFlow chose -- attribute
    paths, `layer_with_weights-N/...` aliases at any depth -- the graph names them.  Paths it cannot reach are left out."""
    try:      # best effort: the key-name scan of load_model_weights covers a graph this reader cannot take
        blob = read_scalar_string(prefix, OBJECT_GRAPH_KEY)
        nodes = parse_object_graph(blob) if blob else []
    except (ValueError, IndexError, struct.error, UnicodeDecodeError) as exc:
        import warnings
        warnings.warn('%s: object graph unreadable (%s); falling back to the variable key names' % (prefix, exc))
        return {}
    if not nodes:
        return {}
    out = {}
    for path in paths:
        nid = 0
        for part in path.split('/'):
            nid = nodes[nid]['children'].get(part) if nid is not None and nid < len(nodes) else None
            if nid is None:
                break
        if nid is not None and nid < len(nodes) and nodes[nid]['key']:
            out[path] = nodes[nid]['key']
    return out


def _string_tensor_bytes(items):
    """On-disk form of a DT_STRING tensor: varint64 lengths | uint32 masked crc of the lengths (each as uint32 while it fits,
    else uint64: _length_word) | bytes; returns (data, unmasked running crc over the length words, the length checksum, the
    string bytes)."""
    crc = 0
    data = bytearray()
    for it in items:
        crc = crc32c(_length_word(len(it)), crc)
        data += put_varint(len(it))
    lcs = struct.pack('<I', mask_crc(crc))
    crc = crc32c(lcs, crc)
    data += lcs
    for it in items:
        crc = crc32c(it, crc)
        data += it
    return bytes(data), crc


def write_bundle(prefix, tensors, strings=None):
    """tensors: {name: numpy array}; strings: {name: bytes} scalar string tensors.  One shard, little-endian."""
    items = sorted([(k.encode(), ('t', np.asarray(v).copy(order='C'))) for k, v in tensors.items()] +
                   [(k.encode() if isinstance(k, str) else k, ('s', v)) for k, v in (strings or {}).items()])
    entries = [(HEADER_KEY, pb_varint(1, 1) + pb_varint(2, 0) + pb_bytes(3, pb_varint(1, 1)))]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as fh:
        for key, (kind, val) in items:
            if kind == 't':
                if val.dtype.byteorder == '>':
                    val = val.astype(val.dtype.newbyteorder('<'))
                code = _NP2DT.get(val.dtype)
                if code is None:
                    raise ValueError('%s: dtype %s has no TensorFlow code here' % (key, val.dtype))
                raw = val.tobytes()
                crc = crc32c(raw)
                shape = b''.join(pb_bytes(2, pb_varint(1, int(s))) for s in val.shape)
            else:
                raw, crc = _string_tensor_bytes([val])
                code, shape = DT_STRING, b''
            fh.write(raw)
            ent = pb_varint(1, code) + pb_bytes(2, shape)
            if offset:
                ent += pb_varint(4, offset)
            ent += pb_varint(5, len(raw)) + pb_fixed32(6, mask_crc(crc))
            entries.append((key, ent))
            offset += len(raw)
    write_table(prefix + '.index', entries)


# ------------------------------------------------------------------------------------------- model <-> checkpoint names
_LOCAL = {'moving_var': 'moving_variance'}


def checkpoint_names(engine):
    """{engine tensor name: object-graph path} for every variable of a ULSTMnet2D engine (reference attribute names:
    Networks.py:44-58,130-139,195-205)."""
    out = {}
    for name in list(engine.P) + list(engine.S):
        side, bi, kind, idx, leaf = name.split('.')
        root = 'DownLayers' if side == 'down' else 'UpLayers'
        leaf = _LOCAL.get(leaf, leaf)
        if kind == 'lstm':
            out[name] = '%s/%s/ConvLSTM/%s/cell/%s' % (root, bi, idx, leaf)
        elif kind == 'conv':
            out[name] = '%s/%s/Conv/%s/%s' % (root, bi, idx, leaf)
        else:
            out[name] = '%s/%s/BN/%s/%s' % (root, bi, idx, leaf)
    return out


def _object_graph(paths):
    """Serialized TrackableObjectGraph (tensorflow/core/protobuf/trackable_object_graph.proto) for variables at the given
    attribute paths: nodes {1: children {1: node_id, 2: local_name}, 2: attributes {1: 'VARIABLE_VALUE', 2: full_name,
    3: checkpoint_key}}; node 0 is the model."""
    nodes = [{'children': [], 'attr': None}]
    index = {(): 0}
    for path in sorted(paths):
        parts = tuple(path.split('/'))
        for d in range(1, len(parts) + 1):
            if parts[:d] not in index:
                index[parts[:d]] = len(nodes)
                nodes.append({'children': [], 'attr': None})
                nodes[index[parts[:d - 1]]]['children'].append((index[parts[:d]], parts[d - 1]))
        nodes[index[parts]]['attr'] = path
    out = b''
    for n in nodes:
        body = b''.join(pb_bytes(1, pb_varint(1, cid) + pb_bytes(2, nm.encode())) for cid, nm in n['children'])
        if n['attr'] is not None:
            body += pb_bytes(2, pb_bytes(1, b'VARIABLE_VALUE') + pb_bytes(2, n['attr'].encode()) +
                             pb_bytes(3, (n['attr'] + SUFFIX).encode()))
        out += pb_bytes(1, body)
    return out


def save_model_weights(model, prefix):
    """ULSTMnet2D -> <prefix>.index + <prefix>.data-00000-of-00001 (`save_weights(prefix, save_format='tf')`)."""
    e = model.engine
    if e.plan is None:
        raise RuntimeError('the model has no variables yet (Keras-style lazy build): call it once first')
    names = checkpoint_names(e)
    params = e.export_params()
    write_bundle(prefix, {names[k] + SUFFIX: params[k] for k in names},
                 strings={OBJECT_GRAPH_KEY: _object_graph(names.values())})


_ALIAS = re.compile(r'^layer_with_weights-(\d+)/')


def load_model_weights(model, prefix):
    """-> {engine tensor name: array} for `engine.load_params`, from a tensor bundle written by tf.keras for the reference's
    ULSTMnet2D (or by save_model_weights).  Optimiser slots, the step counter and the object graph are ignored; every
    model variable must be present with the right shape."""
    e = model.engine
    want = checkpoint_names(e)
    n_down = len(e.plan['down'])
    have = {}
    listed = list_bundle(prefix)
    # first choice: the checkpoint's own object graph (handles key aliasing at any depth: the reference's blocks are k.Model
    # subclasses too, so a TensorFlow that writes `layer_with_weights-N` names does so inside the blocks as well)
    for path, key in resolve_through_object_graph(prefix, want.values()).items():
        if key in listed:
            have[path] = key
    for key in listed:
        if not key.endswith(SUFFIX):
            continue
        path = key[:-len(SUFFIX)]
        m = _ALIAS.match(path)
        if m:       # blocks enumerated in construction order: down blocks first, then up blocks (Networks.py:195-205)
            j = int(m.group(1))
            path = ('DownLayers/%d/' % j if j < n_down else 'UpLayers/%d/' % (j - n_down)) + path[m.end():]
        have.setdefault(path, key)
    missing = [p for p in want.values() if p not in have]
    if missing:
        raise KeyError('checkpoint %s lacks %d model variables, e.g. %s; it holds e.g. %s' %
                       (prefix, len(missing), missing[:3], sorted(have)[:3]))
    data = read_bundle(prefix, names=set(have[p] for p in want.values()))
    out = {}
    for name, path in want.items():
        arr = data[have[path]]
        dst = e.P[name] if name in e.P else e.S[name]
        if tuple(arr.shape) != tuple(dst.shape):
            raise ValueError('%s: checkpoint shape %s, model shape %s' % (path, arr.shape, tuple(dst.shape)))
        out[name] = arr.astype(np.float32)
    return out
