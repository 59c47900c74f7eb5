"""tf_bundle.py: the TensorFlow tensor-bundle reader / writer behind `save_weights(..., save_format='tf')` /
`load_weights` (reference train2D.py:235, Inference2D.py:34), pinned by known answers built BY HAND from the published
format (no TensorFlow exists here): CRC-32C vectors, a bundle assembled byte by byte in this file with encoder choices
the module's own writer never makes, a snappy block, round trips over many blocks, and the model <-> checkpoint name map."""
import struct

import numpy as np
import pytest

import tf_bundle as tb
from conftest import tiny_net
from engine_backend import engine_backend


def test_crc32c_known_answers():
    assert tb.crc32c(b'123456789') == 0xE3069283                  # the standard check value
    assert tb.crc32c(b'\x00' * 32) == 0x8A9136AA                   # RFC 3720 B.4 test vectors
    assert tb.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E
    assert tb.crc32c(b'6789', tb.crc32c(b'12345')) == 0xE3069283    # incremental
    assert tb.unmask_crc(tb.mask_crc(0xE3069283)) == 0xE3069283
    assert tb.mask_crc(0) == 0xA282EAD8


def _hand_block(entries):
    """One table block with NO prefix compression and a restart point at every entry (the module's writer shares prefixes
    and restarts every 16 entries)."""
    out, restarts = b'', []
    for k, v in entries:
        restarts.append(len(out))
        out += bytes([0, len(k), len(v)]) + k + v           # all lengths < 128: one-byte varints
    for r in restarts:
        out += struct.pack('<I', r)
    return out + struct.pack('<I', len(restarts))


def _on_disk(block, ctype=0):
    return block + bytes([ctype]) + struct.pack('<I', tb.mask_crc(tb.crc32c(block + bytes([ctype]))))


def test_hand_assembled_bundle(tmp_path):
    a = np.arange(6, dtype='<f4').reshape(2, 3) * 0.5 - 1
    b = np.array([7, -9], dtype='<i8')
    data = a.tobytes() + b.tobytes()
    header = bytes([0x08, 0x01, 0x10, 0x00, 0x1a, 0x02, 0x08, 0x01])     # num_shards 1, little endian, version {producer 1}
    ent_a = (bytes([0x08, 0x01]) + bytes([0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03]) +
             bytes([0x28, 24]) + bytes([0x35]) + struct.pack('<I', tb.mask_crc(tb.crc32c(a.tobytes()))))
    ent_b = (bytes([0x08, 0x09]) + bytes([0x12, 0x04, 0x12, 0x02, 0x08, 0x02]) + bytes([0x20, 24, 0x28, 16]) +
             bytes([0x35]) + struct.pack('<I', tb.mask_crc(tb.crc32c(b.tobytes()))))
    blk0 = _on_disk(_hand_block([(b'', header), (b'a/x', ent_a)]))       # two data blocks
    blk1 = _on_disk(_hand_block([(b'b', ent_b)]))
    meta = _on_disk(_hand_block([]))
    h0 = bytes([0, len(blk0) - 5])
    h1 = bytes([len(blk0), len(blk1) - 5])
    hm = bytes([len(blk0) + len(blk1), len(meta) - 5])
    index = _on_disk(_hand_block([(b'a/y', h0), (b'b', h1)]))            # separator keys >= last key of the block
    hi = bytes([len(blk0) + len(blk1) + len(meta), len(index) - 5])
    foot = hm + hi
    foot += b'\x00' * (40 - len(foot)) + struct.pack('<Q', 0xdb4775248b80fb57)
    prefix = str(tmp_path / 'hand.ckpt')
    open(prefix + '.index', 'wb').write(blk0 + blk1 + meta + index + foot)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    got = tb.read_bundle(prefix)
    assert sorted(got) == ['a/x', 'b'] and np.array_equal(got['a/x'], a) and np.array_equal(got['b'], b)
    assert got['a/x'].dtype == np.float32 and got['b'].dtype == np.int64
    ls = tb.list_bundle(prefix)
    assert ls['a/x']['shape'] == (2, 3) and ls['b']['offset'] == 24
    bad = bytearray(data)
    bad[3] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        tb.read_bundle(prefix)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[5] ^= 1
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum'):
        tb.list_bundle(prefix)


def test_snappy_block_and_bad_magic(tmp_path):
    # literal "abcd" + copy (offset 4, length 8) -> "abcdabcdabcd"; 2-byte-offset copy form, then a 1-byte-offset copy
    stream = bytes([16]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((8 - 1) << 2) | 2, 4, 0]) + bytes([((4 - 4) << 2) | 1, 12])
    assert tb.snappy_decompress(stream) == b'abcdabcdabcdabcd'
    block = _hand_block([(b'k', b'v' * 40)])
    comp = bytes([len(block)]) + b''.join(bytes([(min(60, len(block) - i) - 1) << 2]) + block[i:i + 60]
                                          for i in range(0, len(block), 60))
    blk = _on_disk(comp, 1)
    meta = _on_disk(_hand_block([]))
    index = _on_disk(_hand_block([(b'k', bytes([0, len(blk) - 5]))]))
    foot = bytes([len(blk), len(meta) - 5]) + bytes([len(blk) + len(meta), len(index) - 5])
    foot += b'\x00' * (40 - len(foot)) + struct.pack('<Q', 0xdb4775248b80fb57)
    p = tmp_path / 's.index'
    p.write_bytes(blk + meta + index + foot)
    assert tb.read_table(str(p)) == [(b'k', b'v' * 40)]
    p.write_bytes(b'\x00' * 64)
    with pytest.raises(ValueError, match='magic'):
        tb.read_table(str(p))


def test_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'layer/%03d/kernel%s' % (i, tb.SUFFIX): rng.standard_normal((3, i % 5 + 1, 2)).astype(np.float32)
               for i in range(300)}
    tensors['step'] = np.array(12345, dtype=np.int64)
    prefix = str(tmp_path / 'rt.ckpt')
    tb.write_bundle(prefix, tensors, strings={tb.OBJECT_GRAPH_KEY: b'\x0a\x00'})
    orig = tb.write_table

    def small_blocks(path, entries, block_size=262144):
        return orig(path, entries, block_size=700)          # force dozens of data blocks + a multi-entry index block
    tb.write_table = small_blocks
    try:
        tb.write_bundle(prefix + '2', tensors)
    finally:
        tb.write_table = orig
    for pre in (prefix, prefix + '2'):
        got = tb.read_bundle(pre)
        assert set(got) == set(tensors)
        for k in tensors:
            assert np.array_equal(got[k], tensors[k]) and got[k].shape == tensors[k].shape
    keys = [k for k, _ in tb.read_table(prefix + '.index')]
    assert keys == sorted(keys) and keys[0] == b'' and tb.OBJECT_GRAPH_KEY in keys
    assert tb.list_bundle(prefix)[tb.OBJECT_GRAPH_KEY.decode()]['dtype'] == tb.DT_STRING


def test_model_checkpoint_names_and_round_trip(tmp_path):
    """save_weights(save_format='tf') -> load_weights restores every variable bit for bit through the reference's
    attribute-path names; a checkpoint that names the blocks `layer_with_weights-<n>` loads too; a missing or
    mis-shaped variable is an error."""
    import Networks
    with engine_backend('emu') as dev:
        net = tiny_net(3)
        m = Networks.ULSTMnet2D(net, 'NHWC', True, seed=3)
        m.engine.build(1, dev)
        names = tb.checkpoint_names(m.engine)
        assert names['down.0.lstm.0.recurrent_kernel'] == 'DownLayers/0/ConvLSTM/0/cell/recurrent_kernel'
        assert names['up.3.conv.2.bias'] == 'UpLayers/3/Conv/2/bias'
        assert names['down.2.bn.1.moving_var'] == 'DownLayers/2/BN/1/moving_variance'
        prefix = str(tmp_path / 'model.ckpt')
        m.save_weights(prefix, save_format='tf')
        keys = tb.list_bundle(prefix)
        assert 'DownLayers/0/ConvLSTM/0/cell/kernel' + tb.SUFFIX in keys and len(keys) == len(names) + 1
        # the object graph: node 0 (the model) has the two block lists as children
        og = tb.read_table(prefix + '.index')
        raw = dict(og)[tb.OBJECT_GRAPH_KEY]
        assert raw
        m2 = Networks.ULSTMnet2D(net, 'NHWC', True, seed=99)
        m2.load_weights(prefix)
        a, b = m.engine.export_params(), m2.engine.export_params()
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
        # alias spelling + extra optimiser / step entries
        blob = tb.read_bundle(prefix)
        alias = {}
        for k, v in blob.items():
            side, bi, rest = k.split('/', 2)
            j = int(bi) + (0 if side == 'DownLayers' else 4)
            alias['layer_with_weights-%d/%s' % (j, rest)] = v
        alias['optimizer/iter' + tb.SUFFIX] = np.array(7, dtype=np.int64)
        tb.write_bundle(prefix + '.alias', alias)
        m3 = Networks.ULSTMnet2D(net, 'NHWC', True, seed=5)
        m3.load_weights(prefix + '.alias')
        c = m3.engine.export_params()
        assert all(np.array_equal(a[k], c[k]) for k in a)
        del alias['layer_with_weights-0/ConvLSTM/0/cell/bias' + tb.SUFFIX]
        tb.write_bundle(prefix + '.missing', alias)
        with pytest.raises(KeyError):
            Networks.ULSTMnet2D(net, 'NHWC', True).load_weights(prefix + '.missing')
        # NESTED aliasing (the reference's blocks are k.Model subclasses too): keys like
        # layer_with_weights-0/layer_with_weights-1/kernel resolve through the checkpoint's own object graph, at any depth
        nested, graph_paths = {}, {}
        for k, v in blob.items():
            path = k[:-len(tb.SUFFIX)]
            side, bi, kind, idx, rest = path.split('/', 4)
            j = int(bi) + (0 if side == 'DownLayers' else 4)
            inner = {'ConvLSTM': 0, 'Conv': 1, 'BN': 2}[kind] * 10 + int(idx)
            key = 'layer_with_weights-%d/layer_with_weights-%d/%s' % (j, inner, rest) + tb.SUFFIX
            nested[key] = v
            graph_paths[path] = key
        # the graph has the attribute paths, its leaves name the aliased keys
        tb.write_bundle(prefix + '.nested', nested, strings={tb.OBJECT_GRAPH_KEY: _regraph(tb, graph_paths)})
        res = tb.resolve_through_object_graph(prefix + '.nested', list(graph_paths))
        assert res == graph_paths
        m4 = Networks.ULSTMnet2D(net, 'NHWC', True, seed=6)
        m4.load_weights(prefix + '.nested')
        d4 = m4.engine.export_params()
        assert all(np.array_equal(a[k], d4[k]) for k in a)
        # and the ADVICE item: weights can be reloaded after the public autograd path has been used
        m.parameters()
        m.load_weights(prefix)


def _regraph(tb, path_to_key):
    """TrackableObjectGraph whose variable leaves sit at the attribute paths but carry arbitrary checkpoint keys."""
    nodes = [{'children': [], 'key': None}]
    index = {(): 0}
    for path in sorted(path_to_key):
        parts = tuple(path.split('/'))
        for d in range(1, len(parts) + 1):
            if parts[:d] not in index:
                index[parts[:d]] = len(nodes)
                nodes.append({'children': [], 'key': None})
                nodes[index[parts[:d - 1]]]['children'].append((index[parts[:d]], parts[d - 1]))
        nodes[index[parts]]['key'] = path_to_key[path]
    out = b''
    for n in nodes:
        body = b''.join(tb.pb_bytes(1, tb.pb_varint(1, cid) + tb.pb_bytes(2, nm.encode())) for cid, nm in n['children'])
        if n['key'] is not None:
            body += tb.pb_bytes(2, tb.pb_bytes(1, b'VARIABLE_VALUE') + tb.pb_bytes(2, b'v') + tb.pb_bytes(3, n['key'].encode()))
        out += tb.pb_bytes(1, body)
    return out


def test_object_graph_proto():
    og = tb._object_graph(['DownLayers/0/Conv/0/kernel', 'DownLayers/0/Conv/0/bias', 'UpLayers/1/BN/0/gamma'])
    nodes = [v for n, _, v in tb.pb_fields(og) if n == 1]
    root_children = [dict((n, v) for n, _, v in tb.pb_fields(c))[2] for n2, _, c in tb.pb_fields(nodes[0]) if n2 == 1]
    assert root_children == [b'DownLayers', b'UpLayers']
    leaves = [tb.pb_fields(v) for node in nodes for n, _, v in tb.pb_fields(node) if n == 2]
    keys = sorted(dict((n, v) for n, _, v in leaf)[3] for leaf in leaves)
    assert keys == [(p + tb.SUFFIX).encode() for p in ['DownLayers/0/Conv/0/bias', 'DownLayers/0/Conv/0/kernel',
                                                       'UpLayers/1/BN/0/gamma']]


def test_string_tensor_checksum_uses_uint32_lengths(tmp_path):
    """tensor_bundle.cc WriteStringTensor / ReadStringTensor: the running checksum extends with each string LENGTH as the
    restored integer -- a uint32 while it fits (checkpoints from before the uint64 lengths stay valid), a uint64 only above
    UINT32_MAX -- then with the masked length checksum, then with the bytes.  Assembled here by hand, independently of the
    writer; a bundle whose string entry carries the uint64-width checksum (the round-3 writer) must be REJECTED by the verifying
    reader and must not break load_model_weights (object graph = best effort, key names = fallback).  Unpinned against a real
    TensorFlow string tensor (none available)."""
    payload = b'\x0a\x02\x12\x00' * 5
    crc = tb.crc32c(struct.pack('<I', len(payload)))
    lcs = struct.pack('<I', tb.mask_crc(crc))
    crc = tb.crc32c(payload, tb.crc32c(lcs, crc))
    raw = tb.put_varint(len(payload)) + lcs + payload
    data, got_crc = tb._string_tensor_bytes([payload])
    assert data == raw and got_crc == crc and tb._string_tensor_crc(raw) == crc
    wide = tb.crc32c(struct.pack('<Q', len(payload)))
    assert wide != tb.crc32c(struct.pack('<I', len(payload)))
    assert tb._length_word(2 ** 32) == struct.pack('<Q', 2 ** 32) and tb._length_word(2 ** 32 - 1) == b'\xff\xff\xff\xff'
    prefix = str(tmp_path / 's.ckpt')
    tb.write_bundle(prefix, {'v' + tb.SUFFIX: np.arange(3, dtype=np.float32)}, strings={tb.OBJECT_GRAPH_KEY: payload})
    assert tb.read_scalar_string(prefix, tb.OBJECT_GRAPH_KEY) == payload
    # the same bundle with a string entry checksummed the old (uint64-length) way
    orig = tb._length_word
    tb._length_word = lambda n: struct.pack('<Q', n)
    try:
        tb.write_bundle(prefix + '.old', {'v' + tb.SUFFIX: np.arange(3, dtype=np.float32)}, strings={tb.OBJECT_GRAPH_KEY: payload})
    finally:
        tb._length_word = orig
    with pytest.raises(ValueError, match='checksum'):
        tb.read_scalar_string(prefix + '.old', tb.OBJECT_GRAPH_KEY)
    with pytest.warns(UserWarning, match='object graph unreadable'):
        assert tb.resolve_through_object_graph(prefix + '.old', ['v']) == {}


def test_ckpt_probe_tool_reports_found_missing_and_misshaped(tmp_path, capsys):
    """tools/ckpt_probe.py: the key list of any model.ckpt diffed against the names and shapes
    tf_bundle expects -- device-free (the static plan), exit code 0 only when every model variable is there."""
    import os
    import pickle
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    try:
        import ckpt_probe
    finally:
        sys.path.pop(0)
    import tf_bundle as tb
    from conftest import tiny_net
    net = tiny_net(3)
    want, _ = ckpt_probe.expected_variables(net, 1)
    rng = np.random.default_rng(0)
    tensors = {p + tb.SUFFIX: rng.standard_normal(s).astype(np.float32) for p, s in want.items()}
    tensors['optimizer/iter' + tb.SUFFIX] = np.array(3, dtype=np.int64)
    good = str(tmp_path / 'model.ckpt')
    tb.write_bundle(good, tensors, strings={tb.OBJECT_GRAPH_KEY: tb._object_graph(want.keys())})
    pk = str(tmp_path / 'model_params.pickle')
    with open(pk, 'wb') as fh:
        pickle.dump({'name': 'ULSTMnet2D', 'params': (net,)}, fh)
    assert ckpt_probe.main([good + '.index', '--params', pk, '--json', str(tmp_path / 'r.json')]) == 0
    out = capsys.readouterr().out
    assert 'found %d' % len(want) in out and 'missing 0' in out
    bad = dict(tensors)
    keys = sorted(want)
    del bad[keys[0] + tb.SUFFIX]
    bad[keys[5] + tb.SUFFIX] = np.zeros((2, 2), np.float32)
    tb.write_bundle(str(tmp_path / 'bad.ckpt'), bad)
    assert ckpt_probe.main([str(tmp_path / 'bad.ckpt'), '--params', pk]) == 1
    out = capsys.readouterr().out
    assert 'missing: ' + keys[0] in out and 'shape:   ' + keys[5] in out
