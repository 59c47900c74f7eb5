"""Minimal TensorBoard event-file writer (no TensorFlow): scalars and images, as train2D.py logs them
(reference train2D.py:119-137,163-176,205-211: 'Loss' / 'SEG' scalars and 'Image' / 'GT' / 'Output' images for the
train and val runs every `write_to_tb_interval` steps).

File format = TFRecord framing of serialized `Event` protos:
    uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)
with Event{1: wall_time double, 2: step int64, 3: file_version string, 5: Summary} and
Summary{1: repeated Value{1: tag, 2: simple_value float, 4: Image{1: height, 2: width, 3: colorspace, 4: png bytes}}}.
The protobuf wire encoding is done by hand (a dozen lines) so that neither TensorFlow nor tensorboard is needed."""
import io
import os
import socket
import struct
import time

import numpy as np

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    """CRC-32C (Castagnoli); crc32c(b'123456789') == 0xE3069283."""
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _bytes_field(num, data):
    return _field(num, 2, _varint(len(data)) + data)


def _event(step, summary=None, file_version=None, wall_time=None):
    ev = _field(1, 1, struct.pack('<d', time.time() if wall_time is None else wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if summary is not None:
        ev += _bytes_field(5, summary)
    return ev


def _png(img):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format='PNG')
    return buf.getvalue()


class SummaryWriter(object):
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        name = 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname())
        self.path = os.path.join(logdir, name)
        self._fh = open(self.path, 'ab')
        self._record(_event(0, file_version='brain.Event:2'))

    def _record(self, data):
        header = struct.pack('<Q', len(data))
        self._fh.write(header + struct.pack('<I', masked_crc(header)) + data + struct.pack('<I', masked_crc(data)))

    def scalar(self, tag, value, step):
        val = _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack('<f', float(value)))
        self._record(_event(int(step), _bytes_field(1, val)))

    def image(self, tag, img, step):
        """img: [H,W] or [H,W,1|3] array; floats are taken as [0,1] (train2D.py normalises the display image)."""
        a = np.asarray(img)
        if a.ndim == 3 and a.shape[2] == 1:
            a = a[..., 0]
        if a.dtype != np.uint8:
            a = np.round(np.clip(np.nan_to_num(a.astype(np.float64)), 0.0, 1.0) * 255.0).astype(np.uint8)
        channels = 1 if a.ndim == 2 else a.shape[2]
        im = (_field(1, 0, _varint(a.shape[0])) + _field(2, 0, _varint(a.shape[1])) + _field(3, 0, _varint(channels)) +
              _bytes_field(4, _png(a)))
        val = _bytes_field(1, tag.encode()) + _bytes_field(4, im)
        self._record(_event(int(step), _bytes_field(1, val)))

    def flush(self):
        self._fh.flush()

    def close(self):
        self._fh.close()


def read_events(path):
    """Test helper: parse an event file back into [(step, {tag: float or (h, w, c, png_bytes)})], verifying every CRC."""
    def parse(buf):
        i, out = 0, []
        while i < len(buf):
            key, sh = 0, 0
            while True:
                b = buf[i]
                i += 1
                key |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, sh = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    v |= (b & 0x7F) << sh
                    sh += 7
                    if not b & 0x80:
                        break
                out.append((num, v))
            elif wire == 1:
                out.append((num, struct.unpack('<d', buf[i:i + 8])[0]))
                i += 8
            elif wire == 5:
                out.append((num, struct.unpack('<f', buf[i:i + 4])[0]))
                i += 4
            else:
                n, sh = 0, 0
                while True:
                    b = buf[i]
                    i += 1
                    n |= (b & 0x7F) << sh
                    sh += 7
                    if not b & 0x80:
                        break
                out.append((num, bytes(buf[i:i + n])))
                i += n
        return out

    events = []
    with open(path, 'rb') as fh:
        blob = fh.read()
    pos = 0
    while pos < len(blob):
        header = blob[pos:pos + 8]
        (n,) = struct.unpack('<Q', header)
        assert struct.unpack('<I', blob[pos + 8:pos + 12])[0] == masked_crc(header), 'length CRC'
        data = blob[pos + 12:pos + 12 + n]
        assert struct.unpack('<I', blob[pos + 12 + n:pos + 16 + n])[0] == masked_crc(data), 'data CRC'
        pos += 16 + n
        fields = dict(parse(data))
        vals = {}
        if 5 in fields:
            for num, payload in parse(fields[5]):
                v = dict(parse(payload))
                if 2 in v:
                    vals[v[1].decode()] = v[2]
                elif 4 in v:
                    im = dict(parse(v[4]))
                    vals[v[1].decode()] = (im[1], im[2], im[3], im[4])
        events.append((fields.get(2, 0), vals, fields.get(3)))
    return events
