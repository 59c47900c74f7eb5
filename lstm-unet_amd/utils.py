"""Small helpers with the names the reference's utils.py exposes (utils.py:33-69)."""
from datetime import datetime

import numpy as np


def log_print(*args):
    print('{}:'.format(datetime.now().strftime('%Y-%m-%d %H:%M:%S')), *args, flush=True)


def get_model(model_name: str):
    """Resolve a model class by name on Networks (utils.py:38-40) -- part of the saved-model contract."""
    import Networks as Nets
    return getattr(Nets, model_name)


def load_model(model_name: str, *args, **kwargs):
    model = get_model(model_name)
    return model(*args, **kwargs) if (args or kwargs) else model


def bbox_crop(img, margin=10):
    """Tight bounding box of the non-zero region, grown by `margin` (clipped)."""
    ys = np.flatnonzero(np.any(img, axis=1))
    xs = np.flatnonzero(np.any(img, axis=0))
    rmin, rmax = max(0, ys[0] - margin), min(img.shape[0], ys[-1] + margin)
    cmin, cmax = max(0, xs[0] - margin), min(img.shape[1], xs[-1] + margin)
    return img[rmin:rmax, cmin:cmax], (rmin, rmax, cmin, cmax)


def bbox_fill(img, crop, loc):
    rmin, rmax, cmin, cmax = loc
    out = img.copy()
    out[rmin:rmax, cmin:cmax] = crop
    return out


def select_gpu(gpu_id):
    """--gpu_id of both drivers: the reference exports CUDA_VISIBLE_DEVICES = gpu_id and computes on '/gpu:0'
    (train2D.py:359-361, Inference2D.py:213-214).  Here: the listed devices become the visible set (HIP_VISIBLE_DEVICES, when
    the HIP runtime has not been initialised yet), otherwise the first listed id is made current.  Data-parallel launches
    (one process per GPU, LOCAL_RANK set) pick their device from the rank instead.  There is no CPU execution path: -1 is
    an error, not a fallback."""
    import os
    import torch
    if gpu_id is None or 'LOCAL_RANK' in os.environ:
        return
    ids = [t for t in str(gpu_id).replace(' ', '').split(',') if t != '']
    if not ids:
        return
    if any(int(t) < 0 for t in ids):
        raise ValueError('gpu_id=%s: this build has no CPU execution path (MI355X only)' % (gpu_id,))
    if not torch.cuda.is_initialized():
        os.environ['HIP_VISIBLE_DEVICES'] = ','.join(ids)
        return
    if int(ids[0]) >= torch.cuda.device_count():
        raise ValueError('gpu_id=%s but only %d device(s) are visible' % (gpu_id, torch.cuda.device_count()))
    torch.cuda.set_device(int(ids[0]))


def write_tiff16(path, arr):
    """Baseline little-endian TIFF of a uint16 image, [H,W] (grayscale) or [H,W,3] (RGB): what the reference writes with
    cv2.imwrite for the label maps and the softmax visualisation (Inference2D.py:104-108,124-130; OpenCV stores the same
    pixels LZW-compressed).  One strip, no compression, so any TIFF reader opens it."""
    import struct
    a = np.ascontiguousarray(arr, dtype='<u2')
    if a.ndim == 2:
        a = a[:, :, None]
    H, W, S = a.shape
    if S not in (1, 3):
        raise ValueError('write_tiff16: 1 or 3 samples per pixel')
    data = a.tobytes()
    entries = [(256, 4, 1, W), (257, 4, 1, H), (258, 3, S, None), (259, 3, 1, 1), (262, 3, 1, 1 if S == 1 else 2),
               (273, 4, 1, None), (277, 3, 1, S), (278, 4, 1, H), (279, 4, 1, len(data)), (284, 3, 1, 1)]
    ifd_off = 8
    ifd_len = 2 + 12 * len(entries) + 4
    bits_off = ifd_off + ifd_len
    data_off = bits_off + (6 if S == 3 else 0)
    data_off += data_off % 2
    out = struct.pack('<2sHI', b'II', 42, ifd_off) + struct.pack('<H', len(entries))
    for tag, typ, cnt, val in entries:
        if tag == 258:
            val_bytes = struct.pack('<HH', 16, 0) if S == 1 else struct.pack('<I', bits_off)
        elif tag == 273:
            val_bytes = struct.pack('<I', data_off)
        elif typ == 3:
            val_bytes = struct.pack('<HH', val, 0)
        else:
            val_bytes = struct.pack('<I', val)
        out += struct.pack('<HHI', tag, typ, cnt) + val_bytes
    out += struct.pack('<I', 0)
    if S == 3:
        out += struct.pack('<HHH', 16, 16, 16)
    out += b'\x00' * (data_off - len(out))
    with open(path, 'wb') as fh:
        fh.write(out + data)


def read_tiff16(path):
    """Reader for write_tiff16's files (tests / round trips)."""
    import struct
    buf = open(path, 'rb').read()
    assert buf[:4] == b'II*\x00'
    off = struct.unpack_from('<I', buf, 4)[0]
    n = struct.unpack_from('<H', buf, off)[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack_from('<HHI', buf, off + 2 + 12 * i)
        val = struct.unpack_from('<H' if typ == 3 else '<I', buf, off + 2 + 12 * i + 8)[0]
        tags[tag] = val
    W, H, S = tags[256], tags[257], tags[277]
    a = np.frombuffer(buf, dtype='<u2', count=H * W * S, offset=tags[273]).reshape(H, W, S)
    return a[:, :, 0].copy() if S == 1 else a.copy()
