"""Clip-stream providers honouring the reference's batch contract (DataHandeling.py:454-493):

    get_batch() -> (image, seg, full_seg, keep)
      image, seg : float32 [B,T,1,H,W] ('NCHW') or [B,T,H,W,1];  image per-frame z-scored (:103)
      seg        : values {-1 unlabeled, 0 background, 1 cell, 2 edge}
      full_seg   : [B,T]
      keep       : [B] 1.0 while slot b's clip continues, 0.0 when this window ended it (:378,471)
    slot b always continues the same clip in the next batch (per-slot FIFO, :447-452)

Only the synthetic provider is implemented (benchmarks / tests).  The RAM readers for
Cell-Tracking-Challenge folders (cv2 + tf.queue in the reference) are SURVEY §8f-2 "next" rows.
"""
import numpy as np
from scipy import ndimage


def instances_to_classes(inst):
    """Instance map -> {0 bg, 1 cell, 2 edge}: class = min(id, 1); pixels whose id differs from the
    3x3 max-filter of the map while that maximum is > 0 become edge (DataHandeling.py:199-211)."""
    seg = np.round(np.asarray(inst, np.float32))
    dil = ndimage.maximum_filter(seg.astype(np.int32), size=3, mode='reflect')
    out = np.minimum(seg, 1)
    out[(seg != dil) & (dil > 0)] = 2
    return out


class SyntheticSequence2D(object):
    """Endless synthetic microscopy clips: N(0,1) frames re-z-scored per frame; ~12 drifting ellipses
    per clip turned into {0,1,2} labels by the edge rule; 10 % of frames fully unlabeled (-1)."""

    def __init__(self, sequence_folder_list=None, image_crop_size=(128, 128), unroll_len=4, deal_with_end=0,
                 batch_size=5, queue_capacity=200, data_format='NCHW', randomize=True, return_dist=False,
                 num_threads=1, seed=1234, rank=0, clip_len=32, n_cells=12):
        self.crop = tuple(image_crop_size)
        self.unroll_len = unroll_len
        self.batch_size = batch_size
        self.data_format = data_format
        self.clip_len = max(unroll_len, clip_len - clip_len % unroll_len)   # clips trimmed to a multiple of T
        self.n_cells = n_cells
        self.rng = np.random.default_rng(seed + rank)
        self.slots = [self._new_clip() for _ in range(batch_size)]
        self.q_stat_list = []

    def start_queues(self, coord=None, debug=False):
        return []

    def _new_clip(self):
        h, w = self.crop
        n = self.n_cells
        return {'t': 0, 'cy': self.rng.uniform(0, h, n), 'cx': self.rng.uniform(0, w, n),
                'ry': self.rng.uniform(6, 20, n) * min(1.0, h / 256 + 0.25), 'rx': self.rng.uniform(6, 20, n) * min(1.0, w / 256 + 0.25),
                'vy': self.rng.uniform(-2, 2, n), 'vx': self.rng.uniform(-2, 2, n)}

    def _frame(self, clip):
        h, w = self.crop
        img = self.rng.standard_normal((h, w)).astype(np.float32)
        yy, xx = np.mgrid[:h, :w]
        inst = np.zeros((h, w), np.float32)
        for i in range(self.n_cells):
            m = ((yy - clip['cy'][i]) / clip['ry'][i]) ** 2 + ((xx - clip['cx'][i]) / clip['rx'][i]) ** 2 <= 1
            inst[m] = i + 1
            img[m] += 1.5
        clip['cy'] = (clip['cy'] + clip['vy']) % h
        clip['cx'] = (clip['cx'] + clip['vx']) % w
        img = (img - img.mean()) / img.std()
        full = self.rng.random() >= 0.1
        seg = instances_to_classes(inst) if full else np.full((h, w), -1, np.float32)
        return img, seg.astype(np.float32), float(full)

    def get_batch(self):
        B, T = self.batch_size, self.unroll_len
        h, w = self.crop
        image = np.empty((B, T, h, w), np.float32)
        seg = np.empty((B, T, h, w), np.float32)
        full = np.empty((B, T), np.float32)
        keep = np.ones(B, np.float32)
        for b in range(B):
            clip = self.slots[b]
            for t in range(T):
                image[b, t], seg[b, t], full[b, t] = self._frame(clip)
            clip['t'] += T
            if clip['t'] >= self.clip_len:
                keep[b] = 0.0
                self.slots[b] = self._new_clip()
        axis = 2 if self.data_format[1] == 'C' else 4
        return np.expand_dims(image, axis), np.expand_dims(seg, axis), full, keep


class CTCRAMReaderSequence2D(object):
    """Placeholder with the reference's class name: the Cell-Tracking-Challenge RAM reader needs OpenCV and
    dataset folders (DataHandeling.py:21-529) and is a SURVEY §8f-2 'next' row, not part of the hot path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('CTCRAMReaderSequence2D (real-data reader) is not part of the MI355X hot-path '
                                  'build yet; use SyntheticSequence2D or feed get_batch()-shaped arrays')


class CTCInferenceReader(object):
    """Sorted frames of a sequence folder with a mirrored warm-up prefix (DataHandeling.py:1572-1604)."""

    def __init__(self, data_path, filename_format='t*.tif', normalize=True, pre_sequence_frames=0):
        import glob
        import os
        files = sorted(glob.glob(os.path.join(data_path, filename_format)))
        self.file_list = files[:pre_sequence_frames][::-1] + files
        self.normalize = normalize
        self.dataset = self._gen()

    def _gen(self):
        from PIL import Image
        for f in self.file_list:
            img = np.asarray(Image.open(f)).astype(np.float32)
            if self.normalize:
                img = (img - img.mean()) / img.std()
            yield img
