"""Clip-stream providers honouring the reference's batch contract (DataHandeling.py:454-493):

    get_batch() -> (image, seg, full_seg, keep)      [+ dist [B,T,2,H,W] with return_dist=True, :371-377,490-491]
      image, seg : float32 [B,T,1,H,W] ('NCHW') or [B,T,H,W,1];  image per-frame z-scored (:103)
      seg        : values {-1 unlabeled, 0 background, 1 cell, 2 edge}
      full_seg   : [B,T]
      keep       : [B] 1.0 while slot b's clip continues, 0.0 when this window ended it (:378,471)
    slot b always continues the same clip in the next batch (per-slot FIFO, :447-452)

SyntheticSequence2D is the benchmark / test provider; CTCRAMReaderSequence2D reads Cell-Tracking-Challenge
folders with Pillow + scipy instead of OpenCV and per-slot generators instead of tf.queue.
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


def _read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def affine_from_points(src, dst):
    """2x3 matrix M with M @ [x, y, 1] = dst for the three point pairs (what cv2.getAffineTransform solves)."""
    a = np.hstack([np.asarray(src, np.float64), np.ones((3, 1))])
    return np.linalg.solve(a, np.asarray(dst, np.float64)).T


def warp_affine(image, matrix, order, mode, cval=0.0):
    """out(x, y) = image(M^-1 [x, y, 1]): forward-matrix warp onto the same canvas (cv2.warpAffine semantics; mode
    'mirror' = BORDER_REFLECT_101, 'constant' + cval = BORDER_CONSTANT; order 1 bilinear / 0 nearest)."""
    full = np.vstack([matrix, [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(full)
    h, w = image.shape
    ys, xs = np.mgrid[:h, :w].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    return ndimage.map_coordinates(image, [sy, sx], order=order, mode=mode, cval=cval)


def adjust_contrast(image, factor):
    """(image - mean) * factor + mean, mean over the whole crop (DataHandeling.py:250-260)."""
    m = image.mean()
    return (image - m) * factor + m


def adjust_brightness(image, delta):
    """image + delta (DataHandeling.py:239-248)."""
    return image + delta


def elastic_affine_points(shape_size, alpha_affine, random_state):
    """The three control-point pairs of the clip's random affine (DataHandeling.py:152-173), float32 like the reference's.
    Kept quirk: `center_square = float32(shape_size) // 2` is (h // 2, w // 2) but is handed to OpenCV as an (x, y) point,
    so on a non-square crop the control triangle is centred at (x = h // 2, y = w // 2)."""
    centre = np.float32(shape_size) // 2
    sq = min(shape_size) // 3
    pts1 = np.float32([centre + sq, [centre[0] + sq, centre[1] - sq], centre - sq])
    pts2 = pts1 + random_state.uniform(-alpha_affine, alpha_affine, size=pts1.shape).astype(np.float32)
    return pts1, pts2


def elastic_indices(shape, alpha, sigma, random_state):
    """Sampling coordinates (rows, cols) of the smooth random displacement field, in the reference's draw order
    (DataHandeling.py:188-197): the x displacement is drawn first, then y; both are uniform(-1, 1) fields smoothed with a
    Gaussian of `sigma` (scipy default boundary 'reflect') and scaled by `alpha`.  Column vectors like the reference's."""
    dx = ndimage.gaussian_filter(random_state.rand(*shape) * 2 - 1, sigma) * alpha
    dy = ndimage.gaussian_filter(random_state.rand(*shape) * 2 - 1, sigma) * alpha
    x, y = np.meshgrid(np.arange(shape[1]), np.arange(shape[0]))
    return np.reshape(y + dy, (-1, 1)), np.reshape(x + dx, (-1, 1))


def transformed_image(image, affine_matrix, indices, seg=False):
    """Affine warp, then the elastic resampling (DataHandeling.py:175-186): labels nearest-neighbour with -1 outside,
    images bilinear with mirrored (warp: BORDER_REFLECT_101) / reflected (resampling) borders."""
    shape = image.shape
    if seg:
        moved = warp_affine(image, affine_matrix, 0, 'constant', -1.0)
        return ndimage.map_coordinates(moved, indices, order=0, mode='constant', cval=-1).reshape(shape)
    moved = warp_affine(image, affine_matrix, 1, 'mirror')
    return ndimage.map_coordinates(moved, indices, order=1, mode='reflect').reshape(shape)


class ClipAugmenter(object):
    """One clip's geometric / photometric augmentation, drawn once per clip and applied to every frame
    (DataHandeling.py:262-300): temporal sub-sampling / reverse, crop offset, flips, 90-degree rotations, and -- when
    `elastic` -- a random affine (three control points jittered by 8 % of the crop width) followed by a smooth random
    displacement field (sigma = 15 % of the crop width, amplitude = 2 crop widths before smoothing), both drawn from ONE
    `numpy.random.RandomState` in the reference's order (affine jitter, x field, y field) so that the helpers are pinned
    by the reference's own (`tests/golden/elastic.npz`).  Labels go through the same maps with nearest-neighbour sampling
    and -1 outside, and are then turned into {0,1,2} by the edge rule."""

    def __init__(self, rng, frame_shape, crop, randomize, elastic):
        H, W = frame_shape
        h, w = crop
        self.crop = crop
        self.step = int(rng.integers(1, 4)) if randomize else 1
        self.reverse = bool(rng.integers(0, 2)) if randomize else False
        self.y0 = int(rng.integers(0, H - h)) if (randomize and H - h > 0) else 0
        self.x0 = int(rng.integers(0, W - w)) if (randomize and W - w > 0) else 0
        self.flip = rng.integers(0, 2, 2) if randomize else np.zeros(2, np.int64)
        self.rot = int(rng.integers(0, 4)) if (randomize and h == w) else 0
        self.randomize = randomize
        self.rng = rng
        self.matrix = self.field = None
        if elastic:
            # the reference seeds RandomState(None) per clip; here the seed comes from the slot's own generator so that the
            # stream stays deterministic per (seed, rank, slot)
            state = np.random.RandomState(int(rng.integers(0, 2 ** 31 - 1)))
            self.matrix = affine_from_points(*elastic_affine_points((h, w), w * 0.08, state))
            self.field = elastic_indices((h, w), w * 2, w * 0.15, state)

    def _geom(self, a, label):
        if self.matrix is not None:
            a = transformed_image(a, self.matrix, self.field, seg=label)
        return a

    def _orient(self, a):
        if self.flip[0]:
            a = a[::-1]
        if self.flip[1]:
            a = a[:, ::-1]
        if self.rot:
            a = np.rot90(a, self.rot)
        return np.ascontiguousarray(a)

    def frame(self, img, seg, img_max):
        """img: z-scored frame, seg: instance ids (0 background, -1 unlabeled) -> (image, {-1,0,1,2} labels)."""
        h, w = self.crop
        img = img[self.y0:self.y0 + h, self.x0:self.x0 + w].astype(np.float64)
        seg = seg[self.y0:self.y0 + h, self.x0:self.x0 + w].astype(np.float32).copy()
        if self.randomize:      # contrast factor in [0.5, 1.5], brightness +-10 % of the sequence maximum (:342-348)
            factor, delta = self.rng.random() + 0.5, (self.rng.random() - 0.5) * 0.2 * img_max
            img = adjust_brightness(adjust_contrast(img, factor), delta)
        img = self._geom(img, False)
        if self.matrix is not None:
            if not np.all(seg == -1):
                unlabeled = (seg == -1).astype(np.float32)
                seg[:, 0] = seg[:, -1] = seg[0, :] = seg[-1, :] = 0       # instances never touch the canvas border
                moved = self._geom(seg, True)
                moved_unl = self._geom(unlabeled, True)
                out = instances_to_classes(moved)
                out[(moved_unl > 0.5) | (moved == -1)] = -1
                seg = out
        else:
            seg = instances_to_classes(seg)
        return self._orient(img).astype(np.float32), self._orient(seg).astype(np.float32)


class CTCRAMReaderSequence2D(object):
    """Cell-Tracking-Challenge training reader with the reference's constructor and batch contract
    (DataHandeling.py:21-45, 454-493).  Every `(folder, sequence)` entry needs the reference's
    `metadata_<sequence>.pickle` (`filelist` rows `(image, seg or None, tra, fully_annotated)`, `shape`); all frames are
    held in RAM, z-scored per frame.

    MI355X-native differences: images are read with Pillow (no OpenCV) and the tf.queue / thread machinery is replaced
    by one deterministic generator per batch slot (`seed`, `rank`), optionally prefetched by `num_threads` > 1 worker
    threads into bounded queues; under data-parallel training every rank builds the reader with its own `rank`, so
    the slots of different GPUs follow different clips."""

    def __init__(self, sequence_folder_list, image_crop_size=(128, 128), unroll_len=7, deal_with_end=0, batch_size=4,
                 queue_capacity=32, num_threads=3, data_format='NCHW', randomize=True, return_dist=False, keep_sample=1,
                 elastic_augmentation=True, seed=1, rank=0):
        self.return_dist = bool(return_dist)      # distance-map targets beside the class map (DataHandeling.py:371-377,479-491)
        self.sequence_folder_list = list(sequence_folder_list)
        self.sub_seq_size = tuple(image_crop_size)
        self.dist_sub_seq_size = (2,) + self.sub_seq_size
        self.unroll_len = unroll_len
        self.deal_with_end = deal_with_end
        self.batch_size = batch_size
        self.queue_capacity = queue_capacity
        self.num_threads = num_threads
        self.data_format = data_format
        self.randomize = randomize
        self.keep_sample = keep_sample
        self.elastic_augmentation = elastic_augmentation
        self.sequence_data = {}
        self._seed = (seed, rank)
        self._load_rng = np.random.default_rng([seed, 7])     # label sub-sampling: identical on every rank
        self._slots = None
        self._queues = None
        self._threads = []
        self._stop = False
        self._error = None
        self._wake = None
        self.q_stat_list = []

    # ---- loading -------------------------------------------------------------------------------------------
    def _read_sequence_to_ram_(self):
        import os
        import pickle
        for entry in self.sequence_folder_list:
            folder, seq, train_set = (tuple(entry) + (True,))[:3] if isinstance(entry, (tuple, list)) else (entry, None, True)
            with open(os.path.join(folder, 'metadata_{}.pickle'.format(seq)), 'rb') as fh:
                meta = pickle.load(fh)
            shape = tuple(meta['shape'])[-2:]
            n = len(meta['filelist'])
            images = np.zeros((n,) + shape, np.float32)
            segs = np.full((n,) + shape, -1.0, np.float32)
            full = np.zeros(n, np.float32)
            for t, row in enumerate(meta['filelist']):
                img = _read_image(os.path.join(folder, row[0])).astype(np.float32)
                images[t] = (img - img.mean()) / img.std()
                keep = (self._load_rng.random() < self.keep_sample) and train_set
                if row[1] is None or not keep:
                    continue                                 # unlabeled frame: all -1
                try:
                    seg = _read_image(os.path.join(folder, row[1])).astype(np.float32)
                except (OSError, ValueError):
                    full[t] = -1
                    continue
                if row[3] is True:                           # every cell annotated: 0 really is background
                    full[t] = 1
                else:                                        # partial annotation: unmarked pixels are unknown
                    seg[seg == 0] = -1
                segs[t] = seg
            self.sequence_data[tuple(entry) if isinstance(entry, list) else entry] = {
                'images': images, 'segs': segs, 'full_seg': full, 'metadata': meta, 'max': float(images.max())}

    @staticmethod
    def _gt2dist_(gt_image):
        """Distance of every pixel to the nearest and to the second-nearest cell EDGE (reference `DataHandeling.py:213-236`):
        cells = 8-connected components of `gt == 1` (`cv2.connectedComponents`), a cell's edge = its pixels that a 3 x 3 erosion
        removes (`cv2.erode`, whose default border does not erode from outside the image), distances by scipy's exact Euclidean
        transform; pixels far from two cells keep the start values H + W + 2 and H + W + 3.  Returns `(stack, (dist_1, dist_2))`
        like the reference.  Restated on scipy.ndimage (no OpenCV here); the result does not depend on the label order."""
        gt_image = np.asarray(gt_image)
        labeled, n = ndimage.label(gt_image == 1, structure=np.ones((3, 3)))
        dist_1 = np.ones_like(gt_image, dtype=np.float64) * (gt_image.shape[0] + gt_image.shape[1]) + 2.
        dist_2 = dist_1 + 1.
        for lab in range(1, n + 1):
            bw = labeled == lab
            edge = bw & ~ndimage.binary_erosion(bw, structure=np.ones((3, 3)), border_value=1)
            dist = ndimage.distance_transform_edt(~edge)
            first = dist < dist_1
            dist_2[first] = dist_1[first]
            second = (dist < dist_2) & ~first
            dist_1[first] = dist[first]
            dist_2[second] = dist[second]
        return np.stack((dist_1, dist_2), 0), (dist_1, dist_2)

    # ---- per-slot clip streams -----------------------------------------------------------------------------
    def _clip_stream(self, slot):
        """Endless frames of slot `slot`: (image, seg, full_seg, keep) with keep = 0 on the last frame of a clip."""
        rng = np.random.default_rng([self._seed[0], self._seed[1], slot])
        keys = list(self.sequence_data.keys())
        T = self.unroll_len
        empty_draws = 0
        while True:
            data = self.sequence_data[keys[int(rng.integers(0, len(keys)))]]
            aug = ClipAugmenter(rng, data['images'].shape[1:], self.sub_seq_size, self.randomize, self.elastic_augmentation)
            idx = list(range(len(data['images'])))
            if aug.reverse:
                idx.reverse()
            idx = idx[::aug.step]
            rem = len(idx) % T
            if rem:                                           # window alignment (DataHandeling.py:296-302)
                if self.deal_with_end == 0:
                    idx = idx[:-rem]
                elif self.deal_with_end == 1:
                    idx += idx[-2:-T + rem - 2:-1]
                else:
                    idx += idx[-1:] * (T - rem)
            if not idx:      # deal_with_end == 0 and fewer than unroll_len (sub-sampled) frames: nothing to yield from this draw
                empty_draws += 1
                if empty_draws > 100 * len(keys):      # ... and from no other sequence either: fail instead of spinning
                    raise ValueError('no sequence holds unroll_len = %d frames after temporal sub-sampling '
                                     '(deal_with_end = 0 trims clips to whole windows)' % T)
                continue
            empty_draws = 0
            for j, t in enumerate(idx):
                img, seg = aug.frame(data['images'][t], data['segs'][t], data['max'])
                if not (np.isfinite(img).all() and np.isfinite(seg).all()):
                    raise ValueError('non-finite values in frame {} after augmentation'.format(t))
                item = (img, seg, max(0.0, float(data['full_seg'][t])), 1.0 if j + 1 < len(idx) else 0.0)
                if self.return_dist:      # (frames without labels carry zeros, DataHandeling.py:372-375)
                    dist = np.zeros(self.dist_sub_seq_size, np.float32) if data['full_seg'][t] == -1 else \
                        self._gt2dist_(seg)[0].astype(np.float32)
                    item = item + (dist,)
                yield item

    def start_queues(self, coord=None, debug=False):
        import queue
        import threading
        self._read_sequence_to_ram_()
        self._slots = [self._clip_stream(b) for b in range(self.batch_size)]
        if self.num_threads > 1 and not debug:
            self._queues = [queue.Queue(maxsize=self.queue_capacity) for _ in range(self.batch_size)]
            self._wake = threading.Event()      # set by stop(): idle workers wake up at once

            def work(slots):
                # An exception in a producer (non-finite frame, unreadable file ...) must reach the training loop: the
                # reference stops its coordinator for ALL threads (DataHandeling.py:425-428).  The error is kept on the
                # reader, `_stop` ends every worker, and get_batch -- which never blocks without a timeout -- re-raises it
                # whichever slot queue it happens to be waiting on.
                try:
                    while not self._stop:
                        busy = False
                        for b in slots:
                            if self._queues[b].full():
                                continue
                            self._queues[b].put(next(self._slots[b]))
                            busy = True
                        if not busy:
                            self._wake.wait(0.005)
                except BaseException as exc:      # noqa: B902 -- re-raised by get_batch in the consumer thread
                    if self._error is None:
                        self._error = exc
                    self._stop = True

            n = min(self.num_threads, self.batch_size)
            for i in range(n):
                th = threading.Thread(target=work, args=(list(range(i, self.batch_size, n)),), daemon=True)
                th.start()
                self._threads.append(th)
            self.q_stat_list = [lambda q=q: q.qsize() / float(self.queue_capacity) for q in self._queues]
        return self._threads

    def stop(self):
        self._stop = True
        if self._wake is not None:
            self._wake.set()

    def _next_item(self, b):
        """Next frame of slot b.  With worker threads: wait on the slot queue in short timeouts, surfacing a producer's
        exception (from ANY worker) and a stopped / dead owner instead of blocking forever."""
        if self._queues is None:
            return next(self._slots[b])
        import queue
        while True:
            try:
                return self._queues[b].get(timeout=0.05)
            except queue.Empty:
                if self._error is not None:
                    raise self._error
                if self._stop or not any(th.is_alive() for th in self._threads):
                    raise RuntimeError('the clip reader was stopped while a batch was being assembled')

    def get_batch(self):
        if self._slots is None:
            self.start_queues()
        B, T = self.batch_size, self.unroll_len
        h, w = self.sub_seq_size
        image = np.empty((B, T, h, w), np.float32)
        seg = np.empty((B, T, h, w), np.float32)
        full = np.empty((B, T), np.float32)
        keep = np.ones(B, np.float32)
        dist = np.empty((B, T) + self.dist_sub_seq_size, np.float32) if self.return_dist else None
        for b in range(B):
            for t in range(T):
                if self._error is not None:              # a producer thread died: surface its error here
                    raise self._error
                item = self._next_item(b)
                image[b, t], seg[b, t], full[b, t], keep[b] = item[:4]      # keep: the flag of the window's last frame
                if self.return_dist:
                    dist[b, t] = item[4]
        axis = 2 if self.data_format[1] == 'C' else 4
        if self.return_dist:      # (image, seg, full_seg, is_last, dist [B, T, 2, H, W]) as DataHandeling.py:490-491
            return np.expand_dims(image, axis), np.expand_dims(seg, axis), full, keep, dist
        return np.expand_dims(image, axis), np.expand_dims(seg, axis), full, keep


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
