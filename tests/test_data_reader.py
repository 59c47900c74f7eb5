"""Real-data clip reader: CTCRAMReaderSequence2D on a tiny Cell-Tracking-Challenge-shaped folder written
to tmp_path -- batch contract of DataHandeling.py:454-493 (shapes, value sets, per-slot clip persistence, keep flags,
window alignment), the annotation rules of :98-129, determinism per (seed, rank, slot), and the OpenCV-free warps."""
import os
import pickle

import numpy as np
import pytest
from PIL import Image

import DataHandeling as D


def _make_ctc(root, n_frames=11, shape=(40, 48), seq='01'):
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, seq), exist_ok=True)
    os.makedirs(os.path.join(root, seq + '_GT', 'SEG'), exist_ok=True)
    rows = []
    for t in range(n_frames):
        img = (rng.random(shape) * 900 + 100).astype(np.uint16)
        inst = np.zeros(shape, np.uint16)
        inst[5 + t % 3:15 + t % 3, 6:18] = 1
        inst[20:30, 25 + t % 4:38 + t % 4] = 2
        img[inst > 0] += 2000
        name = os.path.join('.', seq, 't%03d.tif' % t)
        Image.fromarray(img).save(os.path.join(root, name))
        if t % 3 == 0:
            rows.append((name, None, None, None))                       # unlabeled frame
        else:
            sname = os.path.join('.', seq + '_GT', 'SEG', 'man_seg%03d.tif' % t)
            Image.fromarray(inst).save(os.path.join(root, sname))
            rows.append((name, sname, None, True if t % 3 == 1 else False))   # fully / partially annotated
    with open(os.path.join(root, 'metadata_%s.pickle' % seq), 'wb') as fh:
        pickle.dump({'filelist': rows, 'shape': shape, 'max': 3000, 'min': 100}, fh)
    return rows


def _reader(root, **kw):
    args = dict(sequence_folder_list=[(root, '01')], image_crop_size=(32, 32), unroll_len=3, deal_with_end=0, batch_size=2,
                queue_capacity=16, num_threads=1, data_format='NCHW', randomize=True, elastic_augmentation=True)
    args.update(kw)
    return D.CTCRAMReaderSequence2D(**args)


def test_batch_contract_and_annotation_rules(tmp_path):
    root = str(tmp_path)
    _make_ctc(root)
    r = _reader(root)
    r.start_queues()
    data = r.sequence_data[(root, '01')]
    assert np.allclose(data['images'].reshape(11, -1).mean(1), 0, atol=1e-4)          # per-frame z-score (:103)
    assert np.allclose(data['images'].reshape(11, -1).std(1), 1, atol=1e-3)
    assert (data['segs'][0] == -1).all() and data['full_seg'][0] == 0                  # unlabeled frame
    assert data['full_seg'][1] == 1 and set(np.unique(data['segs'][1])) == {0.0, 1.0, 2.0}     # full: 0 = background
    assert data['full_seg'][2] == 0 and set(np.unique(data['segs'][2])) == {-1.0, 1.0, 2.0}    # partial: 0 -> unknown
    for fmt, shp in (('NCHW', (2, 3, 1, 32, 32)), ('NHWC', (2, 3, 32, 32, 1))):
        r = _reader(root, data_format=fmt)
        img, seg, full, keep = r.get_batch()
        assert img.shape == shp and seg.shape == shp and full.shape == (2, 3) and keep.shape == (2,)
        assert img.dtype == np.float32 and seg.dtype == np.float32
        assert set(np.unique(seg)) <= {-1.0, 0.0, 1.0, 2.0}
        assert set(np.unique(keep)) <= {0.0, 1.0}


def test_slot_persistence_window_alignment_and_determinism(tmp_path):
    root = str(tmp_path)
    _make_ctc(root)
    # without augmentation a slot walks the sequence in order; 11 frames trimmed to 9 = 3 windows of T=3
    r = _reader(root, randomize=False, elastic_augmentation=False, batch_size=1)
    r.start_queues()
    ref = r.sequence_data[(root, '01')]['images']
    keeps = []
    for w in range(6):
        img, seg, full, keep = r.get_batch()
        keeps.append(float(keep[0]))
        for t in range(3):
            assert np.array_equal(img[0, t, 0], ref[(3 * w + t) % 9, :32, :32])        # next window continues the clip
    assert keeps == [1.0, 1.0, 0.0, 1.0, 1.0, 0.0]                                     # keep = 0 ends the clip (:378,471)
    # deal_with_end = 2 pads with the last frame instead of trimming: 12 frames = 4 windows
    r = _reader(root, randomize=False, elastic_augmentation=False, batch_size=1, deal_with_end=2)
    keeps = [float(r.get_batch()[3][0]) for _ in range(4)]
    assert keeps == [1.0, 1.0, 1.0, 0.0]
    # same (seed, rank) -> same stream; another rank -> other clips; threads only prefetch, same per-slot order
    a = _reader(root, seed=5, rank=0).get_batch()
    b = _reader(root, seed=5, rank=0).get_batch()
    c = _reader(root, seed=5, rank=1).get_batch()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[0], c[0])
    th = _reader(root, seed=5, rank=0, num_threads=2)
    d = th.get_batch()
    th.stop()
    assert all(np.array_equal(x, y) for x, y in zip(a, d))


def test_opencv_free_warps():
    src = np.array([[10.0, 12.0], [30.0, 12.0], [10.0, 28.0]])
    dst = np.array([[12.0, 11.0], [31.0, 14.0], [9.0, 30.0]])
    m = D.affine_from_points(src, dst)
    assert np.allclose(m @ np.hstack([src, np.ones((3, 1))]).T, dst.T, atol=1e-9)
    img = np.zeros((20, 24))
    img[5:9, 6:11] = 1.0
    shift = np.array([[1.0, 0.0, 3.0], [0.0, 1.0, 2.0]])                                # x + 3, y + 2
    out = D.warp_affine(img, shift, 0, 'constant', -1.0)
    assert np.array_equal(out[7:11, 9:14], np.ones((4, 5))) and out[0, 0] == -1.0      # content moves with the matrix
    ident = D.warp_affine(img, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]), 1, 'mirror')
    assert np.allclose(ident, img)


def test_reader_feeds_training_shapes(tmp_path):
    """Params.CTCParams can be pointed at the reader (the reference's default provider)."""
    import Params
    root = str(tmp_path / 'Fluo-N2DH-SIM+')
    os.makedirs(root)
    _make_ctc(root)
    p = Params.CTCParams({'data_provider_class': D.CTCRAMReaderSequence2D, 'root_data_dir': str(tmp_path), 'dry_run': True,
                          'train_sequence_list': [('Fluo-N2DH-SIM+', '01')], 'val_sequence_list': [('Fluo-N2DH-SIM+', '01')],
                          'crop_size': (32, 32), 'batch_size': 2, 'unroll_len': 2, 'num_train_threads': 1, 'num_val_threads': 1})
    img, seg, full, keep = p.train_data_provider.get_batch()
    assert img.shape == (2, 2, 1, 32, 32) and seg.shape == img.shape


def test_short_clips_and_producer_errors_surface(tmp_path):
    """Two failure modes that used to hang get_batch: (1) deal_with_end = 0 with every clip shorter than unroll_len trims
    each draw to nothing -- now a ValueError instead of an endless loop; (2) an exception in a producer thread travels
    through the slot queue and is re-raised in the consumer (the reference stops its coordinator, DataHandeling.py:425-428)."""
    root = str(tmp_path)
    _make_ctc(root, n_frames=4)
    r = _reader(root, unroll_len=5, deal_with_end=0)
    with pytest.raises(ValueError, match='unroll_len'):
        r.get_batch()
    ok = _reader(root, unroll_len=5, deal_with_end=2)            # padding modes still serve such clips
    assert ok.get_batch()[0].shape == (2, 5, 1, 32, 32)
    r2 = _reader(root, unroll_len=2, num_threads=2)
    r2._read_sequence_to_ram_()
    key = next(iter(r2.sequence_data))
    r2.sequence_data[key]['images'][1][:] = np.nan               # a non-finite frame -> ValueError inside the worker thread
    r2._read_sequence_to_ram_ = lambda: None                     # (start_queues would re-read the files)
    with pytest.raises(ValueError, match='non-finite'):
        for _ in range(200):
            r2.get_batch()
    r2.stop()


def test_producer_error_never_hangs_get_batch(tmp_path):
    """The consumer may be parked on the queue of a HEALTHY worker when another worker dies: the error must still surface
    (shared reader error + timed waits).  Repeated, because the failure used to depend on thread timing."""
    root = str(tmp_path)
    _make_ctc(root, n_frames=4)
    for rep in range(25):
        r = _reader(root, unroll_len=2, num_threads=2, batch_size=3, queue_capacity=4, seed=rep)
        r._read_sequence_to_ram_()
        key = next(iter(r.sequence_data))
        r.sequence_data[key]['images'][rep % 4][:] = np.nan
        r._read_sequence_to_ram_ = lambda: None
        with pytest.raises(ValueError, match='non-finite'):
            for _ in range(200):
                r.get_batch()
        assert r._stop and all(not th.is_alive() or th.join(1.0) is None for th in r._threads)
    # stop() while a consumer waits: a clean error, not a hang
    r = _reader(root, unroll_len=2, num_threads=2, deal_with_end=2)
    r.get_batch()
    r.stop()
    with pytest.raises(RuntimeError, match='stopped'):
        for _ in range(200):
            r.get_batch()


def test_augmentation_helpers_match_the_reference_goldens():
    """tests/golden/elastic.npz was produced by the reference's own static helpers (DataHandeling.py:152-197,239-260) with a
    seeded RandomState; the build's helpers must reproduce them draw for draw."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'elastic.npz'))
    for ci in range(3):
        shape = tuple(int(v) for v in g['shape%d' % ci])
        state = np.random.RandomState(int(g['seed%d' % ci]))
        p1, p2 = D.elastic_affine_points(shape, shape[1] * 0.08, state)
        assert p1.dtype == np.float32 and np.array_equal(p1, g['pts1_%d' % ci]) and np.array_equal(p2, g['pts2_%d' % ci])
        rows, cols = D.elastic_indices(shape, shape[1] * 2, shape[1] * 0.15, state)         # same state: draws continue
        assert rows.shape == (shape[0] * shape[1], 1)
        assert np.array_equal(rows, g['rows%d' % ci]) and np.array_equal(cols, g['cols%d' % ci])
        ident = np.float64([[1, 0, 0], [0, 1, 0]])
        out_img = D.transformed_image(g['img%d' % ci], ident, (rows, cols), seg=False)
        out_lab = D.transformed_image(g['lab%d' % ci], ident, (rows, cols), seg=True)
        assert np.allclose(out_img, g['out_img%d' % ci], atol=1e-12) and np.array_equal(out_lab, g['out_lab%d' % ci])
        # what the reference asks of cv2.warpAffine: dsize = (w, h); image: BORDER_REFLECT_101 (4), bilinear; labels:
        # BORDER_CONSTANT (0) with -1, INTER_NEAREST (0) -- the modes warp_affine() is called with in transformed_image()
        calls = g['warp_calls%d' % ci]
        assert calls[0].tolist() == [shape[1], shape[0], 4, 0, 1] and calls[1].tolist() == [shape[1], shape[0], 0, -1, 0]
    assert np.array_equal(D.adjust_contrast(g['pc_in'], float(g['pc_factor'])), g['pc_contrast'])
    assert np.array_equal(D.adjust_brightness(D.adjust_contrast(g['pc_in'], float(g['pc_factor'])), float(g['pc_delta'])), g['pc_both'])


def test_return_dist_distance_map_targets(tmp_path):
    """`return_dist=True` (reference DataHandeling.py:213-236,371-377,479-491): a fifth batch element [B, T, 2, H, W] with each
    pixel's distance to the nearest / second-nearest cell edge.  `_gt2dist_` against a brute-force restatement (8-connected
    cells of gt == 1, edge = cell pixels with a non-cell 8-neighbour inside the image, plain Euclidean distances)."""
    rng = np.random.default_rng(5)
    gt = np.zeros((14, 17), np.float32)
    gt[1:5, 1:6] = 1
    gt[7:13, 9:16] = 1
    gt[8, 10] = 2          # an edge-class pixel inside: a hole of the cell mask
    gt[0, 12:15] = 1       # touches the image border: the border itself erodes nothing
    gt[5, 6] = 1           # joined to the first cell through a corner (8-connectivity)
    gt[rng.integers(0, 14, 5), rng.integers(0, 17, 5)] = -1
    out, (d1, d2) = D.CTCRAMReaderSequence2D._gt2dist_(gt)
    fg = gt == 1
    lab, n = D.ndimage.label(fg, structure=np.ones((3, 3)))
    assert n == 3 and out.shape == (2, 14, 17) and np.array_equal(out[0], d1) and np.array_equal(out[1], d2)
    H, W = gt.shape
    per_label = []
    for k in range(1, n + 1):
        edge = [(y, x) for y in range(H) for x in range(W) if lab[y, x] == k and any(
            0 <= y + dy < H and 0 <= x + dx < W and lab[y + dy, x + dx] != k for dy in (-1, 0, 1) for dx in (-1, 0, 1))]
        ey, ex = np.array(edge).T
        yy, xx = np.mgrid[0:H, 0:W]
        per_label.append(np.sqrt((yy[..., None] - ey) ** 2 + (xx[..., None] - ex) ** 2).min(-1))
    two = np.sort(np.stack(per_label + [np.full((H, W), H + W + 2.), np.full((H, W), H + W + 3.)], 0), 0)[:2]
    assert np.allclose(out, two, atol=1e-9)
    # ... through the reader: shapes, zeros on frames whose label file could not be read, the map of the frame's own class map
    root = str(tmp_path)
    _make_ctc(root)
    r = _reader(root, return_dist=True, randomize=False, elastic_augmentation=False)
    img, seg, full, keep, dist = r.get_batch()
    assert dist.shape == (2, 3, 2, 32, 32) and dist.dtype == np.float32 and img.shape == (2, 3, 1, 32, 32)
    for b in range(2):
        for t in range(3):
            assert np.allclose(dist[b, t], D.CTCRAMReaderSequence2D._gt2dist_(seg[b, t, 0])[0], atol=1e-5)
    assert len(_reader(root).get_batch()) == 4
