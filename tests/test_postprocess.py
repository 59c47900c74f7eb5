"""Inference post-processing (reference Inference2D.py:66-123): the device pipeline behind Inference2D.postprocess against
oracle/postprocess_oracle.py, bit for bit -- final uint16 label maps AND the intermediate maps (after edge absorption, after
the per-object hole fill).  The oracle itself is pinned first: its scipy stages ARE scipy calls, the distance-transform
tie-break the GPU search relies on is checked against scipy.ndimage.distance_transform_edt, and OpenCV's label order
(cv2 absent: unpinned) is restated twice -- block scan with union-find vs first-block sort -- plus hand KATs.
'emu' = the same kernels on the host SIMT emulator (small frames); 'hip' = the MI355X (`-m gpu`, incl. 832 x 992)."""
import numpy as np
import pytest
import scipy.ndimage

from oracle import postprocess_oracle as po
from engine_backend import engine_backend

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    with engine_backend(request.param) as d:
        yield d


# ----------------------------------------------------------------------------------------------- oracle pins (CPU)
def test_edt_tiebreak_matches_scipy():
    rng = np.random.default_rng(0)
    for _ in range(150):
        H, W = rng.integers(2, 12), rng.integers(2, 12)
        fg = rng.random((H, W)) < rng.choice([0.05, 0.15, 0.3, 0.5])
        if not fg.any():
            continue
        dist, ind = scipy.ndimage.distance_transform_edt(1 - fg.astype(np.float32), return_indices=True)
        d2, i2 = po.edt_nearest_bruteforce(fg)
        assert np.allclose(dist, np.sqrt(d2)) and np.array_equal(ind[0], i2[0]) and np.array_equal(ind[1], i2[1])


def test_cc_label_order_restatements_and_kats():
    rng = np.random.default_rng(1)
    for _ in range(200):
        m = rng.random((rng.integers(1, 12), rng.integers(1, 12))) < rng.choice([0.2, 0.4, 0.6])
        a, b = po.cc_label_opencv_order(m), po.cc_label_block_scan(m)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        # the partition is scipy's 8-connected partition
        lab, n = scipy.ndimage.label(m, structure=np.ones((3, 3)))
        assert n + 1 == a[0] and len(set(zip(lab[m].tolist(), a[1][m].tolist()))) == n
    # block order != pixel-raster order: (1,0) lies in block (0,0), (0,5) in block (0,2)
    m = np.zeros((4, 8), bool)
    m[1, 0] = m[0, 5] = True
    assert po.cc_label_opencv_order(m)[1][1, 0] == 1 and po.cc_label_opencv_order(m)[1][0, 5] == 2
    assert scipy.ndimage.label(m, structure=np.ones((3, 3)))[0][0, 5] == 1
    # touching diagonals are one component; a U shape is one component whose arms meet late in the scan
    d = np.eye(5, dtype=bool)
    assert po.cc_label_opencv_order(d)[0] == 2 and po.cc_label_block_scan(d)[0] == 2
    u = np.zeros((6, 7), bool)
    u[0:5, 1] = u[0:5, 5] = True
    u[4, 1:6] = True
    lone = u.copy()
    lone[0, 3] = True                       # a pixel between the arms: own component, opened AFTER the left arm
    n, lab, area = po.cc_label_opencv_order(lone)
    assert n == 3 and lab[0, 1] == 1 and lab[0, 5] == 1 and lab[0, 3] == 2 and area[1] == 13 and area[2] == 1
    assert np.array_equal(po.cc_label_block_scan(lone)[1], lab)
    # nested rings: the outer ring comes first, the inner blob is a separate component
    r = np.zeros((9, 9), bool)
    r[1:8, 1:8] = True
    r[2:7, 2:7] = False
    r[4, 4] = True
    n, lab, _ = po.cc_label_opencv_order(r)
    assert n == 3 and lab[1, 1] == 1 and lab[4, 4] == 2


# ----------------------------------------------------------------------------------------------- device pipeline
def _scenario(kind):
    """Hand-built softmax maps for the branches random blobs rarely reach."""
    H, W = 40, 48
    cell, edge = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
    if kind == 'c_ring_closed_by_edge':
        # a C-shaped cell whose gap is bridged by edge pixels: absorption closes the ring -> the object gets a hole that
        # holds zero pixels AND another cell: the additive quirk (inner label m becomes m + n) and the sequential fall-back
        cell[8:28, 10:30] = 1
        cell[12:24, 14:26] = 0
        cell[17:19, 10:14] = 0               # the gap (left side), two pixels wide
        edge[17:19, 10:14] = 1               # ... filled with edge pixels: each is 1 away from the C and absorbed into it
        cell[16:20, 18:22] = 1               # the inner cell
        cell[2:6, 36:44] = 1                 # unrelated objects before / after in label order
        cell[32:38, 4:12] = 1
        cell[30:38, 30:44] = 1
        cell[33:35, 34:40] = 0               # a plain hole, removed by the global fill
    elif kind == 'edge_ties':
        # cells one and two pixels apart with edge pixels in between: equidistant candidates with different labels
        for i, x in enumerate((4, 14, 25, 37)):
            cell[6:30, x:x + 8] = 1
        edge[4:32, 2:46] = 1
        edge[10:14, 0:48] = 1
    elif kind == 'holes_after_absorption':
        # open rings closed by absorbed edge pixels, nothing but zeros inside: the parallel (non-dirty) hole fill
        cell[5:20, 5:20] = 1
        cell[9:16, 9:16] = 0
        cell[12:14, 5:9] = 0                 # gap, bridged by edge pixels
        edge[12:14, 5:9] = 1
        cell[24:37, 22:42] = 1
        cell[28:33, 26:38] = 0
        cell[24:28, 30:32] = 0               # gap on top
        edge[24:28, 30:32] = 1
        edge[30, 30:34] = 1                  # edge pixels deep inside the hole: too far to be absorbed, filled as holes
    elif kind == 'border_objects':
        cell[0:6, 0:9] = 1
        cell[0:5, 30:48] = 1
        cell[34:40, 0:48:3] = 1
        cell[15:25, 44:48] = 1
        edge[6, 0:10] = 1
    logits = np.stack([np.ones((H, W), np.float32), 3 * cell, 3 * edge])
    e = np.exp(logits - logits.max(0))
    return (e / e.sum(0)).astype(np.float32)


def _check(dev, sm, **kw):
    import Inference2D
    import torch
    st_ref, st_got = {}, {}
    ref = po.postprocess(sm, stages=st_ref, **kw)
    got = Inference2D.postprocess(torch.from_numpy(sm).to(dev), stages=st_got, **kw)
    assert got.dtype == np.uint16 and got.shape == ref.shape
    assert np.array_equal(st_got['absorbed'], st_ref['absorbed'].astype(np.int64)), 'after edge absorption'
    assert np.array_equal(st_got['filled'], st_ref['filled'].astype(np.int64)), 'after the per-object hole fill'
    if st_got['areas'] is not None:
        assert np.array_equal(st_got['areas'], st_ref['areas'])
    assert np.array_equal(got, ref)
    # the same frame without `stages`: one lu_post_frame call, nested objects replayed in label order on the device
    assert np.array_equal(Inference2D.postprocess(torch.from_numpy(sm).to(dev), **kw), ref), 'device-driven frame'
    return ref, st_ref


def test_postprocess_scenarios(dev):
    for kind in ('c_ring_closed_by_edge', 'edge_ties', 'holes_after_absorption', 'border_objects'):
        sm = _scenario(kind)
        ref, st = _check(dev, sm, edge_dist=2, min_cell_size=1, max_cell_size=10 ** 6)
        if kind == 'c_ring_closed_by_edge':
            inner = st['absorbed'][17, 19]
            ring = st['absorbed'][9, 11]
            assert inner > 0 and ring > 0 and inner != ring
            assert st['filled'][17, 19] == inner + ring, 'the additive quirk: a labelled pixel inside a hole gets m + n'
            assert st['filled'][14, 15] == ring, 'zero pixels of the hole take the ring label'
        if kind == 'holes_after_absorption':
            assert (st['filled'] != st['absorbed']).sum() >= 2
        _check(dev, sm, edge_dist=3, min_cell_size=10, max_cell_size=300)
        _check(dev, sm, edge_dist=2, min_cell_size=1, max_cell_size=10 ** 6, fov=3)
        _check(dev, sm, edge_dist=2, min_cell_size=1, max_cell_size=10 ** 6, fov=3, fov_fix=True)
    empty = np.zeros((3, 9, 11), np.float32)
    empty[0] = 1.0
    assert _check(dev, empty)[0].max() == 0
    full = np.zeros((3, 7, 8), np.float32)
    full[1] = 1.0
    assert _check(dev, full, min_cell_size=1, max_cell_size=100)[0].min() == 1


def test_postprocess_random_maps(dev):
    big = dev.type == 'cuda'
    rng = np.random.default_rng(7)
    n_maps = 0
    for i in range(22 if big else 8):
        H, W = (int(rng.integers(40, 300)), int(rng.integers(40, 300))) if big else (int(rng.integers(17, 60)), int(rng.integers(17, 60)))
        sm = po.synthetic_softmax(H, W, seed=100 + i, n_cells=int(rng.integers(3, 40 if big else 10)), nested=bool(i % 2),
                                  noise=float(rng.choice([0.05, 0.3, 0.6])))
        kw = dict(edge_dist=int(rng.choice([1, 2, 2, 3])), min_cell_size=int(rng.choice([1, 10])),
                  max_cell_size=int(rng.choice([100, 10 ** 6])), fov=int(rng.choice([0, 0, 4])), fov_fix=bool(i % 3 == 0))
        _check(dev, sm, **kw)
        n_maps += 1
    assert n_maps >= 8


@pytest.mark.gpu
def test_postprocess_full_frame_832x992():
    """Fluo-C2DL-MSC frame size (BASELINE config-4): hundreds of objects, noisy edges."""
    import torch
    for seed, nested in ((1, False), (2, True)):
        sm = po.synthetic_softmax(832, 992, seed=seed, n_cells=400, nested=nested, noise=0.25, rmax=16)
        ref, st = _check(torch.device('cuda', 0), sm, edge_dist=2, min_cell_size=10, max_cell_size=5000, fov=0)
        assert ref.max() > 50


def test_post_pipeline_returns_the_same_label_maps(dev):
    """Inference2D.PostPipeline (post-processing of frame t on a side stream while frame t + 1's forward is in flight)
    delivers, one frame late and in order, exactly the label maps of the synchronous postprocess() -- also when a busy
    main stream keeps producing work between the pushes."""
    import Inference2D
    import torch
    kw = dict(edge_dist=2, min_cell_size=4, max_cell_size=10 ** 6, fov=3)
    sms = [po.synthetic_softmax(64, 80, seed, n_cells=9, nested=(seed % 2 == 0)) for seed in range(6)]
    want = [po.postprocess(sm, **kw) for sm in sms]
    pipe = Inference2D.PostPipeline(kw['edge_dist'], kw['min_cell_size'], kw['max_cell_size'], kw['fov'], graph=True)
    got = []
    busy = torch.randn(512, 512, device=dev)
    for t, sm in enumerate(sms):
        d = torch.from_numpy(sm).to(dev)
        for _ in range(4):
            busy = torch.tanh(busy @ busy * 1e-3)         # main-stream work standing in for the next forward
        got += pipe.push(t, d)
    got += pipe.flush()
    assert [t for t, _, _ in got] == list(range(len(sms)))
    for (t, labels, _), ref in zip(got, want):
        assert np.array_equal(labels, ref), t
    if dev.type == 'cuda':
        # graph=True: each processor's launch sequence is replayed from a hipGraph (capturable: no host decision inside a
        # frame); the default eager pipeline gives the same maps
        assert sum(p.replays for p in pipe._procs) == len(sms)
        eager = Inference2D.PostPipeline(kw['edge_dist'], kw['min_cell_size'], kw['max_cell_size'], kw['fov'], graph=False)
        got2 = []
        for t, sm in enumerate(sms):
            got2 += eager.push(t, torch.from_numpy(sm).to(dev))
        got2 += eager.flush()
        assert all(np.array_equal(a[1], b[1]) for a, b in zip(got, got2)) and sum(p.replays for p in eager._procs) == 0


def test_device_driven_fill_and_its_exact_fallback(dev):
    """The frame is device-driven (lu_post_fill_all + lu_post_newid, one device -> host copy); nested objects -- where the
    reference's additive quirk makes the label order matter -- and crops beyond the kernel's LDS staging fall back to the
    strictly sequential replay.  Both routes must give the oracle's map, and each must actually be taken where expected."""
    import Inference2D
    import torch
    from lu_native.post import PostProcessor
    proc = PostProcessor()
    kw = dict(edge_dist=2, min_cell_size=1, max_cell_size=10 ** 6)
    sm = _scenario('holes_after_absorption')                     # holes, nothing nested: device route
    st = {}
    got = proc(torch.from_numpy(sm).to(dev), stages=st, **kw)
    assert proc.fallbacks == 0 and (st['filled'] != st['absorbed']).sum() >= 2
    assert np.array_equal(got, po.postprocess(sm, **kw))
    sm = _scenario('c_ring_closed_by_edge')                      # a cell inside a closed ring: the quirk -> replay in label order
    got = proc(torch.from_numpy(sm).to(dev), **kw)               # ... by one workgroup on the device
    assert proc.device_replays == 1 and proc.fallbacks == 0 and np.array_equal(got, po.postprocess(sm, **kw))
    st = {}
    got = proc(torch.from_numpy(sm).to(dev), stages=st, **kw)    # ... and object by object from the host (the `stages` form)
    assert proc.fallbacks == 1 and np.array_equal(got, po.postprocess(sm, **kw))
    # several open rings closed by absorbed edge pixels (holes that exist only AFTER absorption), nothing nested: processed
    # concurrently by different workgroups of ONE launch
    def rings(H, W, boxes, gap=2):
        cell, edge = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        for (y0, x0, h, w, t) in boxes:                           # a t-thick rectangular ring with a gap on its left side
            cell[y0:y0 + h, x0:x0 + w] = 1
            cell[y0 + t:y0 + h - t, x0 + t:x0 + w - t] = 0
            gy = y0 + h // 2
            cell[gy:gy + gap, x0:x0 + t] = 0
            edge[gy:gy + gap, x0:x0 + t] = 1
        logits = np.stack([np.ones((H, W), np.float32), 3 * cell, 3 * edge])
        e = np.exp(logits - logits.max(0))
        return (e / e.sum(0)).astype(np.float32)

    boxes = [(3, 3, 16, 18, 3), (4, 30, 20, 14, 4), (5, 52, 14, 30, 3), (30, 6, 24, 22, 4), (32, 40, 20, 40, 5)]
    sm = rings(60, 90, boxes)
    before = proc.fallbacks
    st = {}
    got = proc(torch.from_numpy(sm).to(dev), stages=st, **kw)
    ref = po.postprocess(sm, **kw)
    assert proc.fallbacks == before and np.array_equal(got, ref) and ref.max() == 5
    for (y0, x0, h, w, t) in boxes:                               # every ring's interior took the ring's label in the per-object fill
        assert st['absorbed'][y0 + h // 2 - 1, x0 + w // 2] == 0 and ref[y0 + h // 2 - 1, x0 + w // 2] == ref[y0 + 1, x0 + 1] != 0
    if dev.type == 'cuda':                                       # a ring whose crop exceeds the LDS staging (48 K pixels)
        sm = rings(300, 320, [(10, 12, 270, 290, 12), (100, 100, 40, 40, 5)])
        before, before_host = proc.device_replays, proc.fallbacks      # ... goes to the one-workgroup replay (global scratch)
        got = proc(torch.from_numpy(sm).to(dev), **kw)
        assert proc.device_replays == before + 1 and proc.fallbacks == before_host
        assert np.array_equal(got, po.postprocess(sm, **kw))
