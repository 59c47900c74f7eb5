"""CPU restatement of the reference's inference post-processing (softmax -> uint16 instance labels), Inference2D.py:66-131.
TEST INFRASTRUCTURE ONLY: imported by tests/ (and tests/golden/make_golden.py); the product path is the HIP pipeline in
lstm-unet_amd/csrc/lu_postprocess.hip behind Inference2D.postprocess.

What follows the reference line by line (numpy / scipy calls are THE SAME calls the reference makes -- scipy is installed
here, so these stages are pinned by the real dependency, not by a restatement):
    :66-71   edge threshold, cell mask, binary_fill_holes, edge minus cell
    :77-78   distance_transform_edt(1 - seg_cell, return_indices=True) and the nearest-label absorption of edge pixels
    :80-91   per-object hole filling through bbox_crop / bbox_fill (utils.py:51-69), INCLUDING its additive quirk: a hole
             pixel that already carries another object's label m ends up with m + n
    :93-103  FOV filter, INCLUDING the single-column quirk `fov_im[:, FOV] = 0` (:97)
    :113-123 size / FOV filtering and consecutive relabelling in label order

What is restated because OpenCV is absent (PARITY UNPINNED for this one call -- cv2.connectedComponentsWithStats(img, 8,
CV_32S), :72): the label ORDER.  OpenCV's 8-connectivity labelling (BBDT, Grana et al. 2010, CCL_DEFAULT / CCL_GRANA up to
4.5.1; Spaghetti, Bolelli et al. 2019, from 4.5.2) scans the image in 2 x 2 blocks, block rows top to bottom, blocks left to
right; a block without a foreground 8-neighbour in the already scanned blocks opens a new provisional label; unions keep the
SMALLER label as root; `flattenL` then renumbers the roots 1, 2, ... in increasing provisional order.  All pixels of a
2 x 2 block are mutually 8-adjacent, so a component's smallest provisional label is the one opened at its first block in
block-raster order: components are numbered by the block-raster position of their first block.  Two independent
restatements live here and must agree (tests): `cc_label_block_scan` plays the scan with a union-find;
`cc_label_opencv_order` sorts scipy's components by  min over pixels of (y // 2) * ceil(W / 2) + (x // 2).
(The label PARTITION and the per-label areas do not depend on any of this.)
"""
import numpy as np
import scipy.ndimage


def cc_label_opencv_order(mask):
    """8-connected components of a 2-D boolean mask -> (num_labels incl. background, int32 labels, areas[num_labels])."""
    mask = np.asarray(mask) != 0
    lab, n = scipy.ndimage.label(mask, structure=np.ones((3, 3)))
    H, W = mask.shape
    bw = (W + 1) // 2
    ys, xs = np.nonzero(mask)
    key = (ys // 2) * bw + xs // 2
    first = np.full(n + 1, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, lab[ys, xs], key)
    order = np.argsort(first[1:], kind='stable') + 1            # old labels sorted by first block
    remap = np.zeros(n + 1, np.int32)
    remap[order] = np.arange(1, n + 1, dtype=np.int32)
    out = remap[lab]
    return n + 1, out, np.bincount(out.ravel(), minlength=n + 1)


def cc_label_block_scan(mask):
    """The same labelling by playing the 2 x 2 block scan: provisional labels in block-raster order, min-root unions with
    the previously scanned neighbour blocks (up-left, up, up-right, left), flatten.  Pure-Python: small masks only."""
    mask = np.asarray(mask) != 0
    H, W = mask.shape
    bh, bw = (H + 1) // 2, (W + 1) // 2
    parent = [0]
    blab = np.zeros((bh, bw), np.int64)

    def find(i):
        while parent[i] != i:
            i = parent[i]
        return i

    def pixels(by, bx):
        return [(y, x) for y in (2 * by, 2 * by + 1) for x in (2 * bx, 2 * bx + 1) if y < H and x < W and mask[y, x]]

    for by in range(bh):
        for bx in range(bw):
            mine = pixels(by, bx)
            if not mine:
                continue
            roots = set()
            for (ny, nx) in ((by - 1, bx - 1), (by - 1, bx), (by - 1, bx + 1), (by, bx - 1)):
                if 0 <= ny < bh and 0 <= nx < bw and blab[ny, nx]:
                    theirs = pixels(ny, nx)
                    if any(abs(y - v) <= 1 and abs(x - u) <= 1 for (y, x) in mine for (v, u) in theirs):
                        roots.add(find(blab[ny, nx]))
            if not roots:
                parent.append(len(parent))
                blab[by, bx] = len(parent) - 1
            else:
                r = min(roots)
                for q in roots:
                    parent[q] = r
                blab[by, bx] = r
    final, k = {}, 1
    for i in range(1, len(parent)):          # flattenL: roots renumbered in increasing provisional order
        if parent[i] == i:
            final[i] = k
            k += 1
    out = np.zeros((H, W), np.int32)
    for by in range(bh):
        for bx in range(bw):
            if blab[by, bx]:
                for (y, x) in pixels(by, bx):
                    out[y, x] = final[find(blab[by, bx])]
    return k, out, np.bincount(out.ravel(), minlength=k)


def bbox_crop(img, margin=10):
    """utils.py:51-62."""
    rows, cols = np.any(img, axis=1), np.any(img, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    rmin, cmin = max(0, rmin - margin), max(0, cmin - margin)
    rmax, cmax = min(img.shape[0], rmax + margin), min(img.shape[1], cmax + margin)
    return img[rmin:rmax, cmin:cmax], (rmin, rmax, cmin, cmax)


def bbox_fill(img, crop, loc):
    """utils.py:65-69."""
    rmin, rmax, cmin, cmax = loc
    img = img.copy()
    img[rmin:rmax, cmin:cmax] = crop
    return img


def postprocess(softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, stages=None):
    """softmax [3,H,W] -> uint16 labels, following Inference2D.py:66-123 statement by statement.
    fov_fix=False keeps the reference's `fov_im[:, FOV] = 0` (one column); True zeroes columns [:FOV].
    stages: optional dict that receives the intermediate arrays (seg_cell, labels after each stage, areas)."""
    sm = np.asarray(softmax_chw)
    seg_edge = np.greater_equal(sm[2], 0.2)
    seg_cell = np.logical_and(np.equal(np.argmax(sm, 0), 1).astype(np.float32), np.logical_not(seg_edge))
    seg_edge = seg_edge.astype(np.float32)
    seg_cell = scipy.ndimage.binary_fill_holes(seg_cell).astype(np.float32)
    seg_edge = np.maximum((seg_edge - seg_cell), 0)
    num_cells, labels, areas = cc_label_opencv_order(seg_cell.astype(np.uint8))
    if stages is not None:
        stages.update(seg_cell=seg_cell.copy(), seg_edge=seg_edge.copy(), cc=labels.copy(), areas=areas.copy())
    dist, ind = scipy.ndimage.distance_transform_edt(1 - seg_cell, return_indices=True)
    labels = labels[ind[0, :], ind[1, :]] * seg_edge * (dist < edge_dist) + labels
    if stages is not None:
        stages['absorbed'] = labels.copy()
    for n in range(1, num_cells):
        bw = labels == n
        if not np.any(bw):
            continue
        bw_crop, loc = bbox_crop(bw)
        fill_crop = scipy.ndimage.binary_fill_holes(bw_crop).astype(np.float32)
        fill_diff = fill_crop - bw_crop
        bw_fill = bbox_fill(bw, fill_diff, loc)
        labels = labels + bw_fill * n
    if stages is not None:
        stages['filled'] = labels.copy()
    if fov:
        fov_im = np.ones_like(labels)
        fov_im[:fov, :] = 0
        fov_im[-fov:, :] = 0
        if fov_fix:
            fov_im[:, :fov] = 0
        else:
            fov_im[:, fov] = 0
        fov_im[:, -fov:] = 0
        unique_fov_labels = np.unique((labels * fov_im).flatten())
        remove_ind = np.setdiff1d(np.arange(num_cells), unique_fov_labels)
    else:
        remove_ind = []
    labels_out = np.zeros_like(labels, dtype=np.uint16)
    p = 0
    for n in range(1, num_cells):
        if min_cell_size <= areas[n] <= max_cell_size and not (n in remove_ind):
            p += 1
            labels_out[labels == n] = p
    return labels_out


def edt_nearest_bruteforce(fg):
    """Nearest foreground pixel of every pixel with scipy's tie-break (tests pin it against
    scipy.ndimage.distance_transform_edt): among equidistant candidates the smallest column, then the smallest row."""
    fg = np.asarray(fg) != 0
    H, W = fg.shape
    ys, xs = np.nonzero(fg)
    ind = np.zeros((2, H, W), np.int64)
    d2o = np.zeros((H, W), np.int64)
    for y in range(H):
        for x in range(W):
            d2 = (ys - y) ** 2 + (xs - x) ** 2
            m = d2.min()
            c = np.nonzero(d2 == m)[0]
            b = min(c, key=lambda i: (xs[i], ys[i]))
            ind[0, y, x], ind[1, y, x], d2o[y, x] = ys[b], xs[b], m
    return d2o, ind


def synthetic_softmax(H, W, seed, n_cells=12, nested=False, noise=0.15, rmax=None):
    """A [3,H,W] softmax of blobby cells with edge rings, touching neighbours, specks and (nested=True) ring-shaped cells
    with a smaller cell inside -- exercises hole filling, edge absorption ties and the additive quirk."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    cell = np.zeros((H, W), np.float32)
    edge = np.zeros((H, W), np.float32)
    for i in range(n_cells):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ry, rx = rng.uniform(3, rmax or max(4, H / 8)), rng.uniform(3, rmax or max(4, W / 8))
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        ring = nested and i % 3 == 0
        inner = 0.45 if ring else -1.0
        cell = np.maximum(cell, ((d < 1.0) & (d > inner)).astype(np.float32))
        edge = np.maximum(edge, ((d >= 1.0) & (d < 1.0 + rng.uniform(0.2, 0.6))).astype(np.float32))
        if ring:
            edge = np.maximum(edge, ((d <= inner) & (d > inner - 0.12)).astype(np.float32) * (rng.random() < 0.5))
            cell = np.maximum(cell, (d < 0.08).astype(np.float32))
    logits = np.stack([np.ones((H, W), np.float32), 2.5 * cell, 2.5 * edge * (1 - 0.5 * cell)]) + \
        noise * rng.standard_normal((3, H, W)).astype(np.float32)
    speck = rng.random((H, W)) < 0.01
    logits[1][speck] += 3.0
    e = np.exp(logits - logits.max(0))
    return (e / e.sum(0)).astype(np.float32)
