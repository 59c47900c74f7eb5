"""Streaming inference driver with the command line of the reference's Inference2D.py (flags :179-207).

Per frame (Inference2D.py:45-62): reshape to [1,1,1,H,W] (NCHW) / [1,1,H,W,1], one stateful forward with
training=False and pad_image=True, take the softmax [3,H,W]; the first `pre_sequence_frames` frames (a
mirrored prefix) only warm the recurrent state.  The forward runs on the gfx950 kernels.

Post-processing to instance label maps (Inference2D.py:66-131) is host code in the reference too
(numpy / scipy / OpenCV).  OpenCV is not available here, so 8-connected labelling uses
scipy.ndimage.label: the label PARTITION is the same, the label NUMBERING may differ from
cv2.connectedComponentsWithStats (unpinned) -- output ids are relabelled consecutively
after filtering in both implementations.
"""
import argparse
import os
import pickle

import numpy as np
import scipy.ndimage

from utils import log_print, get_model, bbox_crop, bbox_fill


def postprocess(softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0):
    """softmax [3,H,W] -> uint16 instance labels (0 = background)."""
    seg_edge = np.greater_equal(softmax_chw[2], 0.2)
    seg_cell = np.logical_and(np.equal(np.argmax(softmax_chw, 0), 1), np.logical_not(seg_edge))
    seg_cell = scipy.ndimage.binary_fill_holes(seg_cell).astype(np.float32)
    seg_edge = np.maximum(seg_edge.astype(np.float32) - seg_cell, 0)
    labels, n = scipy.ndimage.label(seg_cell.astype(np.uint8), structure=np.ones((3, 3)))
    num_cells = n + 1
    areas = np.bincount(labels.ravel(), minlength=num_cells)
    labels = labels.astype(np.float32)
    dist, ind = scipy.ndimage.distance_transform_edt(1 - seg_cell, return_indices=True)
    labels = labels[ind[0], ind[1]] * seg_edge * (dist < edge_dist) + labels
    for lab in range(1, num_cells):
        bw = labels == lab
        if not np.any(bw):
            continue
        crop, loc = bbox_crop(bw)
        fill = scipy.ndimage.binary_fill_holes(crop).astype(np.float32) - crop
        labels = labels + bbox_fill(bw, fill, loc) * lab
    remove = []
    if fov:
        inside = np.ones_like(labels)
        inside[:fov, :] = 0
        inside[-fov:, :] = 0
        inside[:, :fov] = 0       # the reference zeroes a single column here (Inference2D.py:97, a typo)
        inside[:, -fov:] = 0
        remove = np.setdiff1d(np.arange(num_cells), np.unique(labels * inside))
    out = np.zeros(labels.shape, np.uint16)
    nxt = 0
    for lab in range(1, num_cells):
        if min_cell_size <= areas[lab] <= max_cell_size and lab not in remove:
            nxt += 1
            out[labels == lab] = nxt
    return out


def stream_softmax(model, frames, data_format='NCHW', pre_sequence_frames=0):
    """Yield (t, softmax [3,H,W]) for every real frame; warm-up frames are consumed silently."""
    nchw = data_format[1] == 'C'
    for T, image in enumerate(frames):
        t = T - pre_sequence_frames
        image = np.asarray(image, np.float32)
        if image.ndim == 2:
            image = image.reshape((1, 1, 1) + image.shape) if nchw else image.reshape((1, 1) + image.shape + (1,))
        elif image.ndim == 3:
            image = image.reshape((1, 1) + image.shape)
        else:
            raise ValueError()
        _, sm = model(image, training=False)
        if t < 0:
            continue
        sm = sm.cpu().numpy()[0, 0]
        yield t, (sm if nchw else np.transpose(sm, (2, 0, 1)))


def inference(params):
    from PIL import Image
    with open(os.path.join(params.model_path, 'model_params.pickle'), 'rb') as fobj:
        model_dict = pickle.load(fobj)
    model_cls = get_model(model_dict['name'])
    model = model_cls(*model_dict['params'], data_format=params.data_format, pad_image=True,
                      precision=getattr(params, 'precision', 'fp32'))
    model.load_weights(os.path.join(params.model_path, 'model.ckpt'))
    log_print('Restored from {}'.format(os.path.join(params.model_path, 'model.ckpt')))
    dataset = params.data_reader(params.sequence_path, params.filename_format,
                                 pre_sequence_frames=params.pre_sequence_frames).dataset
    # Post-processing (scipy, ~15 ms per 256x256 frame) and file output run on worker threads while the GPU computes the
    # next frames: at bf16 rates the forward is 2.6 ms per frame and the host side would otherwise set the pace.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=int(getattr(params, 'num_post_threads', 4)))

    def finish(t, sm):
        labels = postprocess(sm, params.edge_dist, params.min_cell_size, params.max_cell_size, params.FOV)
        out_fname = os.path.join(params.output_path, 'mask{time:03d}.tif'.format(time=t))
        Image.fromarray(labels).save(out_fname)
        log_print('Saved File: {}'.format(out_fname))
        if params.save_intermediate:
            vis = np.round(np.transpose(sm, (1, 2, 0)) * (2 ** 16 - 1)).astype(np.uint16)
            np.save(os.path.join(params.save_intermediate_vis_path, 'softmax{time:03d}.npy'.format(time=t)), vis)
            Image.fromarray(labels).save(os.path.join(params.save_intermediate_label_path,
                                                      'mask{time:03d}.tif'.format(time=t)))

    pending = []
    try:
        for t, sm in stream_softmax(model, dataset, params.data_format, params.pre_sequence_frames):
            if params.dry_run:
                continue
            pending.append(pool.submit(finish, t, sm))
            while len(pending) > 16:          # bounded backlog
                pending.pop(0).result()
        for job in pending:
            job.result()
    except (KeyboardInterrupt, ValueError) as err:
        print('Error: {}'.format(str(err)))
    finally:
        pool.shutdown(wait=True)
        print('Done!')


FLAGS = [
    (('--gpu_id',), dict(dest='gpu_id', type=str, help="Visible GPUs: example, '0,2,3', use -1 for CPU")),
    (('--model_path',), dict(dest='model_path', type=str, help='Path to trained model generated by train2D.py')),
    (('--sequence_path',), dict(dest='sequence_path', type=str, help='Path to sequence images')),
    (('--output_path',), dict(dest='output_path', type=str, help='Directory to save outputs')),
    (('--filename_format',), dict(dest='filename_format', type=str, help="Format of file names ('t*.tif')")),
    (('--data_format',), dict(dest='data_format', type=str, choices=['NCHW', 'NWHC', 'NHWC'],
                              help='Data format NCHW or NHWC')),
    (('--min_cell_size',), dict(dest='min_cell_size', type=int, help='Minimum cell size')),
    (('--max_cell_size',), dict(dest='max_cell_size', type=int, help='Maximum cell size')),
    (('--num_iterations',), dict(dest='num_iterations', type=int, help='Maximum number of training iterations')),
    (('--edge_dist',), dict(dest='edge_dist', type=int, help='Maximum edge width to add to cell object')),
    (('--pre_sequence_frames',), dict(dest='pre_sequence_frames', type=int,
                                      help='Number of frames to run before sequence, uses mirror of first N frames.')),
    (('--save_intermediate',), dict(dest='save_intermediate', action='store_const', const=True,
                                    help='Save intermediate files')),
    (('--save_intermediate_path',), dict(dest='save_intermediate_path', type=str,
                                         help='Path to save intermediate files, used only with --save_intermediate')),
    (('--dry_run',), dict(dest='dry_run', action='store_const', const=True, help='Do not write any outputs')),
    (('--precision',), dict(dest='precision', choices=['fp32', 'bf16'],
                            help='[MI355X] fp32 (default) or bf16 MFMA operands for the wide convolutions')),
]


def build_arg_parser():
    parser = argparse.ArgumentParser(description='Run Inference LSTMUnet Segmentation (MI355X-native)')
    for names, kw in FLAGS:
        parser.add_argument(*names, **kw)
    return parser


if __name__ == '__main__':
    from Params import CTCInferenceParams
    args = build_arg_parser().parse_args()
    args_dict = {key: val for key, val in vars(args).items() if val is not None}
    print(args_dict)
    inference(CTCInferenceParams(args_dict))
