"""Streaming inference driver with the command line of the reference's Inference2D.py (flags :179-207).

Per frame (Inference2D.py:45-62): reshape to [1,1,1,H,W] (NCHW) / [1,1,H,W,1], one stateful forward with
training=False and pad_image=True, take the softmax [3,H,W]; the first `pre_sequence_frames` frames (a
mirrored prefix) only warm the recurrent state.  The forward runs on the gfx950 kernels.

Post-processing to instance label maps (Inference2D.py:66-131: host numpy / scipy / OpenCV in the reference) runs on the
GPU here: union-find connected components numbered in cv2.connectedComponentsWithStats' order, binary_fill_holes, nearest-label
edge absorption with scipy's distance-transform tie-break, the per-object hole fill with the reference's additive quirk, the
FOV / size filters and the consecutive relabel (csrc/lu_postprocess.hip, lu_native/post.py).  Label maps are bit-identical to
the reference's algorithm as restated in oracle/postprocess_oracle.py (scipy stages run through scipy itself there; the
OpenCV label ORDER is a restatement of its block scan: OpenCV is not installed -- SURVEY §8f-1).  The reference's FOV quirk
(`fov_im[:, FOV] = 0` masks ONE column, Inference2D.py:97) is kept by default; `--fov_fix` masks columns [0, FOV).
"""
import argparse
import os
import pickle

import numpy as np

from utils import log_print, get_model, select_gpu, write_tiff16


_POST = None


def postprocess(softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, stages=None):
    """softmax [3,H,W] (device tensor, or host array -- moved to the device) -> uint16 instance labels (0 = background),
    Inference2D.py:66-123 on the GPU.  fov_fix=False reproduces the reference's single-column FOV mask (:97)."""
    import torch
    import Networks
    from lu_native.post import PostProcessor
    global _POST
    if _POST is None:
        _POST = PostProcessor()
    sm = softmax_chw
    if not torch.is_tensor(sm) or sm.device.type == 'cpu':
        sm = torch.as_tensor(np.asarray(sm), dtype=torch.float32).to(Networks._device())
    return _POST(sm, edge_dist, min_cell_size, max_cell_size, fov, fov_fix, stages)


class PostPipeline(object):
    """Software pipeline of the per-frame path: the post-processing of frame t (lu_native/post.py: ONE foreign call that
    enqueues a device-driven chain of small kernels ending in one device -> host copy) goes to its OWN HIP stream right after
    the forward that produced its softmax, and runs there while the forwards of the next frames keep the chip busy.
    push(t, softmax) enqueues frame t and returns the frames that are certainly finished by then as [(t, labels, softmax)]
    -- TWO frames late -- and flush() the rest.  Two frames, because the host must never wait for a frame whose kernels still
    share the GPU with the current forward: at bf16 rates (1.7 ms per frame, host and GPU neck and neck) collecting frame
    t - 1 stalled the launch thread until the forward of frame t had drained, and the frame rate halved at random (290 vs 580
    frames/s measured).  Two processors alternate; frame t - 2's buffers are collected before frame t re-uses them.
    Results are those of postprocess(): same kernels, same order per frame."""

    def __init__(self, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, graph=False, depth=2):
        self.args = (edge_dist, min_cell_size, max_cell_size, fov, fov_fix)
        self.depth = int(depth)      # frames in flight behind the forward (>= 2 for overlap: see the class comment; 1 serialises)
        if self.depth < 1:
            raise ValueError('PostPipeline depth must be >= 1 (2 or more for overlap), got %r' % (depth,))
        self.stream = None
        self.pending = []
        self._procs = None
        self._n = 0
        self.graph = graph       # replay the frame's launches from a hipGraph (lu_native.post: measured slower, off by default)

    def _finish(self):
        import torch
        t, sm, proc, job = self.pending.pop(0)
        with torch.cuda.stream(self.stream):
            labels = proc.collect(job)                # waits for that frame's copy (oversize crops: replays it from the host)
        return [(t, labels, sm)]

    def push(self, t, softmax_chw):
        import torch
        if softmax_chw.device.type != 'cuda':        # the host emulator of the test-suite: nothing to overlap
            return [(t, postprocess(softmax_chw, *self.args), softmax_chw)]
        if self.stream is None:
            from lu_native.post import PostProcessor
            self.stream = torch.cuda.Stream()
            self._procs = [PostProcessor(graph=self.graph) for _ in range(self.depth)]
        done = self._finish() if len(self.pending) == self.depth else []      # frame t - depth (its processor is the one re-used now)
        ready = torch.cuda.Event()
        ready.record()                                # after the forward that produced softmax_chw (current stream)
        softmax_chw.record_stream(self.stream)
        proc = self._procs[self._n % self.depth]
        self._n += 1
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            job = proc.enqueue(softmax_chw, *self.args)
        self.pending.append((t, softmax_chw, proc, job))
        return done

    def flush(self):
        out = []
        while self.pending:
            out += self._finish()
        return out


def stream_softmax(model, frames, data_format='NCHW', pre_sequence_frames=0, on_device=False, graph=False):
    """Yield (t, softmax [3,H,W]) for every real frame; warm-up frames are consumed silently.  on_device=True yields the
    device tensor (what the GPU post-processing consumes) instead of a host array.
    graph=True (GPU only; the model must be at the start of a sequence): the per-frame launch sequence is captured once
    into a hipGraph (lu_native.graph.GraphedFrame, bit-identical to the eager forward) and replayed per frame, which takes
    the ~70 host-side launches per frame off the host.  Measured (MI355X, B = 1, 256x256): no gain in fp32 (GPU-bound) and
    530 vs 551 frames/s in bf16 mode (the fixed state buffers cost eight small copies per frame) -- an option, off by default."""
    nchw = data_format[1] == 'C'
    replay = None
    for T, image in enumerate(frames):
        t = T - pre_sequence_frames
        image = np.asarray(image, np.float32)
        if image.ndim == 2:
            image = image.reshape((1, 1, 1) + image.shape) if nchw else image.reshape((1, 1) + image.shape + (1,))
        elif image.ndim == 3:
            image = image.reshape((1, 1) + image.shape)
        else:
            raise ValueError()
        if graph and replay is None:
            from lu_native.graph import GraphedFrame
            replay = GraphedFrame(model, image)
            replay.reset_states()                 # the capture's warm-up frames are not history
        if replay is not None and tuple(replay.x.shape) == image.shape:
            sm = replay(image)[1].clone()         # the graph owns its output buffer: the next replay overwrites it
        else:
            _, sm = model(image, training=False)
        if t < 0:
            continue
        sm = sm[0, 0] if nchw else sm[0, 0].permute(2, 0, 1).contiguous()
        yield t, (sm if on_device else sm.cpu().numpy())


def resolve_resize(model_dict, asked=None):
    """The bilinear convention the saved weights were trained under: model_params.pickle records it since round 4.
    A pickle without it comes from the reference itself (or an older build): 'tf2.0' -- the TensorFlow release the reference
    pins -- is assumed, loudly, because a model trained under the other convention runs without any error and segments worse.
    `asked` (--resize) overrides; a contradiction with the record is a warning, not an error."""
    recorded = model_dict.get('resize')
    if recorded is None and asked is None:
        log_print("WARNING: model_params.pickle does not record the bilinear resize convention (written by the reference or an "
                  "older build): assuming 'tf2.0' (TensorFlow 2.0 / 2.1); pass --resize half_pixel for a model trained "
                  "under a later TensorFlow")
        return 'tf2.0'
    if recorded is not None and asked is not None and asked != recorded:
        log_print("WARNING: the model was trained with resize='{}', running it with --resize {}".format(recorded, asked))
    return asked if asked is not None else recorded


def inference(params):
    select_gpu(getattr(params, 'gpu_id', None))
    with open(os.path.join(params.model_path, 'model_params.pickle'), 'rb') as fobj:
        model_dict = pickle.load(fobj)
    model_cls = get_model(model_dict['name'])
    model = model_cls(*model_dict['params'], data_format=params.data_format, pad_image=True,
                      precision=getattr(params, 'precision', 'fp32'),
                      resize=resolve_resize(model_dict, getattr(params, 'resize', None)))
    model.load_weights(os.path.join(params.model_path, 'model.ckpt'))
    log_print('Restored from {}'.format(os.path.join(params.model_path, 'model.ckpt')))
    dataset = params.data_reader(params.sequence_path, params.filename_format,
                                 pre_sequence_frames=params.pre_sequence_frames).dataset
    # The forward AND the post-processing run on the GPU; only the uint16 label map (and, with --save_intermediate, the
    # softmax) comes back.  File output runs on worker threads so that TIFF encoding never stalls the device.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=int(getattr(params, 'num_post_threads', 4)))

    def write(t, labels, sm):
        out_fname = os.path.join(params.output_path, 'mask{time:03d}.tif'.format(time=t))
        if sm is not None:        # Inference2D.py:104-108: HWC softmax scaled to 16 bit; cv2.imwrite of the channel-flipped array
            vis = np.round(np.transpose(sm, (1, 2, 0)) * (2 ** 16 - 1)).astype(np.uint16)      # stores R,G,B = bg, cell, edge
            write_tiff16(os.path.join(params.save_intermediate_vis_path, 'softmax{time:03d}.tif'.format(time=t)), vis)
        write_tiff16(out_fname, labels)
        log_print('Saved File: {}'.format(out_fname))
        if sm is not None:
            write_tiff16(os.path.join(params.save_intermediate_label_path, 'mask{time:03d}.tif'.format(time=t)), labels)

    pending = []
    pipe = PostPipeline(params.edge_dist, params.min_cell_size, params.max_cell_size, params.FOV,
                        bool(getattr(params, 'fov_fix', False)))

    def emit(done):
        for (t_done, labels, sm_done) in done:
            pending.append(pool.submit(write, t_done, labels, sm_done.cpu().numpy() if params.save_intermediate else None))
            while len(pending) > 16:          # bounded backlog
                pending.pop(0).result()

    try:
        import Networks
        use_graph = Networks._device().type == 'cuda' and bool(getattr(params, 'graph', False))
        for t, sm in stream_softmax(model, dataset, params.data_format, params.pre_sequence_frames, on_device=True,
                                    graph=use_graph):
            if params.dry_run:
                continue
            emit(pipe.push(t, sm))            # labels of frame t - 2, computed while the next forwards run
        emit(pipe.flush())
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
    (('--precision',), dict(dest='precision', choices=['fp32', 'bf16', 'bf16x3'],
                            help='[MI355X] fp32 (default) or bf16 MFMA operands for the wide convolutions')),
    (('--resize',), dict(dest='resize', choices=['tf2.0', 'half_pixel'],
                         help='[MI355X] bilinear convention of the up blocks; default: what model_params.pickle recorded')),
    (('--graph',), dict(dest='graph', action='store_const', const=True,
                        help='[MI355X] replay the per-frame launch sequence from a captured hipGraph (bit-identical; measured '
                             'neutral to slightly slower: the frame is bound by the GPU, not by the host launches)')),
    (('--fov_fix',), dict(dest='fov_fix', action='store_const', const=True,
                          help='[MI355X] mask columns [0, FOV) instead of the reference\'s single column FOV (Inference2D.py:97)')),
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
