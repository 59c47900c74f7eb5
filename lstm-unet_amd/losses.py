"""Loss and metric with the public surface of the reference's losses.py:
WeightedCELoss (losses.py:8-27) on the gfx950 softmax+CE kernels, seg_measure (losses.py:29-88) on
the host (numpy/scipy, exactly as the reference runs it through tf.py_function on /cpu:0)."""
import numpy as np
import torch
from scipy import ndimage

from lu_native import ops

__all__ = ['WeightedCELoss', 'seg_measure']


class _WCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_cl, gt, class_w, reducer):
        lg = logits_cl.contiguous().view(-1, 3)
        g = gt.contiguous().view(-1)
        sums, _ = ops.wce_forward(lg, g, class_w, False)
        if reducer is not None:      # DP: the reference normalises by the GLOBAL valid-pixel count
            reducer(sums)
        ctx.save_for_backward(lg, g, class_w, sums)
        ctx.shape = logits_cl.shape
        return ops.wce_loss(sums)[0]

    @staticmethod
    def backward(ctx, grad_out):
        lg, g, class_w, sums = ctx.saved_tensors
        dl = ops.wce_backward(lg, g, class_w, sums, 1.0)
        return (dl.view(ctx.shape) * grad_out), None, None, None


class WeightedCELoss(object):
    """loss = sum(ce * w[gt] * valid) / (sum(valid) + 1e-5), valid = gt > -1  (losses.py:13-27).
    channel_axis is the axis of the 5-D tensors holding classes: 2 for NCHW, 4 for NHWC."""

    def __init__(self, channel_axis, class_weights, reducer=None):
        self.channel_axis = channel_axis
        self.class_weights = class_weights
        self._reducer = reducer
        self._cw = None

    def _weights(self, device):
        if self._cw is None or self._cw.device != device:
            self._cw = torch.tensor(list(self.class_weights), dtype=torch.float32, device=device)
        return self._cw

    def __call__(self, gt_sequence, output_sequence):
        out = output_sequence
        if not torch.is_tensor(out) or not out.is_cuda:
            # host logits (numpy / CPU tensor, as a caller of the reference may hold them): moved to the device -- the
            # arithmetic still runs in the HIP kernels, there is no CPU path
            import Networks
            out = torch.as_tensor(np.asarray(out) if not torch.is_tensor(out) else out).to(device=Networks._device(),
                                                                                            dtype=torch.float32)
        ops._chk(out)
        gt = torch.as_tensor(np.asarray(gt_sequence) if not torch.is_tensor(gt_sequence) else gt_sequence)
        gt = gt.to(device=out.device, dtype=torch.float32).squeeze(self.channel_axis)
        if self.channel_axis == 2:
            out = out.permute(0, 1, 3, 4, 2)
        return _WCEFn.apply(out, gt, self._weights(out.device), self._reducer)


def seg_measure(channel_axis, three_d=False, foreground_class_index=1):
    """SEG: mean over ground-truth objects of IoU with the predicted object covering >50 % of it
    (4-connected components per frame).  Returns callable(gt_sequence, output_sequence) -> float.
    three_d (losses.py:33-36): sequences of VOLUMES [B, T, D, H, W] (+ the channel axis), components 6-connected inside each
    (b, t) volume -- the 3 x 3 x 3 element the reference stores at strel[1][1].  (The reference embeds it in a 5-D array and hands
    that to ndimage.label together with the 3-D volumes of its (b, t) loop, which scipy rejects -- "structure and input must
    have equal rank": its branch cannot run as written and no shipped model is 3-D; this is the evident intent.)"""
    if three_d:
        strel = ndimage.generate_binary_structure(3, 1)       # == the reference's strel[1][1]
    else:
        strel = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])

    def components(stack):
        lab = np.zeros(stack.shape, dtype=np.uint16)
        for b, frames in enumerate(stack):
            for t, frame in enumerate(frames):
                lab[b, t], _ = ndimage.label(frame, structure=strel)
        return lab

    def seg_numpy(gt_fg, out_fg):
        gl_all, sl_all = components(gt_fg), components(out_fg)
        ious = []
        for gl_b, sl_b in zip(gl_all, sl_all):
            for gl, sl in zip(gl_b, sl_b):
                for lab in np.unique(gl):
                    if lab == 0:
                        continue
                    ious.append(0.)
                    bw = gl == lab
                    area = np.sum(bw).astype(np.float32)
                    hit = sl[bw]
                    for s in np.unique(hit):
                        if s == 0:
                            continue
                        inter = np.sum(hit == s).astype(np.float32)
                        if inter / area > 0.5:
                            s_area = np.sum(sl == s).astype(np.float32)
                            ious[-1] = inter / (area + s_area - inter)
        if not len(ious):
            return np.nan
        return np.mean(ious)

    def calc_seg(gt_sequence, output_sequence):
        gt = gt_sequence.detach().cpu().numpy() if torch.is_tensor(gt_sequence) else np.asarray(gt_sequence)
        out = output_sequence.detach().cpu().numpy() if torch.is_tensor(output_sequence) else np.asarray(output_sequence)
        gt = np.squeeze(gt, channel_axis)
        valid = (gt > -1).astype(np.float32)
        gt_fg = (gt * valid).astype(np.float32) == foreground_class_index
        out_fg = np.argmax(out, axis=channel_axis) == foreground_class_index
        return np.float32(seg_numpy(gt_fg, out_fg))

    return calc_seg
