"""Host sequencing of the device post-processing (include/lstm_unet_hip.h `lu_post_*`, csrc/lu_postprocess.hip):
softmax [3,H,W] on the device -> uint16 instance labels, reference Inference2D.py:66-123.

Two small device -> host reads per frame (label count + per-label statistics in one, and the FOV presence flags; the label
map itself has to come back for the TIFF anyway).  Objects with holes are found for ALL labels at once from the bit-quad Euler numbers
(holes = components - Euler number); the reference's per-object loop (`for n in range(1, num_cells)`, :80-91) is then run
only over those, in label order.  Its additive quirk (a hole pixel that already carries label m becomes m + n) can change
which pixels later labels own; the fill kernel reports that (`dirty`) and the remaining labels are then processed strictly
one by one from the current map, exactly like the reference.  No CPU arithmetic path: device tensors in, device kernels."""
import numpy as np
import torch

from . import calls, ops


class PostProcessor(object):
    def __init__(self):
        self._shape = None

    def _alloc(self, H, W, dev):
        if self._shape == (H, W, dev):
            return
        lib = ops.lib()
        self._shape = (H, W, dev)
        self.nmax = int(lib.lu_post_max_labels(H, W))
        self.ws = torch.empty(int(lib.lu_post_workspace_bytes(H, W)) // 4 + 4, dtype=torch.int32, device=dev)
        self.labels = torch.empty((H, W), dtype=torch.int32, device=dev)
        # one buffer for everything the host reads: num_labels | dirty | area | bbox | e4 | ncomp | present
        n = self.nmax
        self.small = torch.zeros(2 + 8 * n, dtype=torch.int32, device=dev)
        self.off = {'num': 0, 'dirty': 1, 'area': 2, 'bbox': 2 + n, 'e4': 2 + 5 * n, 'ncomp': 2 + 6 * n, 'present': 2 + 7 * n}
        self.newid = torch.zeros(n, dtype=torch.int32, device=dev)
        self.box = torch.zeros(4, dtype=torch.int32, device=dev)
        self.out = torch.empty((H, W), dtype=torch.int16, device=dev)

    def _p(self, name):
        return self.small.data_ptr() + 4 * self.off[name]

    def __call__(self, softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, stages=None):
        """softmax_chw: [3,H,W] float32 device tensor -> numpy uint16 [H,W].  stages (dict): receives intermediate label maps."""
        sm = softmax_chw
        if sm.dim() != 3 or sm.shape[0] != 3:
            raise ValueError('expected a [3, H, W] softmax, got %s' % (tuple(sm.shape),))
        ops._chk(sm)
        sm = sm.contiguous()
        H, W = int(sm.shape[1]), int(sm.shape[2])
        self._alloc(H, W, sm.device)
        lib, st, ck = ops.lib(), ops._stream(), calls.check
        ws, L = self.ws.data_ptr(), self.labels.data_ptr()
        self.small[:2].zero_()
        ck(lib, lib.lu_post_label(sm.data_ptr(), H, W, 0.2, float(edge_dist), ws, L, self._p('num'), self._p('area'), st),
           'lu_post_label')
        # the statistics pass is sized by the BOUND on the label count (a few thousand idle table entries), so that the
        # count itself and the statistics come back in ONE device -> host read
        ck(lib, lib.lu_post_label_stats(L, H, W, self.nmax, ws, self._p('bbox'), self._p('e4'), self._p('ncomp'), st),
           'lu_post_label_stats')
        host = self.small.cpu().numpy()                     # sync 1
        num = int(host[0])
        if num > self.nmax:
            raise calls.NativeError('label count %d exceeds the bound %d' % (num, self.nmax))
        if stages is not None:
            stages['absorbed'] = self.labels.cpu().numpy().copy()
        areas = None
        if num > 1:
            o, n = self.off, self.nmax
            areas = host[o['area']:o['area'] + num].astype(np.int64)
            bbox = host[o['bbox']:o['bbox'] + 4 * num].reshape(num, 4)
            e4, ncomp = host[o['e4']:o['e4'] + num], host[o['ncomp']:o['ncomp'] + num]
            holes = ncomp - e4 // 4
            assert not np.any(e4[1:] % 4), 'bit-quad Euler count not a multiple of 4'
            todo = [v for v in range(1, num) if holes[v] > 0]
            sequential_from = None
            for v in todo:
                x0, y0, x1, y1 = [int(t) for t in bbox[v]]
                self._fill(lib, st, ws, L, H, W, v, x0, y0, x1, y1)
                if int(self.small[1].item()):               # a hole held another label: strict reference order from here on
                    sequential_from = v + 1
                    break
            if sequential_from is not None:
                for v in range(sequential_from, num):
                    ck(lib, lib.lu_post_bbox_of_label(L, H, W, v, self.box.data_ptr(), st), 'lu_post_bbox_of_label')
                    x0, y0, x1, y1 = [int(t) for t in self.box.cpu().numpy()]
                    if x1 < 0:
                        continue                            # `if not np.any(bw): continue`
                    self._fill(lib, st, ws, L, H, W, v, x0, y0, x1, y1)
        if stages is not None:
            stages['filled'] = self.labels.cpu().numpy().copy()
            stages['areas'] = None if areas is None else areas.copy()
        newid = np.zeros(self.nmax, np.int32)
        if num > 1:
            present = None
            if fov:
                ck(lib, lib.lu_post_present(L, H, W, int(fov), 0 if fov_fix else 1, num, self._p('present'), st),
                   'lu_post_present')
                o = self.off['present']
                present = self.small[o:o + num].cpu().numpy()       # sync 2
            p = 0
            for v in range(1, num):
                if min_cell_size <= areas[v] <= max_cell_size and (present is None or present[v]):
                    p += 1
                    newid[v] = p
        self.newid.copy_(torch.from_numpy(newid))
        ck(lib, lib.lu_post_relabel(L, H, W, self.newid.data_ptr(), max(num, 1), self.out.data_ptr(), st), 'lu_post_relabel')
        return self.out.cpu().numpy().view(np.uint16).copy()

    def _fill(self, lib, st, ws, L, H, W, v, x0, y0, x1, y1):
        cx0, cy0, cx1, cy1 = max(0, x0 - 1), max(0, y0 - 1), min(W - 1, x1 + 1), min(H - 1, y1 + 1)
        calls.check(lib, lib.lu_post_fill_object(L, H, W, v, cx0, cy0, cx1 - cx0 + 1, cy1 - cy0 + 1, ws, self._p('dirty'), st),
                    'lu_post_fill_object')
