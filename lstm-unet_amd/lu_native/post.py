"""Host sequencing of the device post-processing (include/lstm_unet_hip.h `lu_post_*`, csrc/lu_postprocess.hip):
softmax [3,H,W] on the device -> uint16 instance labels, reference Inference2D.py:66-123.

The frame is DEVICE-DRIVEN: labelling, per-label statistics, the hole filling of every object (`lu_post_fill_all`: one
workgroup per object with holes, found from the bit-quad Euler numbers, holes = components - Euler number), the field-of-view
presence table, the size filter + consecutive numbering (`lu_post_newid`) and the relabelling are enqueued back to back with
no host decision in between, and ONE device -> host copy brings back the uint16 map together with three words {label count,
dirty, oversize}.  `enqueue()` returns at once (the copy lands in pinned memory behind an event); `collect()` waits for it.

The reference's per-object loop (`for n in range(1, num_cells)`, :80-91) has an additive quirk -- a hole pixel that already
carries label m becomes m + n -- which makes its label ORDER matter when objects are nested.  The concurrent fill reports that
(`dirty`); the library then restores the snapshot of the map and replays the reference's strict order with ONE workgroup, still
on the device and inside the same call (`replayed`).  Only a crop too large for the kernels' LDS staging (`oversize`) -- or a
`stages` run, which keeps the step-by-step form -- makes collect() replay the frame object by object from the host
(`lu_post_fill_object`).
No CPU arithmetic path: device tensors in, device kernels."""
import numpy as np
import torch

from . import calls, ops


class _Job(object):
    __slots__ = ('args', 'event', 'stages', 'H', 'W')


class PostProcessor(object):
    """A frame is ONE foreign call (lu_post_frame: ~35 short kernels and copies enqueued by the library itself).
    graph=True (GPU, optional): that launch sequence is captured once per (frame size, parameters) into a hipGraph and replayed.
    Measured SLOWER on MI355X / ROCm 7.2 (fp32 streaming with post-processing 96.7 vs 121 frames/s, bf16 307 vs 433: the
    graph's ~35 nodes cost more to launch than the eager calls and do not overlap the forward stream as well) -- kept as an
    option and as a test of capturability, off by default."""

    def __init__(self, graph=False):
        self._shape = None
        self.fallbacks = 0       # frames replayed object by object from the HOST (oversize crops; `stages` runs)
        self.device_replays = 0  # frames with nested objects, replayed in label order by ONE workgroup on the device
        self.use_graph = bool(graph)
        self._graphs = {}        # (H, W, parameters) -> (CUDAGraph, static softmax buffer) | None when capture failed
        self.replays = 0
        self.library_copy = True     # the device -> host copy of the frame is issued by lu_post_frame itself

    def _alloc(self, H, W, dev):
        if self._shape == (H, W, dev):
            return
        lib = ops.lib()
        self._shape = (H, W, dev)
        self.nmax = int(lib.lu_post_max_labels(H, W))
        self.ws = torch.empty(int(lib.lu_post_workspace_bytes(H, W)) // 4 + 4, dtype=torch.int32, device=dev)
        self.labels = torch.empty((H, W), dtype=torch.int32, device=dev)
        self.snapshot = torch.empty((H, W), dtype=torch.int32, device=dev)
        # per-label tables: num_labels | dirty | oversize | pad | area | bbox | e4 | ncomp | present
        n = self.nmax
        self.small = torch.zeros(4 + 8 * n, dtype=torch.int32, device=dev)
        self.off = {'num': 0, 'dirty': 1, 'area': 4, 'bbox': 4 + n, 'e4': 4 + 5 * n, 'ncomp': 4 + 6 * n, 'present': 4 + 7 * n}
        self.newid = torch.zeros(n, dtype=torch.int32, device=dev)
        self.box = torch.zeros(4, dtype=torch.int32, device=dev)
        # the frame's result: uint16 map, then (4-byte aligned) {label count, dirty, oversize, 0}
        self.map_words = (H * W + 1) // 2
        self.out = torch.zeros(self.map_words + 4, dtype=torch.int32, device=dev)
        self.host = torch.zeros(self.map_words + 4, dtype=torch.int32)
        if dev.type == 'cuda':
            self.host = self.host.pin_memory()

    def _p(self, name):
        return self.small.data_ptr() + 4 * self.off[name]

    # ------------------------------------------------------------------------------------------------------------------
    def enqueue(self, softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, stages=None):
        """Launch the whole frame on the current stream; no host synchronisation (unless `stages` asks for intermediate maps)."""
        sm = softmax_chw
        if sm.dim() != 3 or sm.shape[0] != 3:
            raise ValueError('expected a [3, H, W] softmax, got %s' % (tuple(sm.shape),))
        ops._chk(sm)
        sm = sm.contiguous()
        H, W = int(sm.shape[1]), int(sm.shape[2])
        self._alloc(H, W, sm.device)
        job = _Job()
        job.args = (min_cell_size, max_cell_size, fov, fov_fix)
        job.stages, job.H, job.W = stages, H, W
        if self.use_graph and stages is None and sm.device.type == 'cuda':
            key = (H, W, float(edge_dist), min_cell_size, max_cell_size, fov, bool(fov_fix))
            if key not in self._graphs:
                self._graphs[key] = self._capture(sm, edge_dist, job)
            hit = self._graphs[key]
            if hit is not None:
                graph, static_sm = hit
                static_sm.copy_(sm, non_blocking=True)
                graph.replay()
                self.replays += 1
                job.event = torch.cuda.Event()
                job.event.record()
                return job
        self._body(sm, edge_dist, job)
        self._record(job)
        return job

    def _capture(self, sm, edge_dist, job):
        """One eager frame (first-use work: kernel attributes, allocations), then the same sequence into a hipGraph reading a
        static copy of the softmax.  -> (graph, static softmax) or None (capture unavailable: the eager path stays)."""
        static_sm = sm.clone()
        self._body(static_sm, edge_dist, job)
        torch.cuda.current_stream().synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            cur = torch.cuda.current_stream()
            cap = torch.cuda.Stream()
            cap.wait_stream(cur)
            with torch.cuda.graph(graph, stream=cap):
                self._body(static_sm, edge_dist, job)
            cur.wait_stream(cap)
            return graph, static_sm
        except Exception:      # noqa: BLE001 -- e.g. a runtime without capturable device -> pinned-host copies
            torch.cuda.synchronize()
            return None

    def _body(self, sm, edge_dist, job):
        lib, st, ck = ops.lib(), ops._stream(), calls.check
        H, W, stages = job.H, job.W, job.stages
        ws, L = self.ws.data_ptr(), self.labels.data_ptr()
        if stages is None:       # the whole frame in ONE foreign call (the step-by-step form below serves `stages`)
            min_cell_size, max_cell_size, fov, fov_fix = job.args
            big = 2 ** 31 - 1
            ck(lib, lib.lu_post_frame(sm.data_ptr(), H, W, 0.2, float(edge_dist), int(min(max(min_cell_size, -big), big)),
                                      int(min(max_cell_size, big)), int(fov), 0 if fov_fix else 1, ws, L, self.snapshot.data_ptr(),
                                      self.small.data_ptr(), self.newid.data_ptr(), self.out.data_ptr(),
                                      self.host.data_ptr() if self.library_copy else 0, st), 'lu_post_frame')
            if not self.library_copy:
                self.host.copy_(self.out, non_blocking=True)
            return
        self.small[:4].zero_()
        ck(lib, lib.lu_post_label(sm.data_ptr(), H, W, 0.2, float(edge_dist), ws, L, self._p('num'), self._p('area'), st),
           'lu_post_label')
        # sized by the BOUND on the label count: the count itself stays on the device
        ck(lib, lib.lu_post_label_stats(L, H, W, self.nmax, ws, self._p('bbox'), self._p('e4'), self._p('ncomp'), st),
           'lu_post_label_stats')
        self.snapshot.copy_(self.labels)
        if stages is not None:
            stages['absorbed'] = self.labels.cpu().numpy().copy()
        ck(lib, lib.lu_post_fill_all(L, H, W, self._p('num'), self._p('bbox'), self._p('e4'), self._p('ncomp'),
                                     self._p('dirty'), st), 'lu_post_fill_all')
        self._tail(job)

    def _record(self, job):
        if self.out.device.type == 'cuda':
            job.event = torch.cuda.Event()
            job.event.record()
        else:
            job.event = None

    def _tail(self, job):
        """FOV presence, numbering, relabel, and the one copy to the host."""
        lib, st, ck = ops.lib(), ops._stream(), calls.check
        H, W, L = job.H, job.W, self.labels.data_ptr()
        min_cell_size, max_cell_size, fov, fov_fix = job.args
        if job.stages is None:
            big = 2 ** 31 - 1
            ck(lib, lib.lu_post_frame_tail(H, W, int(min(max(min_cell_size, -big), big)), int(min(max_cell_size, big)), int(fov),
                                           0 if fov_fix else 1, L, self.small.data_ptr(), self.newid.data_ptr(), self.out.data_ptr(),
                                           self.host.data_ptr() if self.library_copy else 0, st), 'lu_post_frame_tail')
            if not self.library_copy:
                self.host.copy_(self.out, non_blocking=True)
            return
        present = 0
        if fov:
            ck(lib, lib.lu_post_present(L, H, W, int(fov), 0 if fov_fix else 1, self.nmax, self._p('present'), st),
               'lu_post_present')
            present = self._p('present')
        big = 2 ** 31 - 1
        ck(lib, lib.lu_post_newid(self._p('num'), self._p('area'), present, int(min(max(min_cell_size, -big), big)),
                                  int(min(max_cell_size, big)), self.nmax, self.newid.data_ptr(), self._p('dirty'),
                                  self.out.data_ptr() + 4 * self.map_words, st), 'lu_post_newid')
        ck(lib, lib.lu_post_relabel(L, H, W, self.newid.data_ptr(), self.nmax, self.out.data_ptr(), st), 'lu_post_relabel')
        self.host.copy_(self.out, non_blocking=True)

    def collect(self, job):
        """Wait for the frame enqueued as `job` (the most recent enqueue of THIS processor) -> numpy uint16 [H,W]."""
        if job.event is not None:
            job.event.synchronize()
        H, W = job.H, job.W
        tail = self.host[self.map_words:].numpy()
        num, dirty, oversize, replayed = int(tail[0]), int(tail[1]), int(tail[2]), int(tail[3])
        if num > self.nmax:
            raise calls.NativeError('label count %d exceeds the bound %d' % (num, self.nmax))
        self.device_replays += 1 if (replayed and not oversize) else 0
        if oversize or (dirty and not replayed):
            self.fallbacks += 1
            self._replay_in_reference_order(job, num)
            self._record(job)
            if job.event is not None:
                job.event.synchronize()
        if job.stages is not None:
            job.stages['filled'] = self.labels.cpu().numpy().copy()
            o = self.off['area']
            job.stages['areas'] = None if num <= 1 else self.small[o:o + num].cpu().numpy().astype(np.int64)
        flat = self.host[:self.map_words].numpy().view(np.uint16)[:H * W]
        return flat.reshape(H, W).copy()

    def __call__(self, softmax_chw, edge_dist=2, min_cell_size=10, max_cell_size=100, fov=0, fov_fix=False, stages=None):
        """softmax_chw: [3,H,W] float32 device tensor -> numpy uint16 [H,W].  stages (dict): receives intermediate label maps."""
        return self.collect(self.enqueue(softmax_chw, edge_dist, min_cell_size, max_cell_size, fov, fov_fix, stages))

    # ------------------------------------------------------------------------------------------------------------------
    def _replay_in_reference_order(self, job, num):
        """Nested objects: start again from the map before any fill and walk the labels as Inference2D.py:80-91 does.  Labels
        with holes (from the statistics) are filled in order until one reports the additive quirk; from there on every label
        is processed one by one from the CURRENT map (its pixel set may have changed)."""
        lib, st, ck = ops.lib(), ops._stream(), calls.check
        H, W = job.H, job.W
        ws, L = self.ws.data_ptr(), self.labels.data_ptr()
        self.labels.copy_(self.snapshot)
        self.small[1:3].zero_()
        o, n = self.off, self.nmax
        host = self.small.cpu().numpy()
        bbox = host[o['bbox']:o['bbox'] + 4 * num].reshape(num, 4)
        e4, ncomp = host[o['e4']:o['e4'] + num], host[o['ncomp']:o['ncomp'] + num]
        holes = ncomp - e4 // 4
        assert not np.any(e4[1:] % 4), 'bit-quad Euler count not a multiple of 4'
        sequential_from = None
        for v in [v for v in range(1, num) if holes[v] > 0]:
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
        self._tail(job)

    def _fill(self, lib, st, ws, L, H, W, v, x0, y0, x1, y1):
        cx0, cy0, cx1, cy1 = max(0, x0 - 1), max(0, y0 - 1), min(W - 1, x1 + 1), min(H - 1, y1 + 1)
        calls.check(lib, lib.lu_post_fill_object(L, H, W, v, cx0, cy0, cx1 - cx0 + 1, cy1 - cy0 + 1, ws, self._p('dirty'), st),
                    'lu_post_fill_object')
