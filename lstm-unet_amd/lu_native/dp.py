"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed with the
"nccl" backend (= RCCL over xGMI on ROCm).  Independent clip slots shard across ranks;
the only data-path exchange per step is the gradient all-reduce, issued per bucket while backward
is still running (buckets are contiguous ranges of the flat gradient buffer in backward-completion
order).  Small scalar/vector reductions (loss sums, SyncBN statistics) go through all_reduce_.

The module is device-agnostic plumbing (CPU tensors + "gloo" are used by the world_size-2 tests).
"""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, backend=None, bucket_bytes=64 << 20, force=None):
        """force (or LU_DP_FORCE=1): a world of ONE still initialises the process group and sends every gradient bucket, the
        loss sums and the SyncBN statistics through the backend's all-reduce (RCCL on a GPU): a 1-GPU box then exercises
        everything of the multi-GPU step except the wire -- stream ordering between the compute stream, the weight-gradient
        side stream and RCCL's stream, the async handles, the flat-buffer slicing."""
        self.rank = int(os.environ.get('RANK', '0'))
        self.world_size = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.bucket_bytes = bucket_bytes
        self._pending = []
        self.launched = 0    # collectives issued so far (tests / logs)
        self._carry = None   # (start, end) range waiting to reach bucket size
        self.last_ranges = []  # bucket ranges of the most recent step (bench.py times all-reduces of these sizes)
        self._ranges = []
        self.flat = None
        # trace = [] switches per-bucket time stamps on (bench.py's overlap proof): device events on the compute stream when a
        # bucket is handed to the collective layer, when backward has ended (finish()) and when each all-reduce has been waited for
        self.trace = None
        self._trace_cur = []
        if force is None:
            force = os.environ.get('LU_DP_FORCE', '0') not in ('', '0')
        self.collectives = self.world_size > 1 or bool(force)      # False: every call below is a no-op (a plain single-process step)
        if self.collectives and not dist.is_initialized():
            if backend is None:
                # LU_DP_BACKEND=gloo lets several ranks share ONE GPU (control-flow checks on a 1-GPU box)
                backend = os.environ.get('LU_DP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)

    # -- small synchronous-on-stream reductions ------------------------------------------
    def all_reduce_(self, t):
        if self.collectives:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def broadcast_(self, t, src=0):
        if self.collectives:
            dist.broadcast(t, src)
        return t

    def barrier(self):
        if self.collectives:
            dist.barrier()

    # -- bucketed gradient all-reduce, overlapped with backward -----------------------------
    def attach(self, flat_grads):
        self.flat = flat_grads

    def bucket_ready(self, start, end):
        """Called by the engine as soon as flat[start:end] holds final local gradients."""
        if not self.collectives:
            return
        # adjacent ranges grow one bucket, whichever way the caller walks the flat buffer (the engine lays its parameters out
        # in backward-completion order, i.e. ascending): fewer, larger collectives -- 3 per step at Params.py widths
        if self._carry is not None and self._carry[1] == start:
            start = self._carry[0]
        elif self._carry is not None and self._carry[0] == end:
            end = self._carry[1]
        elif self._carry is not None:
            self._launch(*self._carry)
        self._carry = (start, end)
        if (end - start) * 4 >= self.bucket_bytes:
            self._launch(start, end)
            self._carry = None

    @classmethod
    def solo(cls):
        """A world of one inside a multi-rank process (bench.py --check: the single-process reference step): no collectives."""
        self = cls.__new__(cls)
        self.rank, self.world_size, self.local_rank, self.collectives = 0, 1, 0, False
        self.bucket_bytes, self._pending, self.launched, self._carry = 64 << 20, [], 0, None
        self.last_ranges, self._ranges, self.flat, self.trace, self._trace_cur = [], [], None, None, []
        return self

    def _stamp(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _launch(self, start, end):
        self.launched += 1
        self._ranges.append((start, end))
        tracing = self.trace is not None and self.flat.is_cuda
        if tracing:
            self._trace_cur.append({'bytes': 4 * (end - start), 'issue': self._stamp()})
        self._pending.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Flush the tail bucket and make the current stream wait for every outstanding all-reduce."""
        tracing = self.trace is not None and self.flat is not None and self.flat.is_cuda
        end_of_backward = self._stamp() if tracing else None
        if self._carry is not None:
            self._launch(*self._carry)
            self._carry = None
        for i, w in enumerate(self._pending):
            w.wait()
            if tracing:
                self._trace_cur[i]['done'] = self._stamp()
        self._pending = []
        if tracing:
            self.trace.append({'backward_end': end_of_backward, 'buckets': self._trace_cur})
            self._trace_cur = []
        if self._ranges:
            self.last_ranges, self._ranges = self._ranges, []

    def trace_report(self):
        """Per traced step, per bucket (after a device synchronize): how long before the end of backward the bucket was handed
        over (> 0: its all-reduce had that much of the remaining backward to hide behind), and how long after the end of
        backward the compute stream could continue past it (the EXPOSED part; the last bucket's value is the step's)."""
        out = []
        for st in self.trace or []:
            e0 = st['backward_end']
            out.append([{'bytes': b['bytes'], 'issued_ms_before_backward_end': round(b['issue'].elapsed_time(e0), 3),
                         'done_ms_after_backward_end': round(e0.elapsed_time(b['done']), 3)} for b in st['buckets']])
        return out

    def shard_slots(self, n_slots):
        """Batch slots owned by this rank: [rank*n/W, (rank+1)*n/W)."""
        per = n_slots // self.world_size
        return range(self.rank * per, (self.rank + 1) * per)
