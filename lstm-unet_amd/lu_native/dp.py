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
    def __init__(self, backend=None, bucket_bytes=64 << 20):
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
        if self.world_size > 1 and not dist.is_initialized():
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
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def broadcast_(self, t, src=0):
        if self.world_size > 1:
            dist.broadcast(t, src)
        return t

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()

    # -- bucketed gradient all-reduce, overlapped with backward -----------------------------
    def attach(self, flat_grads):
        self.flat = flat_grads

    def bucket_ready(self, start, end):
        """Called by the engine as soon as flat[start:end] holds final local gradients."""
        if self.world_size == 1:
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

    def _launch(self, start, end):
        self.launched += 1
        self._ranges.append((start, end))
        self._pending.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Flush the tail bucket and make the current stream wait for every outstanding all-reduce."""
        if self._carry is not None:
            self._launch(*self._carry)
            self._carry = None
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._ranges:
            self.last_ranges, self._ranges = self._ranges, []

    def shard_slots(self, n_slots):
        """Batch slots owned by this rank: [rank*n/W, (rank+1)*n/W)."""
        per = n_slots // self.world_size
        return range(self.rank * per, (self.rank + 1) * per)
