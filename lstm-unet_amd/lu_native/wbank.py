"""Derived weight images (input-gradient kernels = flipped / transposed, bf16 MFMA-fragment packs) kept ACROSS steps and
refreshed by two launches after every optimiser step.

The reference never needs this (Keras layers read their variables, Networks.py:48-58); here each convolution streams a
prepared image of its kernel, and preparing ~80 of them one 8-10 us launch at a time was 0.7 ms of a 75 ms bf16 step
(0.25 ms of the fp32 one).  An image is built on first use by the single-operation entry point (lu_weight_flip_transpose /
lu_pack_weights_bf16) into a buffer the bank keeps; `refresh()` re-runs every recorded operation on the same buffers through
lu_weight_prep_batch: one launch for the flips, one for the packs (which include packs OF flips).  Sources are views of the
engine's flat parameter buffer (stable addresses), so a record stays valid for the life of the engine."""
import ctypes as C

import torch

from . import cabi, calls
from . import ops


class WeightBank(object):
    def __init__(self):
        self._flips = {}        # key -> (record fields, source tensor, image)
        self._packs = {}
        self._tables = None     # (device table of the flips | None, n, blocks), (... packs ...): rebuilt when a record is added
        self.refreshes = 0      # (tests) batch refreshes so far

    def __len__(self):
        return len(self._flips) + len(self._packs)

    @staticmethod
    def _key(w, *extra):
        return (w.data_ptr(), tuple(w.shape), tuple(w.stride())) + extra

    def flip(self, w, c_off=0, c_sub=None):
        """ops.flip_transpose(w, c_off, c_sub), remembered."""
        c_sub = w.shape[2] - c_off if c_sub is None else c_sub
        key = self._key(w, c_off, c_sub)
        hit = self._flips.get(key)
        if hit is None:
            wt = ops.flip_transpose(w, c_off, c_sub)
            k, _, Ctot, N = w.shape
            rec = dict(kind=0, k=k, src=w.data_ptr(), dst=wt.data_ptr(), C=c_sub, N=N, C_tot=Ctot, c_off=c_off,
                       nblk=k * k * (-(-c_sub // 32)) * (-(-N // 32)))
            hit = self._flips[key] = (rec, w, wt)
            self._tables = None
        return hit[2]

    def pack(self, w):
        """ops.pack_bf16(w), remembered."""
        key = self._key(w)
        hit = self._packs.get(key)
        if hit is None:
            pw = ops.pack_bf16(w)
            k, _, Cc, N = w.shape
            threads = k * k * (-(-Cc // 32)) * (-(-N // 32)) * 128        # one per 8 packed elements
            rec = dict(kind=1, k=k, src=w.data_ptr(), dst=pw.data.data_ptr(), tap_stride=w.stride(1), row_stride=w.stride(2),
                       kk=k * k, C=Cc, N=N, nblk=max(1, -(-threads // (256 * 4))))
            hit = self._packs[key] = (rec, w, pw)
            self._tables = None
        return hit[2]

    def flip_pack(self, w, c_off=0, c_sub=None):
        return self.pack(self.flip(w, c_off, c_sub))

    @staticmethod
    def _table(records, device):
        if not records:
            return None, 0, 0
        arr = (cabi.PrepOp * len(records))()
        blk = 0
        for i, (rec, _, _) in enumerate(records):
            o = arr[i]
            for name, val in rec.items():
                setattr(o, name, val)
            o.blk0 = blk
            blk += rec['nblk']
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(device), len(records), blk

    def refresh(self):
        """Recompute every image from the current parameters (call after the weights changed, before the next use)."""
        if not len(self):
            return
        if self._tables is None:
            dev = next(iter({**self._flips, **self._packs}.values()))[1].device
            self._tables = (self._table(list(self._flips.values()), dev), self._table(list(self._packs.values()), dev))
        lib = ops.lib()
        for table, n, blocks in self._tables:      # flips first: some packs read them
            if n:
                calls.check(lib, lib.lu_weight_prep_batch(table.data_ptr(), n, blocks, ops._stream()), 'lu_weight_prep_batch')
        self.refreshes += 1
