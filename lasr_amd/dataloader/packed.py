"""A sequence's frame pairs packed for a one-launch batch gather (input side of row a1).

PackedTable holds, per distinct frame pair, the model's batch dictionary of that pair
(/root/reference/nnutils/train_utils.py:164-178: 15 keys, fp32) as ONE row of a [pairs, W] tensor in HBM.  A batch of B pairs is
a single launch of lasr_gather_rows into a persistent buffer; the dictionary handed to the model is a set of views into that
buffer, in the interleaved layout of train_utils.py:179-180.  Because the buffer is the same every iteration, the trainer's
HIP graph reads its inputs from it directly -- no per-key index_select, no per-key copy into static graph inputs (that was
39 launches per iteration, tools/step_sequence.py).
"""
import ctypes

import torch

from .. import _lib


class PackedBatch(dict):
    """dict of views into PackedTable's persistent output buffer; `persistent` tells the trainer the storage never moves."""
    persistent = True


class PackedTable:
    def __init__(self, rows, device):
        """rows: list (one per distinct pair) of dicts key -> tensor [2, ...] (frame t, frame t'), all with the same keys/shapes."""
        keys = list(rows[0].keys())
        if len(keys) > 24:
            raise ValueError('at most 24 keys (LASR_GATHER_MAX_KEYS)')
        self.keys, self.device, self.n_pairs = keys, torch.device(device), len(rows)
        self.shapes = {k: tuple(rows[0][k].shape) for k in keys}              # (2, ...)
        off, self.seg_off, self.seg_len = 0, [], []
        for k in keys:
            n = int(rows[0][k].numel())
            self.seg_off.append(off)
            self.seg_len.append(n)
            off += (n + 3) // 4 * 4                                           # 16-byte aligned segments
        self.W = off
        table = torch.zeros(len(rows), self.W, dtype=torch.float32)
        for r, row in enumerate(rows):
            for k, o, n in zip(keys, self.seg_off, self.seg_len):
                table[r, o:o + n] = row[k].detach().reshape(-1).float().cpu()
        self.table = table.to(self.device)
        self._B = None

    def _alloc(self, B):
        self._B, off, self.out_off = B, 0, []
        for n in self.seg_len:
            self.out_off.append(off)
            off += (B * n + 3) // 4 * 4
        self.out = torch.zeros(off, dtype=torch.float32, device=self.device)
        views = PackedBatch()
        for k, o, n in zip(self.keys, self.out_off, self.seg_len):
            views[k] = self.out[o:o + B * n].view(2 * B, *self.shapes[k][1:])   # [B, 2, ...] flattened: pair-major
        self.views = views
        arr = ctypes.c_longlong * len(self.keys)
        self._c = (arr(*self.seg_off), arr(*self.seg_len), arr(*self.out_off))

    def gather(self, ids, clone=False):
        """ids: int64 tensor [B] on the table's device -> the batch dictionary.

        ALIASING CONTRACT: by default every call returns the SAME PackedBatch of views into one persistent buffer (that is what
        lets the trainer's HIP graph read its inputs in place): the previous batch's tensors are overwritten by this call.  A
        caller that holds two batches at once (prefetching, comparing, accumulating over micro-batches) passes clone=True and
        gets an independent copy (a plain dict, `persistent` False).  A change of B re-allocates the buffer: views of the old
        size keep the old storage."""
        out = self._gather(ids)
        if clone:
            return {k: v.clone() for k, v in out.items()}
        return out

    def _gather(self, ids):
        B = int(ids.shape[0])
        if self._B != B:
            self._alloc(B)
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        if self.device.type != 'cuda':              # host-side logic on CPU tensors (layout tests); on a GPU always the kernel
            sel = self.table.index_select(0, ids)
            for o, n, oo in zip(self.seg_off, self.seg_len, self.out_off):
                self.out[oo:oo + B * n] = sel[:, o:o + n].reshape(-1)
            return self.views
        guard, st = _lib.stream_of(self.table)
        with guard:
            rc = _lib.lib().lasr_gather_rows(self.table.data_ptr(), self.W, self.n_pairs, ids.data_ptr(), B, len(self.keys),
                                             self._c[0], self._c[1], self._c[2], self.out.data_ptr(), st)
        _lib.check(rc, 'lasr_gather_rows')
        return self.views
