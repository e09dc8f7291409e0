"""Mesh regularisers with the reference's module API, on sparse adjacency.

ARAPLoss      /root/reference/nnutils/loss_utils.py:29-64                 (dense [N,V,V] x 6 in the reference)
LaplacianLoss /root/reference/third_party/ext_nnutils/loss_utils.py:34-65 (dense [V,V] matmul in the reference)
FlattenLoss   /root/reference/third_party/ext_nnutils/loss_utils.py:67-152 (edge based; ~60 eager launches there)
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib
from . import fused_ops


def adjacency_csr(faces, nv):
    """Unique, symmetric vertex adjacency of a triangle list as CSR (row_ptr [V+1], col [nnz]) int32 arrays."""
    f = np.asarray(faces.detach().cpu().numpy() if torch.is_tensor(faces) else faces).astype(np.int64).reshape(-1, 3)
    a = np.concatenate([f[:, [0, 1]], f[:, [1, 0]], f[:, [1, 2]], f[:, [2, 1]], f[:, [2, 0]], f[:, [0, 2]]], 0)
    a = a[a[:, 0] != a[:, 1]]
    key = np.unique(a[:, 0] * nv + a[:, 1])
    rows, cols = key // nv, key % nv
    row_ptr = np.zeros(nv + 1, np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return np.cumsum(row_ptr).astype(np.int32), cols.astype(np.int32)


class _ARAP(Function):
    @staticmethod
    def forward(ctx, dx, x, row_ptr, col):
        _lib.need_cuda(dx, x, row_ptr, col)
        N, V = x.shape[:2]
        dx, x = dx.contiguous().float(), x.contiguous().float()
        loss = torch.empty(N, dtype=torch.float32, device=x.device)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_arap_forward(dx.data_ptr(), x.data_ptr(), row_ptr.data_ptr(), col.data_ptr(),
                                              loss.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_arap_forward')
        ctx.save_for_backward(dx, x, row_ptr, col)
        return loss

    @staticmethod
    def backward(ctx, g):
        dx, x, row_ptr, col = ctx.saved_tensors
        N, V = x.shape[:2]
        g = g.contiguous().float()
        gdx, gx = torch.empty_like(dx), torch.empty_like(x)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_arap_backward(dx.data_ptr(), x.data_ptr(), row_ptr.data_ptr(), col.data_ptr(),
                                               g.data_ptr(), gdx.data_ptr(), gx.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_arap_backward')
        return gdx, gx, None, None


class _Laplacian(Function):
    @staticmethod
    def forward(ctx, x, row_ptr, col):
        _lib.need_cuda(x, row_ptr, col)
        N, V = x.shape[:2]
        x = x.contiguous().float()
        loss = torch.empty(N, dtype=torch.float32, device=x.device)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_laplacian_forward(x.data_ptr(), row_ptr.data_ptr(), col.data_ptr(), loss.data_ptr(),
                                                   N, V, st)
        _lib.check(rc, 'lasr_laplacian_forward')
        ctx.save_for_backward(x, row_ptr, col)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, row_ptr, col = ctx.saved_tensors
        N, V = x.shape[:2]
        g = g.contiguous().float()
        gx, lx = torch.empty_like(x), torch.empty_like(x)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_laplacian_backward(x.data_ptr(), row_ptr.data_ptr(), col.data_ptr(), g.data_ptr(),
                                                    gx.data_ptr(), lx.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_laplacian_backward')
        return gx, None, None


class _SparseMeshLoss(nn.Module):
    def __init__(self, vertex, faces, average=False):
        super().__init__()
        self.nv, self.nf, self.average = vertex.size(0), faces.size(0), average
        row_ptr, col = adjacency_csr(faces, self.nv)
        self.register_buffer('row_ptr', torch.from_numpy(row_ptr))
        self.register_buffer('col', torch.from_numpy(col))


class ARAPLoss(_SparseMeshLoss):
    """forward(dx, x) -> [N]: mean over directed edges of | |x_b - x_a|^2 - |dx_b - dx_a|^2 |."""

    def forward(self, dx, x):
        return _ARAP.apply(dx, x, self.row_ptr.to(x.device), self.col.to(x.device))


class LaplacianLoss(_SparseMeshLoss):
    """forward(x) -> [N] (or their mean over the batch with average=True): sum_v |x_v - mean_nbr x|^2."""

    def forward(self, x):
        out = _Laplacian.apply(x, self.row_ptr.to(x.device), self.col.to(x.device))
        return out.sum() / x.size(0) if self.average else out


class FlattenLoss(nn.Module):
    """Dihedral-angle smoothness over interior edges (ext_nnutils/loss_utils.py:67-152)."""

    def __init__(self, faces, average=False):
        super().__init__()
        self.nf, self.average = faces.size(0), average
        f = faces.detach().cpu().numpy().astype(np.int64)
        # The reference only enumerates the (v0,v1) and (v1,v2) edges of every face (:72); an edge that is the
        # (v2,v0) edge of both its faces is therefore not regularised.  Same edge set here; for each edge the
        # opposite vertices come from the faces containing it, in face-index order (:87-101).
        opp, listed = {}, set()
        for t in f:
            for a, b, c in ((t[0], t[1], t[2]), (t[1], t[2], t[0]), (t[2], t[0], t[1])):
                opp.setdefault((min(a, b), max(a, b)), []).append(c)
            listed.add((min(t[0], t[1]), max(t[0], t[1])))
            listed.add((min(t[1], t[2]), max(t[1], t[2])))
        quads = np.array([[e[0], e[1], opp[e][0], opp[e][1]] for e in sorted(listed) if len(opp[e]) >= 2],
                         np.int64).reshape(-1, 4)
        for i, name in enumerate(('v0s', 'v1s', 'v2s', 'v3s')):
            self.register_buffer(name, torch.from_numpy(quads[:, i].copy()))
        # for the HIP kernels: the same quads as int32 and, for the backward gather, the (edge, slot) pairs touching
        # each vertex in ascending order
        nv = int(f.max()) + 1 if f.size else 0
        flat = quads.reshape(-1)
        order = np.argsort(flat, kind='stable')
        ptr = np.zeros(nv + 1, np.int64)
        np.add.at(ptr, flat + 1, 1)
        self.register_buffer('quads', torch.from_numpy(quads.astype(np.int32)), persistent=False)
        self.register_buffer('inc_ptr', torch.from_numpy(np.cumsum(ptr).astype(np.int32)), persistent=False)
        self.register_buffer('inc', torch.from_numpy(order.astype(np.int32)), persistent=False)

    @staticmethod
    def _rejection(a, b, eps):
        al2 = a.pow(2).sum(-1)
        bl1 = (b.pow(2).sum(-1) + eps).sqrt()
        ab = (a * b).sum(-1)
        cos = ab / ((al2 + eps).sqrt() * bl1 + eps)
        sin = (1 - cos.pow(2) + eps).sqrt()
        return b - a * (ab / (al2 + eps))[:, :, None], bl1 * sin

    def forward(self, vertices, eps=1e-6):
        if vertices.is_cuda and eps == 1e-6:                              # one HIP kernel forward, two backward
            dev, ptr = vertices.device, self.inc_ptr
            if ptr.numel() < vertices.shape[1] + 1:                       # trailing vertices no face refers to
                ptr = torch.cat([ptr, ptr[-1:].repeat(vertices.shape[1] + 1 - ptr.numel())])
            loss = fused_ops.flatten_loss(vertices, self.quads.to(dev), ptr.to(dev), self.inc.to(dev))
            return loss.sum() / vertices.size(0) if self.average else loss
        v0, v1 = vertices[:, self.v0s], vertices[:, self.v1s]
        a = v1 - v0
        cb1, n1 = self._rejection(a, vertices[:, self.v2s] - v0, eps)
        cb2, n2 = self._rejection(a, vertices[:, self.v3s] - v0, eps)
        cos = (cb1 * cb2).sum(-1) / (n1 * n2 + eps)
        loss = (cos + 1).pow(2).sum(tuple(range(1, cos.ndimension())))
        return loss.sum() / vertices.size(0) if self.average else loss
