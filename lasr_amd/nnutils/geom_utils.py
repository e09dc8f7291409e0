"""Geometry helpers of the LASR forward pass, same signatures as /root/reference/nnutils/geom_utils.py:27-71,
running as fused HIP kernels (include/lasr_ops.h) instead of K-1 bmm launches and chains of elementwise ops."""
import torch
from torch.autograd import Function

from .. import _lib


class _LBS(Function):
    @staticmethod
    def forward(ctx, verts, Rmat, Tmat, skin, K, tocam):
        _lib.need_cuda(verts, Rmat, Tmat, skin)
        N, V = verts.shape[:2]
        # shapes are normalised by obj_to_cam() below (autograd then maps the gradients back)
        verts, Rmat, Tmat = verts.contiguous().float(), Rmat.contiguous().float(), Tmat.contiguous().float()
        if K > 1:
            skin = skin.contiguous().float()
        out = torch.empty(N, V, 3, dtype=torch.float32, device=verts.device)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = _lib.lib().lasr_lbs_forward(verts.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(),
                                             skin.data_ptr() if K > 1 else None, out.data_ptr(),
                                             N, V, K, 1 if tocam else 0, st)
        _lib.check(rc, 'lasr_lbs_forward')
        ctx.save_for_backward(verts, Rmat, Tmat, skin if K > 1 else verts.new_empty(0))
        ctx.meta = (N, V, K, tocam)
        return out

    @staticmethod
    def backward(ctx, gout):
        verts, Rmat, Tmat, skin = ctx.saved_tensors
        N, V, K, tocam = ctx.meta
        gout = gout.contiguous().float()
        # only what the caller differentiates: LASR's joint / control-point call passes detached transforms and a constant skin
        # (mesh_net.py:285-288), so its backward is the g_verts part alone -- no transposed contraction, no fold launch
        need_v, need_R, need_T, need_s = ctx.needs_input_grad[:4]
        want_rt = need_R or need_T
        gv = torch.empty_like(verts) if need_v else None
        gR = torch.empty_like(Rmat) if want_rt else None
        gT = torch.empty_like(Tmat) if want_rt else None
        gs = torch.empty(N, K - 1, V, dtype=torch.float32, device=verts.device) if (K > 1 and need_s) else None
        h = _lib.lib()
        scratch = torch.empty(h.lasr_lbs_backward_scratch_floats(N, V, K), dtype=torch.float32, device=verts.device) if want_rt else None
        ptr = lambda t: t.data_ptr() if t is not None else None                  # noqa: E731
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = h.lasr_lbs_backward(verts.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(),
                                     skin.data_ptr() if K > 1 else None, gout.data_ptr(), ptr(gv), ptr(gR), ptr(gT), ptr(gs),
                                     ptr(scratch), N, V, K, 1 if tocam else 0, st)
        _lib.check(rc, 'lasr_lbs_backward')
        return gv, gR if need_R else None, gT if need_T else None, gs, None, None


class _LBSBoth(Function):
    """One blend, two results: (camera-space vertices, blended vertices before the body transform)."""

    @staticmethod
    def forward(ctx, verts, Rmat, Tmat, skin, K):
        _lib.need_cuda(verts, Rmat, Tmat, skin)
        N, V = verts.shape[:2]
        verts, Rmat, Tmat = verts.contiguous().float(), Rmat.contiguous().float(), Tmat.contiguous().float()
        if K > 1:
            skin = skin.contiguous().float()
        out = torch.empty(2, N, V, 3, dtype=torch.float32, device=verts.device)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = _lib.lib().lasr_lbs_forward_both(verts.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(),
                                                  skin.data_ptr() if K > 1 else None, out[0].data_ptr(), out[1].data_ptr(),
                                                  N, V, K, st)
        _lib.check(rc, 'lasr_lbs_forward_both')
        ctx.save_for_backward(verts, Rmat, Tmat, skin if K > 1 else verts.new_empty(0))
        ctx.meta = (N, V, K)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, gcam, gblend):
        verts, Rmat, Tmat, skin = ctx.saved_tensors
        N, V, K = ctx.meta
        gcam, gblend = gcam.contiguous().float(), gblend.contiguous().float()
        gv, gR, gT = torch.empty_like(verts), torch.empty_like(Rmat), torch.empty_like(Tmat)
        gs = torch.empty(N, K - 1, V, dtype=torch.float32, device=verts.device) if K > 1 else None
        h = _lib.lib()
        scratch = torch.empty(h.lasr_lbs_backward_scratch_floats(N, V, K), dtype=torch.float32, device=verts.device)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = h.lasr_lbs_backward_both(verts.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(), skin.data_ptr() if K > 1 else None,
                                          gcam.data_ptr(), gblend.data_ptr(), gv.data_ptr(), gR.data_ptr(), gT.data_ptr(),
                                          gs.data_ptr() if K > 1 else None, scratch.data_ptr(), N, V, K, st)
        _lib.check(rc, 'lasr_lbs_backward_both')
        return gv, gR, gT, gs, None


def _lbs_args(verts, Rmat, Tmat, nmesh, skin):
    verts = verts.view(-1, verts.shape[1], 3)
    N = verts.shape[0]
    V, K = verts.shape[1], int(nmesh)
    Rm = Rmat.reshape(-1, 9)
    Tm = Tmat.reshape(-1, 3)
    if Rm.shape[0] != N * K or Tm.shape[0] != N * K:
        raise ValueError('Rmat/Tmat hold %d/%d transforms, expected %d meshes x %d bones'
                         % (Rm.shape[0], Tm.shape[0], N, K))
    sk = None
    if K > 1:
        sk = skin.reshape(skin.shape[0], K - 1, V)
        if sk.shape[0] != N:                             # e.g. the identity skin of the joints (mesh_net.py:285)
            sk = sk.expand(N, K - 1, V)
    return verts, Rm, Tm, sk, K


def obj_to_cam_both(verts, Rmat, Tmat, nmesh, n_hypo, skin):
    """(obj_to_cam(..., tocam=True), obj_to_cam(..., tocam=False)) from ONE blend launch and one backward launch: the two calls
    LASR.forward makes back to back on the same arguments (/root/reference/nnutils/mesh_net.py:291 and :298)."""
    if verts.device.type != 'cuda':
        return (obj_to_cam(verts, Rmat, Tmat, nmesh, n_hypo, skin), obj_to_cam(verts, Rmat, Tmat, nmesh, n_hypo, skin, tocam=False))
    verts, Rm, Tm, sk, K = _lbs_args(verts, Rmat, Tmat, nmesh, skin)
    return _LBSBoth.apply(verts, Rm, Tm, sk, K)


def obj_to_cam(verts, Rmat, Tmat, nmesh, n_hypo, skin, tocam=True):
    """Canonical object coordinates -> camera coordinates with linear-blend skinning (geom_utils.py:45-71).

    verts [N,V,3]; Rmat [N*nmesh,3,3] / Tmat [N*nmesh,1,3] with the body transform first, then nmesh-1 part
    bones, per mesh; skin [N or 1, nmesh-1, V, 1].  Gradients flow to all four tensor arguments.
    """
    verts = verts.view(-1, verts.shape[1], 3)
    N = verts.shape[0]
    V, K = verts.shape[1], int(nmesh)
    Rm = Rmat.reshape(-1, 9)
    Tm = Tmat.reshape(-1, 3)
    if Rm.shape[0] != N * K or Tm.shape[0] != N * K:
        raise ValueError('Rmat/Tmat hold %d/%d transforms, expected %d meshes x %d bones'
                         % (Rm.shape[0], Tm.shape[0], N, K))
    sk = None
    if K > 1:
        sk = skin.reshape(skin.shape[0], K - 1, V)
        if sk.shape[0] != N:                             # e.g. the identity skin of the joints (mesh_net.py:285)
            sk = sk.expand(N, K - 1, V)
    # the reshapes/expands above are autograd ops, so gradients come back in the callers' shapes
    return _LBS.apply(verts, Rm, Tm, sk, K, bool(tocam))


class _Pinhole(Function):
    @staticmethod
    def forward(ctx, verts, pp, fl):
        _lib.need_cuda(verts, pp, fl)
        N, V = verts.shape[:2]
        verts = verts.contiguous().float()
        pp = pp.contiguous().float()
        fl = fl.contiguous().float()
        out = torch.empty_like(verts)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = _lib.lib().lasr_pinhole_forward(verts.data_ptr(), pp.data_ptr(), fl.data_ptr(), out.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_pinhole_forward')
        ctx.save_for_backward(verts, pp, fl)
        return out

    @staticmethod
    def backward(ctx, gout):
        verts, pp, fl = ctx.saved_tensors
        N, V = verts.shape[:2]
        gout = gout.contiguous().float()
        gv, gpp, gfl = torch.empty_like(verts), torch.empty_like(pp), torch.empty_like(fl)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = _lib.lib().lasr_pinhole_backward(verts.data_ptr(), pp.data_ptr(), fl.data_ptr(), gout.data_ptr(),
                                                  gv.data_ptr(), gpp.data_ptr(), gfl.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_pinhole_backward')
        return gv, gpp, gfl


def pinhole_cam(verts, pp, fl):
    """x,y <- pp + (x,y) * fl / z for homogeneous [N,V,4] vertices (geom_utils.py:27-34).
    pp [2B,2] is shared by the n_hypo = N // 2B hypotheses of a frame; fl has N entries."""
    n_hypo = verts.shape[0] // pp.shape[0]
    pp = pp[:, None].expand(-1, n_hypo, -1).reshape(-1, 2)
    return _Pinhole.apply(verts, pp, fl.reshape(-1))


def orthographic_cam(verts, pp, fl):
    """geom_utils.py:36-43 (not on the LASR training path; plain torch)."""
    n_hypo = verts.shape[0] // pp.shape[0]
    pp = pp[:, None].expand(-1, n_hypo, -1).reshape(-1, 2)
    fl = fl.reshape(-1, 1)
    x = pp[:, 0:1] + verts[:, :, 0] * fl
    y = pp[:, 1:2] + verts[:, :, 1] * fl
    return torch.cat([x[..., None], y[..., None], verts[:, :, 2:]], -1)
