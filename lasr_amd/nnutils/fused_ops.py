"""Launch-bound chains of LASR.forward as single HIP kernel pairs (include/lasr_ops.h, lasr_amd/csrc/fused.hip).

Each function replaces a run of eager elementwise ops of the reference (file:line in its docstring) and has a torch
restatement in oracle/path_oracle.py that the GPU tests compare against."""
import torch
from torch.autograd import Function

from .. import _lib


class _FlowReproject(Function):
    @staticmethod
    def forward(ctx, px, pp0, pp1, fl0, fl1):
        _lib.need_cuda(px, pp0, pp1, fl0, fl1)
        N = px.shape[0]
        P = px[0, 0].numel()
        px = px.contiguous().float()
        pp0, pp1 = pp0.contiguous().float(), pp1.contiguous().float()
        fl0, fl1 = fl0.contiguous().float(), fl1.contiguous().float()
        flow = torch.empty(px.shape[0], *px.shape[2:], 2, dtype=torch.float32, device=px.device)
        bg = torch.empty(px.shape[0], *px.shape[2:], dtype=torch.uint8, device=px.device)
        guard, st = _lib.stream_of(px)
        with guard:
            rc = _lib.lib().lasr_flow_reproject_forward(px.data_ptr(), pp0.data_ptr(), pp1.data_ptr(), fl0.data_ptr(),
                                                        fl1.data_ptr(), flow.data_ptr(), bg.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_forward')
        ctx.save_for_backward(px, fl1)
        bg = bg.view(torch.bool)
        ctx.mark_non_differentiable(bg)
        return flow, bg

    @staticmethod
    def backward(ctx, gflow, _gbg):
        px, fl1 = ctx.saved_tensors
        N = px.shape[0]
        P = px[0, 0].numel()
        gflow = gflow.contiguous().float()
        gpx = torch.empty_like(px)
        gpp1 = torch.empty(N, 2, dtype=torch.float32, device=px.device)
        gfl1 = torch.empty(N, dtype=torch.float32, device=px.device)
        h = _lib.lib()
        scratch = torch.empty(h.lasr_flow_reproject_scratch_floats(N, P), dtype=torch.float32, device=px.device)
        guard, st = _lib.stream_of(px)
        with guard:
            rc = h.lasr_flow_reproject_backward(px.data_ptr(), fl1.data_ptr(), gflow.data_ptr(), gpx.data_ptr(),
                                                gpp1.data_ptr(), gfl1.data_ptr(), scratch.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_backward')
        return gpx, None, gpp1, None, gfl1


def flow_reproject(px, pp0, pp1, fl0, fl1):
    """Tail of render_flow_soft_2 (/root/reference/nnutils/mesh_net.py:93-104): px [N,7,IS,IS] (position of frame t,
    of frame t', alpha), pp0/pp1 [N,2], fl0/fl1 [N] or [N,1] -> flow [N,IS,IS,2], bgmask [N,IS,IS] bool."""
    N = px.shape[0]
    return _FlowReproject.apply(px, pp0, pp1, fl0.reshape(N, -1)[:, 0], fl1.reshape(N, -1)[:, 0])


class _QuatToRotmat(Function):
    @staticmethod
    def forward(ctx, q):
        _lib.need_cuda(q)
        q = q.contiguous().float()
        M = q.shape[0]
        R = torch.empty(M, 9, dtype=torch.float32, device=q.device)
        guard, st = _lib.stream_of(q)
        with guard:
            rc = _lib.lib().lasr_quat_to_rotmat_forward(q.data_ptr(), R.data_ptr(), M, st)
        _lib.check(rc, 'lasr_quat_to_rotmat_forward')
        ctx.save_for_backward(q)
        return R

    @staticmethod
    def backward(ctx, gR):
        q, = ctx.saved_tensors
        gR = gR.contiguous().float()
        gq = torch.empty_like(q)
        guard, st = _lib.stream_of(q)
        with guard:
            rc = _lib.lib().lasr_quat_to_rotmat_backward(q.data_ptr(), gR.data_ptr(), gq.data_ptr(), q.shape[0], st)
        _lib.check(rc, 'lasr_quat_to_rotmat_backward')
        return gq


def quat_to_rotmat(q):
    """(x,y,z,w) quaternions [...,4] -> rotation matrices [...,3,3], normalising first (kornia 0.5.3
    quaternion_to_rotation_matrix as called at /root/reference/nnutils/mesh_net.py:232,250,265)."""
    return _QuatToRotmat.apply(q.reshape(-1, 4)).reshape(q.shape[:-1] + (3, 3))


class _SkinWeights(Function):
    @staticmethod
    def forward(ctx, ts, rs, lc, verts):
        _lib.need_cuda(ts, rs, lc, verts)
        H, V = verts.shape[:2]
        J = ts.shape[0] // H
        ts, rs, lc, verts = (t.contiguous().float() for t in (ts, rs, lc, verts))
        skin = torch.empty(H, J, V, dtype=torch.float32, device=ts.device)
        guard, st = _lib.stream_of(ts)
        with guard:
            rc = _lib.lib().lasr_skin_weights_forward(ts.data_ptr(), rs.data_ptr(), lc.data_ptr(), verts.data_ptr(),
                                                      skin.data_ptr(), H, J, V, st)
        _lib.check(rc, 'lasr_skin_weights_forward')
        ctx.save_for_backward(ts, rs, lc, verts, skin)
        return skin

    @staticmethod
    def backward(ctx, g):
        ts, rs, lc, verts, skin = ctx.saved_tensors
        H, J, V = skin.shape
        g = g.contiguous().float()
        gts, grs, glc = torch.empty_like(ts), torch.empty_like(rs), torch.empty_like(lc)
        scratch = torch.empty(H * V + 1, dtype=torch.float32, device=ts.device)
        guard, st = _lib.stream_of(ts)
        with guard:
            rc = _lib.lib().lasr_skin_weights_backward(ts.data_ptr(), rs.data_ptr(), lc.data_ptr(), verts.data_ptr(),
                                                       skin.data_ptr(), g.data_ptr(), gts.data_ptr(), grs.data_ptr(),
                                                       glc.data_ptr(), scratch.data_ptr(), H, J, V, st)
        _lib.check(rc, 'lasr_skin_weights_backward')
        return gts, grs, glc, None


def skin_weights(ctl_ts, ctl_rs, log_ctl, verts):
    """GMM skinning weights (/root/reference/nnutils/mesh_net.py:264-271): ctl_ts, log_ctl [H*J,3], ctl_rs [H*J,4],
    verts [H,V,3] (treated as a constant, :266) -> [H,J,V]."""
    return _SkinWeights.apply(ctl_ts, ctl_rs, log_ctl, verts.detach())


class _Flatten(Function):
    @staticmethod
    def forward(ctx, x, quads, inc_ptr, inc):
        _lib.need_cuda(x, quads, inc_ptr, inc)
        x = x.contiguous().float()
        N, V = x.shape[:2]
        E = quads.shape[0]
        loss = torch.empty(N, dtype=torch.float32, device=x.device)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_flatten_forward(x.data_ptr(), quads.data_ptr(), loss.data_ptr(), N, V, E, st)
        _lib.check(rc, 'lasr_flatten_forward')
        ctx.save_for_backward(x, quads, inc_ptr, inc)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, quads, inc_ptr, inc = ctx.saved_tensors
        N, V = x.shape[:2]
        E = quads.shape[0]
        g = g.contiguous().float()
        gx = torch.empty_like(x)
        scratch = torch.empty(N * E * 12 + 1, dtype=torch.float32, device=x.device)
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_flatten_backward(x.data_ptr(), quads.data_ptr(), inc_ptr.data_ptr(), inc.data_ptr(),
                                                  g.data_ptr(), gx.data_ptr(), scratch.data_ptr(), N, V, E, st)
        _lib.check(rc, 'lasr_flatten_backward')
        return gx, None, None, None


def flatten_loss(x, quads, inc_ptr, inc):
    """x [N,V,3], quads [E,4] int32, incidence CSR -> [N] (/root/reference/third_party/ext_nnutils/loss_utils.py:110-152)."""
    return _Flatten.apply(x, quads, inc_ptr, inc)


def nearest_point(a, b):
    """a [N,P,3], b [N,Q,3] -> (squared distance [N,P], index [N,P] int64) of the nearest b point; not differentiable
    (chamfer3D's dist1/idx1, /root/reference/nnutils/mesh_net.py:477)."""
    _lib.need_cuda(a, b)
    a, b = a.detach().contiguous().float(), b.detach().contiguous().float()
    N, P = a.shape[:2]
    d2 = torch.empty(N, P, dtype=torch.float32, device=a.device)
    idx = torch.empty(N, P, dtype=torch.int32, device=a.device)
    guard, st = _lib.stream_of(a)
    with guard:
        rc = _lib.lib().lasr_nearest_point(a.data_ptr(), b.data_ptr(), d2.data_ptr(), idx.data_ptr(), N, P, b.shape[1], st)
    _lib.check(rc, 'lasr_nearest_point')
    return d2, idx.long()


class _PointMesh(Function):
    @staticmethod
    def forward(ctx, verts, faces, points):
        _lib.need_cuda(verts, faces, points)
        verts, points = verts.contiguous().float(), points.contiguous().float()
        faces = faces.contiguous().long()
        N, V = verts.shape[:2]
        F_, P = faces.shape[0], points.shape[1]
        dev = verts.device
        dp, df = torch.empty(N, P, device=dev), torch.empty(N, F_, device=dev)
        ap = torch.empty(N, P, dtype=torch.int32, device=dev)
        af = torch.empty(N, F_, dtype=torch.int32, device=dev)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = _lib.lib().lasr_point_mesh_forward(verts.data_ptr(), faces.data_ptr(), points.data_ptr(), dp.data_ptr(),
                                                    ap.data_ptr(), df.data_ptr(), af.data_ptr(), N, V, F_, P, st)
        _lib.check(rc, 'lasr_point_mesh_forward')
        ctx.save_for_backward(verts, faces, points, ap, af)
        return (dp.mean(1) + df.mean(1)).mean()

    @staticmethod
    def backward(ctx, g):
        verts, faces, points, ap, af = ctx.saved_tensors
        N, V = verts.shape[:2]
        F_, P = faces.shape[0], points.shape[1]
        h = _lib.lib()
        gtri = torch.empty(N, F_, 3, 3, dtype=torch.float32, device=verts.device)
        gpts = torch.empty_like(points)
        gverts = torch.empty_like(verts)
        faces_n = faces[None].expand(N, F_, 3).contiguous()
        guard, st = _lib.stream_of(verts)
        with guard:
            # unit weights here; the incoming scalar gradient multiplies the results (it may live on the device)
            rc = h.lasr_point_mesh_backward(verts.data_ptr(), faces.data_ptr(), points.data_ptr(), ap.data_ptr(),
                                            af.data_ptr(), 1.0 / (N * P), 1.0 / (N * F_), gtri.data_ptr(), gpts.data_ptr(),
                                            N, V, F_, P, st)
            _lib.check(rc, 'lasr_point_mesh_backward')
            rc = h.lasr_face_gather_backward(gtri.data_ptr(), faces_n.data_ptr(), gverts.data_ptr(), N, V, F_, 3, st)
        _lib.check(rc, 'lasr_face_gather_backward')
        return gverts * g, None, gpts * g


def point_mesh_face_distance(verts, faces, points):
    """mean_n [ mean_p min_f d2(p, f) + mean_f min_p d2(p, f) ]  (pytorch3d.loss.point_mesh_face_distance as used at
    /root/reference/nnutils/mesh_net.py:470-471); verts [N,V,3], faces [F,3], points [N,P,3]."""
    return _PointMesh.apply(verts, faces, points)


class _CosDist(Function):
    @staticmethod
    def forward(ctx, fa, fb, rep):
        _lib.need_cuda(fa, fb)
        fa, fb = fa.detach().contiguous().float(), fb.contiguous().float()
        N, C = fb.shape[:2]
        P = fb[0, 0].numel()
        h = _lib.lib()
        d = torch.empty(N, dtype=torch.float32, device=fb.device)
        scratch = torch.empty(h.lasr_cosdist_scratch_floats(N, P), dtype=torch.float32, device=fb.device)
        guard, st = _lib.stream_of(fb)
        with guard:
            rc = h.lasr_cosdist_forward(fa.data_ptr(), fb.data_ptr(), d.data_ptr(), scratch.data_ptr(), N, C, P, rep, st)
        _lib.check(rc, 'lasr_cosdist_forward')
        ctx.save_for_backward(fa, fb)
        ctx.rep = rep
        return d

    @staticmethod
    def backward(ctx, g):
        fa, fb = ctx.saved_tensors
        N, C = fb.shape[:2]
        P = fb[0, 0].numel()
        g = g.contiguous().float()
        gb = torch.empty_like(fb)
        guard, st = _lib.stream_of(fb)
        with guard:
            rc = _lib.lib().lasr_cosdist_backward(fa.data_ptr(), fb.data_ptr(), g.data_ptr(), gb.data_ptr(), N, C, P, ctx.rep, st)
        _lib.check(rc, 'lasr_cosdist_backward')
        return None, gb, None


def cosine_distance(feat_obs, feat_rnd, repeat=1):
    """1 - mean over pixels of the cosine between channel vectors (/root/reference/third_party/PerceptualSimilarity/
    util/util.py:71-83, models/networks_basic.py:51-52): feat_obs [N/repeat,C,h,w] (no gradient), feat_rnd [N,C,h,w]
    -> [N]."""
    return _CosDist.apply(feat_obs, feat_rnd, repeat)
