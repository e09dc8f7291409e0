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


class _FlowReprojectPlanes(Function):
    @staticmethod
    def forward(ctx, pos6, pp0, pp1, fl0, fl1):
        _lib.need_cuda(pos6, pp0, pp1, fl0, fl1)
        N, P = pos6.shape[0], pos6[0, 0].numel()
        if pos6.dtype != torch.float32 or pos6.shape[1] != 6 or (N > 0 and not pos6[0].is_contiguous()):
            raise ValueError('pos6 must be float32 [N,6,IS,IS] with the six planes of an image contiguous')
        stride = pos6.stride(0) if N > 1 else 6 * P
        pp0, pp1 = pp0.contiguous().float(), pp1.contiguous().float()
        fl0, fl1 = fl0.contiguous().float(), fl1.contiguous().float()
        flow = torch.empty(N, *pos6.shape[2:], 2, dtype=torch.float32, device=pos6.device)
        bg = torch.empty(N, *pos6.shape[2:], dtype=torch.uint8, device=pos6.device)
        guard, st = _lib.stream_of(pos6)
        with guard:
            rc = _lib.lib().lasr_flow_reproject_planes_forward(pos6.data_ptr(), stride, pp0.data_ptr(), pp1.data_ptr(), fl0.data_ptr(),
                                                               fl1.data_ptr(), flow.data_ptr(), bg.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_planes_forward')
        ctx.save_for_backward(pos6, fl1)
        ctx.stride = stride
        bg = bg.view(torch.bool)
        ctx.mark_non_differentiable(bg)
        return flow, bg

    @staticmethod
    def backward(ctx, gflow, _gbg):
        pos6, fl1 = ctx.saved_tensors
        N, P = pos6.shape[0], pos6[0, 0].numel()
        gflow = gflow.contiguous().float()
        gpos = torch.empty(pos6.shape, dtype=torch.float32, device=pos6.device)
        gpp1 = torch.empty(N, 2, dtype=torch.float32, device=pos6.device)
        gfl1 = torch.empty(N, dtype=torch.float32, device=pos6.device)
        h = _lib.lib()
        scratch = torch.empty(h.lasr_flow_reproject_scratch_floats(N, P), dtype=torch.float32, device=pos6.device)
        guard, st = _lib.stream_of(pos6)
        with guard:
            rc = h.lasr_flow_reproject_planes_backward(pos6.data_ptr(), ctx.stride, fl1.data_ptr(), gflow.data_ptr(), gpos.data_ptr(),
                                                       gpp1.data_ptr(), gfl1.data_ptr(), scratch.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_planes_backward')
        return gpos, None, gpp1, None, gfl1


class _RenderTables(Function):
    """lasr_render_tables_*: see render_tables()."""

    @staticmethod
    def forward(ctx, px, masks, occ, flow_obs, obspair, pp, fl, wt, want_pair, from_imgs=False):
        _lib.need_cuda(px, masks, occ, flow_obs, obspair, pp, fl)
        N, I = px.shape[0], masks.shape[0]
        H, P = N // max(I, 1), masks[0].numel()
        if px.dtype != torch.float32 or px.shape[1] != 10 or N != I * H or obspair.shape[0] != (I if from_imgs else 2 * I):
            raise ValueError('px must be float32 [I*H,10,IS,IS], obspair [2I,3,IS,IS] (or the images [I,3,IS,IS])')
        px = px.contiguous()
        masks, occ, flow_obs = masks.contiguous().float(), occ.contiguous().float(), flow_obs.contiguous().float()
        obspair, pp, fl = obspair.detach().contiguous().float(), pp.contiguous().float(), fl.contiguous().float()
        imgs = obspair if from_imgs else None
        if from_imgs:                                      # the pair is formed by the pass itself (lasr_render_tables_forward_imgs)
            obspair = torch.empty(2 * I, *imgs.shape[1:], dtype=torch.float32, device=px.device)
        dev, hw = px.device, px.shape[2:]
        h = _lib.lib()
        tabs = torch.empty(3, I, H, dtype=torch.float32, device=dev)
        flow = torch.empty(N, *hw, 2, dtype=torch.float32, device=dev)
        bg = torch.empty(N, *hw, dtype=torch.uint8, device=dev)
        fmap = torch.empty(N, *hw, dtype=torch.float32, device=dev)
        vis = torch.empty(N, *hw, dtype=torch.uint8, device=dev)
        pair = torch.empty(2 * N, 3, *hw, dtype=torch.float32, device=dev) if want_pair else None
        scratch = torch.empty(h.lasr_render_tables_scratch_floats(I, H, P), dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(px)
        with guard:
            tail = (pp.data_ptr(), fl.data_ptr(), float(wt), tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), flow.data_ptr(),
                    bg.data_ptr(), fmap.data_ptr(), vis.data_ptr(), pair.data_ptr() if want_pair else None, scratch.data_ptr(), I, H, P, st)
            head = (px.data_ptr(), masks.data_ptr(), occ.data_ptr(), flow_obs.data_ptr(), flow_obs[0].numel())
            if from_imgs:
                rc = h.lasr_render_tables_forward_imgs(*head, imgs.data_ptr(), obspair.data_ptr(), *tail)
            else:
                rc = h.lasr_render_tables_forward(*head, obspair.data_ptr(), obspair[I:].data_ptr(), *tail)
        _lib.check(rc, 'lasr_render_tables_forward')
        ctx.save_for_backward(px, masks, occ, flow_obs, obspair, pp, fl, scratch)
        ctx.wt, ctx.geom = float(wt), (I, H, P)
        ctx.from_imgs, ctx.want_pair = bool(from_imgs), bool(want_pair)
        bg, vis = bg.view(torch.bool), vis.view(torch.bool)
        ctx.mark_non_differentiable(flow, bg, fmap, vis)
        ctx.set_materialize_grads(False)
        out = (tabs[0], tabs[1], tabs[2], flow, bg, fmap, vis) + ((pair,) if want_pair else ())
        if from_imgs:
            ctx.mark_non_differentiable(obspair)
            out = out + (obspair,)
        return out

    @staticmethod
    def backward(ctx, g_mask, g_flow, g_tex, _gf=None, _gb=None, _gm=None, _gv=None, g_pair=None, _g_obs=None):
        if ctx.from_imgs and not ctx.want_pair:
            g_pair = None                                  # (the eighth output is then the observed pair: data)
        px, masks, occ, flow_obs, obspair, pp, fl, scratch = ctx.saved_tensors
        I, H, P = ctx.geom
        dev = px.device
        zero = None

        def tab(g):
            nonlocal zero
            if g is None:
                if zero is None:
                    zero = torch.zeros(I, H, dtype=torch.float32, device=dev)
                return zero
            return g.contiguous().float()
        g_mask, g_flow, g_tex = tab(g_mask), tab(g_flow), tab(g_tex)
        g_pair = g_pair.contiguous().float() if g_pair is not None else None
        gpx = torch.empty_like(px)
        gpp = torch.empty_like(pp)
        gfl = torch.empty_like(fl)
        guard, st = _lib.stream_of(px)
        with guard:
            rc = _lib.lib().lasr_render_tables_backward(
                px.data_ptr(), masks.data_ptr(), occ.data_ptr(), flow_obs.data_ptr(), flow_obs[0].numel(), obspair.data_ptr(),
                obspair[I:].data_ptr(), pp.data_ptr(), fl.data_ptr(), ctx.wt, g_mask.data_ptr(), g_flow.data_ptr(), g_tex.data_ptr(),
                g_pair.data_ptr() if g_pair is not None else None, scratch.data_ptr(), gpx.data_ptr(), gpp.data_ptr(),
                gfl.data_ptr(), I, H, P, st)
        _lib.check(rc, 'lasr_render_tables_backward')
        return gpx, None, None, None, None, gpp, gfl, None, None, None


def render_tables(px, masks, occ, flow_obs, obspair, pp, fl, l1tex_wt=1.0, want_pair=False):
    """Everything between the nine-attribute render and the loss sum (/root/reference/nnutils/mesh_net.py:87-104, :374-441) in one
    pass over px [N,10,IS,IS] (texture colours | own position | other frame's position | alpha; N = I*H, first half of the batch
    frame t, second half frame t') and one pass back -- no contiguous copies of channel slices, no split / cat / accumulate
    kernels in the backward.  masks / occ [I,IS,IS], flow_obs [I,>=2,IS,IS], obspair [2I,3,IS,IS] (fused_ops.obs_pair),
    pp [N,2], fl [N] (image n reprojects the other frame's position with the intrinsics of image (n + N/2) % N).
    -> mask table [I,H], flow table [I,H], texture L1 table [I,H] (= 2 wt (mean1 + mean2)), flow_rd [N,IS,IS,2], bgmask,
       weighted flow error map, vis_mask [N,IS,IS] (+ rndpair [2N,3,IS,IS] = (render * alpha | render) when want_pair)."""
    return _RenderTables.apply(px, masks, occ, flow_obs, obspair, pp, fl, l1tex_wt, want_pair)


def render_tables_imgs(px, masks, occ, flow_obs, imgs, pp, fl, l1tex_wt=1.0, want_pair=False):
    """render_tables from the observed images [I,3,IS,IS] themselves: the object on black | on white (obs_pair's values) is formed
    inside the pass and returned as the LAST element [2I,3,IS,IS] -- one launch and three input planes less."""
    return _RenderTables.apply(px, masks, occ, flow_obs, imgs, pp, fl, l1tex_wt, want_pair, True)


class _RasterInputs(Function):
    @staticmethod
    def forward(ctx, verts_cam, tex, pp, fl, eye):
        _lib.need_cuda(verts_cam, tex, pp, fl)
        import ctypes
        N, V = verts_cam.shape[:2]
        verts_cam, tex = verts_cam.contiguous().float(), tex.contiguous().float()
        pp, fl = pp.contiguous().float(), fl.contiguous().float()
        dev = verts_cam.device
        pre = torch.empty(N, V, 3, dtype=torch.float32, device=dev)
        attrs = torch.empty(N, V, 9, dtype=torch.float32, device=dev)
        nf = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = torch.empty(2 * N, dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(verts_cam)
        with guard:
            rc = _lib.lib().lasr_raster_inputs_forward(verts_cam.data_ptr(), tex.data_ptr(), pp.data_ptr(), fl.data_ptr(),
                                                       (ctypes.c_float * 3)(*[float(e) for e in eye]), pre.data_ptr(),
                                                       attrs.data_ptr(), nf.data_ptr(), scratch.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_raster_inputs_forward')
        ctx.save_for_backward(verts_cam, fl)
        ctx.mark_non_differentiable(nf)
        return pre, attrs, nf

    @staticmethod
    def backward(ctx, g_pre, g_attrs, _g_nf=None):
        verts_cam, fl = ctx.saved_tensors
        N, V = verts_cam.shape[:2]
        dev = verts_cam.device
        g_pre = g_pre.contiguous().float() if g_pre is not None else torch.zeros(N, V, 3, device=dev)
        g_attrs = g_attrs.contiguous().float() if g_attrs is not None else torch.zeros(N, V, 9, device=dev)
        g_cam, g_tex = torch.empty_like(verts_cam), torch.empty_like(verts_cam)
        g_pp, g_fl = torch.empty(N, 2, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(verts_cam)
        with guard:
            rc = _lib.lib().lasr_raster_inputs_backward(verts_cam.data_ptr(), fl.data_ptr(), g_pre.data_ptr(), g_attrs.data_ptr(),
                                                        g_cam.data_ptr(), g_tex.data_ptr(), g_pp.data_ptr(), g_fl.data_ptr(), N, V, st)
        _lib.check(rc, 'lasr_raster_inputs_backward')
        return g_cam, g_tex, g_pp, g_fl, None


def raster_inputs(verts_cam, tex, pp, fl, eye):
    """What LASR.forward builds from the camera-space vertices before the render (/root/reference/nnutils/mesh_net.py:298-311,
    :350-356; geom_utils.py:27-34) in one launch: verts_cam / tex [N,V,3], pp [N,2], fl [N], eye (3 Python floats) ->
    verts_pre [N,V,3] = (pinhole(verts_cam) + eye) * (1,-1,1), attrs [N,V,9] = (tex | verts_cam | the other frame's verts_cam;
    other = (n + N/2) % N), near_far [2] device floats (zmin - r/2, zmax + r/2 over all meshes; no gradient)."""
    return _RasterInputs.apply(verts_cam, tex, pp, fl, eye)


def face_incidence(faces, num_vertices):
    """CSR vertex -> incident corners of a face tensor [N,F,3] (or [1,F,3]): (inc_ptr int32 [N,V+1], inc int32 [N,3F]) with the
    corner ids (3 f + c) of a vertex in ASCENDING order -- the order lasr_face_gather_backward sums in.  Built with a handful of
    torch ops; callers cache it per connectivity (LASR.forward: with its repeated face tensor)."""
    N = faces.shape[0]
    flat = faces.reshape(N, -1).long()
    if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= num_vertices):      # (built once per connectivity: one sync)
        raise ValueError('face indices must lie in [0, %d)' % num_vertices)
    order = torch.argsort(flat, dim=1, stable=True)                  # stable: ascending corner id inside a vertex
    counts = torch.zeros(N, num_vertices, dtype=torch.long, device=faces.device)
    counts.scatter_add_(1, flat, torch.ones_like(flat))
    ptr = torch.zeros(N, num_vertices + 1, dtype=torch.long, device=faces.device)
    ptr[:, 1:] = counts.cumsum(1)
    return ptr.int().contiguous(), order.int().contiguous()


class _RasterFaces(Function):
    @staticmethod
    def forward(ctx, verts_cam, tex, pp, fl, eye, faces, inc_ptr, inc):
        import ctypes
        _lib.need_cuda(verts_cam, tex, pp, fl, faces, inc_ptr, inc)
        N, V = verts_cam.shape[:2]
        verts_cam, tex = verts_cam.contiguous().float(), tex.contiguous().float()
        pp, fl = pp.contiguous().float(), fl.contiguous().float()
        faces = faces.contiguous().long()
        shared = int(faces.shape[0] == 1 and N != 1)
        if faces.shape[0] not in (1, N) or inc_ptr.shape[0] != faces.shape[0] or inc.shape[0] != faces.shape[0]:
            raise ValueError('faces / incidence must be [N,...] or [1,...] (shared connectivity)')
        F_ = faces.shape[1]
        # the backward walks inc_ptr / inc without bounds checks: a structure built for another mesh (or left as int64) must not
        # reach the kernels
        if (inc_ptr.dtype != torch.int32 or inc.dtype != torch.int32 or not inc_ptr.is_contiguous() or not inc.is_contiguous()
                or tuple(inc_ptr.shape[1:]) != (V + 1,) or tuple(inc.shape[1:]) != (3 * F_,)):
            raise ValueError('incidence must be face_incidence(faces, V): int32, contiguous, [Nf, V + 1] and [Nf, 3 F] '
                             '(got %s %s and %s %s for V = %d, F = %d)' % (inc_ptr.dtype, tuple(inc_ptr.shape), inc.dtype,
                                                                            tuple(inc.shape), V, F_))
        dev = verts_cam.device
        h = _lib.lib()
        fv = torch.empty(N, F_, 3, 3, dtype=torch.float32, device=dev)
        fa = torch.empty(N, F_, 3, 9, dtype=torch.float32, device=dev)
        nf = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = torch.empty(h.lasr_raster_faces_scratch_floats(N, V, F_), dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(verts_cam)
        with guard:
            rc = h.lasr_raster_faces_forward(verts_cam.data_ptr(), tex.data_ptr(), pp.data_ptr(), fl.data_ptr(),
                                             (ctypes.c_float * 3)(*[float(e) for e in eye]), faces.data_ptr(), shared,
                                             fv.data_ptr(), fa.data_ptr(), nf.data_ptr(), scratch.data_ptr(), N, V, F_, st)
        _lib.check(rc, 'lasr_raster_faces_forward')
        ctx.save_for_backward(verts_cam, fl, inc_ptr, inc)
        ctx.dims = (N, V, F_, shared)
        ctx.mark_non_differentiable(nf)
        return fv, fa, nf

    @staticmethod
    def backward(ctx, g_fv, g_fa, _g_nf=None):
        verts_cam, fl, inc_ptr, inc = ctx.saved_tensors
        N, V, F_, shared = ctx.dims
        dev = verts_cam.device
        g_fv = g_fv.contiguous().float() if g_fv is not None else torch.zeros(N, F_, 3, 3, device=dev)
        g_fa = g_fa.contiguous().float() if g_fa is not None else torch.zeros(N, F_, 3, 9, device=dev)
        g_cam, g_tex = torch.empty_like(verts_cam), torch.empty_like(verts_cam)
        g_pp, g_fl = torch.empty(N, 2, dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev)
        h = _lib.lib()
        scratch = torch.empty(h.lasr_raster_faces_scratch_floats(N, V, F_), dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(verts_cam)
        with guard:
            rc = h.lasr_raster_faces_backward(verts_cam.data_ptr(), fl.data_ptr(), inc_ptr.data_ptr(), inc.data_ptr(), shared,
                                              g_fv.data_ptr(), g_fa.data_ptr(), g_cam.data_ptr(), g_tex.data_ptr(),
                                              g_pp.data_ptr(), g_fl.data_ptr(), scratch.data_ptr(), N, V, F_, st)
        _lib.check(rc, 'lasr_raster_faces_backward')
        return g_cam, g_tex, g_pp, g_fl, None, None, None, None


def raster_faces(verts_cam, tex, pp, fl, eye, faces, incidence):
    """raster_inputs + the camera stage's eye shift + both face gathers in one launch each way: verts_cam / tex [N,V,3], pp [N,2],
    fl [N], eye (3 floats, the constant look_at eye: rotation = identity), faces int64 [N,F,3] or [1,F,3] (shared), incidence =
    face_incidence(faces, V) -> (face_vertices [N,F,3,3] ready for soft_rasterize, face_attrs [N,F,3,9], near_far [2])."""
    return _RasterFaces.apply(verts_cam, tex, pp, fl, eye, faces, incidence[0], incidence[1])


def flow_reproject_planes(pos6, pp0, pp1, fl0, fl1):
    """flow_reproject on the six position planes [N,6,IS,IS] of a wider render (a channel slice of the [N,10,IS,IS] output of the
    9-attribute pass: consecutive images further apart than 6 planes) -> flow [N,IS,IS,2], bgmask [N,IS,IS] bool."""
    N = pos6.shape[0]
    return _FlowReprojectPlanes.apply(pos6, pp0, pp1, fl0.reshape(N, -1)[:, 0], fl1.reshape(N, -1)[:, 0])


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


class _MeshReg(Function):
    @staticmethod
    def forward(ctx, x, dx, ax, ca, cb, lap_ptr, lap_col, arap_ptr, arap_col, quads, inc_ptr, inc):
        _lib.need_cuda(x, dx, ax, lap_ptr, lap_col, arap_ptr, arap_col, quads, inc_ptr, inc)
        x, dx, ax = x.contiguous().float(), dx.contiguous().float(), ax.contiguous().float()
        N, V = x.shape[:2]
        NA, E = dx.shape[0], quads.shape[0]
        if dx.shape != ax.shape or (NA and dx.shape[1] != V):
            raise ValueError('the ARAP pair must be two [NA,V,3] tensors on the same vertex set')
        dev = x.device
        NC = P = Q = 0
        nn = None
        if ca is not None:
            _lib.need_cuda(ca, cb)
            ca, cb = ca.contiguous().float(), cb.contiguous().float()
            NC, P, Q = ca.shape[0], ca.shape[1], cb.shape[1]
            nn = torch.empty(NC * (P + Q), dtype=torch.int32, device=dev)
        out = torch.empty(2 * N + NA + NC, dtype=torch.float32, device=dev)
        lx = torch.empty_like(x)
        ptr = lambda t_: t_.data_ptr() if t_ is not None else None            # noqa: E731
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_step_regularisers_forward(
                x.data_ptr(), dx.data_ptr(), ax.data_ptr(), lap_ptr.data_ptr(), lap_col.data_ptr(), arap_ptr.data_ptr(), arap_col.data_ptr(),
                quads.data_ptr(), out.data_ptr(), lx.data_ptr(), out[N:].data_ptr(), out[2 * N:].data_ptr(), N, NA, V, E,
                ptr(ca), ptr(cb), out[2 * N + NA:].data_ptr() if NC else None, ptr(nn), (nn.data_ptr() + 4 * NC * P) if NC else None,
                NC, P, Q, st)
        _lib.check(rc, 'lasr_step_regularisers_forward')
        ctx.save_for_backward(x, dx, ax, lx, lap_ptr, lap_col, arap_ptr, arap_col, quads, inc_ptr, inc, ca, cb, nn)
        if NC:
            return out[:N], out[N:2 * N], out[2 * N:2 * N + NA], out[2 * N + NA:]
        return out[:N], out[N:2 * N], out[2 * N:]

    @staticmethod
    def backward(ctx, g_lap, g_flat, g_arap, g_cham=None):
        x, dx, ax, lx, lap_ptr, lap_col, arap_ptr, arap_col, quads, inc_ptr, inc, ca, cb, nn = ctx.saved_tensors
        N, V = x.shape[:2]
        NA, E = dx.shape[0], quads.shape[0]
        dev = x.device
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)     # noqa: E731
        g_lap = g_lap.contiguous().float() if g_lap is not None else zeros(N)
        g_flat = g_flat.contiguous().float() if g_flat is not None else zeros(N)
        g_arap = g_arap.contiguous().float() if g_arap is not None else zeros(NA)
        gx, gdx, gax = torch.empty_like(x), torch.empty_like(dx), torch.empty_like(ax)
        NC = P = Q = 0
        gca = gcb = None
        if ca is not None:
            NC, P, Q = ca.shape[0], ca.shape[1], cb.shape[1]
            g_cham = g_cham.contiguous().float() if g_cham is not None else zeros(NC)
            gca, gcb = torch.empty_like(ca), torch.empty_like(cb)
        ptr = lambda t_: t_.data_ptr() if t_ is not None else None            # noqa: E731
        guard, st = _lib.stream_of(x)
        with guard:
            rc = _lib.lib().lasr_step_regularisers_backward(
                x.data_ptr(), dx.data_ptr(), ax.data_ptr(), lap_ptr.data_ptr(), lap_col.data_ptr(), arap_ptr.data_ptr(), arap_col.data_ptr(),
                quads.data_ptr(), inc_ptr.data_ptr(), inc.data_ptr(), lx.data_ptr(), g_lap.data_ptr(), g_flat.data_ptr(), g_arap.data_ptr(),
                gx.data_ptr(), gdx.data_ptr(), gax.data_ptr(), N, NA, V, E, ptr(ca), ptr(cb), ptr(nn),
                (nn.data_ptr() + 4 * NC * P) if NC else None, ptr(g_cham) if NC else None, ptr(gca), ptr(gcb), NC, P, Q, st)
        _lib.check(rc, 'lasr_step_regularisers_backward')
        return (gx, gdx, gax, gca, gcb) + (None,) * 7


def mesh_regularisers(x, arap_dx, arap_x, laplacian, flatten, arap, chamfer_pair=None):
    """(laplacian(x), flatten(x), arap(arap_dx, arap_x)) for the criteria LaplacianLoss / FlattenLoss / ARAPLoss of
    nnutils/loss_utils.py (average = False), in one launch each way instead of three forward and five backward launches; values and
    gradients bit-identical to the three calls (/root/reference/nnutils/mesh_net.py:449-459, :494-497).
    chamfer_pair = (a [NC,P,3], b [NC,Q,3]): a fourth result, chamfer(a, b) [NC] (:500-503), rides on the same two launches."""
    V = x.shape[1]
    dev = x.device
    ptr = flatten.inc_ptr
    if ptr.numel() < V + 1:                                           # trailing vertices no face refers to
        ptr = torch.cat([ptr, ptr[-1:].repeat(V + 1 - ptr.numel())])
    ca, cb = chamfer_pair if chamfer_pair is not None else (None, None)
    return _MeshReg.apply(x, arap_dx, arap_x, ca, cb, laplacian.row_ptr.to(dev), laplacian.col.to(dev), arap.row_ptr.to(dev),
                          arap.col.to(dev), flatten.quads.to(dev), ptr.to(dev), flatten.inc.to(dev))


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
        h = _lib.lib()
        scratch = torch.empty(h.lasr_point_mesh_scratch_floats(N, F_, P), dtype=torch.float32, device=dev)
        guard, st = _lib.stream_of(verts)
        with guard:
            rc = h.lasr_point_mesh_forward(verts.data_ptr(), faces.data_ptr(), points.data_ptr(), dp.data_ptr(),
                                           ap.data_ptr(), df.data_ptr(), af.data_ptr(), scratch.data_ptr(), N, V, F_, P, st)
        _lib.check(rc, 'lasr_point_mesh_forward')
        ctx.save_for_backward(verts, faces, points, ap, af)
        from ..soft_renderer.functional.geometry import _incidence_of
        ctx.inc = _incidence_of(faces, V)                # the model's face tensor is seen every step: vertex -> corner lists, built once
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
        faces_n = faces[None].expand(N, F_, 3).contiguous() if ctx.inc is None else None
        guard, st = _lib.stream_of(verts)
        with guard:
            # unit weights here; the incoming scalar gradient multiplies the results (it may live on the device)
            rc = h.lasr_point_mesh_backward(verts.data_ptr(), faces.data_ptr(), points.data_ptr(), ap.data_ptr(),
                                            af.data_ptr(), 1.0 / (N * P), 1.0 / (N * F_), gtri.data_ptr(), gpts.data_ptr(),
                                            N, V, F_, P, st)
            _lib.check(rc, 'lasr_point_mesh_backward')
            if ctx.inc is not None:                      # same sums, same order, no scan of the face tensor (and no expanded copy of it)
                rc = h.lasr_face_gather_backward_csr(gtri.data_ptr(), ctx.inc[0].data_ptr(), ctx.inc[1].data_ptr(), 1, gverts.data_ptr(),
                                                     N, V, F_, 3, st)
            else:
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


class _CosDistMulti(Function):
    """lasr_cosdist_multi_*: every feature layer in one launch each way (see cosine_distance_layers)."""

    @staticmethod
    def forward(ctx, rep, *feats):
        import ctypes
        L = len(feats) // 2
        _lib.need_cuda(*feats)
        fa = [f.detach().contiguous().float() for f in feats[:L]]
        fb = [f.contiguous().float() for f in feats[L:]]
        N = fb[0].shape[0]
        Cs = (ctypes.c_int * L)(*[f.shape[1] for f in fb])
        Ps = (ctypes.c_int * L)(*[f[0, 0].numel() for f in fb])
        for a, b in zip(fa, fb):
            if b.shape[0] != N or a.shape[0] * rep != N or a.shape[1:] != b.shape[1:]:
                raise ValueError('feature maps must be [N/rep,C,h,w] / [N,C,h,w] per layer')
        h = _lib.lib()
        dev = fb[0].device
        d = torch.empty(N, dtype=torch.float32, device=dev)
        scratch = torch.empty(h.lasr_cosdist_multi_scratch_floats(Ps, L, N), dtype=torch.float32, device=dev)
        pa = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fa])
        pb = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fb])
        guard, st = _lib.stream_of(fb[0])
        with guard:
            rc = h.lasr_cosdist_multi_forward(pa, pb, Cs, Ps, L, d.data_ptr(), scratch.data_ptr(), N, rep, st)
        _lib.check(rc, 'lasr_cosdist_multi_forward')
        ctx.save_for_backward(*fa, *fb)
        ctx.rep, ctx.dims = rep, (Cs, Ps, L, N)
        return d

    @staticmethod
    def backward(ctx, g):
        import ctypes
        Cs, Ps, L, N = ctx.dims
        fa, fb = ctx.saved_tensors[:L], ctx.saved_tensors[L:]
        g = g.contiguous().float()
        gb = [torch.empty_like(f) for f in fb]
        pa = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fa])
        pb = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fb])
        pg = (ctypes.c_void_p * L)(*[f.data_ptr() for f in gb])
        guard, st = _lib.stream_of(fb[0])
        with guard:
            rc = _lib.lib().lasr_cosdist_multi_backward(pa, pb, Cs, Ps, L, g.data_ptr(), pg, N, ctx.rep, st)
        _lib.check(rc, 'lasr_cosdist_multi_backward')
        return (None,) + (None,) * L + tuple(gb)


def cosine_distance_layers(feats_obs, feats_rnd, repeat=1):
    """sum over the feature layers of cosine_distance(feats_obs[l], feats_rnd[l], repeat) -> [N]: the perceptual distance of
    /root/reference/third_party/PerceptualSimilarity/models/networks_basic.py:51-64 (unweighted layers) as ONE launch each way
    (+ one fold launch) instead of one reduce + one fold + one backward launch per layer.  Each layer's value, and the left-to-right sum of the
    layers, are bit-identical to the per-layer calls."""
    if len(feats_obs) != len(feats_rnd) or not 1 <= len(feats_rnd) <= 8:
        raise ValueError('1..8 feature layers, the same number on both sides')
    return _CosDistMulti.apply(int(repeat), *feats_obs, *feats_rnd)


# ---- small-tensor glue of LASR.forward (lasr_amd/csrc/glue.hip): one launch where the reference runs a chain of tiny ops --------
class _Geodesic(Function):
    @staticmethod
    def forward(ctx, m1, m2):
        _lib.need_cuda(m1, m2)
        m1, m2 = m1.contiguous().float(), m2.contiguous().float()
        n = m1.numel() // 9
        angle = torch.empty(n, dtype=torch.float32, device=m1.device)
        guard, st = _lib.stream_of(m1)
        with guard:
            rc = _lib.lib().lasr_geodesic_forward(m1.data_ptr(), m2.data_ptr(), angle.data_ptr(), n, st)
        _lib.check(rc, 'lasr_geodesic_forward')
        ctx.save_for_backward(m1, m2)
        return angle

    @staticmethod
    def backward(ctx, g):
        m1, m2 = ctx.saved_tensors
        g = g.contiguous().float()
        g1, g2 = torch.empty_like(m1), torch.empty_like(m2)
        guard, st = _lib.stream_of(m1)
        with guard:
            rc = _lib.lib().lasr_geodesic_backward(m1.data_ptr(), m2.data_ptr(), g.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                                                   m1.numel() // 9, st)
        _lib.check(rc, 'lasr_geodesic_backward')
        return g1, g2


def geodesic_distance(m1, m2):
    """Rotation angle between two batches of 3x3 matrices [n,3,3] -> [n] (/root/reference/third_party/ext_utils/util_rot.py:27-37)."""
    return _Geodesic.apply(m1, m2)


class _WeightedMeans(Function):
    @staticmethod
    def forward(ctx, weights, groups, n_groups, *xs):
        import ctypes
        _lib.need_cuda(*xs)
        xs = [x.contiguous().float() for x in xs]
        n = len(xs)
        ctx.numels = (ctypes.c_int * n)(*[x.numel() for x in xs])
        ctx.weights = (ctypes.c_float * n)(*[float(w) for w in weights])
        ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
        grp = (ctypes.c_int * n)(*[int(g) for g in groups])
        out = torch.empty(n_groups + 1, dtype=torch.float32, device=xs[0].device)
        guard, st = _lib.stream_of(xs[0])
        with guard:
            rc = _lib.lib().lasr_weighted_means_forward(ptrs, ctx.numels, ctx.weights, grp, n, n_groups, out.data_ptr(), st)
        _lib.check(rc, 'lasr_weighted_means_forward')
        ctx.shapes = [x.shape for x in xs]
        ctx.device = xs[0].device
        total, sums = out[n_groups], out[:n_groups]
        ctx.mark_non_differentiable(sums)
        return total, sums

    @staticmethod
    def backward(ctx, g, _gs):
        n = len(ctx.shapes)
        g = g.contiguous().float()
        coef = torch.empty(n, dtype=torch.float32, device=ctx.device)
        guard, st = _lib.stream_of(coef)
        with guard:
            rc = _lib.lib().lasr_weighted_means_backward(ctx.numels, ctx.weights, n, g.data_ptr(), coef.data_ptr(), st)
        _lib.check(rc, 'lasr_weighted_means_backward')
        return (None, None, None) + tuple(coef[t].expand(ctx.shapes[t]) for t in range(n))


def weighted_mean_sum(terms, n_groups=None):
    """terms: [(tensor, weight, group)] -> (total, group totals [n_groups]) with total = sum_t weight_t * tensor_t.mean()
    accumulated in list order -- the reference's `total_loss += w * x.mean()` chain (nnutils/mesh_net.py:374-530) in one launch.
    The group totals (the per-loss scalars LASR logs) carry no gradient."""
    xs = [t[0] for t in terms]
    groups = [t[2] for t in terms]
    n_groups = max(groups) + 1 if n_groups is None else n_groups
    return _WeightedMeans.apply(tuple(float(t[1]) for t in terms), tuple(groups), n_groups, *xs)


class _Intrinsics(Function):
    @staticmethod
    def forward(ctx, cams, pp, scale, depth, ppoint, half):
        _lib.need_cuda(cams, pp, scale, depth, ppoint)
        cams, pp = cams.contiguous().float(), pp.contiguous().float()
        scale, depth, ppoint = scale.contiguous().float(), depth.contiguous().float(), ppoint.contiguous().float()
        n2, H, K = scale.shape[0], scale.shape[1], depth.shape[1]
        so, do, po = torch.empty_like(scale), torch.empty_like(depth), torch.empty_like(ppoint)
        guard, st = _lib.stream_of(scale)
        with guard:
            rc = _lib.lib().lasr_intrinsics_forward(cams.data_ptr(), cams.shape[1], pp.data_ptr(), scale.data_ptr(), depth.data_ptr(),
                                                    ppoint.data_ptr(), so.data_ptr(), do.data_ptr(), po.data_ptr(), n2 // 2, H, K,
                                                    float(half), st)
        _lib.check(rc, 'lasr_intrinsics_forward')
        ctx.save_for_backward(cams)
        ctx.dims = (n2 // 2, H, K)
        return so, do, po

    @staticmethod
    def backward(ctx, gs, gd, gp):
        cams, = ctx.saved_tensors
        B, H, K = ctx.dims
        gs, gd, gp = gs.contiguous().float(), gd.contiguous().float(), gp.contiguous().float()
        a, b, c = torch.empty_like(gs), torch.empty_like(gd), torch.empty_like(gp)
        guard, st = _lib.stream_of(gs)
        with guard:
            rc = _lib.lib().lasr_intrinsics_backward(cams.data_ptr(), cams.shape[1], gs.data_ptr(), gd.data_ptr(), gp.data_ptr(),
                                                     a.data_ptr(), b.data_ptr(), c.data_ptr(), B, H, K, st)
        _lib.check(rc, 'lasr_intrinsics_backward')
        return None, None, a, b, c, None


def intrinsics(cams, pp, scale, depth, ppoint, img_size):
    """Crop-aware intrinsics of an image pair (/root/reference/nnutils/mesh_net.py:204-217): cams [2B,>=1] (column 0 = crop
    scale), pp [2B,2], predicted scale [2B,H], depth [2B,K], ppoint [2B,2] -> (scale [2B,H], depth [2B,K], ppoint [2B,2])."""
    return _Intrinsics.apply(cams, pp, scale, depth, ppoint, img_size / 2.)


class _BoneFixup(Function):
    @staticmethod
    def forward(ctx, quat, trans, depth, rest_ts, H, K, pair):
        _lib.need_cuda(quat, trans, depth, rest_ts)
        quat, trans, depth = quat.contiguous().float(), trans.contiguous().float(), depth.contiguous().float()
        rest = rest_ts.contiguous().float() if rest_ts is not None else None
        MK = quat.numel() // 9
        M = MK // K
        rmat = torch.empty(MK, 3, 3, dtype=torch.float32, device=quat.device)
        tmat = torch.empty(MK, 3, dtype=torch.float32, device=quat.device)
        angle = torch.empty(MK // 2, dtype=torch.float32, device=quat.device) if pair else None
        h = _lib.lib()
        guard, st = _lib.stream_of(quat)
        with guard:
            args = (quat.data_ptr(), trans.data_ptr(), depth.data_ptr(), rest.data_ptr() if rest is not None else None,
                    rmat.data_ptr(), tmat.data_ptr())
            rc = h.lasr_bone_fixup_pair_forward(*args, angle.data_ptr(), M, H, K, st) if pair else \
                h.lasr_bone_fixup_forward(*args, M, H, K, st)
        _lib.check(rc, 'lasr_bone_fixup_forward')
        ctx.save_for_backward(quat, rest)
        ctx.dims = (M, H, K, bool(pair))
        ctx.in_shapes = (quat.shape, trans.shape, depth.shape, None if rest is None else rest_ts.shape)
        return (rmat, tmat, angle) if pair else (rmat, tmat)

    @staticmethod
    def backward(ctx, gR, gT, g_angle=None):
        quat, rest = ctx.saved_tensors
        M, H, K, pair = ctx.dims
        dev = quat.device
        gR = gR.contiguous().float() if gR is not None else torch.zeros(M * K, 3, 3, device=dev)
        gT = gT.contiguous().float() if gT is not None else torch.zeros(M * K, 3, device=dev)
        gq = torch.empty(M * K, 9, dtype=torch.float32, device=dev)
        gt = torch.empty(M * K, 2, dtype=torch.float32, device=dev)
        gd = torch.empty(M * K, dtype=torch.float32, device=dev)
        gr = torch.empty(H, K - 1, 3, dtype=torch.float32, device=dev) if rest is not None else None
        h = _lib.lib()
        guard, st = _lib.stream_of(quat)
        with guard:
            head = (quat.data_ptr(), rest.data_ptr() if rest is not None else None, gR.data_ptr(), gT.data_ptr())
            tail = (gq.data_ptr(), gt.data_ptr(), gd.data_ptr(), gr.data_ptr() if gr is not None else None, M, H, K, st)
            if pair:
                g_angle = g_angle.contiguous().float() if g_angle is not None else torch.zeros(M * K // 2, device=dev)
                rc = h.lasr_bone_fixup_pair_backward(*head, g_angle.data_ptr(), *tail)
            else:
                rc = h.lasr_bone_fixup_backward(*head, *tail)
        _lib.check(rc, 'lasr_bone_fixup_backward')
        qs, ts, ds, rs = ctx.in_shapes
        return gq.view(qs), gt.view(ts), gd.view(ds), (gr.view(rs) if gr is not None else None), None, None, None


def bone_fixup(quat, trans, depth, rest_ts, H, K, pair_angle=False):
    """Bone-transform fix-up (/root/reference/nnutils/mesh_net.py:259-283): quat [M*K,9] (or [M*K,3,3]) predicted matrices,
    trans [M*K,2], depth [M*K,1], rest_ts [H,(K-1)*3] joint centres (None for K == 1) -> (Rmat [M*K,3,3], Tmat [M*K,3]):
    root = transposed prediction; bones rotate about their joint: T' = -Q^T c + T + c, R' = Q.
    pair_angle: also return the rotation distance [M*K/2] between every matrix of the first half of the batch and its counterpart
    in the second half (:514-516: geodesic_distance(quat[:half], quat[half:])) -- same values and gradients, no launches of its own."""
    return _BoneFixup.apply(quat, trans, depth, rest_ts if K > 1 else None, H, K, bool(pair_angle))


class _ProjectPoints(Function):
    @staticmethod
    def forward(ctx, rest, ctl, Rmat, Tmat, pp, fl, H, K):
        _lib.need_cuda(rest, ctl, Rmat, Tmat, pp, fl)
        rest, ctl = rest.contiguous().float(), ctl.contiguous().float()
        Rmat, Tmat = Rmat.detach().contiguous().float(), Tmat.detach().contiguous().float()
        pp, fl = pp.detach().contiguous().float(), fl.detach().contiguous().float()
        M = Rmat.numel() // (9 * K)
        if Tmat.numel() != 3 * M * K or fl.numel() != M or pp.numel() != 2 * (M // H) or rest.numel() != 3 * H * (K - 1) \
                or ctl.numel() != rest.numel():
            raise ValueError('project_points: rest_ts / ctl_ts [H,K-1,3], Rmat [M*K,3,3], Tmat [M*K,3], pp [M/H,2], fl [M]')
        proj = torch.empty(M, 2 * (K - 1), 4, dtype=torch.float32, device=rest.device)
        guard, st = _lib.stream_of(rest)
        with guard:
            rc = _lib.lib().lasr_project_points_forward(rest.data_ptr(), ctl.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(), pp.data_ptr(),
                                                        fl.data_ptr(), proj.data_ptr(), M, H, K, st)
        _lib.check(rc, 'lasr_project_points_forward')
        ctx.save_for_backward(rest, ctl, Rmat, Tmat, fl)
        ctx.dims = (M, H, K)
        return proj

    @staticmethod
    def backward(ctx, g):
        rest, ctl, Rmat, Tmat, fl = ctx.saved_tensors
        M, H, K = ctx.dims
        g = g.contiguous().float()
        grest, gctl = torch.empty_like(rest), torch.empty_like(ctl)
        guard, st = _lib.stream_of(rest)
        with guard:
            rc = _lib.lib().lasr_project_points_backward(rest.data_ptr(), ctl.data_ptr(), Rmat.data_ptr(), Tmat.data_ptr(), fl.data_ptr(),
                                                         g.data_ptr(), grest.data_ptr(), gctl.data_ptr(), M, H, K, st)
        _lib.check(rc, 'lasr_project_points_backward')
        return grest, gctl, None, None, None, None, None, None


def project_points(rest_ts, ctl_ts, Rmat, Tmat, ppoint, scale, n_hypo, n_bones):
    """Joint centres and control points of every (image, hypothesis) projected into the image -- the two identity-skin obj_to_cam
    calls and the pinhole_cam of /root/reference/nnutils/mesh_net.py:285-288, :302 -- in one launch each way: rest_ts / ctl_ts
    [H(K-1),3], Rmat [M*K,3,3], Tmat [M*K,3] (constants: no gradient), ppoint [2B,2], scale [2B,H] -> proj [M, 2(K-1), 4]
    (joints first, then control points; (u, v, z, 1)).  Gradient flows to rest_ts / ctl_ts only."""
    return _ProjectPoints.apply(rest_ts, ctl_ts, Rmat, Tmat, ppoint, scale.reshape(-1), int(n_hypo), int(n_bones))


class _PoseChain(Function):
    @staticmethod
    def forward(ctx, cams, pp, scale, depth, ppoint, quat4, trans, rest, ctl, H, K, half, pair):
        _lib.need_cuda(cams, pp, scale, depth, ppoint, quat4, trans)
        f = lambda t: None if t is None else t.contiguous().float()                                   # noqa: E731
        cams, pp, scale, depth, ppoint, quat4, trans, rest, ctl = map(f, (cams, pp, scale, depth, ppoint, quat4, trans, rest, ctl))
        n2 = scale.shape[0]
        M, MK = n2 * H, n2 * H * K
        if scale.numel() != M or depth.numel() != n2 * K or ppoint.numel() != 2 * n2 or quat4.numel() != 4 * MK or \
                trans.numel() != 2 * n2 * K or n2 % 2 or (K > 1 and (rest is None or ctl is None or rest.numel() != 3 * H * (K - 1)
                                                                    or ctl.numel() != rest.numel())):
            raise ValueError('pose_chain: scale [2B,H], depth [2B,K], ppoint [2B,2], quat4 [2B*H*K,4], trans [2B*K,2], rest_ts / ctl_ts [H,K-1,3]')
        dev = scale.device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)                     # noqa: E731
        so, do, po = new(n2, H), new(n2, K), new(n2, 2)
        trep, drep, rmat, tmat = new(MK, 2), new(MK, 1), new(MK, 3, 3), new(MK, 3)
        angle = new(MK // 2) if pair else None
        proj = new(M, 2 * (K - 1), 4) if K > 1 else None
        ptr = lambda t: None if t is None else t.data_ptr()                                           # noqa: E731
        guard, st = _lib.stream_of(scale)
        with guard:
            rc = _lib.lib().lasr_pose_chain_forward(cams.data_ptr(), cams.shape[1], pp.data_ptr(), scale.data_ptr(), depth.data_ptr(),
                                                    ppoint.data_ptr(), quat4.data_ptr(), trans.data_ptr(), ptr(rest), ptr(ctl),
                                                    so.data_ptr(), do.data_ptr(), po.data_ptr(), trep.data_ptr(), drep.data_ptr(),
                                                    rmat.data_ptr(), tmat.data_ptr(), ptr(angle), ptr(proj), n2 // 2, H, K,
                                                    float(half), st)
        _lib.check(rc, 'lasr_pose_chain_forward')
        ctx.save_for_backward(cams, quat4, rest, ctl, rmat, tmat, so)
        ctx.dims = (n2 // 2, H, K)
        ctx.in_shapes = (scale.shape, depth.shape, ppoint.shape, quat4.shape, trans.shape)
        return so, po, rmat, tmat, trep, drep, angle, proj

    @staticmethod
    def backward(ctx, g_so, g_po, gR, gT, g_trep, g_drep, g_angle, g_proj):
        cams, quat4, rest, ctl, rmat, tmat, so = ctx.saved_tensors
        B, H, K = ctx.dims
        n2, MK = 2 * B, 2 * B * H * K
        dev = quat4.device
        f = lambda t: None if t is None else t.contiguous().float()                                   # noqa: E731
        g_so, g_po, g_trep, g_drep, g_angle, g_proj = map(f, (g_so, g_po, g_trep, g_drep, g_angle, g_proj))
        gR = f(gR) if gR is not None else torch.zeros(MK, 3, 3, device=dev)
        gT = f(gT) if gT is not None else torch.zeros(MK, 3, device=dev)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)                     # noqa: E731
        gs, gd, gp, gq, gt = new(n2, H), new(n2, K), new(n2, 2), new(MK, 4), new(n2 * K, 2)
        gr = torch.empty_like(rest) if rest is not None else None
        gc = torch.empty_like(ctl) if ctl is not None else None
        scratch = new(MK, 3)
        ptr = lambda t: None if t is None else t.data_ptr()                                           # noqa: E731
        guard, st = _lib.stream_of(quat4)
        with guard:
            rc = _lib.lib().lasr_pose_chain_backward(cams.data_ptr(), cams.shape[1], quat4.data_ptr(), ptr(rest), ptr(ctl), rmat.data_ptr(),
                                                     tmat.data_ptr(), so.data_ptr(), ptr(g_so), ptr(g_po), ptr(g_trep), ptr(g_drep),
                                                     gR.data_ptr(), gT.data_ptr(), ptr(g_angle), ptr(g_proj), gs.data_ptr(), gd.data_ptr(),
                                                     gp.data_ptr(), gq.data_ptr(), gt.data_ptr(), ptr(gr), ptr(gc), scratch.data_ptr(),
                                                     B, H, K, st)
        _lib.check(rc, 'lasr_pose_chain_backward')
        ss, ds, ps, qs, ts = ctx.in_shapes
        return None, None, gs.view(ss), gd.view(ds), gp.view(ps), gq.view(qs), gt.view(ts), gr, gc, None, None, None, None


def pose_chain(cams, pp, scale, depth, ppoint, quat4, trans, rest_ts, ctl_ts, n_hypo, n_bones, img_size, pair_angle=True):
    """intrinsics -> quaternion matrices -> bone fix-up (+ the rotation distance of the frame pair) -> joint / control-point
    projection, the chain of /root/reference/nnutils/mesh_net.py:204-217, :232, :259-289, :302, :514-516 in one launch each way
    (`intrinsics`, `quat_to_rotmat`, `bone_fixup`, `project_points` of this module are its phases; same values and gradients).
    scale [2B,H], depth [2B,K], ppoint [2B,2], quat4 [2B*H*K,4] unit quaternions (x, y, z, w), trans [2B*K,2] ->
    (scale [2B,H], ppoint [2B,2], Rmat [M*K,3,3], Tmat [M*K,3], trans repeated over the hypotheses [M*K,2], depth likewise
    [M*K,1], pair_angle [M*K/2] or None, proj [M, 2(K-1), 4] or None for K == 1)."""
    K = int(n_bones)
    return _PoseChain.apply(cams, pp, scale, depth, ppoint, quat4, trans, rest_ts if K > 1 else None, ctl_ts if K > 1 else None,
                            int(n_hypo), K, img_size / 2., bool(pair_angle))


class _Chamfer(Function):
    @staticmethod
    def forward(ctx, a, b):
        _lib.need_cuda(a, b)
        a, b = a.contiguous().float(), b.contiguous().float()
        N, P, Q = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty(N, dtype=torch.float32, device=a.device)
        nn = torch.empty(N * (P + Q), dtype=torch.int32, device=a.device)
        guard, st = _lib.stream_of(a)
        with guard:
            rc = _lib.lib().lasr_chamfer_forward(a.data_ptr(), b.data_ptr(), out.data_ptr(), nn.data_ptr(),
                                                 nn.data_ptr() + 4 * N * P, N, P, Q, st)
        _lib.check(rc, 'lasr_chamfer_forward')
        ctx.save_for_backward(a, b, nn)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, nn = ctx.saved_tensors
        N, P, Q = a.shape[0], a.shape[1], b.shape[1]
        g = g.contiguous().float()
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        guard, st = _lib.stream_of(a)
        with guard:
            rc = _lib.lib().lasr_chamfer_backward(a.data_ptr(), b.data_ptr(), nn.data_ptr(), nn.data_ptr() + 4 * N * P, g.data_ptr(),
                                                  ga.data_ptr(), gb.data_ptr(), N, P, Q, st)
        _lib.check(rc, 'lasr_chamfer_backward')
        return ga, gb


def chamfer(a, b):
    """Symmetric squared Chamfer distance per batch item, a [N,P,3], b [N,Q,3] -> [N]
    (pytorch3d.loss.chamfer_distance as used at /root/reference/nnutils/mesh_net.py:500-503; the caller takes the batch mean)."""
    return _Chamfer.apply(a, b)


class _MeanShape(Function):
    @staticmethod
    def forward(ctx, mean_v, tex, flip, mask, R, S):
        _lib.need_cuda(mean_v, tex, flip, mask)
        mean_v, tex = mean_v.contiguous().float(), tex.contiguous().float()
        flip = flip.contiguous().float().reshape(-1) if flip is not None else None
        mask = mask.contiguous().float() if mask is not None else None
        H, Vp = mean_v.shape[0], mean_v.shape[1]
        out_v = torch.empty(R * H, Vp + S, 3, dtype=torch.float32, device=mean_v.device)
        out_t = torch.empty_like(out_v)
        guard, st = _lib.stream_of(mean_v)
        with guard:
            rc = _lib.lib().lasr_mean_shape_forward(mean_v.data_ptr(), tex.data_ptr(), flip.data_ptr() if flip is not None else None,
                                                    mask.data_ptr() if mask is not None else None, out_v.data_ptr(), out_t.data_ptr(),
                                                    R, H, Vp, S, st)
        _lib.check(rc, 'lasr_mean_shape_forward')
        ctx.save_for_backward(tex, flip, mask)
        ctx.dims = (R, H, Vp, S)
        return out_v, out_t

    @staticmethod
    def backward(ctx, gv, gt):
        tex, flip, mask = ctx.saved_tensors
        R, H, Vp, S = ctx.dims
        need_v, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gv = gv.contiguous().float() if need_v else None
        gt = gt.contiguous().float() if need_t else None
        gm = torch.empty(H, Vp, 3, dtype=torch.float32, device=tex.device) if need_v else None
        gp = torch.empty(H, Vp, 3, dtype=torch.float32, device=tex.device) if need_t else None
        guard, st = _lib.stream_of(tex)
        ptr = lambda t: t.data_ptr() if t is not None else None                  # noqa: E731
        with guard:
            rc = _lib.lib().lasr_mean_shape_backward(tex.data_ptr(), ptr(flip), ptr(mask), ptr(gv), ptr(gt), ptr(gm), ptr(gp),
                                                     R, H, Vp, S, st)
        _lib.check(rc, 'lasr_mean_shape_backward')
        return gm, gp, None, None, None, None


def mean_shape(mean_v, tex, flip, mask, repeats, num_sym):
    """Per-hypothesis mean shape and colours -> the batch's meshes (/root/reference/third_party/ext_nnutils/mesh_net.py:128-149,
    171-185): mean_v / tex [H,Vp,3] -> (verts [repeats*H, Vp+num_sym, 3], sigmoid colours, same shape); the last num_sym vertices
    are mirrored (flip [1,3]) and appended, positions times mask [Vp+num_sym,3].  num_sym = 0: no symmetry (flip / mask None)."""
    return _MeanShape.apply(mean_v, tex, flip, mask, int(repeats), int(num_sym))


def obs_pair(imgs, masks):
    """Observed image on black and on white for the texture losses (/root/reference/nnutils/mesh_net.py:364-366):
    imgs [n,3,IS,IS], masks [n,IS,IS] -> [2n,3,IS,IS] = cat(imgs * fg, 1 - fg + imgs * fg), fg = masks > 0.  Data, no gradient."""
    _lib.need_cuda(imgs, masks)
    imgs, masks = imgs.detach().contiguous().float(), masks.detach().contiguous().float()
    n, P = imgs.shape[0], masks[0].numel()
    out = torch.empty(2 * n, *imgs.shape[1:], dtype=torch.float32, device=imgs.device)
    guard, st = _lib.stream_of(imgs)
    with guard:
        rc = _lib.lib().lasr_obs_pair(imgs.data_ptr(), masks.data_ptr(), out.data_ptr(), n, P, st)
    _lib.check(rc, 'lasr_obs_pair')
    return out


# ---- plumbing around the raster calls ------------------------------------------------------------------------------------------
def fill_planes(dst, values):
    """dst [N, C, ...] (contiguous fp32, on the GPU): channel plane c of every image := values[c], in place -- the background fill
    in front of the forward raster kernel (/root/reference/third_party/softras/soft_renderer/functional/soft_rasterize.py:50-53)."""
    import ctypes
    _lib.need_cuda(dst)
    C = dst.shape[1]
    if len(values) != C or dst.dtype != torch.float32 or not dst.is_contiguous():
        raise ValueError('fill_planes: dst must be contiguous fp32 [N, %d, ...]' % len(values))
    N = dst.shape[0]
    P = dst[0, 0].numel() if N else 0
    guard, st = _lib.stream_of(dst)
    with guard:
        rc = _lib.lib().lasr_fill_planes(dst.data_ptr(), (ctypes.c_float * C)(*[float(v) for v in values]), C, N, P, st)
    _lib.check(rc, 'lasr_fill_planes')
    return dst
