"""Whole-forward oracle for `LASR.forward` -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

Eager CPU (torch fp32) composition of the rendering + loss section of the reference model,
/root/reference/nnutils/mesh_net.py:152-556, restated statement by statement in the reference's order so that torch
autograd reproduces the reference's gradient flow: the pair interleave (:155-156), the intrinsics bookkeeping
(:204-217), which half of `ppoint` / `scale` feeds which render (:318-331), every `.detach()` (:264, :285-288, :101-102),
the bone fix-up (:275-283), the loss weights (:374-530).  The pieces it composes are the restatements of
oracle/path_oracle.py (obj_to_cam, pinhole_cam, skinning, loss tables, regularisers) and, for the three render calls,
the C restatement of the reference rasteriser (oracle/sr_oracle.c, fp32) behind a torch.autograd.Function that mirrors
third_party/softras/soft_renderer/functional/soft_rasterize.py:9-102.

What is injected instead of computed: the outputs of the encoder + code predictor (ResNet-18 / FC heads: SURVEY.md
section 2 rows marked OUT) enter as the leaf tensors `code = (scale, trans, quat, depth, ppoint)`; the perceptual
network is off (`ptex_loss = None`); the pose-noise branch (:220-235, random) is not taken (epoch 0).

PARITY STATUS: PINNED.  oracle/gen_forward_golden.py imports the reference's nnutils/mesh_net.py in the build container (with
placeholder modules for absl / torchvision / trimesh, the encoder and the perceptual network injected, the compiled rasteriser
served by oracle/sr_oracle.c, and the same three third-party restatements this file uses: its manifest lists them), runs
`LASR.forward` + backward on three small configurations and stores inputs and outputs in tests/golden/lasr_forward.npz;
tests/test_forward_oracle_vs_reference_golden.py holds this file to it: total loss bit for bit, tables and gradients to ~1e-6.
(That run also showed what mesh_net.py:281 does on a CPU -- `Rmat[:,1:] = Rmat[:,1:].permute(0,1,3,2)` copies a tensor onto
itself through a transposed view; see the generator's note -- this file restates the statement out of place, as the transpose
the code means and the device computes.)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import path_oracle as po
from . import sr_oracle


RASTER = 'c_oracle'


def set_raster(which):
    """Which rasteriser sits behind the three render calls: 'c_oracle' = oracle/sr_oracle.c on the host cores (default);
    'reference_build' = the reference's own kernels built for gfx950 (oracle/_ref/sr_ref_nofma.so, oracle/build_ref.py) on
    cuda:0 -- the same arithmetic (tests/test_oracle_vs_reference_vectors.py) three orders of magnitude faster, which is what
    makes the BASELINE-size comparisons (96 meshes per render call at dog15's stage 0) take seconds.  Returns the old value."""
    global RASTER
    assert which in ('c_oracle', 'reference_build')
    old, RASTER = RASTER, which
    return old


class SoftRasterizeOracle(torch.autograd.Function):
    """soft_rasterize.py:9-102 with the CUDA extension replaced by oracle/sr_oracle.c (fp32), or by the reference
    extension itself running on the GPU (set_raster)."""

    @staticmethod
    def forward(ctx, face_vertices, textures, image_size, kw):
        ctx.kw, ctx.image_size, ctx.which = kw, image_size, RASTER
        ctx.shapes = (face_vertices.shape, textures.shape)
        if RASTER == 'reference_build':
            from . import sr_ref
            saved = sr_ref.forward(face_vertices.detach().cuda(), textures.detach().cuda(), image_size,
                                   variant='sr_ref_nofma', **kw)
            ctx.saved = saved
            return saved['soft_colors'].cpu()
        fv = face_vertices.detach().numpy()
        tx = textures.detach().numpy()
        saved = sr_oracle.forward(fv, tx, image_size, **kw)
        ctx.saved = saved
        return torch.from_numpy(saved['soft_colors'].copy())

    @staticmethod
    def backward(ctx, grad):
        if ctx.which == 'reference_build':
            from . import sr_ref
            gf, gt = sr_ref.backward(ctx.saved, grad.contiguous().cuda(), ctx.image_size, variant='sr_ref_nofma', **ctx.kw)
            return gf.cpu().view(ctx.shapes[0]), gt.cpu().view(ctx.shapes[1]), None, None
        gf, gt = sr_oracle.backward(ctx.saved, grad.contiguous().numpy(), ctx.image_size, **ctx.kw)
        return (torch.from_numpy(gf).view(ctx.shapes[0]), torch.from_numpy(gt).view(ctx.shapes[1]), None, None)


def look_at(vertices, eye, at=(0., 0., 0.), up=(0., 1., 0.)):
    """third_party/softras/soft_renderer/functional/look_at.py:46-61."""
    n = vertices.shape[0]
    eye = torch.tensor(eye, dtype=torch.float32)[None].repeat(n, 1)
    at = torch.tensor(at, dtype=torch.float32)[None].repeat(n, 1)
    up = torch.tensor(up, dtype=torch.float32)[None].repeat(n, 1)
    z_axis = F.normalize(at - eye, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    r = torch.cat((x_axis[:, None], y_axis[:, None], z_axis[:, None]), 1)
    return torch.matmul(vertices - eye[:, None], r.transpose(1, 2))


def face_vertices(vertices, faces):
    """third_party/softras/soft_renderer/functional/face_vertices.py:4-22."""
    bs, nv = vertices.shape[:2]
    faces = faces + (torch.arange(bs) * nv)[:, None, None]
    return vertices.reshape(bs * nv, -1)[faces.long()]


def render_mesh(verts, faces, textures, eye, image_size, kw):
    """SoftRenderer.render_mesh (renderer.py:94-98) for LASR's renderers (mesh_net.py:132-149): vertex textures,
    ambient light 1 / directional 0 (lighting.py: textures * (1*[1,1,1] + 0*...) -- the directional term is computed
    and multiplied by intensity 0 in the reference, contributing +0 and zero gradient), look_at camera without
    perspective, then the rasteriser."""
    textures = textures * 1.0
    verts = look_at(verts, eye)
    return SoftRasterizeOracle.apply(face_vertices(verts, faces), face_vertices(textures, faces), image_size, kw)


def inject_values(x, values):
    """Straight-through value injection for tests: returns a tensor whose VALUE is `values` bit for bit and whose autograd
    graph is x's (x + (values - x).detach(): the difference of two nearby floats is exact, so the sum lands on `values`).
    The soft rasteriser is ill-conditioned in its geometry input (the same kernel in fp32 and fp64 differs
    by 1e-2 on 0.3 % of the pixels), so a tight image / gradient comparison needs bit-identical vertices on both sides;
    everything up to the rasteriser is compared directly."""
    if values is None:
        return x
    return x + (values.to(x.dtype) - x).detach()


def render_flow_soft_2(eye, image_size, kw, verts, faces, verts_pos0, verts_pos1, pp0, pp1, proj_cam0, proj_cam1,
                       inject=None, record=None):
    """nnutils/mesh_net.py:75-104 (two stacked 3-channel renders, as the reference does it)."""
    n_hypo = verts.shape[0] // faces.shape[0]
    faces = faces[:, None].repeat(1, n_hypo, 1, 1).view(-1, faces.shape[1], 3)
    offset = torch.tensor(eye, dtype=torch.float32)[None, None]
    verts_pre = verts[:, :, :3] + offset
    verts_pre = torch.cat([verts_pre[:, :, :1], -1 * verts_pre[:, :, 1:2], verts_pre[:, :, 2:]], -1)
    if record is not None:
        record.append(verts_pre.detach().clone())
    verts_pre = inject_values(verts_pre, inject)
    nb = verts.shape[0]
    px = render_mesh(torch.cat([verts_pre, verts_pre], 0), torch.cat([faces, faces], 0),
                     torch.cat([verts_pos0[:, :, :3], verts_pos1[:, :, :3]], 0), eye, image_size, kw)
    fgmask = px[:nb, -1]
    stacked = torch.cat([px[:nb, :3], px[nb:, :3], px[:nb, 3:]], 1)            # (pos0, pos1, alpha) per pixel
    flow, bgmask = po.flow_reproject(stacked, pp0, pp1, proj_cam0, proj_cam1)   # :93-104
    return flow, bgmask, fgmask


def symmetrize(V, num_indept, num_sym, symidx):
    """third_party/ext_nnutils/mesh_net.py:128-140 (no-batch branch)."""
    flip = torch.ones(1, 3)
    flip[0, symidx] = -1
    verts = torch.cat([V, flip * V[-num_sym:]], 0)
    keep = torch.ones_like(verts)
    keep[:num_indept, symidx] = 0                 # `verts[:num_indept, symidx] = 0` written out of place
    return verts * keep


def reg_decay(curr_steps, max_steps, min_wt, max_wt):
    """nnutils/mesh_net.py:106-113."""
    if curr_steps > max_steps:
        return min_wt
    return float(np.exp(curr_steps / float(max_steps) * (np.log(min_wt) - np.log(max_wt))) * max_wt)


def geodesic(m1, m2):
    """third_party/ext_utils/util_rot.py:27-37 (compute_geodesic_distance_from_two_matrices)."""
    m = torch.bmm(m1, m2.transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    cos = torch.min(cos, torch.ones_like(cos))
    cos = torch.max(cos, -torch.ones_like(cos))
    return torch.acos(cos)


def flatten_loss(vertices, faces, eps=1e-6):
    """third_party/ext_nnutils/loss_utils.py:67-152 (average=False): [N]."""
    f = np.asarray(faces)
    edges = sorted(set(tuple(v) for v in np.sort(np.concatenate((f[:, 0:2], f[:, 1:3]), 0))))
    vert_face = {}
    for k, v in enumerate(f):
        for vx in v:
            vert_face.setdefault(int(vx), []).append(k)
    v0s, v1s, v2s, v3s = [], [], [], []
    for v0, v1 in edges:
        others = []
        for fid in sorted(set(vert_face[int(v0)]) & set(vert_face[int(v1)])):
            face = f[fid]
            others.append(int(face[(face != v0) & (face != v1)][0]))
        v0s.append(int(v0)); v1s.append(int(v1)); v2s.append(others[0]); v3s.append(others[1])
    v0, v1, v2, v3 = (vertices[:, torch.tensor(ix)] for ix in (v0s, v1s, v2s, v3s))

    def side(a, b):
        al2, bl2 = a.pow(2).sum(-1), b.pow(2).sum(-1)
        al1, bl1 = (al2 + eps).sqrt(), (bl2 + eps).sqrt()
        ab = (a * b).sum(-1)
        cos = ab / (al1 * bl1 + eps)
        sin = (1 - cos.pow(2) + eps).sqrt()
        c = a * (ab / (al2 + eps))[:, :, None]
        return b - c, bl1 * sin
    cb1, cb1l1 = side(v1 - v0, v2 - v0)
    cb2, cb2l1 = side(v1 - v0, v3 - v0)
    cos = (cb1 * cb2).sum(-1) / (cb1l1 * cb2l1 + eps)
    return (cos + 1).pow(2).sum(1)


def lasr_forward(P, code, batch_input, cfg):
    """The training branch of LASR.forward, nnutils/mesh_net.py:152-556.

    P     : dict of leaf tensors -- mean_v [H,Vs,3], tex [H,Vs,3] (Vs = stored vertices), ctl_rs [H(K-1),4],
            rest_ts / ctl_ts / log_ctl [H(K-1),3]
    code  : (scale [2B,H], trans [2B*K,2], quat [2B*H*K,9], depth [2B,K], ppoint [2B,2]) -- the code predictor's output
    batch_input : the trainer's dict of [2B,...] tensors in loader order (pair-interleaved), CPU
    cfg['inject'] (optional, tests): dict(flow_fw, flow_bw, tex = geometry values for the three render calls, near_far)
    cfg   : dict(n_hypo, n_bones, img_size, subdivide, num_epochs, l1tex_wt, sigval, symmetric, symmetric_loss, opt_tex,
                 use_gtpose, epoch, iters, faces [F,3] int array, num_indept, num_sym, symidx, eye)
    -> (total_loss, dict of intermediate tensors)
    """
    H, K, IS = cfg['n_hypo'], cfg['n_bones'], cfg['img_size']
    out = {}
    B = batch_input['input_imgs  '].shape[0] // 2                                                     # :154
    bi = {k: v.view(B, 2, -1).permute(1, 0, 2).reshape(v.shape) for k, v in batch_input.items()}     # :155-156
    imgs, masks, cams = bi['imgs        '], bi['masks       '], bi['cams        ']
    depth_gt, flow_obs, ddts_barrier = bi['depth_gt    '], bi['flow        '], bi['ddts_barrier']
    pp, occ, oriimg_shape = bi['pp          '], bi['occ         '], bi['oriimg_shape']

    # ---- get_mean_shape (ext_nnutils/mesh_net.py:171-185)
    if cfg['symmetric']:
        mean_v = torch.cat([symmetrize(v, cfg['num_indept'], cfg['num_sym'], cfg['symidx'])[None] for v in P['mean_v']], 0)
        tex = torch.cat([torch.cat([t, t[-cfg['num_sym']:]], 0)[None] for t in P['tex']], 0)
    else:
        mean_v, tex = P['mean_v'], P['tex']
    faces_np = np.asarray(cfg['faces'])
    faces = torch.from_numpy(faces_np.astype(np.int64))[None].repeat(2 * B, 1, 1)
    pred_v = mean_v[None].repeat(2 * B, 1, 1, 1).view(2 * B * H, -1, 3)
    tex = tex[None].repeat(2 * B, 1, 1, 1).sigmoid().view(2 * B * H, -1, 3)
    # :183-188 picks tex[i, dataid[i]] out of a [2B,1,H,V,3] view: the identity for the single-video datasets (dataid 0)

    scale, trans, quat, depth, ppoint = [c.clone() for c in code]                # the reference edits them in place
    scale = cams[:, :1] * scale                                                  # :206
    depth = torch.cat([cams[:, :1] * depth[:, :1], depth[:, 1:]], 1).view(-1, 1)  # :209
    ppb1 = cams[:B, :1] * pp[:B] / (IS / 2.)                                     # :213-214
    ppb2 = cams[B:, :1] * pp[B:] / (IS / 2.)
    ppa1 = ppoint[:B] + ppb1 + 1                                                 # :217-219
    ppa2 = ppa1 * (cams[B:, :1] / cams[:B, :1])
    ppoint = torch.cat([ppoint[:B], ppa2 - ppb2 - 1], 0)
    quat = quat.view(-1, 9)
    assert not (cfg.get('noise', True) and cfg['epoch'] > 0 and 1 < cfg['iters'] < 100), 'random branch :222-235 not restated'
    depth = depth.view(B * 2, 1, K, 1).repeat(1, H, 1, 1).view(-1, 1)            # :237-238
    trans = trans.view(B * 2, 1, K, 2).repeat(1, H, 1, 1).view(-1, 2)
    if cfg['use_gtpose']:                                                        # :240-253
        quat_pred, scale_pred, trans_pred = quat.clone(), scale.clone(), trans.clone()
        ppoint_pred, depth_pred = ppoint.clone(), depth.clone()
        scale = 10 * cams[:, :1]
        trans = cams[:, 1:3]
        quat = po.quaternion_to_rotation_matrix(torch.cat((cams[:, 4:], cams[:, 3:4]), -1)).view(-1, 9)
        depth = depth_gt[:]
        halforisize = 0.5 * IS / cams[:, :1]
        ppoint = (0.5 * oriimg_shape - pp[:]) / halforisize - 1

    Rmat = quat.view(-1, 3, 3).permute(0, 2, 1)                                  # :259-260
    Tmat = torch.cat([trans, depth], 1)
    if K > 1:
        skin = po.skin_weights(P['ctl_ts'], P['ctl_rs'], P['log_ctl'], pred_v.view(2 * B, H, -1, 3)[0])[..., None]   # :264-267
        out['skin'] = skin
        skin = skin.repeat(B * 2, 1, 1, 1)                                       # :271
        rest_ts = P['rest_ts'][:, None, :, None].repeat(B * 2, 1, 1, 1).view(-1, K - 1, 3, 1)        # :275-276
        ctl_ts = P['ctl_ts'][:, None, :, None].repeat(B * 2, 1, 1, 1).view(-1, K - 1, 3, 1)
        Rmat = Rmat.reshape(-1, K, 3, 3)
        Tmat = Tmat.view(-1, K, 3, 1)
        Tmat = torch.cat([Tmat[:, :1], -Rmat[:, 1:].matmul(rest_ts) + Tmat[:, 1:] + rest_ts], 1)    # :280
        Rmat = torch.cat([Rmat[:, :1], Rmat[:, 1:].permute(0, 1, 3, 2)], 1)                          # :281
        Rmat = Rmat.reshape(-1, 3, 3)
        Tmat = Tmat.reshape(-1, 3)
        out['Rmat'], out['Tmat'] = Rmat, Tmat
        eye_k = torch.eye(K - 1)[None, :, :, None]
        jp = po.obj_to_cam(rest_ts[:, :, :, 0], Rmat.detach(), Tmat[:, None].detach(), K, H, eye_k)  # :285-288
        joints_proj = po.pinhole_cam(torch.cat([jp, torch.ones_like(jp[:, :, :1])], -1), ppoint.detach(), scale.detach())
        cp = po.obj_to_cam(ctl_ts[:, :, :, 0], Rmat.detach(), Tmat[:, None].detach(), K, H, eye_k)
        ctl_proj = po.pinhole_cam(torch.cat([cp, torch.ones_like(cp[:, :, :1])], -1), ppoint.detach(), scale.detach())
        out['joints_proj'], out['ctl_proj'] = joints_proj, ctl_proj
    else:
        skin = None
    deform_v = po.obj_to_cam(pred_v, Rmat.view(-1, 3, 3), Tmat[:, None, :], K, H, skin, tocam=False)   # :291
    out['deform_v'] = deform_v

    # ---- 1) flow rendering (:298-335)
    verts_fl = po.obj_to_cam(pred_v, Rmat, Tmat[:, None, :], K, H, skin)
    out['verts_cam'] = verts_fl
    verts_fl = torch.cat([verts_fl, torch.ones_like(verts_fl[:, :, 0:1])], -1)
    verts_pos0 = verts_fl.view(2 * B, H, -1, 4)[:B].clone().view(B * H, -1, 4)
    verts_pos1 = verts_fl.view(2 * B, H, -1, 4)[B:].clone().view(B * H, -1, 4)
    verts_fl = po.pinhole_cam(verts_fl, ppoint, scale)
    dmax, dmin = verts_fl[:, :, -2].max(), verts_fl[:, :, -2].min()               # :304-311 (-> Python floats at the call)
    near, far = float((dmin - (dmax - dmin) / 2).detach()), float((dmax + (dmax - dmin) / 2).detach())
    out['near_far'] = (near, far)
    inj = cfg.get('inject') or {}
    if 'near_far' in inj:
        near, far = inj['near_far']
    out['pre_raster'] = rec_geo = []                 # the geometry handed to the three render calls, before any injection
    eye = cfg['eye']
    kw = dict(background_color=[0, 0, 0], near=near, far=far, fill_back=True, eps=1e-3, sigma_val=cfg['sigval'],
              dist_func='euclidean', dist_eps=1e-4, gamma_val=1e-2, aggr_func_rgb='softmax', aggr_func_alpha='prod',
              texture_type='vertex')
    vf = verts_fl.view(2 * B, H, -1, 4)
    flow_fw, bgmask_fw, _ = render_flow_soft_2(
        eye, IS, kw, vf[:B].reshape(-1, verts_fl.shape[1], 4), faces[:B], verts_pos0, verts_pos1,
        ppoint[:, None][:B].repeat(1, H, 1).view(-1, 2), ppoint[:, None][B:].repeat(1, H, 1).view(-1, 2),
        scale[:, None][:B].reshape(-1, 1), scale[:, None][B:].reshape(-1, 1), inj.get('flow_fw'), rec_geo)
    flow_bw, bgmask_bw, _ = render_flow_soft_2(
        eye, IS, kw, vf[B:].reshape(-1, verts_fl.shape[1], 4), faces[B:], verts_pos1, verts_pos0,
        ppoint[:, None][B:].repeat(1, H, 1).view(-1, 2), ppoint[:, None][:B].repeat(1, H, 1).view(-1, 2),
        scale[:, None][B:].reshape(-1, 1), scale[:, None][:B].reshape(-1, 1), inj.get('flow_bw'), rec_geo)
    bgmask = torch.cat([bgmask_fw, bgmask_bw], 0)
    flow_rd = torch.cat([flow_fw, flow_bw], 0)
    out['flow_rd'], out['bgmask'] = flow_rd, bgmask

    # ---- 3) texture rendering (:348-363); verts_tex is the same LBS + projection recomputed from a clone of Rmat
    verts_tex = po.obj_to_cam(pred_v, Rmat.clone(), Tmat[:, None, :], K, H, skin)
    verts_tex = torch.cat([verts_tex, torch.ones_like(verts_tex[:, :, 0:1])], -1)
    verts_tex = po.pinhole_cam(verts_tex, ppoint, scale)
    verts_pre = verts_tex[:, :, :3] + torch.tensor(eye, dtype=torch.float32)[None, None]
    verts_pre = torch.cat([verts_pre[:, :, :1], -1 * verts_pre[:, :, 1:2], verts_pre[:, :, 2:]], -1)
    rec_geo.append(verts_pre.detach().clone())
    verts_pre = inject_values(verts_pre, inj.get('tex'))
    kw_tex = dict(kw, background_color=[1, 1, 1])
    texture_render = render_mesh(verts_pre, faces[:, None].repeat(1, H, 1, 1).view(-1, faces.shape[1], 3), tex, eye, IS, kw_tex)
    mask_pred = texture_render[:, -1]
    fgmask_tex = texture_render[:, -1]
    texture_render = texture_render[:, :3]
    img_obs = imgs[:] * (masks[:] > 0).float()[:, None]
    img_white = 1 - (masks[:] > 0).float()[:, None] + img_obs
    out['mask_pred'], out['texture_render'] = mask_pred, texture_render

    # ---- losses.  1) mask (:374-390)
    mask_loss_sub = po.mask_loss_table(mask_pred.view(2 * B, -1, IS, IS), masks, occ)
    total = mask_loss_sub.mean().clone()
    # 2) flow (:393-416)
    flow_rd_loss_sub, flow_rd_map = po.flow_loss_table(flow_rd.view(2 * B, -1, IS, IS, 2), flow_obs,
                                                       bgmask.view(2 * B, -1, IS, IS), occ, masks)
    total = total + flow_rd_loss_sub.mean()
    # 3) texture (:419-447), perceptual term off
    tmplist = po.tex_loss_table(img_obs, img_white, texture_render.view(2 * B, -1, 3, IS, IS),
                                fgmask_tex.view(2 * B, -1, IS, IS), occ, cfg['l1tex_wt'])
    texture_loss_sub = 0.25 * tmplist
    if cfg['opt_tex']:
        total = total + texture_loss_sub.mean()
    # 4) smoothness (:449-459)
    factor = 1 if H > 1 else reg_decay(cfg['epoch'], cfg['num_epochs'], 0.05, 0.5)
    tri = factor * 0.005 * po.laplacian(pred_v, faces_np) * (4 ** cfg['subdivide']) / 64.
    tri = tri + factor * 5e-4 * flatten_loss(pred_v, faces_np) * (2 ** cfg['subdivide'] / 8.0)
    triangle_loss_sub = tri.view(2 * B, H)
    total = total + triangle_loss_sub.mean()
    if (not cfg['symmetric']) and cfg['symmetric_loss']:                         # :461-478
        pointa = pred_v.view(2 * B, H, -1, 3)[0]
        pointb = torch.tensor([[[-1., 1., 1.]]]) * pointa
        fac = torch.from_numpy(faces_np.astype(np.int64))
        total = total + po.point_mesh_face_distance(pointa, fac, pointb)
        total = total + po.point_mesh_face_distance(pointb, fac, pointa)
        if cfg['opt_tex']:
            pa = pred_v[:1].detach()
            pb = torch.tensor([[[-1., 1., 1.]]]) * pa
            idx1 = (pa[:, :, None] - pb[:, None]).pow(2).sum(-1).argmin(2)       # chamfer3D idx1
            total = total + (P['tex'][0][idx1[0]].detach() - P['tex'][0]).abs().mean() * 1e-3
    # 5) deformation (:481-497)
    if K > 1:
        lmotion_loss_sub = factor * (deform_v - pred_v).norm(2, -1).mean(-1).view(2 * B, H)
        total = total + lmotion_loss_sub.mean()
        arap = po.arap(deform_v[:B * H], deform_v[B * H:], faces_np).mean() * (4 ** cfg['subdivide']) / 64.
        total = total + arap
        out['lmotion_loss_sub'], out['arap_loss'] = lmotion_loss_sub, arap
    if K > 1 and cfg['symmetric_loss']:                                          # :500-503
        pointa = P['ctl_ts'].view(H, -1, 3)
        total = total + 0.1 * po.chamfer_distance(pointa, torch.tensor([[[-1., 1., 1.]]]) * pointa)
    # 7) camera (:506-522)
    if cfg['use_gtpose']:
        cam_loss = geodesic(quat.view(-1, 3, 3), quat_pred.view(-1, 3, 3)).mean()
        cam_loss = cam_loss + (scale_pred - scale).abs().mean() + (trans_pred - trans).abs().mean()
        cam_loss = cam_loss + (depth_pred - depth).abs().mean() + (ppoint_pred - ppoint).abs().mean()
        cam_loss = 0.2 * cam_loss
    else:
        q = quat.view(-1, H, K, 9)
        cam_loss = 0.001 * geodesic(q[:B].reshape(-1, 3, 3), q[B:].reshape(-1, 3, 3)).mean()
        if K > 1:
            t = trans.view(-1, H, K, 2)
            d = depth.view(-1, H, K, 1)
            cam_loss = cam_loss + 0.01 * (t[:B, :, 1:] - t[B:, :, 1:]).abs().mean()
            cam_loss = cam_loss + 0.01 * (d[:B, :, 1:] - d[B:, :, 1:]).abs().mean()
    total = total + cam_loss
    # 8) aux (:524-530)
    total = total + 0.02 * F.relu(2 - Tmat.view(-1, 1, K, 3)[:, :, :1, -1]).mean()
    if K > 1:
        barrier = ddts_barrier.repeat(1, H, 1, 1).view(-1, 1, IS, IS)
        bone_loc = 0.1 * F.grid_sample(barrier, joints_proj[:, :, :2].reshape(-1, K - 1, 1, 2), padding_mode='border',
                                       align_corners=False).mean()
        ctl_loc = 0.1 * F.grid_sample(barrier, ctl_proj[:, :, :2].reshape(-1, K - 1, 1, 2), padding_mode='border',
                                      align_corners=False).mean()
        total = total + 100 * (bone_loc + ctl_loc)
    out.update(mask_loss_sub=mask_loss_sub, flow_rd_loss_sub=flow_rd_loss_sub, flow_rd_map=flow_rd_map,
               texture_loss_sub=texture_loss_sub, triangle_loss_sub=triangle_loss_sub, cam_loss=cam_loss, total_loss=total)
    return total, out
