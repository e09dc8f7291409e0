#!/usr/bin/env python3
"""Pin the COMPOSITION of the step: run the reference's own `LASR.forward` (nnutils/mesh_net.py:152-556) on the CPU of this
container and store its losses, tables and gradients as tests/golden/lasr_forward.npz.

    python oracle/gen_forward_golden.py          # needs /root/reference; never runs on the GPU box

TEST INFRASTRUCTURE.  The fixture holds inputs + expected outputs only (arrays and a JSON manifest, no source).  What is
EXECUTED here is the reference's Python: nnutils/mesh_net.py (LASR.forward, render_flow_soft_2, reg_decay),
third_party/ext_nnutils/mesh_net.py (MeshNet: mean shape, symmetrisation), nnutils/geom_utils.py (obj_to_cam, pinhole_cam),
nnutils/loss_utils.py (ARAPLoss), third_party/ext_nnutils/loss_utils.py (LaplacianLoss, FlattenLoss),
third_party/ext_utils/{mesh,meshzoo,util_rot}.py and the whole soft_renderer Python package (renderer, lighting, look_at,
rasterizer, the autograd Function of soft_rasterize.py).  What is NOT available in this image and is replaced -- every
replacement is listed in the fixture's manifest:

  placeholders (modules the reference imports at the top of its files and does not need for the captured values):
      absl.app / absl.flags (the options arrive as a plain namespace), torchvision, trimesh, skimage, cv2, png;
  out of scope, injected (SURVEY.md section 2, rows marked OUT): ext_nnutils.net_blocks -- the ResNet-18 encoder and the
      code predictor; their OUTPUT (scale, trans, quat, depth, ppoint) is a set of leaf tensors of the fixture;
      the perceptual network `ptex_loss` (AlexNet) returns zeros, as in oracle/lasr_forward_oracle.py;
  the compiled extension soft_renderer.cuda.soft_rasterize: its two entry points (soft_rasterize_cuda.cpp:59-76, 94-114) are
      served by oracle/sr_oracle.c, which tests/test_oracle_vs_reference_vectors.py holds to the reference's own kernels;
  arithmetic stand-ins for three third-party functions whose sources are not in the image (oracle/path_oracle.py restates
      their published definitions; the same three the oracle uses): kornia.quaternion_to_rotation_matrix,
      pytorch3d.loss.point_mesh_face_distance, pytorch3d.loss.chamfer_distance; and chamfer3D's nearest-neighbour index
      (brute force; the product's kernel is held to chamfer3D's own kernel in tests/test_side_kernels_vs_reference_vectors.py).

tests/test_forward_oracle_vs_reference_golden.py holds oracle/lasr_forward_oracle.py to this fixture on the CPU.
"""
import ctypes
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

MANIFEST = {
    'executed': ['nnutils/mesh_net.py LASR.forward :152-556, render_flow_soft_2 :75-104, reg_decay :106-113',
                 'third_party/ext_nnutils/mesh_net.py MeshNet.__init__/get_mean_shape/symmetrize',
                 'nnutils/geom_utils.py obj_to_cam, pinhole_cam', 'nnutils/loss_utils.py ARAPLoss',
                 'third_party/ext_nnutils/loss_utils.py LaplacianLoss, FlattenLoss',
                 'third_party/ext_utils/mesh.py, meshzoo.py, util_rot.py',
                 'third_party/softras/soft_renderer/*.py (renderer, lighting, transform, rasterizer, mesh, functional/*)'],
    'placeholder_modules': ['absl', 'absl.app', 'absl.flags', 'torchvision', 'trimesh', 'skimage', 'skimage.io', 'cv2', 'png',
                            'soft_renderer.cuda.load_textures', 'soft_renderer.cuda.create_texture_image',
                            'soft_renderer.cuda.voxelization'],
    'injected_out_of_scope': ['ext_nnutils.net_blocks.Encoder / CodePredictor -> the fixture\'s code_* leaf tensors',
                              'ptex_loss.forward_pair (perceptual network) -> zeros'],
    'environment': 'tmp/sphere_N.npy (MeshNet.__init__ cache) written as an object array: numpy 2 rejects the ragged np.save of :75',
    'device_semantics': 'Tensor.__setitem__ clones its right-hand side: mesh_net.py:281 copies a tensor onto itself through a transposed view (read-before-write on CUDA = the transpose the code means; a sequential CPU copy corrupts it)',
    'extension': 'soft_renderer.cuda.soft_rasterize.{forward,backward}_soft_rasterize -> oracle/sr_oracle.c (fp32)',
    'arithmetic_stand_ins': ['kornia.quaternion_to_rotation_matrix -> oracle.path_oracle.quaternion_to_rotation_matrix',
                             'pytorch3d.loss.point_mesh_face_distance -> oracle.path_oracle.point_mesh_face_distance',
                             'pytorch3d.loss.chamfer_distance -> oracle.path_oracle.chamfer_distance',
                             'chamfer3D.dist_chamfer_3D.chamfer_3DDist (index only) -> brute-force argmin'],
}


def install_placeholders():
    from oracle import path_oracle as po
    from oracle import sr_oracle
    for p in (REF, os.path.join(REF, 'third_party'), os.path.join(REF, 'third_party', 'softras')):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in MANIFEST['placeholder_modules']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['skimage.io'].imread = sys.modules['skimage.io'].imsave = None
    fl = sys.modules['absl.flags']
    for n in ('DEFINE_boolean', 'DEFINE_integer', 'DEFINE_string', 'DEFINE_float', 'DEFINE_bool'):
        setattr(fl, n, lambda *a, **k: None)
    fl.FLAGS = types.SimpleNamespace()
    sys.modules['absl'].app, sys.modules['absl'].flags = sys.modules['absl.app'], fl
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # mesh_net.py:281 `Rmat[:,1:] = Rmat[:,1:].permute(0,1,3,2)` copies a tensor ONTO ITSELF through a transposed view.  The
    # reference runs on CUDA, whose copy kernel loads its elements before it stores them: the statement transposes the bone
    # rotations, which is what the code means (R <- R^T, next to T <- -R rest + T + rest).  The CPU copy is one sequential loop and
    # overwrites elements it still has to read (the result is neither R nor R^T).  Indexed assignment therefore takes a private
    # copy of its right-hand side first, which gives this CPU run the read-before-write semantics of the device.
    _setitem = torch.Tensor.__setitem__
    torch.Tensor.__setitem__ = lambda self, idx, val: _setitem(self, idx, val.clone() if isinstance(val, torch.Tensor) else val)

    # ---- the compiled extension, served by the C oracle (same entry points, same argument order: soft_rasterize.py:55-62, 95-103)
    ext = types.ModuleType('soft_renderer.cuda.soft_rasterize')
    lib = sr_oracle.lib()
    f, i = ctypes.c_float, ctypes.c_int

    def ptr(t):
        assert t.is_contiguous() and t.dtype == torch.float32
        return ctypes.c_void_p(t.data_ptr())

    def forward_soft_rasterize(fv, tx, faces_info, aggrs_info, soft_colors, image_size, near, far, eps, sigma_val,
                               func_dist, dist_eps, gamma_val, func_rgb, func_alpha, tex_type, fill_back):
        N, F = fv.shape[:2]
        fv, tx = fv.contiguous(), tx.contiguous()
        rc = lib.oracle_sr_forward_f32(ptr(fv), ptr(tx), ptr(faces_info), ptr(aggrs_info), ptr(soft_colors), i(N), i(F),
                                       i(tx.numel() // (N * F * 3)), i(int(image_size)), f(float(near)), f(float(far)),
                                       f(float(eps)), f(float(sigma_val)), i(func_dist), f(float(dist_eps)), f(float(gamma_val)),
                                       i(func_rgb), i(func_alpha), i(tex_type), i(1 if fill_back else 0))
        assert rc == 0
        return faces_info, aggrs_info, soft_colors

    def backward_soft_rasterize(fv, tx, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors,
                                image_size, near, far, eps, sigma_val, func_dist, dist_eps, gamma_val, func_rgb, func_alpha,
                                tex_type, fill_back):
        N, F = fv.shape[:2]
        fv, tx, g = fv.contiguous(), tx.contiguous(), grad_soft_colors.contiguous()
        rc = lib.oracle_sr_backward_f32(ptr(fv), ptr(tx), ptr(soft_colors), ptr(faces_info), ptr(aggrs_info), ptr(grad_faces),
                                        ptr(grad_textures), ptr(g), i(N), i(F), i(tx.numel() // (N * F * 3)), i(int(image_size)),
                                        f(float(near)), f(float(far)), f(float(eps)), f(float(sigma_val)), i(func_dist),
                                        f(float(dist_eps)), f(float(gamma_val)), i(func_rgb), i(func_alpha), i(tex_type),
                                        i(1 if fill_back else 0))
        assert rc == 0
        return grad_faces, grad_textures
    ext.forward_soft_rasterize, ext.backward_soft_rasterize = forward_soft_rasterize, backward_soft_rasterize
    sys.modules.setdefault('soft_renderer.cuda', types.ModuleType('soft_renderer.cuda'))
    sys.modules['soft_renderer.cuda.soft_rasterize'] = ext

    # ---- third-party arithmetic (not in the image): the published definitions, as restated in oracle/path_oracle.py
    kornia = types.ModuleType('kornia')
    kornia.quaternion_to_rotation_matrix = po.quaternion_to_rotation_matrix
    sys.modules['kornia'] = kornia
    p3 = types.ModuleType('pytorch3d')
    p3.loss = types.ModuleType('pytorch3d.loss')
    p3.structures = types.ModuleType('pytorch3d.structures')
    p3.structures.meshes = types.ModuleType('pytorch3d.structures.meshes')

    class Meshes(object):
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    class Pointclouds(object):
        def __init__(self, points):
            self.points = points
    p3.structures.meshes.Meshes = p3.structures.Meshes = Meshes
    p3.structures.Pointclouds = Pointclouds
    p3.loss.point_mesh_face_distance = lambda meshes, pcls: po.point_mesh_face_distance(meshes.verts, meshes.faces[0], pcls.points)
    p3.loss.chamfer_distance = lambda a, b: (po.chamfer_distance(a, b), None)
    for n, m in (('pytorch3d', p3), ('pytorch3d.loss', p3.loss), ('pytorch3d.structures', p3.structures),
                 ('pytorch3d.structures.meshes', p3.structures.meshes)):
        sys.modules[n] = m

    # ---- out of scope: encoder + code predictor (their outputs are injected)
    nb = types.ModuleType('ext_nnutils.net_blocks')

    class Encoder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, img):
            return img

    class CodePredictor(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.code = None

        def forward(self, feat):
            return tuple(c * 1.0 for c in self.code)          # non-leaf copies: the reference edits them in place
    nb.Encoder, nb.CodePredictor = Encoder, CodePredictor
    sys.modules['ext_nnutils.net_blocks'] = nb


def make_case(name, seed, B, H, K, IS, subdivide, symmetric, use_gtpose, epoch):
    """One fixture case: the options, parameters, code and batch are generated here; everything after them is the reference."""
    from nnutils import mesh_net, loss_utils
    from ext_nnutils import loss_utils as ext_loss
    rng = np.random.default_rng(seed)
    opts = types.SimpleNamespace(
        symmetric=symmetric, symmetric_texture=True, symmetric_loss=True, subdivide=subdivide, symidx=0, only_mean_sym=False,
        opt_tex='yes', dataname='none', n_hypo=H, n_bones=K, img_size=IS, noise=True, use_gtpose=use_gtpose, sigval=1e-4,
        l1tex_wt=1.0, n_faces=str(20 * 4 ** subdivide), num_epochs=20, rscale=1.0, local_rank=0)
    # (MeshNet.__init__ :66-75 caches the icosphere as np.save('tmp/sphere_N.npy', [verts, faces]): a ragged list this image's
    # numpy 2 refuses to save.  The cache file is written here with the reference's own create_sphere, as an object array, so the
    # constructor takes its `exists` branch and reads the same two arrays back.)
    from ext_utils import mesh as ref_mesh
    if not os.path.exists('tmp/sphere_%d.npy' % subdivide):
        os.makedirs('tmp', exist_ok=True)
        sv, sf = ref_mesh.create_sphere(subdivide)
        cache = np.empty(2, dtype=object)
        cache[0], cache[1] = sv, sf
        np.save('tmp/sphere_%d.npy' % subdivide, cache, allow_pickle=True)
    model = mesh_net.LASR((IS, IS), opts, nz_feat=8)
    model.train()
    model.epoch, model.iters, model.optim_idx = epoch, 1, 0
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    P = {}
    with torch.no_grad():
        model.mean_v.data = model.mean_v.data * t(rng.uniform(0.85, 1.15, model.mean_v.shape)) + t(rng.normal(0, 0.02, model.mean_v.shape))
        model.tex.data = t(rng.normal(0, 1, model.tex.shape))
        if K > 1:
            q = rng.normal(0, 1, model.ctl_rs.shape); q[:, 3] += 3.0
            model.ctl_rs.data = t(q / np.linalg.norm(q, axis=1, keepdims=True))
            model.rest_ts.data = t(rng.uniform(-0.4, 0.4, model.rest_ts.shape))
            model.ctl_ts.data = t(rng.uniform(-0.5, 0.5, model.ctl_ts.shape))
            model.log_ctl.data = t(rng.uniform(-0.5, 0.5, model.log_ctl.shape))
    for n in ('mean_v', 'tex') + (('ctl_rs', 'rest_ts', 'ctl_ts', 'log_ctl') if K > 1 else ()):
        P[n] = getattr(model, n)
    # the loss objects as the trainer installs them (nnutils/train_utils.py:113-123)
    mean_v, tex, faces = model.get_mean_shape(B)
    model.triangle_loss_fn_sr = ext_loss.LaplacianLoss(mean_v[0].detach(), faces[0])
    model.arap_loss_fn = loss_utils.ARAPLoss(mean_v[0].detach(), faces[0])
    model.flatten_loss = ext_loss.FlattenLoss(faces[0])
    model.ptex_loss = types.SimpleNamespace(forward_pair=lambda a, b: torch.zeros(a.shape[0]))
    model.chamLoss = lambda a, b: (None, None, (a[:, :, None] - b[:, None]).pow(2).sum(-1).argmin(2), None)

    # the code predictor's output: a camera that puts the unit-sized shape well inside the crop
    def quats(n):
        R = []
        for _ in range(n):
            a = rng.normal(0, 1, (3, 3)); qq, rr = np.linalg.qr(a); qq = qq * np.sign(np.diag(rr)); qq[:, 0] *= np.linalg.det(qq)
            R.append(qq)
        return np.stack(R)
    body = quats(2 * B * H)
    small = np.stack([np.eye(3) + 0.15 * rng.normal(0, 1, (3, 3)) for _ in range(2 * B * H * max(K - 1, 1))])
    small = np.stack([np.linalg.qr(m)[0] * np.sign(np.diag(np.linalg.qr(m)[1])) for m in small])
    quat = np.zeros((2 * B * H, K, 3, 3))
    quat[:, 0] = body
    if K > 1:
        quat[:, 1:] = small.reshape(2 * B * H, K - 1, 3, 3)
    depth = np.concatenate([rng.uniform(7, 9, (2 * B, 1)), rng.uniform(-0.05, 0.05, (2 * B, K - 1))], 1)
    code = dict(scale=rng.uniform(4.2, 5.2, (2 * B, H)), trans=np.concatenate([rng.uniform(-0.1, 0.1, (2 * B, 1, 2)), rng.uniform(-0.03, 0.03, (2 * B, K - 1, 2))], 1).reshape(2 * B * K, 2),
                quat=quat.reshape(2 * B * H * K, 9), depth=depth, ppoint=rng.uniform(-0.05, 0.05, (2 * B, 2)))
    code = {k: t(v).requires_grad_(True) for k, v in code.items()}
    model.code_predictor.code = (code['scale'], code['trans'], code['quat'], code['depth'], code['ppoint'])

    # the trainer's batch, loader order (pair-interleaved, keys of nnutils/train_utils.py:164-178)
    yy, xx = np.meshgrid(np.linspace(-1, 1, IS), np.linspace(-1, 1, IS), indexing='ij')
    masks = np.stack([((xx - rng.uniform(-.1, .1)) ** 2 / 0.45 ** 2 + (yy - rng.uniform(-.1, .1)) ** 2 / 0.6 ** 2 < 1) for _ in range(2 * B)]).astype(np.float32)
    occ = rng.uniform(0.2, 2.0, (2 * B, IS, IS)).astype(np.float32)
    occ[rng.uniform(0, 1, occ.shape) < 0.15] = 0
    cams = np.concatenate([rng.uniform(0.9, 1.2, (2 * B, 1)), rng.uniform(-0.1, 0.1, (2 * B, 2)), np.tile([[1., 0, 0, 0]], (2 * B, 1)) + rng.normal(0, 0.2, (2 * B, 4))], 1)
    cams[:, 3:] /= np.linalg.norm(cams[:, 3:], axis=1, keepdims=True)
    batch = {
        'input_imgs  ': rng.uniform(0, 1, (2 * B, 3, IS, IS)), 'imgs        ': rng.uniform(0, 1, (2 * B, 3, IS, IS)),
        'masks       ': masks, 'cams        ': cams, 'depth_gt    ': rng.uniform(7, 9, (2 * B, 1)),
        'flow        ': np.concatenate([rng.normal(0, 0.05, (2 * B, 2, IS, IS)), np.ones((2 * B, 1, IS, IS))], 1),
        'dts_barrier ': rng.uniform(0, 1, (2 * B, 1, IS, IS)), 'ddts_barrier': rng.uniform(0, 1, (2 * B, 1, IS, IS)),
        'mask_contour': rng.uniform(0, 1, (2 * B, 1, IS, IS)), 'pp          ': rng.uniform(IS / 2 - 2, IS / 2 + 2, (2 * B, 2)),
        'occ         ': occ, 'oriimg_shape': np.tile([[IS * 1.5, IS * 1.25]], (2 * B, 1)),
        'frameid': np.arange(2 * B).reshape(2 * B, 1), 'dataid': np.zeros((2 * B, 1)), 'is_canonical': np.zeros((2 * B, 1))}
    batch_t = {k: (torch.from_numpy(np.asarray(v)).long() if k in ('frameid', 'dataid', 'is_canonical') else t(v)) for k, v in batch.items()}
    if use_gtpose:                                                # :247: depth = self.depth_gt[:] must line up with [2B*H*K, 1]
        assert H == 1 and K == 1
    total, aux = model({k: v.clone() for k, v in batch_t.items()})
    total.backward()

    out = {'cfg': json.dumps(dict(n_hypo=H, n_bones=K, img_size=IS, subdivide=subdivide, num_epochs=opts.num_epochs, l1tex_wt=1.0,
                                  sigval=1e-4, symmetric=symmetric, symmetric_loss=True, opt_tex=True, use_gtpose=use_gtpose,
                                  epoch=epoch, iters=1, symidx=0, B=B,
                                  num_indept=int(getattr(model, 'num_indept', 0)), num_sym=int(getattr(model, 'num_sym', 0)),
                                  eye=[float(v) for v in model.renderer_softtex.transform.transformer._eye]))}
    out['faces'] = model.faces.numpy().astype(np.int64)
    for k, v in P.items():
        out['P_' + k] = v.detach().numpy().copy()
        out['gP_' + k] = v.grad.numpy().copy()
    for k, v in code.items():
        out['code_' + k] = v.detach().numpy().copy()
        out['gcode_' + k] = v.grad.numpy().copy()
    for k, v in batch_t.items():
        out['batch_' + k.strip()] = v.numpy().copy()
    out['total_loss'] = np.float32(total.item())
    for n in ('mask_loss_sub', 'flow_rd_loss_sub', 'texture_loss_sub', 'triangle_loss_sub', 'cam_loss') + \
            (('lmotion_loss_sub', 'arap_loss') if K > 1 else ()):
        out['ref_' + n] = getattr(model, n).detach().numpy().copy()
    for n in ('flow_rd_map', 'flow_rd', 'vis_mask', 'mask_pred'):
        out['aux_' + n] = aux[n].detach().numpy().copy()
    out['aux_texture_render'] = model.texture_render.detach().numpy().copy()      # (aux_output's own entry sits behind part_render's try)
    out['aux_deform_v'] = model.deform_v.detach().numpy().copy()
    out['near_far'] = np.float32([float(model.renderer_softtex.rasterizer.near), float(model.renderer_softtex.rasterizer.far)])
    print('%-28s total_loss %.6f  mask %.4f flow %.4f tex %.4f tri %.4f  covered px %.2f' % (
        name, total.item(), model.mask_loss.item(), model.flow_rd_loss.item(), model.texture_loss.item(), model.triangle_loss.item(),
        float((aux['mask_pred'] > 0.5).float().mean())))
    return {name + '/' + k: v for k, v in out.items()}


def main():
    install_placeholders()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    here = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:          # MeshNet.__init__ caches tmp/sphere_N.npy under the working directory
        os.chdir(tmp)
        try:
            data = {}
            data.update(make_case('articulated_two_hypotheses', 11, B=1, H=2, K=3, IS=24, subdivide=1, symmetric=True, use_gtpose=False, epoch=0))
            data.update(make_case('single_hypothesis_unsymmetric', 12, B=1, H=1, K=3, IS=24, subdivide=1, symmetric=False, use_gtpose=False, epoch=3))
            data.update(make_case('rigid_ground_truth_cameras', 13, B=2, H=1, K=1, IS=20, subdivide=1, symmetric=True, use_gtpose=True, epoch=5))
        finally:
            os.chdir(here)
    data['manifest'] = np.frombuffer(json.dumps(MANIFEST, indent=1).encode(), np.uint8)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'lasr_forward.npz'), **data)
    print('wrote', os.path.join(OUT, 'lasr_forward.npz'), '%d arrays' % len(data))


if __name__ == '__main__':
    main()
