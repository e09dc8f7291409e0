#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the Python reference (CPU, this container only).

    python oracle/gen_golden.py            # needs /root/reference; never runs on the GPU box

TEST INFRASTRUCTURE.  The fixtures hold inputs + expected outputs only (data, no source).
What can be pinned this way is the Python part of the hot path:
  - soft_renderer pre-raster pipeline (lighting -> look_at -> orthogonal/perspective ->
    face_vertices / face_textures), vertex_normals, surface_normals
  - nnutils.geom_utils.obj_to_cam / pinhole_cam (+ autograd gradients)
  - ext_nnutils.loss_utils.LaplacianLoss / FlattenLoss and nnutils.loss_utils.ARAPLoss (+ gradients)
  - ext_utils.meshzoo.iso_sphere (the mesh of the SURVEY App. B known-answer test)
  - ext_utils.{mesh.make_symmetric, util_flow.readPFM/write_pfm, image.compute_dt*, util_rot geodesic distance}
The compiled CUDA extension modules (soft_renderer.cuda.*) and skimage are absent here; they
are replaced by empty placeholder modules so that `import soft_renderer` succeeds -- none of the
functions captured below call into them (the rasterise call itself is intercepted to record its
inputs; the CUDA kernels themselves are pinned separately: oracle/build_ref.py + oracle/gen_ref_vectors.py).
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def import_reference():
    for p in (REF, os.path.join(REF, 'third_party'), os.path.join(REF, 'third_party', 'softras')):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in ('soft_renderer.cuda', 'soft_renderer.cuda.soft_rasterize', 'soft_renderer.cuda.load_textures',
                 'soft_renderer.cuda.create_texture_image', 'soft_renderer.cuda.voxelization', 'skimage', 'skimage.io',
                 'cv2', 'png'):          # imported at module level by ext_utils/{image,util_flow}.py, not used below
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['skimage.io'].imread = sys.modules['skimage.io'].imsave = None
    torch.Tensor.cuda = lambda self, *a, **k: self          # ARAPLoss.forward calls .cuda() (loss_utils.py:49-50)
    import soft_renderer as sr
    from nnutils import geom_utils, loss_utils
    from ext_nnutils import loss_utils as ext_loss
    from ext_utils import meshzoo
    return sr, geom_utils, loss_utils, ext_loss, meshzoo


def t(a):
    return torch.from_numpy(np.asarray(a))


def main():
    os.makedirs(OUT, exist_ok=True)
    sr, geom_utils, loss_utils, ext_loss, meshzoo = import_reference()
    import soft_renderer.functional as srf
    rng = np.random.default_rng(1234)

    # ---- 1. icospheres of the reference (ordering differs from lasr_amd.synth) -------------
    v2, f2 = meshzoo.iso_sphere(2)
    v3, f3 = meshzoo.iso_sphere(3)
    np.savez_compressed(os.path.join(OUT, 'meshzoo_icosphere.npz'), v2=v2.astype(np.float32), f2=f2.astype(np.int64),
                        v3=v3.astype(np.float32), f3=f3.astype(np.int64))

    # ---- 2. pre-raster pipeline: capture what reaches srf.soft_rasterize ----------------
    captured = {}

    def fake_rasterize(face_vertices, textures, *args, **kw):
        captured['fv'] = face_vertices.detach().numpy().copy()
        captured['ft'] = textures.detach().numpy().copy()
        captured['args'] = args
        return torch.zeros(face_vertices.shape[0], 4, 4, 4)

    real_soft_rasterize = srf.soft_rasterize
    srf.soft_rasterize = fake_rasterize
    import soft_renderer.rasterizer as sr_rast
    sr_rast.srf.soft_rasterize = fake_rasterize

    V, F = v2.shape[0], f2.shape[0]
    verts = (v2[None].repeat(2, 0) * 0.5 + 0.03 * rng.standard_normal((2, V, 3))).astype(np.float32)
    verts[..., 2] += 3.0
    faces = f2[None].repeat(2, 0).astype(np.int64)
    vtex = rng.uniform(0, 1, (2, V, 3)).astype(np.float32)
    stex = rng.uniform(0, 1, (2, F, 4, 3)).astype(np.float32)
    pre = dict(verts=verts, faces=faces, vtex=vtex, stex=stex)

    # (a) LASR's renderer configuration (mesh_net.py:136-138) incl. its eye offset + y flip (mesh_net.py:81-82)
    r = sr.SoftRenderer(image_size=32, sigma_val=1e-4, gamma_val=1e-2, camera_mode='look_at', perspective=False,
                        aggr_func_rgb='softmax', light_mode='vertex', light_intensity_ambient=1.,
                        light_intensity_directionals=0.)
    eye = np.asarray(r.transform.transformer._eye, np.float32)
    vpre = verts + eye[None, None]
    vpre[:, :, 1] *= -1
    r.render_mesh(sr.Mesh(t(vpre.copy()), t(faces), textures=t(vtex), texture_type='vertex'))
    pre.update(lasr_eye=eye, lasr_vpre=vpre, lasr_fv=captured['fv'], lasr_ft=captured['ft'])

    # (b) default lighting (ambient .5 + directional .5), perspective look_at, vertex textures
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=30, light_mode='vertex')
    r.render_mesh(sr.Mesh(t(verts - np.float32([0, 0, 3])), t(faces), textures=t(vtex), texture_type='vertex'))
    pre.update(persp_vertex_fv=captured['fv'], persp_vertex_ft=captured['ft'])

    # (c) surface textures + surface lighting, look_at from an elevated eye (camera_mode='look' is
    #     unusable in the reference: Transform passes Look() its arguments in the wrong order, transform.py:85)
    eye_c = srf.get_points_from_angles(2.732, 30., 40.)
    r = sr.SoftRenderer(image_size=32, camera_mode='look_at', perspective=True, viewing_angle=25, light_mode='surface',
                        light_directions=[0.3, 0.8, -0.5], eye=list(eye_c))
    r.render_mesh(sr.Mesh(t(verts - np.float32([0, 0, 3])), t(faces), textures=t(stex), texture_type='surface'))
    pre.update(look_surface_fv=captured['fv'], look_surface_ft=captured['ft'])

    m = sr.Mesh(t(verts), t(faces), textures=t(vtex), texture_type='vertex')
    pre.update(vertex_normals=m.vertex_normals.numpy(), surface_normals=m.surface_normals.numpy(),
               face_vertices=m.face_vertices.numpy())
    pre.update(points_from_angles=np.asarray(srf.get_points_from_angles(2.732, 30., 40.), np.float32))
    np.savez_compressed(os.path.join(OUT, 'softras_pre_raster.npz'), **pre)

    # ---- 3. obj_to_cam / pinhole_cam with gradients (geom_utils.py:27-71) ------------------
    g = {}
    N, Vn, K, H = 4, 50, 5, 2           # N = 2B*H meshes, K bones (1 body + 4 parts)
    pv = t(rng.standard_normal((N, Vn, 3)).astype(np.float32)).requires_grad_(True)
    Rm = t(rng.standard_normal((N * K, 3, 3)).astype(np.float32)).requires_grad_(True)
    Tm = t(rng.standard_normal((N * K, 1, 3)).astype(np.float32)).requires_grad_(True)
    sk = torch.softmax(t(rng.standard_normal((N, K - 1, Vn, 1)).astype(np.float32)), 1).requires_grad_(True)
    up = t(rng.standard_normal((N, Vn, 3)).astype(np.float32))
    for tocam in (True, False):
        out = geom_utils.obj_to_cam(pv, Rm, Tm, K, H, sk, tocam=tocam)
        grads = torch.autograd.grad((out * up).sum(), [pv, Rm, Tm, sk], allow_unused=True)
        tag = 'cam' if tocam else 'obj'
        g['o2c_%s_out' % tag] = out.detach().numpy()
        for name, gr, ref in zip(('verts', 'Rmat', 'Tmat', 'skin'), grads, (pv, Rm, Tm, sk)):
            g['o2c_%s_g_%s' % (tag, name)] = (gr if gr is not None else torch.zeros_like(ref)).numpy()
    # single bone (nmesh == 1): skin unused
    out1 = geom_utils.obj_to_cam(pv, Rm[:N], Tm[:N], 1, H, None)
    g['o2c_k1_out'] = out1.detach().numpy()
    g.update(o2c_verts=pv.detach().numpy(), o2c_Rmat=Rm.detach().numpy(), o2c_Tmat=Tm.detach().numpy(),
             o2c_skin=sk.detach().numpy(), o2c_up=up.numpy(), o2c_K=K, o2c_H=H)
    v4 = t(np.concatenate([rng.standard_normal((N, Vn, 2)), rng.uniform(2, 5, (N, Vn, 1)),
                           np.ones((N, Vn, 1))], -1).astype(np.float32)).requires_grad_(True)
    pp = t(rng.uniform(-0.1, 0.1, (N // H, 2)).astype(np.float32)).requires_grad_(True)
    fl = t(rng.uniform(2, 4, (N // H, H)).astype(np.float32)).requires_grad_(True)
    up4 = t(rng.standard_normal((N, Vn, 4)).astype(np.float32))
    outp = geom_utils.pinhole_cam(v4, pp, fl)
    gp = torch.autograd.grad((outp * up4).sum(), [v4, pp, fl])
    g.update(pin_verts=v4.detach().numpy(), pin_pp=pp.detach().numpy(), pin_fl=fl.detach().numpy(), pin_up=up4.numpy(),
             pin_out=outp.detach().numpy(), pin_g_verts=gp[0].numpy(), pin_g_pp=gp[1].numpy(), pin_g_fl=gp[2].numpy())
    np.savez_compressed(os.path.join(OUT, 'geom_utils.npz'), **g)

    # ---- 4. mesh regularisers ---------------------------------------------------------------
    L = {}
    base = t(v2.astype(np.float32))
    fcs = t(f2.astype(np.int64))
    x = t((v2[None].repeat(3, 0) + 0.05 * rng.standard_normal((3, V, 3))).astype(np.float32)).requires_grad_(True)
    dxv = t((v2[None].repeat(3, 0) + 0.05 * rng.standard_normal((3, V, 3))).astype(np.float32)).requires_grad_(True)
    lap = ext_loss.LaplacianLoss(base, fcs)
    fla = ext_loss.FlattenLoss(fcs)
    arap = loss_utils.ARAPLoss(base, fcs)
    wl = t(rng.uniform(0.5, 1.5, 3).astype(np.float32))
    for name, fn, args in (('lap', lap, (x,)), ('flat', fla, (x,)), ('arap', arap, (dxv, x))):
        out = fn(*args)
        grads = torch.autograd.grad((out * wl).sum(), list(args))
        L[name + '_out'] = out.detach().numpy()
        for i, gr in enumerate(grads):
            L['%s_g%d' % (name, i)] = gr.numpy()
    L.update(x=x.detach().numpy(), dx=dxv.detach().numpy(), w=wl.numpy(), base=v2.astype(np.float32), faces=f2.astype(np.int64))
    np.savez_compressed(os.path.join(OUT, 'mesh_losses.npz'), **L)

    # ---- 5. host-side utilities either side of the path ----------
    import tempfile
    from ext_utils import image as ref_image
    from ext_utils import mesh as ref_mesh
    from ext_utils import util_flow as ref_flow
    from ext_utils import util_rot as ref_rot
    E = {}
    sv, sf, n_ind, n_sym, _, _, order = ref_mesh.make_symmetric(v2, f2, 0)
    E.update(sym_verts=sv.astype(np.float32), sym_faces=sf.astype(np.int64), sym_counts=np.array([n_ind, n_sym]),
             sym_order=order.astype(np.int64))
    flow = rng.standard_normal((7, 5, 3)).astype(np.float32)
    occ = rng.standard_normal((7, 5)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        for name, arr in (('flow', flow), ('occ', occ)):
            ref_flow.write_pfm(os.path.join(d, name + '.pfm'), arr)
            E['pfm_%s_bytes' % name] = np.frombuffer(open(os.path.join(d, name + '.pfm'), 'rb').read(), np.uint8)
            back, scale = ref_flow.readPFM(os.path.join(d, name + '.pfm'))
            E['pfm_%s_read' % name] = np.ascontiguousarray(back)
            assert scale == 1.0
    E.update(pfm_flow=flow, pfm_occ=occ)
    mask = np.zeros((40, 48))
    mask[8:30, 10:37] = 1
    mask[12:18, 20:25] = 0
    E.update(dt_mask=mask, dt0=ref_image.compute_dt(mask, iters=0), dt10=ref_image.compute_dt(mask, iters=10),
             dt_barrier=ref_image.compute_dt_barrier(mask, k=50))
    qa = rng.standard_normal((6, 4))
    qb = qa + 0.3 * rng.standard_normal((6, 4))

    def rot(q):                                              # plain numpy, only to make valid rotation inputs
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        x, y, z, w = q.T
        return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w),
                         1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w),
                         1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3).astype(np.float32)
    ra, rb = rot(qa), rot(qb)
    E.update(rot_a=ra, rot_b=rb, rot_geodesic=ref_rot.compute_geodesic_distance_from_two_matrices(t(ra), t(rb)).numpy())
    np.savez_compressed(os.path.join(OUT, 'ext_utils.npz'), **E)

    # ---- 6. public API surface of the operator package and the geometry / loss helpers (names + call signatures) -------
    import inspect
    import json

    def sig(obj):
        target = obj.__init__ if inspect.isclass(obj) else obj
        params = list(inspect.signature(target).parameters.values())
        if inspect.isclass(obj):
            params = params[1:]
        return [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in params]
    api = {}
    for mod_name, mod, names in (
            ('soft_renderer', sr, ['SoftRenderer', 'SoftRasterizer', 'Mesh', 'Lighting', 'AmbientLighting', 'DirectionalLighting',
                                   'Transform', 'LookAt', 'Look', 'Projection']),
            ('soft_renderer.functional', srf, ['soft_rasterize', 'face_vertices', 'vertex_normals', 'look_at', 'look', 'orthogonal',
                                               'perspective', 'projection', 'ambient_lighting', 'directional_lighting',
                                               'get_points_from_angles', 'load_obj', 'save_obj']),
            ('nnutils.geom_utils', geom_utils, ['obj_to_cam', 'pinhole_cam', 'orthographic_cam']),
            ('nnutils.loss_utils', loss_utils, ['ARAPLoss']),
            ('ext_nnutils.loss_utils', ext_loss, ['LaplacianLoss', 'FlattenLoss'])):
        for n in names:
            api['%s.%s' % (mod_name, n)] = sig(real_soft_rasterize if n == 'soft_rasterize' else getattr(mod, n))
    api['soft_renderer.SoftRenderer.methods'] = sorted(n for n in vars(sr.SoftRenderer) if not n.startswith('_'))
    api['soft_renderer.Mesh.methods'] = sorted(n for n in dir(sr.Mesh) if not n.startswith('_'))
    with open(os.path.join(OUT, 'api_surface.json'), 'w') as fh:
        json.dump(api, fh, indent=1, sort_keys=True)

    # ---- 7. command-line flags of optimize.py (absl definitions scattered over four modules; parsed, not imported) -------
    import ast
    import re
    flags_found = {}
    for rel in ('optimize.py', 'nnutils/mesh_net.py', 'nnutils/train_utils.py', 'dataloader/vid.py'):
        src = open(os.path.join(REF, rel)).read()
        for kind, name, default in re.findall(r"flags\.DEFINE_(\w+)\(\s*'(\w+)'\s*,\s*([^,]+),", src):
            flags_found[name] = [kind, ast.literal_eval(default.strip())]
    with open(os.path.join(OUT, 'cli_flags.json'), 'w') as fh:
        json.dump(flags_found, fh, indent=1, sort_keys=True)

    for n in sorted(os.listdir(OUT)):
        print('%-28s %8d bytes' % (n, os.path.getsize(os.path.join(OUT, n))))


if __name__ == '__main__':
    main()
