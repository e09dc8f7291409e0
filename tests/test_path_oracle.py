"""oracle/path_oracle.py (torch restatements of the Python path pieces) against fixtures captured from the
imported reference, plus hand-computable cases for the loss tables that cannot be captured."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import path_oracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GEO = np.load(os.path.join(G, 'geom_utils.npz'))
ML = np.load(os.path.join(G, 'mesh_losses.npz'))


def t(a, grad=False):
    return torch.from_numpy(np.asarray(a)).clone().requires_grad_(grad)


def test_obj_to_cam_matches_reference_outputs_and_gradients():
    K, H = int(GEO['o2c_K']), int(GEO['o2c_H'])
    for tag, tocam in (('cam', True), ('obj', False)):
        v, R, T, s = (t(GEO['o2c_' + n], True) for n in ('verts', 'Rmat', 'Tmat', 'skin'))
        out = po.obj_to_cam(v, R, T, K, H, s, tocam=tocam)
        np.testing.assert_allclose(out.detach().numpy(), GEO['o2c_%s_out' % tag], atol=1e-6)
        grads = torch.autograd.grad((out * t(GEO['o2c_up'])).sum(), [v, R, T, s], allow_unused=True)
        for name, g in zip(('verts', 'Rmat', 'Tmat', 'skin'), grads):
            ref = GEO['o2c_%s_g_%s' % (tag, name)]
            np.testing.assert_allclose(np.zeros_like(ref) if g is None else g.numpy(), ref, atol=2e-5)
    N = GEO['o2c_verts'].shape[0]
    out1 = po.obj_to_cam(t(GEO['o2c_verts']), t(GEO['o2c_Rmat'])[:N], t(GEO['o2c_Tmat'])[:N], 1, H, None)
    np.testing.assert_allclose(out1.numpy(), GEO['o2c_k1_out'], atol=1e-6)


def test_pinhole_cam_matches_reference():
    v, pp, fl = t(GEO['pin_verts'], True), t(GEO['pin_pp'], True), t(GEO['pin_fl'], True)
    out = po.pinhole_cam(v, pp, fl)
    np.testing.assert_allclose(out.detach().numpy(), GEO['pin_out'], atol=1e-6)
    g = torch.autograd.grad((out * t(GEO['pin_up'])).sum(), [v, pp, fl])
    for a, name in zip(g, ('verts', 'pp', 'fl')):
        np.testing.assert_allclose(a.numpy(), GEO['pin_g_' + name], rtol=1e-5, atol=1e-5)


def test_mesh_regularisers_match_reference():
    x, dx = t(ML['x'], True), t(ML['dx'], True)
    w = t(ML['w'])
    lap = po.laplacian(x, ML['faces'])
    np.testing.assert_allclose(lap.detach().numpy(), ML['lap_out'], rtol=1e-5)
    np.testing.assert_allclose(torch.autograd.grad((lap * w).sum(), x)[0].numpy(), ML['lap_g0'], atol=1e-5)
    ar = po.arap(dx, x, ML['faces'])
    np.testing.assert_allclose(ar.detach().numpy(), ML['arap_out'], rtol=1e-5)
    g = torch.autograd.grad((ar * w).sum(), [dx, x])
    np.testing.assert_allclose(g[0].numpy(), ML['arap_g0'], atol=1e-6)
    np.testing.assert_allclose(g[1].numpy(), ML['arap_g1'], atol=1e-6)


def test_mask_loss_hand_case():
    pred = torch.tensor([[[[0.5, 1.0], [0.0, 0.25]]]])            # I=1,H=1,2x2
    masks = torch.tensor([[[1.0, 1.0], [0.0, 1.0]]])
    occ = torch.tensor([[[1.0, 0.0], [2.0, -1.0]]])               # pixel (0,1) is invalid
    # selected squared errors: .25, 0, .5625 -> mean .27083.. -> x0.5
    assert abs(po.mask_loss_table(pred, masks, occ).item() - 0.5 * (0.25 + 0 + 0.5625) / 3) < 1e-7


def test_flow_loss_hand_case():
    flow_rd = torch.zeros(1, 2, 1, 2, 2)
    flow_rd[0, 0, 0, 0] = torch.tensor([3.0, 4.0])                # error norm 5 at pixel 0, hypothesis 0
    flow_rd[0, 1, 0, 1] = torch.tensor([0.0, 2.0])                # error norm 2 at pixel 1, hypothesis 1
    obs = torch.zeros(1, 3, 1, 2)
    bg = torch.zeros(1, 2, 1, 2, dtype=torch.bool)
    bg[0, 1, 0, 0] = True                                         # hypothesis 1 does not cover pixel 0
    occ = torch.tensor([[[1.0, 2.0]]])
    masks = torch.ones(1, 1, 2)
    s = lambda v: 1 / (1 + np.exp(v))                             # sigmoid(-v)
    wmean = (s(1) + s(2) + s(2)) / 3                              # selected (j,p): (0,0),(0,1),(1,1)
    loss, fmap = po.flow_loss_table(flow_rd, obs, bg, occ, masks)
    assert abs(loss[0, 0].item() - 0.5 * (5 * s(1) / wmean + 0) / 2) < 1e-6
    assert abs(loss[0, 1].item() - 0.5 * (2 * s(2) / wmean)) < 1e-6
    bg[:] = True                                                  # nothing selected -> 0, not NaN (mesh_net.py:412)
    assert po.flow_loss_table(flow_rd, obs, bg, occ, masks)[0].abs().sum().item() == 0


def test_tex_loss_hand_case():
    obs = torch.full((1, 3, 1, 2), 0.5)
    white = torch.ones(1, 3, 1, 2)
    rnd = torch.full((1, 1, 3, 1, 2), 0.25)
    fg = torch.tensor([[[[1.0, 0.0]]]])
    occ = torch.tensor([[[1.0, 1.0]]])
    # |0.5-0.25| and |0.5-0| -> mean .375 ; |1-.25| -> .75 ; (0.375+0.75)*2
    assert abs(po.tex_loss_table(obs, white, rnd, fg, occ).item() - 2.25) < 1e-6


def test_perceptual_pairing_with_shared_observations():
    # mesh_net.py:436-442 feeds H identical copies of every observed image through the feature network; computing
    # the features once per image and repeating them pairs the same (observation, render) couples
    import torch
    from lasr_amd.nnutils.mesh_net import PerceptualDistance
    torch.manual_seed(0)
    net = PerceptualDistance()
    H = 3
    obs = torch.rand(4, 3, 64, 64) * 2 - 1
    rnd = torch.rand(4 * H, 3, 64, 64) * 2 - 1
    full = net.forward_pair(obs[:, None].repeat(1, H, 1, 1, 1).view(-1, 3, 64, 64), rnd)
    shared = net.forward_pair(obs, rnd, repeat=H)
    assert torch.allclose(full, shared, rtol=1e-5, atol=1e-6)


def test_star_remesh_lands_on_the_surface():
    # nnutils/remesh.py (stands in for the Manifold binaries of train_utils.py:419-428): vertices of the new tessellation are
    # ray hits on the old surface, face count = 20 nu^2 nearest to the request
    import numpy as np
    from lasr_amd import synth
    from lasr_amd.nnutils import remesh
    v, f = synth.geodesic_sphere(6)
    v = v * np.array([1.0, 0.6, 0.8], np.float32)                    # an ellipsoid: every ray from the centre hits once
    nv, nf = remesh.remesh_star(v, f, 1600)
    assert nf.shape == (1620, 3) and nv.shape == (812, 3)
    q = (nv / np.array([1.0, 0.6, 0.8])) ** 2                        # on the inscribed polyhedron: just inside the ellipsoid
    r = np.sqrt(q.sum(1))
    assert r.max() <= 1 + 1e-6 and r.min() > 0.97
    t = remesh.ray_mesh_outermost(np.zeros(3), np.array([[0., 0., 1.], [0., 0., -1.]]), v.astype(np.float64), f)
    assert np.allclose(t, 0.8, atol=1e-6)


def test_star_remesh_survives_directions_that_miss_the_surface():
    # an open surface (a hemisphere without its cap): rays through the opening miss; those vertices fall back to the mean hit
    # radius instead of NaN
    import numpy as np
    from lasr_amd import synth
    from lasr_amd.nnutils import remesh
    v, f = synth.geodesic_sphere(5)
    keep = (v[f][:, :, 2] > -0.2).all(1)
    nv, nf = remesh.remesh_star(v, f[keep], 320)
    assert np.isfinite(nv).all() and nf.shape == (320, 3)
    r = np.linalg.norm(nv - nv.mean(0), axis=1)
    assert r.min() > 0.3 and r.max() < 1.5


def test_star_remesh_warns_when_the_surface_is_not_star_shaped():
    # a dumbbell (two blobs joined along x, seen from the centroid of the first one) is crossed more than once by
    # many rays; remesh_star keeps the outermost crossing and must say so.  Snapping --n_faces to 20 nu^2 is reported too.
    import warnings
    from lasr_amd import synth
    from lasr_amd.nnutils import remesh
    v, f = synth.geodesic_sphere(6)
    v = v.astype(np.float64)
    two = np.concatenate([v * 0.5, v * 0.5 + np.array([1.6, 0, 0])])      # disjoint second blob far off-centre
    ff = np.concatenate([f, f + len(v)])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        remesh.remesh_star(two, ff, 1600)
    msgs = ' | '.join(str(x.message) for x in w)
    assert 'not star-shaped' in msgs and 'snapped to 1620' in msgs
    with warnings.catch_warnings(record=True) as w:                        # a convex surface at an exact face count: silent
        warnings.simplefilter('always')
        remesh.remesh_star(v, f, 1620)
    assert not w


def test_pose_noise_follows_the_reference_law():
    # mesh_net.py:220-232 draws a uniformly random rotation (quatlib.q_rnd_m) and slerps it towards the
    # identity by the decay factor (quatlib.q_scale_m).  For t = 1 the angle theta has density (1 - cos theta) / pi on
    # [0, pi] (mean pi/2 + 2/pi); for general t the angle is t * theta and the axis stays uniform on the sphere.
    import math
    import torch
    from lasr_amd.nnutils.mesh_net import pose_noise_quat
    torch.manual_seed(0)
    n = 200000
    for t in (1.0, 0.2, 0.02):
        q = pose_noise_quat(n, t, torch.device('cpu'))
        assert q.shape == (n, 4) and torch.allclose(q.norm(dim=1), torch.ones(n), atol=1e-5)
        ang = 2 * torch.atan2(q[:, :3].norm(dim=1), q[:, 3].abs())
        assert abs(float(ang.mean()) - t * (math.pi / 2 + 2 / math.pi)) < 0.01 * t * 3      # E[theta] = pi/2 + 2/pi
        assert float(ang.max()) <= t * math.pi + 1e-4
        # P(theta <= x) = (x - sin x) / pi for the unscaled angle
        for x in (0.5, 1.5, 2.5):
            emp = float((ang / t <= x).float().mean())
            assert abs(emp - (x - math.sin(x)) / math.pi) < 5e-3, (t, x, emp)
        axis = torch.nn.functional.normalize(q[:, :3], dim=1)
        assert float(axis.mean(0).abs().max()) < 0.01                                        # uniform axis: zero mean ...
        assert abs(float((axis[:, 2] ** 2).mean()) - 1 / 3) < 0.01                           # ... and isotropic second moment
    # reference formulas, statement by statement (quatlib.py:29-50), on the same uniform draws
    u, v, w = (x.numpy() for x in torch.rand(3, 1000, dtype=torch.float64).unbind(0))
    q = np.stack([np.sqrt(1 - u) * np.sin(2 * np.pi * v), np.sqrt(1 - u) * np.cos(2 * np.pi * v),
                  np.sqrt(u) * np.sin(2 * np.pi * w), np.sqrt(u) * np.cos(2 * np.pi * w)], 1)
    q[q[:, 0] < 0] *= -1
    d = q[:, 0].copy()
    t0 = np.arccos(d)
    tt = t0 * 0.3
    s1 = np.sin(tt) / np.sin(t0)
    ref = (np.cos(tt) - d * s1)[:, None] * np.array([1., 0, 0, 0]) + s1[:, None] * q
    ang_ref = 2 * np.arccos(np.clip(ref[:, 0], -1, 1))
    assert np.allclose(ang_ref, 0.3 * 2 * t0, atol=1e-9)       # slerp by t multiplies the rotation angle by t


def test_load_textures_restatement_on_analytic_atlases():
    # load_textures_cuda_kernel.cu:8-66: a constant atlas gives constant texels; on an atlas that is linear in (x, y) the
    # bilinear sample equals the atlas function at uv * (size - 1), i.e. at the texel's barycentric point of the face's uv
    import numpy as np
    from oracle import path_oracle as po
    H, W, R = 9, 17, 3
    yy, xx = np.mgrid[:H, :W].astype(np.float32)
    lin = np.stack([2 * xx + 1, 3 * yy - 2, xx + yy], -1).astype(np.float32)
    uv = np.array([[[0.1, 0.2], [0.9, 0.3], [0.4, 0.8]], [[0, 0], [1, 0], [0, 1]]], np.float32)
    t = po.load_textures(lin, uv, R)
    assert t.shape == (2, R * R, 3)
    for i in range(R * R):
        w_y, w_x = i // R, i % R
        if w_x + w_y < R:
            w0, w1 = (w_x + 1 / 3) / R, (w_y + 1 / 3) / R
        else:
            w0, w1 = ((R - 1 - w_x) + 2 / 3) / R, ((R - 1 - w_y) + 2 / 3) / R
        p = (uv[:, 0] * w0 + uv[:, 1] * w1 + uv[:, 2] * (1 - w0 - w1)) * np.array([W - 1, H - 1])
        np.testing.assert_allclose(t[:, i, 0], 2 * p[:, 0] + 1, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(t[:, i, 1], 3 * p[:, 1] - 2, rtol=1e-5, atol=1e-5)
    const = po.load_textures(np.full((H, W, 3), 0.25, np.float32), uv, R)
    np.testing.assert_allclose(const, 0.25, atol=1e-7)
    skipped = po.load_textures(lin, uv, R, is_update=[1, 0])
    assert np.array_equal(skipped[0], t[0]) and not skipped[1].any()


# ---- restatements behind the glue kernels (lasr_amd/csrc/glue.hip): hand-checkable cases -----------------------------------
def test_intrinsics_restatement_hand_case():
    # uncropped pair (crop scale 1, centre offset 0): focal length and depth pass through, frame t' gets frame t's principal point
    B, H, K = 2, 3, 4
    cams = torch.ones(2 * B, 7)
    pp = torch.zeros(2 * B, 2)
    scale, depth, ppoint = torch.rand(2 * B, H), torch.rand(2 * B, K), torch.rand(2 * B, 2)
    s, d, p = po.intrinsics(cams, pp, scale, depth, ppoint, 256)
    assert torch.equal(s, scale) and torch.equal(d, depth)
    assert torch.allclose(p[B:], ppoint[:B], atol=1e-6) and torch.equal(p[:B], ppoint[:B])
    # a frame t' cropped twice as tightly (scale 2) with a shifted crop centre
    cams[B:, 0] = 2.
    pp[B:] = torch.tensor([12.8, -25.6])
    s, d, p = po.intrinsics(cams, pp, scale, depth, ppoint, 256)
    assert torch.allclose(s[B:], 2 * scale[B:]) and torch.allclose(d[B:, 0], 2 * depth[B:, 0]) and torch.equal(d[:, 1:], depth[:, 1:])
    want = (ppoint[:B] + 1) * 2 - 2 * torch.tensor([12.8, -25.6]) / 128 - 1
    assert torch.allclose(p[B:], want, atol=1e-6)


def test_bone_fixup_restatement_hand_case():
    # one image, one hypothesis, root + one bone; the bone turns 90 degrees about z around the joint c = (1, 0, 0)
    Rz = torch.tensor([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]])
    quat = torch.stack([torch.eye(3), Rz.t()]).reshape(2, 9)            # the head predicts the TRANSPOSE of the applied rotation
    trans, depth = torch.tensor([[0.1, 0.2], [0., 0.]]), torch.tensor([[5.], [0.]])
    rest = torch.tensor([[1., 0., 0.]])
    R, T = po.bone_fixup(quat, trans, depth, rest, 1, 1, 2)
    assert torch.equal(R[0], torch.eye(3)) and torch.equal(T[0], torch.tensor([0.1, 0.2, 5.]))
    assert torch.equal(R[1], Rz.t())                                     # bones are transposed back (mesh_net.py:281)
    assert torch.allclose(T[1], rest[0] - Rz @ rest[0])                  # x -> R (x - c) + c: the joint stays where it is
    assert torch.allclose(Rz @ rest[0] + T[1], rest[0])


def test_geodesic_chamfer_and_loss_sum_restatements_hand_cases():
    th = 0.7
    Rz = torch.tensor([[math.cos(th), -math.sin(th), 0.], [math.sin(th), math.cos(th), 0.], [0., 0., 1.]])
    ang = po.geodesic_distance(torch.stack([torch.eye(3), Rz]), torch.stack([torch.eye(3), torch.eye(3)]))
    assert abs(float(ang[0])) < 1e-3 and abs(float(ang[1]) - th) < 1e-6
    a = torch.tensor([[[0., 0., 0.], [1., 0., 0.]]])
    b = torch.tensor([[[0., 0., 1.], [1., 0., 0.], [5., 0., 0.]]])
    # a -> b: (1 + 0) / 2; b -> a: (1 + 0 + 16) / 3
    assert abs(float(po.chamfer_distance(a, b)) - (0.5 + 17. / 3)) < 1e-6
    total, groups = po.weighted_mean_sum([(torch.tensor([1., 3.]), 2., 0), (torch.tensor([[4.]]), 0.5, 1), (torch.tensor([2., 2., 2.]), 1., 0)])
    assert float(total) == 2 * 2 + 0.5 * 4 + 2 and [float(g) for g in groups] == [6., 2.]


def test_glue_kernels_reject_cpu_tensors():
    from lasr_amd.nnutils import fused_ops
    z = torch.zeros
    calls = [lambda: fused_ops.intrinsics(z(2, 7), z(2, 2), z(2, 1), z(2, 1), z(2, 2), 64),
             lambda: fused_ops.bone_fixup(z(2, 9), z(2, 2), z(2, 1), None, 1, 1),
             lambda: fused_ops.geodesic_distance(z(2, 3, 3), z(2, 3, 3)),
             lambda: fused_ops.chamfer(z(1, 2, 3), z(1, 2, 3)),
             lambda: fused_ops.mean_shape(z(1, 4, 3), z(1, 4, 3), None, None, 2, 0),
             lambda: fused_ops.obs_pair(z(1, 3, 4, 4), z(1, 4, 4)),
             lambda: fused_ops.weighted_mean_sum([(z(3), 1., 0)]),
             lambda: fused_ops.flow_reproject_planes(z(1, 6, 4, 4), z(1, 2), z(1, 2), z(1, 1), z(1, 1))]
    for call in calls:
        with pytest.raises(TypeError):                                   # no CPU fallback behind the HIP operators
            call()


def _write_textured_quad(d):
    """A two-triangle quad with uv coordinates, a two-material .mtl (one atlas image, one flat colour) and a 4x4 atlas."""
    from PIL import Image
    img = (np.arange(4 * 4 * 3).reshape(4, 4, 3) * 5).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(d, 'atlas.png'))
    open(os.path.join(d, 'quad.mtl'), 'w').write('newmtl skin\nKd 0.2 0.4 0.6\nmap_Kd atlas.png\n\nnewmtl flat\nKd 0.9 0.1 0.3\n')
    open(os.path.join(d, 'quad.obj'), 'w').write(
        'mtllib quad.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 0 0\n'
        'vt 0.0 0.0\nvt 1.0 0.0\nvt 1.0 1.0\nvt 0.0 1.0\nvt 1.5 0.25\n'
        'usemtl skin\nf 1/1 2/2 3/3 4/4\nusemtl flat\nf 2/2 5/5 3/3\nf 1 2 3\n')
    return os.path.join(d, 'quad.obj'), img


def test_obj_material_parsing(tmp_path):
    # load_obj.py:28-71: fan triangulation of the uv indices, material per triangle, uv > 1 wraps, a corner without a texture index
    # takes the LAST vt entry (index 0 - 1), Kd colours and atlas paths from the .mtl
    from lasr_amd.soft_renderer.functional import obj_io
    path, _ = _write_textured_quad(str(tmp_path))
    uv, mats, colors, files = obj_io.parse_obj_materials(path)
    assert uv.shape == (4, 3, 2) and mats == ['skin', 'skin', 'flat', 'flat']
    np.testing.assert_allclose(uv[0], [[0, 0], [1, 0], [1, 1]])
    np.testing.assert_allclose(uv[1], [[0, 0], [1, 1], [0, 1]])
    np.testing.assert_allclose(uv[2], [[1, 0], [0.5, 0.25], [1, 1]])          # 1.5 wraps to 0.5
    np.testing.assert_allclose(uv[3], [[0.5, 0.25]] * 3)                       # no texture index: the last vt entry (wrapped)
    assert set(colors) == {'skin', 'flat'} and list(files) == ['skin'] and files['skin'].endswith('atlas.png')
    np.testing.assert_allclose(colors['flat'], [0.9, 0.1, 0.3])
    open(os.path.join(str(tmp_path), 'bare.obj'), 'w').write('v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n')
    with pytest.raises(Exception):
        obj_io.parse_obj_materials(os.path.join(str(tmp_path), 'bare.obj'))


@pytest.mark.skipif(not os.path.exists('/root/reference/database/misc/spot/spot_triangulated.obj'), reason='reference data not present')
def test_obj_material_parsing_on_the_reference_model():
    # the model scripts/render_syn.py:71 of the reference renders (read where it lies; not copied, not needed on the GPU box)
    from lasr_amd.soft_renderer.functional import obj_io
    path = '/root/reference/database/misc/spot/spot_triangulated.obj'
    v, f = obj_io.load_obj(path, device='cpu')
    uv, mats, colors, files = obj_io.parse_obj_materials(path)
    assert v.shape == (2930, 3) and f.shape == (5856, 3) and uv.shape == (5856, 3, 2)
    assert set(mats) == {'material_1'} and os.path.basename(files['material_1']) == 'spot_texture.png' and os.path.exists(files['material_1'])
    assert float(uv.max()) <= 1.0 and float(uv.min()) > -0.1
