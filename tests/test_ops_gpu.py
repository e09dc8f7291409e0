"""HIP geometry / loss kernels (include/lasr_ops.h, called through lasr_amd.nnutils) vs the golden fixtures
captured from the reference and vs the torch oracle on random shapes.  Tolerances are fp32 round-off:
the kernels reassociate sums (MFMA blend, tree reductions) but use the same formulas."""
import os

import numpy as np
import pytest
import torch

from lasr_amd.nnutils import geom_utils, image_losses, loss_utils
from lasr_amd import synth
from oracle import path_oracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GEO = np.load(os.path.join(G, 'geom_utils.npz'))
ML = np.load(os.path.join(G, 'mesh_losses.npz'))


def dev_t(a, dev, grad=False):
    return torch.from_numpy(np.asarray(a)).to(dev).requires_grad_(grad)


def close(a, b, tol, what=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else b
    assert np.array_equal(np.isnan(a), np.isnan(b)), '%s: NaN pattern differs' % what
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    scale = max(float(np.abs(b).max()), 1e-12)
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


def test_obj_to_cam_vs_reference_fixture(cuda):
    K, H = int(GEO['o2c_K']), int(GEO['o2c_H'])
    for tag, tocam in (('cam', True), ('obj', False)):
        v, R, T, s = (dev_t(GEO['o2c_' + n], cuda, True) for n in ('verts', 'Rmat', 'Tmat', 'skin'))
        out = geom_utils.obj_to_cam(v, R, T, K, H, s, tocam=tocam)
        close(out, GEO['o2c_%s_out' % tag], 1e-6, 'obj_to_cam ' + tag)
        (out * dev_t(GEO['o2c_up'], cuda)).sum().backward()
        for name, x in zip(('verts', 'Rmat', 'Tmat', 'skin'), (v, R, T, s)):
            close(x.grad, GEO['o2c_%s_g_%s' % (tag, name)], 1e-5, 'grad ' + name)
    N = GEO['o2c_verts'].shape[0]
    out1 = geom_utils.obj_to_cam(dev_t(GEO['o2c_verts'], cuda), dev_t(GEO['o2c_Rmat'], cuda)[:N],
                                 dev_t(GEO['o2c_Tmat'], cuda)[:N], 1, H, None)
    close(out1, GEO['o2c_k1_out'], 1e-6, 'obj_to_cam K=1')


@pytest.mark.parametrize('N,V,K', [(16, 642, 21), (2, 802, 26), (4, 1212, 36), (3, 17, 2), (1, 1, 5), (2, 70, 1)])
def test_obj_to_cam_vs_oracle_at_lasr_sizes(cuda, N, V, K):
    rng = np.random.default_rng(N * 1000 + V + K)
    v = rng.standard_normal((N, V, 3)).astype(np.float32)
    R = rng.standard_normal((N * K, 3, 3)).astype(np.float32)
    T = rng.standard_normal((N * K, 1, 3)).astype(np.float32)
    s = torch.softmax(torch.from_numpy(rng.standard_normal((N, max(K - 1, 1), V, 1)).astype(np.float32)), 1).numpy()
    up = rng.standard_normal((N, V, 3)).astype(np.float32)
    for tocam in (True, False):
        ref_in = [torch.from_numpy(a).requires_grad_(True) for a in (v, R, T, s)]
        ref = po.obj_to_cam(ref_in[0], ref_in[1], ref_in[2], K, 1, ref_in[3] if K > 1 else None, tocam=tocam)
        (ref * torch.from_numpy(up)).sum().backward()
        hip_in = [dev_t(a, cuda, True) for a in (v, R, T, s)]
        out = geom_utils.obj_to_cam(hip_in[0], hip_in[1], hip_in[2], K, 1, hip_in[3] if K > 1 else None, tocam=tocam)
        close(out, ref, 2e-6, 'out')
        (out * dev_t(up, cuda)).sum().backward()
        for name, a, b in zip(('verts', 'Rmat', 'Tmat', 'skin'), hip_in, ref_in):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
                continue
            close(a.grad, b.grad, 2e-5, 'grad %s tocam=%s' % (name, tocam))


@pytest.mark.parametrize('N,V,K', [(16, 642, 21), (4, 1282, 36), (3, 17, 2), (2, 70, 1)])
def test_obj_to_cam_both_equals_the_two_calls(cuda, N, V, K):
    # nnutils/mesh_net.py:291,298: deform_v = obj_to_cam(..., tocam=False), verts = obj_to_cam(...) on the same arguments; one
    # blend launch yields both (values bit-identical to the separate calls, gradients equal to the sum of their gradients)
    rng = np.random.default_rng(N + V + K)
    v = rng.standard_normal((N, V, 3)).astype(np.float32)
    R = rng.standard_normal((N * K, 3, 3)).astype(np.float32)
    T = rng.standard_normal((N * K, 1, 3)).astype(np.float32)
    s = torch.softmax(torch.from_numpy(rng.standard_normal((N, max(K - 1, 1), V, 1)).astype(np.float32)), 1).numpy()
    up1, up2 = rng.standard_normal((2, N, V, 3)).astype(np.float32)
    a_in = [dev_t(a, cuda, True) for a in (v, R, T, s)]
    cam, blend = geom_utils.obj_to_cam_both(a_in[0], a_in[1], a_in[2], K, 1, a_in[3] if K > 1 else None)
    b_in = [dev_t(a, cuda, True) for a in (v, R, T, s)]
    cam2 = geom_utils.obj_to_cam(b_in[0], b_in[1], b_in[2], K, 1, b_in[3] if K > 1 else None)
    blend2 = geom_utils.obj_to_cam(b_in[0], b_in[1], b_in[2], K, 1, b_in[3] if K > 1 else None, tocam=False)
    assert torch.equal(cam, cam2) and torch.equal(blend, blend2)
    ((cam * dev_t(up1, cuda)).sum() + (blend * dev_t(up2, cuda)).sum()).backward()
    ((cam2 * dev_t(up1, cuda)).sum() + (blend2 * dev_t(up2, cuda)).sum()).backward()
    for name, a, b in zip(('verts', 'Rmat', 'Tmat', 'skin'), a_in, b_in):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0
            continue
        close(a.grad, b.grad, 2e-5, 'grad ' + name)


def test_joint_projection_call_pattern_with_broadcast_identity_skin(cuda):
    # mesh_net.py:285: obj_to_cam(rest_ts[:,:,:,0], Rmat, Tmat[:,None], n_bones, n_hypo, eye(K-1)[None,:,:,None])
    rng = np.random.default_rng(5)
    N, K = 4, 6
    joints = rng.standard_normal((N, K - 1, 3)).astype(np.float32)
    R = rng.standard_normal((N * K, 3, 3)).astype(np.float32)
    T = rng.standard_normal((N * K, 1, 3)).astype(np.float32)
    eye = torch.eye(K - 1)[None, :, :, None]
    ref = po.obj_to_cam(torch.from_numpy(joints), torch.from_numpy(R), torch.from_numpy(T), K, 1, eye)
    out = geom_utils.obj_to_cam(dev_t(joints, cuda), dev_t(R, cuda), dev_t(T, cuda), K, 1, eye.to(cuda))
    close(out, ref, 2e-6)


def test_pinhole_vs_reference_fixture(cuda):
    v, pp, fl = dev_t(GEO['pin_verts'], cuda, True), dev_t(GEO['pin_pp'], cuda, True), dev_t(GEO['pin_fl'], cuda, True)
    out = geom_utils.pinhole_cam(v, pp, fl)
    close(out, GEO['pin_out'], 1e-6, 'pinhole')
    (out * dev_t(GEO['pin_up'], cuda)).sum().backward()
    close(v.grad, GEO['pin_g_verts'], 1e-5)
    close(pp.grad, GEO['pin_g_pp'], 1e-5)
    close(fl.grad, GEO['pin_g_fl'], 1e-5)


def test_mesh_regularisers_vs_reference_fixture(cuda):
    faces = torch.from_numpy(ML['faces'])
    base = torch.from_numpy(ML['base'])
    w = dev_t(ML['w'], cuda)
    x = dev_t(ML['x'], cuda, True)
    lap = loss_utils.LaplacianLoss(base, faces).to(cuda)(x)
    close(lap, ML['lap_out'], 1e-5, 'laplacian')
    (lap * w).sum().backward()
    close(x.grad, ML['lap_g0'], 1e-5, 'laplacian grad')
    x, dx = dev_t(ML['x'], cuda, True), dev_t(ML['dx'], cuda, True)
    ar = loss_utils.ARAPLoss(base, faces).to(cuda)(dx, x)
    close(ar, ML['arap_out'], 1e-5, 'arap')
    (ar * w).sum().backward()
    close(dx.grad, ML['arap_g0'], 1e-5, 'arap grad dx')
    close(x.grad, ML['arap_g1'], 1e-5, 'arap grad x')
    x = dev_t(ML['x'], cuda, True)
    fl = loss_utils.FlattenLoss(faces).to(cuda)(x)
    close(fl, ML['flat_out'], 1e-5, 'flatten')
    (fl * w).sum().backward()
    close(x.grad, ML['flat_g0'], 1e-5, 'flatten grad')


def test_arap_at_full_mesh_size_is_cheap_and_finite(cuda):
    v, f, _ = synth.blobby_mesh(11)                       # V=1212: the reference builds 6 x [N,1212,1212] here
    base, faces = torch.from_numpy(v), torch.from_numpy(f)
    x = torch.from_numpy(v)[None].repeat(4, 1, 1).to(cuda)
    dx = (x + 0.01 * torch.randn_like(x)).requires_grad_(True)
    out = loss_utils.ARAPLoss(base, faces).to(cuda)(dx, x)
    out.sum().backward()
    assert out.shape == (4,) and torch.isfinite(out).all() and torch.isfinite(dx.grad).all()


def _loss_inputs(rng, I, H, S):
    pred = rng.uniform(0, 1, (I, H, S, S)).astype(np.float32)
    masks = (rng.uniform(0, 1, (I, S, S)) > 0.4).astype(np.float32)
    occ = rng.standard_normal((I, S, S)).astype(np.float32)
    occ[rng.uniform(0, 1, occ.shape) < 0.2] = 0.0
    return pred, masks, occ


@pytest.mark.parametrize('I,H,S', [(2, 8, 32), (4, 1, 17), (2, 3, 256)])
def test_mask_loss_table(cuda, I, H, S):
    rng = np.random.default_rng(I + H + S)
    pred, masks, occ = _loss_inputs(rng, I, H, S)
    w = rng.uniform(0.5, 1.5, (I, H)).astype(np.float32)
    a = torch.from_numpy(pred).requires_grad_(True)
    ref = po.mask_loss_table(a, torch.from_numpy(masks), torch.from_numpy(occ))
    (ref * torch.from_numpy(w)).sum().backward()
    b = dev_t(pred, cuda, True)
    out = image_losses.mask_loss_table(b, dev_t(masks, cuda), dev_t(occ, cuda))
    close(out, ref, 1e-5, 'mask loss')
    (out * dev_t(w, cuda)).sum().backward()
    close(b.grad, a.grad, 1e-5, 'mask loss grad')


@pytest.mark.parametrize('I,H,S', [(2, 8, 32), (4, 1, 17), (2, 2, 256)])
def test_flow_loss_table(cuda, I, H, S):
    rng = np.random.default_rng(10 + I + H + S)
    _, masks, occ = _loss_inputs(rng, I, H, S)
    flow_rd = rng.standard_normal((I, H, S, S, 2)).astype(np.float32)
    obs = rng.standard_normal((I, 3, S, S)).astype(np.float32)
    bg = rng.uniform(0, 1, (I, H, S, S)) < 0.5
    bg[0, 0] = True                                        # one (image, hypothesis) with nothing selected
    w = rng.uniform(0.5, 1.5, (I, H)).astype(np.float32)
    a = torch.from_numpy(flow_rd).requires_grad_(True)
    ref, ref_map = po.flow_loss_table(a, torch.from_numpy(obs), torch.from_numpy(bg), torch.from_numpy(occ),
                                      torch.from_numpy(masks))
    (ref * torch.from_numpy(w)).sum().backward()
    b = dev_t(flow_rd, cuda, True)
    out, fmap = image_losses.flow_loss_table(b, dev_t(obs, cuda), dev_t(bg, cuda), dev_t(occ, cuda), dev_t(masks, cuda))
    close(out, ref, 1e-5, 'flow loss')
    close(fmap, ref_map, 1e-5, 'flow map')
    (out * dev_t(w, cuda)).sum().backward()
    close(b.grad, a.grad, 1e-5, 'flow loss grad')
    # the selection mask the kernel evaluates anyway == the reference's vis_mask (mesh_net.py:405)
    out2, fmap2, vis = image_losses.flow_loss_table(b.detach(), dev_t(obs, cuda), dev_t(bg, cuda), dev_t(occ, cuda),
                                                    dev_t(masks, cuda), with_vis=True)
    want = ~torch.from_numpy(bg) & ((torch.from_numpy(occ) != 0) & (torch.from_numpy(masks) > 0))[:, None]
    assert vis.dtype == torch.bool and torch.equal(vis.cpu(), want) and torch.equal(out2, out.detach())


@pytest.mark.parametrize('I,H,S', [(2, 8, 32), (4, 1, 17), (2, 2, 256)])
def test_tex_loss_table(cuda, I, H, S):
    rng = np.random.default_rng(20 + I + H + S)
    _, _, occ = _loss_inputs(rng, I, H, S)
    obs = rng.uniform(0, 1, (I, 3, S, S)).astype(np.float32)
    white = rng.uniform(0, 1, (I, 3, S, S)).astype(np.float32)
    rnd = rng.uniform(0, 1, (I, H, 3, S, S)).astype(np.float32)
    fg = rng.uniform(0, 1, (I, H, S, S)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (I, H)).astype(np.float32)
    a, af = torch.from_numpy(rnd).requires_grad_(True), torch.from_numpy(fg).requires_grad_(True)
    ref = po.tex_loss_table(torch.from_numpy(obs), torch.from_numpy(white), a, af, torch.from_numpy(occ), 0.7)
    (ref * torch.from_numpy(w)).sum().backward()
    b, bf = dev_t(rnd, cuda, True), dev_t(fg, cuda, True)
    out = image_losses.tex_loss_table(dev_t(obs, cuda), dev_t(white, cuda), b, bf, dev_t(occ, cuda), 0.7)
    close(out, ref, 1e-5, 'tex loss')
    (out * dev_t(w, cuda)).sum().backward()
    close(b.grad, a.grad, 1e-5, 'tex grad rnd')
    close(bf.grad, af.grad, 1e-5, 'tex grad fg')


def test_cpu_tensors_are_rejected():
    with pytest.raises(TypeError):
        geom_utils.pinhole_cam(torch.zeros(1, 2, 4), torch.zeros(1, 2), torch.zeros(1, 1))


def test_flow_reproject_matches_restatement(cuda):
    # mesh_net.py:87-104; px as the 6-attribute render delivers it, with a background region (depth 0)
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(5)
    N, S = 3, 40
    px = torch.rand(N, 7, S, S, generator=g) * 2 - 1
    px[:, 2] = px[:, 2] * 0.5 + 10.0
    px[:, 5] = px[:, 5] * 0.5 + 10.5
    px[:, :, :7] = 0.0                                  # background rows: both depths 0
    px[:, 5, 20, :5] = 0.0                              # only the frame-t' depth missing
    pp0, pp1 = torch.randn(N, 2, generator=g) * 0.1, torch.randn(N, 2, generator=g) * 0.1
    fl0, fl1 = torch.rand(N, 1, generator=g) + 8.5, torch.rand(N, 1, generator=g) + 8.5
    gout = torch.randn(N, S, S, 2, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (px, pp0, pp1, fl0, fl1)]
    rflow, rbg = po.flow_reproject(*ref_in)
    (rflow * gout).sum().backward()
    dev_in = [t.clone().to(cuda).requires_grad_(True) for t in (px, pp0, pp1, fl0, fl1)]
    flow, bg = fused_ops.flow_reproject(*dev_in)
    (flow * gout.to(cuda)).sum().backward()
    assert torch.equal(bg.cpu(), rbg) and bg.dtype == torch.bool
    assert torch.equal(flow.detach().cpu(), rflow.detach())          # same operations in the same order: bit-exact
    for name, a, b in zip(('px', 'pp0', 'pp1', 'fl0', 'fl1'), dev_in, ref_in):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0, name
            continue
        scale = float(b.grad.abs().max()) + 1e-30
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-6 * scale, name


def test_quat_to_rotmat_matches_restatement(cuda):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(11)
    q = torch.randn(37, 4, generator=g)
    q[3] = torch.tensor([0., 0., 0., 1.])
    q[4] *= 1e-3                                        # far from unit length: the normalisation matters
    gout = torch.randn(37, 3, 3, generator=g)
    a = q.clone().requires_grad_(True)
    ra = po.quaternion_to_rotation_matrix(a)
    (ra * gout).sum().backward()
    b = q.clone().to(cuda).requires_grad_(True)
    rb = fused_ops.quat_to_rotmat(b)
    (rb * gout.to(cuda)).sum().backward()
    assert float((rb.detach().cpu() - ra.detach()).abs().max()) <= 1e-6
    eye = torch.eye(3).expand(37, 3, 3)
    assert float((rb.detach().cpu() @ rb.detach().cpu().transpose(1, 2) - eye).abs().max()) <= 1e-5
    rel = (b.grad.cpu() - a.grad).abs().max(1)[0] / (a.grad.abs().max(1)[0] + 1e-12)
    assert float(rel.max()) <= 1e-4


@pytest.mark.parametrize('H,J,V', [(1, 1, 70), (2, 5, 300), (8, 20, 642), (1, 35, 1282)])
def test_skin_weights_match_restatement(cuda, H, J, V):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(H * 100 + J)
    verts = torch.randn(H, V, 3, generator=g) * 0.4
    ts = verts[:, torch.randperm(V, generator=g)[:J]].reshape(-1, 3) + 0.01 * torch.randn(H * J, 3, generator=g)
    rs = torch.randn(H * J, 4, generator=g) * 0.2 + torch.tensor([0., 0., 0., 1.])
    lc = torch.randn(H * J, 3, generator=g) * 0.3 + 1.0
    gout = torch.randn(H, J, V, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (ts, rs, lc)]
    ref = po.skin_weights(*ref_in, verts)
    (ref * gout).sum().backward()
    dev_in = [t.clone().to(cuda).requires_grad_(True) for t in (ts, rs, lc)]
    out = fused_ops.skin_weights(*dev_in, verts.to(cuda))
    (out * gout.to(cuda)).sum().backward()
    assert out.shape == (H, J, V)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2e-6
    assert float((out.sum(1) - 1).abs().max()) <= 1e-5
    for name, a, b in zip(('ctl_ts', 'ctl_rs', 'log_ctl'), dev_in, ref_in):
        scale = float(b.grad.abs().max()) + 1e-20
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-4 * scale, name


@pytest.mark.parametrize('level,N', [(1, 2), (3, 5)])
def test_flatten_loss_hip_matches_torch_path(cuda, level, N):
    # the torch path is the restatement pinned to the reference by tests/golden/mesh_losses.npz (CPU suite)
    v, f = synth.geodesic_sphere(2 ** level)
    crit = loss_utils.FlattenLoss(torch.from_numpy(np.asarray(f, np.int64)))
    g = torch.Generator().manual_seed(level)
    x = (torch.from_numpy(v).float()[None] * torch.tensor([1.0, 0.6, 0.8]) + 0.03 * torch.randn(N, len(v), 3, generator=g))
    gout = torch.rand(N, generator=g) + 0.5
    a = x.clone().requires_grad_(True)
    la = crit(a)
    (la * gout).sum().backward()
    b = x.clone().to(cuda).requires_grad_(True)
    lb = crit.to(cuda)(b)
    (lb * gout.to(cuda)).sum().backward()
    assert float(((lb.detach().cpu() - la.detach()).abs() / la.detach().abs()).max()) <= 1e-5
    assert float((b.grad.cpu() - a.grad).abs().max()) <= 1e-4 * float(a.grad.abs().max())


@pytest.mark.parametrize('C', [3, 6])
def test_face_gather_matches_index_select(cuda, C):
    # face_vertices.py:4-22; a fan of 40 faces around vertex 0 exceeds the kernel's per-vertex list (ordered rescan path)
    import lasr_amd.soft_renderer.functional as srf
    g = torch.Generator().manual_seed(C)
    v, f = synth.geodesic_sphere(4)
    f = np.asarray(f, np.int64)
    fan = np.stack([np.zeros(40, np.int64), 1 + np.arange(40), 2 + np.arange(40)], 1)
    faces = torch.from_numpy(np.concatenate([f, fan], 0))[None].repeat(3, 1, 1)
    faces[1] = faces[1].flip(0)                                  # a different face order per mesh
    attr = torch.randn(3, len(v), C, generator=g)
    gout = torch.randn(3, faces.shape[1], 3, C, generator=g)
    a = attr.clone().requires_grad_(True)
    ra = srf.face_vertices(a, faces)                             # CPU tensors: the index_select restatement
    (ra * gout).sum().backward()
    b = attr.clone().to(cuda).requires_grad_(True)
    rb = srf.face_vertices(b, faces.to(cuda))
    (rb * gout.to(cuda)).sum().backward()
    assert torch.equal(rb.detach().cpu(), ra.detach())
    assert float((b.grad.cpu() - a.grad).abs().max()) <= 1e-5 * float(a.grad.abs().max())
    b2 = attr.clone().to(cuda).requires_grad_(True)              # run-to-run identical: no float atomics
    (srf.face_vertices(b2, faces.to(cuda)) * gout.to(cuda)).sum().backward()
    assert torch.equal(b2.grad, b.grad)
    i32 = srf.face_vertices(attr.to(cuda), faces.to(cuda).int())  # int32 faces as Mesh builds from numpy
    assert torch.equal(i32.cpu(), ra.detach())


def test_point_mesh_distance_known_answers(cuda):
    # one triangle in z = 0; points above its interior, beyond an edge, beyond a vertex
    from lasr_amd.nnutils import fused_ops
    verts = torch.tensor([[[0., 0., 0.], [2., 0., 0.], [0., 2., 0.]]], device=cuda)
    faces = torch.tensor([[0, 1, 2]], device=cuda)
    pts = torch.tensor([[[0.5, 0.5, 3.0], [1.0, -2.0, 0.0], [-1.0, -1.0, 1.0], [2.0, 2.0, 0.0]]], device=cuda)
    d2 = torch.tensor([9.0, 4.0, 3.0, 2.0])                       # interior, edge ab, vertex a, edge bc (closest (1,1,0))
    out = fused_ops.point_mesh_face_distance(verts, faces, pts)
    assert abs(float(out) - float(d2.mean() + d2.min())) < 1e-6


@pytest.mark.parametrize('seed', [0, 1])
def test_point_mesh_distance_matches_restatement(cuda, seed):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(seed)
    v, f = synth.geodesic_sphere(3)
    faces = torch.from_numpy(np.asarray(f, np.int64))
    verts = torch.from_numpy(v).float()[None].repeat(2, 1, 1) * torch.tensor([0.7, 0.45, 0.5])
    verts = verts + 0.02 * torch.randn(verts.shape, generator=g)
    pts = verts * torch.tensor([-1., 1., 1.]) + 0.05 * torch.randn(verts.shape, generator=g)    # the mirrored mesh (:470)
    a = [verts.clone().requires_grad_(True), pts.clone().requires_grad_(True)]
    ra = po.point_mesh_face_distance(a[0], faces, a[1])
    (ra * 1.7).backward()
    b = [verts.clone().to(cuda).requires_grad_(True), pts.clone().to(cuda).requires_grad_(True)]
    rb = fused_ops.point_mesh_face_distance(b[0], faces.to(cuda), b[1])
    (rb * 1.7).backward()
    assert abs(float(rb) - float(ra)) <= 1e-5 * abs(float(ra))
    for name, x, y in zip(('verts', 'points'), b, a):
        assert float((x.grad.cpu() - y.grad).abs().max()) <= 2e-4 * float(y.grad.abs().max()), name
    # a face tensor that is seen again (the model's is, every step): from the second call on the backward's face -> vertex
    # reduction runs over the cached incidence lists -- same sums, same order, same bits
    from lasr_amd.soft_renderer.functional import geometry
    fc = faces.to(cuda)
    grads = []
    for _ in range(3):
        c = [verts.clone().to(cuda).requires_grad_(True), pts.clone().to(cuda).requires_grad_(True)]
        (fused_ops.point_mesh_face_distance(c[0], fc, c[1]) * 1.7).backward()
        grads.append((c[0].grad, c[1].grad))
    assert geometry._INC_CACHE[id(fc)][2] is not None
    assert all(torch.equal(gv, b[0].grad) and torch.equal(gp, b[1].grad) for gv, gp in grads)


def test_nearest_point_and_chamfer(cuda):
    from lasr_amd.nnutils import fused_ops, mesh_net
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 300, 3, generator=g), torch.randn(2, 421, 3, generator=g)
    d2, idx = fused_ops.nearest_point(a.to(cuda), b.to(cuda))
    ref = (a[:, :, None] - b[:, None]).pow(2).sum(-1)
    assert torch.equal(idx.cpu(), ref.argmin(2))
    assert float((d2.cpu() - ref.min(2)[0]).abs().max()) <= 1e-6
    ca = torch.randn(8, 20, 3, generator=g)
    x = ca.clone().requires_grad_(True)
    rx = po.chamfer_distance(x, x * torch.tensor([-1., 1., 1.]))
    rx.backward()
    y = ca.clone().to(cuda).requires_grad_(True)
    ry = mesh_net.chamfer_distance(y, y * torch.tensor([-1., 1., 1.], device=cuda))
    ry.backward()
    assert abs(float(ry) - float(rx)) <= 1e-6 * abs(float(rx)) + 1e-9
    assert float((y.grad.cpu() - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())


@pytest.mark.parametrize('N,C,h,rep', [(6, 64, 15, 3), (4, 192, 7, 1), (8, 5, 33, 4)])
def test_perceptual_cosine_reduction_matches_reference_formula(cuda, N, C, h, rep):
    # PerceptualSimilarity/util/util.py:71-83 + networks_basic.py:51-52, restated in mesh_net.cos_sim (torch)
    from lasr_amd.nnutils import fused_ops, mesh_net
    g = torch.Generator().manual_seed(N * C)
    fa = torch.relu(torch.randn(N // rep, C, h, h, generator=g))
    fb = torch.relu(torch.randn(N, C, h, h, generator=g))
    fb[0, :, 0, 0] = 0                                             # a dead pixel: all channels zero
    gout = torch.randn(N, generator=g)
    b_ref = fb.clone().requires_grad_(True)
    ref = 1. - mesh_net.cos_sim(fa.repeat_interleave(rep, 0), b_ref)
    (ref * gout).sum().backward()
    b_dev = fb.clone().to(cuda).requires_grad_(True)
    out = fused_ops.cosine_distance(fa.to(cuda), b_dev, rep)
    (out * gout.to(cuda)).sum().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 2e-6
    gr, gd = b_ref.grad.clone(), b_dev.grad.cpu()
    assert torch.isnan(gr[0, :, 0, 0]).all() and torch.isfinite(gd).all()         # autograd: sqrt'(0); kernel: term dropped
    dead = torch.isnan(gr)                                         # every all-zero channel vector of fb (few channels: several)
    assert float((gd - gr)[~dead].abs().max()) <= 1e-5 * float(gr[~dead].abs().max())


def test_load_textures_kernel_matches_restatement(cuda):
    # texture atlas -> per-face surface texels (soft_renderer.cuda.load_textures), then rendered
    # through the surface-texture mode of the rasteriser
    from lasr_amd.soft_renderer import functional as srf
    rng = np.random.default_rng(3)
    for (H, W, F, R) in ((9, 17, 5, 3), (64, 48, 300, 5), (7, 7, 2, 1)):
        img = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
        uv = rng.uniform(0, 1, (F, 3, 2)).astype(np.float32)
        uv[0] = [[0, 0], [1, 0], [1, 1]]                                  # corners of the atlas, incl. uv == 1
        upd = (rng.uniform(0, 1, F) > 0.2).astype(np.int32)
        upd[0] = 1
        ref = po.load_textures(img, uv, R, upd)
        got = srf.load_textures(torch.from_numpy(img).to(cuda), torch.from_numpy(uv).to(cuda), R, torch.from_numpy(upd)).cpu().numpy()
        assert got.shape == (F, R * R, 3)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    with pytest.raises(TypeError):
        srf.load_textures(torch.from_numpy(img), torch.from_numpy(uv), R)


# ---- small-tensor glue kernels (lasr_amd/csrc/glue.hip) vs the line-by-line torch restatements ------------------------------
def _grads(outs, cots, leaves):
    return torch.autograd.grad([o for o in outs], leaves, [c for c in cots], allow_unused=True)


def test_geodesic_distance_values_and_gradients(cuda):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(3)
    q1, q2 = torch.randn(300, 4, generator=g), torch.randn(300, 4, generator=g)
    m1 = po.quaternion_to_rotation_matrix(q1).reshape(-1, 3, 3)
    m2 = po.quaternion_to_rotation_matrix(q2).reshape(-1, 3, 3)
    m2[:4] = m1[:4]                                                     # coincident rotations: angle 0, gradient 0 (not NaN)
    a, b = m1.clone().requires_grad_(), m2.clone().requires_grad_()
    ref = po.geodesic_distance(a.double(), b.double())
    cot = torch.randn(300, generator=g)
    cot[:4] = 1.0
    ga_ref, gb_ref = _grads([ref[4:]], [cot[4:].double()], [a, b])      # the reference's gradient is NaN on the coincident rows
    A, B_ = m1.to(cuda).requires_grad_(), m2.to(cuda).requires_grad_()
    out = fused_ops.geodesic_distance(A, B_)
    assert out.shape == (300,)
    # acos is ill-conditioned at the ends: an fp32 rounding of cos (6e-8) moves the angle by 6e-8 / sin(angle)
    sin = ref.detach().sin().abs().clamp_min(1e-3).numpy()
    assert (np.abs(out.cpu().detach().numpy() - ref.detach().numpy())[4:] <= 2e-6 + 5e-7 / sin[4:]).all()
    assert float(out[:4].abs().max()) <= 2e-3                           # acos(1 - a few ulp) = sqrt(2 * 2.4e-7) = 7e-4
    gA, gB = _grads([out], [cot.to(cuda)], [A, B_])
    assert torch.isfinite(gA).all() and torch.isfinite(gB).all()
    mid = torch.from_numpy(sin > 0.1)                                   # d acos / d cos = -1 / sin: compare away from the poles
    mid[:4] = False
    np.testing.assert_allclose(gA[mid].cpu().numpy(), ga_ref[mid].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gB[mid].cpu().numpy(), gb_ref[mid].numpy(), rtol=1e-4, atol=1e-5)
    assert int(mid.sum()) > 250
    # opposite rotations (cos = -1 exactly): pi, zero gradient
    flip = torch.diag(torch.tensor([1., -1., -1.]))[None].to(cuda).requires_grad_()
    eye = torch.eye(3)[None].to(cuda)
    ang = fused_ops.geodesic_distance(flip, eye)
    assert abs(float(ang) - np.pi) < 1e-6 and float(_grads([ang], [torch.ones(1, device=cuda)], [flip])[0].abs().max()) == 0


@pytest.mark.parametrize('B,H,K', [(1, 8, 21), (2, 1, 1), (3, 16, 26)])
def test_intrinsics_kernel_matches_the_reference_lines(cuda, B, H, K):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(B * 100 + H)
    cams = torch.rand(2 * B, 7, generator=g) + 0.5
    pp = torch.randn(2 * B, 2, generator=g) * 20
    scale, depth, ppoint = torch.rand(2 * B, H, generator=g) + 1, torch.randn(2 * B, K, generator=g), torch.randn(2 * B, 2, generator=g) * 0.1
    leaves = [t.clone().requires_grad_() for t in (scale, depth, ppoint)]
    ref = po.intrinsics(cams, pp, *leaves, 256)
    cots = [torch.randn(r.shape, generator=g) for r in ref]
    gref = _grads(ref, cots, leaves)
    dl = [t.clone().to(cuda).requires_grad_() for t in (scale, depth, ppoint)]
    out = fused_ops.intrinsics(cams.to(cuda), pp.to(cuda), *dl, 256)
    for o, r in zip(out, ref):
        np.testing.assert_allclose(o.detach().cpu().numpy(), r.detach().numpy(), rtol=1e-6, atol=1e-6)
    gout = _grads(out, [c.to(cuda) for c in cots], dl)
    for a, b in zip(gout, gref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-6, atol=1e-6)
    assert float(gout[2][B:].abs().max()) == 0                          # frame t' principal point prediction is discarded


@pytest.mark.parametrize('n_images,H,K', [(2, 8, 21), (4, 1, 1), (6, 16, 26), (2, 2, 5)])
def test_bone_fixup_kernel_matches_the_reference_lines(cuda, n_images, H, K):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(n_images * 10 + K)
    M = n_images * H
    quat = po.quaternion_to_rotation_matrix(torch.randn(M * K, 4, generator=g)).reshape(-1, 9) + 0.01 * torch.randn(M * K, 9, generator=g)
    trans, depth = torch.randn(M * K, 2, generator=g), torch.randn(M * K, 1, generator=g) + 5
    rest = torch.randn(H, max(K - 1, 0) * 3, generator=g)
    leaves = [t.clone().requires_grad_() for t in (quat, trans, depth, rest)]
    R, T = po.bone_fixup(*leaves, n_images, H, K)
    cR, cT = torch.randn(R.shape, generator=g), torch.randn(T.shape, generator=g)
    gref = _grads([R, T], [cR, cT], leaves)
    dl = [t.clone().to(cuda).requires_grad_() for t in (quat, trans, depth, rest)]
    Rd, Td = fused_ops.bone_fixup(*dl, H, K)
    assert Rd.shape == R.shape and Td.shape == T.shape
    assert torch.equal(Rd.cpu(), R.detach())                            # pure data movement
    np.testing.assert_allclose(Td.detach().cpu().numpy(), T.detach().numpy(), rtol=1e-6, atol=2e-6)
    gout = _grads([Rd, Td], [cR.to(cuda), cT.to(cuda)], dl)
    for k, (a, b) in enumerate(zip(gout, gref)):
        if K == 1 and k == 3:
            assert a is None or not a.any()
            continue
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-5, atol=1e-5)


def test_weighted_mean_sum_matches_the_reference_chain(cuda):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(9)
    shapes = [(2, 8), (2, 8), (16, 642), (1,), (168,), (16, 20, 1, 1), (5000,), (3, 3)]
    weights = [1.0, 1.0, 0.25, 0.1, 0.001, 10.0, 0.02, 5.0]
    groups = [0, 1, 2, 2, 3, 4, 4, 5]
    xs = [torch.randn(s, generator=g) for s in shapes]
    leaves = [x.clone().double().requires_grad_() for x in xs]
    total, sums = po.weighted_mean_sum(list(zip(leaves, weights, groups)))
    gref = torch.autograd.grad(total * 3.0, leaves)
    dl = [x.clone().to(cuda).requires_grad_() for x in xs]
    dl_in = list(dl)
    dl_in[2] = dl[2].t().contiguous().t()                               # a non-contiguous term is accepted (copied)
    tot, sm = fused_ops.weighted_mean_sum(list(zip(dl_in, weights, groups)))
    assert sm.shape == (6,) and not sm.requires_grad
    np.testing.assert_allclose(float(tot), float(total), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(sm.cpu().numpy(), np.array([float(s) for s in sums]), rtol=2e-6, atol=1e-7)
    gout = torch.autograd.grad(tot * 3.0, dl)
    for a, b in zip(gout, gref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-6, atol=1e-12)
    with pytest.raises(TypeError):
        fused_ops.weighted_mean_sum([(xs[0], 1.0, 0)])                  # CPU tensor: no fallback


@pytest.mark.parametrize('N,P,Q', [(8, 20, 20), (2, 300, 421), (1, 1, 5)])
def test_chamfer_kernel_values_indices_and_gradients(cuda, N, P, Q):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(N * 7 + P)
    a, b = torch.randn(N, P, 3, generator=g), torch.randn(N, Q, 3, generator=g)
    x, y = a.clone().double().requires_grad_(), b.clone().double().requires_grad_()
    d = (x[:, :, None] - y[:, None]).pow(2).sum(-1)
    ref = d.min(2)[0].mean(1) + d.min(1)[0].mean(1)                     # per batch item; po.chamfer_distance is its mean
    cot = torch.randn(N, generator=g)
    gx, gy = torch.autograd.grad(ref, [x, y], cot.double())
    u, v = a.clone().to(cuda).requires_grad_(), b.clone().to(cuda).requires_grad_()
    out = fused_ops.chamfer(u, v)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    gu, gv = torch.autograd.grad(out, [u, v], cot.to(cuda))
    np.testing.assert_allclose(gu.cpu().numpy(), gx.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gv.cpu().numpy(), gy.numpy(), rtol=1e-4, atol=1e-6)
    assert abs(float(out.mean()) - float(po.chamfer_distance(a, b))) <= 1e-5 * float(ref.mean().abs()) + 1e-6


def test_perceptual_unit_range_fold_equals_the_explicit_map(cuda):
    # mesh_net.py:436-441 feeds 2 * img - 1 to the perceptual network; the product folds that map into the input normalisation
    from lasr_amd.nnutils import mesh_net
    pd = mesh_net.PerceptualDistance().to(cuda)
    g = torch.Generator().manual_seed(4)
    a = torch.rand(2, 3, 64, 64, generator=g).to(cuda)
    b = torch.rand(6, 3, 64, 64, generator=g).to(cuda)
    b1, b2 = b.clone().requires_grad_(), b.clone().requires_grad_()
    d1 = pd.forward_pair(a, b1, repeat=3, unit_range=True)
    d2 = pd.forward_pair(2 * a - 1, 2 * b2 - 1, repeat=3)
    assert d1.shape == d2.shape == (6,)
    np.testing.assert_allclose(d1.detach().cpu().numpy(), d2.detach().cpu().numpy(), rtol=2e-5, atol=1e-6)
    cot = torch.randn(6, generator=g).to(cuda)
    g1, = torch.autograd.grad(d1, b1, cot)
    g2, = torch.autograd.grad(d2, b2, cot)
    assert float((g1 - g2).abs().max()) <= 2e-4 * float(g2.abs().max())


@pytest.mark.parametrize('H,Vp,S,R', [(8, 337, 305, 2), (1, 40, 0, 4), (3, 10, 10, 6)])
def test_mean_shape_kernel_matches_symmetrize_sigmoid_tile(cuda, H, Vp, S, R):
    # third_party/ext_nnutils/mesh_net.py:128-149 (symmetrize) + :171-185 (get_mean_shape), written out with torch ops
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(H * 31 + S)
    mv, tx = torch.randn(H, Vp, 3, generator=g), torch.randn(H, Vp, 3, generator=g)
    flip = torch.tensor([[-1., 1., 1.]])
    mask = torch.ones(Vp + S, 3)
    mask[:max(Vp - S, 0) // 2, 0] = 0

    def restatement(mean_v, tex):
        if S > 0:
            mean_v = torch.cat([mean_v, flip * mean_v[..., -S:, :]], -2) * mask
            tex = torch.cat([tex, tex[..., -S:, :]], -2)
        V = mean_v.shape[1]
        return (mean_v[None].repeat(R, 1, 1, 1).view(R * H, V, 3), tex.sigmoid()[None].repeat(R, 1, 1, 1).view(R * H, V, 3))
    a, b = mv.clone().requires_grad_(), tx.clone().requires_grad_()
    rv, rt = restatement(a, b)
    cv, ct = torch.randn(rv.shape, generator=g), torch.randn(rt.shape, generator=g)
    ga, gb = torch.autograd.grad([rv, rt], [a, b], [cv, ct])
    x, y = mv.clone().to(cuda).requires_grad_(), tx.clone().to(cuda).requires_grad_()
    ov, ot = fused_ops.mean_shape(x, y, flip.to(cuda) if S else None, mask.to(cuda) if S else None, R, S)
    assert torch.equal(ov.cpu(), rv.detach())                          # copies, sign flips and 0/1 masks: exact
    np.testing.assert_allclose(ot.detach().cpu().numpy(), rt.detach().numpy(), rtol=2e-6, atol=1e-7)
    gx, gy = torch.autograd.grad([ov, ot], [x, y], [cv.to(cuda), ct.to(cuda)])
    np.testing.assert_allclose(gx.cpu().numpy(), ga.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gy.cpu().numpy(), gb.numpy(), rtol=1e-5, atol=1e-6)
    # colours without gradient (--opt_tex no): only the shape gradient is produced
    ov2, ot2 = fused_ops.mean_shape(x, y.detach(), flip.to(cuda) if S else None, mask.to(cuda) if S else None, R, S)
    gx2, = torch.autograd.grad([ov2], [x], [cv.to(cuda)])
    assert torch.equal(gx2, gx)


def test_obs_pair_kernel(cuda):
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(3, 3, 20, 20, generator=g)
    masks = (torch.rand(3, 20, 20, generator=g) > 0.4).float() * torch.rand(3, 20, 20, generator=g)
    fg = (masks > 0).float()[:, None]
    out = fused_ops.obs_pair(imgs.to(cuda), masks.to(cuda)).cpu()
    assert torch.equal(out[:3], imgs * fg) and torch.equal(out[3:], 1 - fg + imgs * fg)


def test_fill_planes(cuda):
    from lasr_amd.nnutils import fused_ops
    for shape, vals in (((3, 4, 16, 16), (1., 1., 1., 1.)), ((2, 10, 7, 5), tuple(float(k) / 4 for k in range(10))), ((1, 2, 1, 1), (3., -2.)),
                        ((2, 4, 256, 256), (0.25, 0.5, 0.75, 1.))):
        t = torch.full(shape, float('nan'), device=cuda)
        fused_ops.fill_planes(t, vals)
        want = torch.tensor(vals).view(1, -1, 1, 1).expand(shape)
        assert torch.equal(t.cpu(), want)


def test_textured_obj_loads_surface_textures_like_the_reference(tmp_path, cuda):
    # load_obj.py:28-101: per-face surface texels of a textured .obj = ones, then the material's Kd colour, then its atlas image
    # (flipped vertically, /255) sampled by the load_textures kernel; rendered afterwards through the surface-texture raster mode
    from lasr_amd.soft_renderer import functional as srf
    from lasr_amd import soft_renderer as sr
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_path_oracle import _write_textured_quad
    path, img = _write_textured_quad(str(tmp_path))
    R = 5
    v, f, tex = srf.load_obj(path, load_texture=True, texture_res=R, texture_type='surface', device=cuda)
    assert v.shape == (5, 3) and f.shape == (4, 3) and tex.shape == (4, R * R, 3) and tex.is_cuda
    uv, mats, colors, files = srf.obj_io.parse_obj_materials(path)
    atlas = (img.astype(np.float32) / 255.)[::-1]
    want = np.ones((4, R * R, 3), np.float32)
    want[2:] = colors['flat']
    want[:2] = po.load_textures(atlas, uv, R, np.array([1, 1, 0, 0], np.int32))[:2]
    np.testing.assert_allclose(tex.cpu().numpy(), want, rtol=0, atol=1e-6)
    # render the model with its texture: the image shows atlas colours on the quad, background elsewhere
    r = sr.SoftRenderer(image_size=32, sigma_val=1e-12, camera_mode='look_at', perspective=False, aggr_func_rgb='hard',
                        dist_func='hard', aggr_func_alpha='hard', light_intensity_ambient=1., light_intensity_directionals=0.,
                        light_mode='surface')
    vv = (v - v.mean(0)) * 0.6
    out = r.render_mesh(sr.Mesh(vv[None], f[None].long(), textures=tex[None], texture_type='surface'))
    assert out.shape == (1, 4, 32, 32) and torch.isfinite(out).all()
    covered = out[0, 3] > 0.5
    assert 0.05 < float(covered.float().mean()) < 0.9
    lo, hi = float(tex.min()), float(tex.max())
    assert float(out[0, :3][:, covered].min()) >= lo - 1e-5 and float(out[0, :3][:, covered].max()) <= hi + 1e-5
