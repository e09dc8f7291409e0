"""Round-5 launch fusions of the optimisation step: each one must give what the launches it replaces gave.

  cosine_distance_layers   == the per-layer cosine_distance calls added up (bit-identical, values and gradients)
  raster_faces             == raster_inputs -> `vertices - eye` -> two face gathers (values bit-identical; the intrinsics'
                              gradients are the same sums in another association)
  LBS backward on MFMA     == autograd of the reference formula in float64 (fp32 round-off: the contractions reassociate)
  mesh_regularisers        == LaplacianLoss / FlattenLoss / ARAPLoss called one by one
References for the composed operators themselves (reference file:line) are in the tests of those operators
(tests/test_ops_gpu.py, tests/test_render_tables_gpu.py)."""
import numpy as np
import pytest
import torch

from lasr_amd import _lib, synth
from lasr_amd.nnutils import fused_ops, geom_utils, loss_utils
from lasr_amd.soft_renderer import functional as srf

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)


@pytest.mark.parametrize('N,rep,sizes', [(32, 8, ((64, 63), (192, 31), (384, 15), (256, 15), (256, 15))),      # AlexNet at 256x256, S0
                                         (6, 3, ((8, 5), (40, 9), (7, 33))),                                   # odd channel counts / tiles
                                         (2, 1, ((64, 4),))])
def test_all_layers_in_one_launch_equal_the_per_layer_calls(cuda, N, rep, sizes):
    g = torch.Generator().manual_seed(N + len(sizes))
    fa = [torch.relu(torch.randn(N // rep, C, h, h, generator=g)).to(cuda) for C, h in sizes]
    fb = [torch.relu(torch.randn(N, C, h, h, generator=g)).to(cuda) for C, h in sizes]
    fb[0][0, :, 0, 0] = 0                                            # a dead pixel
    gout = torch.randn(N, generator=g).to(cuda)
    one = [b.clone().requires_grad_(True) for b in fb]
    d = 0
    for a, b in zip(fa, one):
        d = d + fused_ops.cosine_distance(a, b, rep)
    (d * gout).sum().backward()
    many = [b.clone().requires_grad_(True) for b in fb]
    m = fused_ops.cosine_distance_layers(fa, many, rep)
    (m * gout).sum().backward()
    assert torch.equal(m.detach(), d.detach())
    for x, y in zip(many, one):
        assert torch.equal(x.grad, y.grad)
    # twice in a row: the ticket word is left zero
    assert torch.equal(fused_ops.cosine_distance_layers(fa, fb, rep), d.detach())


@pytest.mark.parametrize('n2,H,level', [(2, 8, 3), (4, 1, 2), (2, 2, 1)])
def test_raster_faces_equals_the_five_launches_it_replaces(cuda, n2, H, level):
    v, f = synth.geodesic_sphere(2 ** level)
    V, N = v.shape[0], n2 * H
    g = torch.Generator().manual_seed(level * 10 + H)
    cam = (torch.from_numpy(v)[None] * 0.5 + 0.05 * torch.randn(N, V, 3, generator=g))
    cam[:, :, 2] += 8
    tex = torch.rand(N, V, 3, generator=g)
    pp = torch.randn(N, 2, generator=g) * 0.1
    fl = torch.rand(N, generator=g) + 8
    eye = [0.0, 0.0, -2.732]
    faces = torch.from_numpy(np.asarray(f, np.int64))[None].repeat(N, 1, 1).to(cuda)
    up_v = torch.randn(N, faces.shape[1], 3, 3, generator=g).to(cuda)
    up_a = torch.randn(N, faces.shape[1], 3, 9, generator=g).to(cuda)

    a = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    pre, attrs, nf = fused_ops.raster_inputs(*a, eye)
    fv0 = srf.face_vertices(srf.look_at(pre, eye), faces)
    fa0 = srf.face_vertices(attrs, faces)
    ((fv0 * up_v).sum() + (fa0 * up_a).sum()).backward()

    b = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    inc = fused_ops.face_incidence(faces[:1], V)
    fv1, fa1, nf1 = fused_ops.raster_faces(*b, eye, faces[:1].contiguous(), inc)
    ((fv1 * up_v).sum() + (fa1 * up_a).sum()).backward()
    assert torch.equal(fv1.detach(), fv0.detach()) and torch.equal(fa1.detach(), fa0.detach()) and torch.equal(nf1, nf)
    assert torch.equal(b[0].grad, a[0].grad) and torch.equal(b[1].grad, a[1].grad)        # same corner order, same expressions
    assert rel(b[2].grad, a[2].grad) <= 2e-5 and rel(b[3].grad, a[3].grad) <= 2e-5          # block sums in another association

    # per-mesh connectivity (not shared): a permutation of the faces of every second mesh
    perm = torch.randperm(faces.shape[1], generator=g).to(cuda)
    faces2 = faces.clone()
    faces2[1::2] = faces[1::2][:, perm]
    c = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    pre, attrs, _ = fused_ops.raster_inputs(*c, eye)
    ((srf.face_vertices(srf.look_at(pre, eye), faces2) * up_v).sum() + (srf.face_vertices(attrs, faces2) * up_a).sum()).backward()
    e = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    fv2, fa2, _ = fused_ops.raster_faces(*e, eye, faces2, fused_ops.face_incidence(faces2, V))
    ((fv2 * up_v).sum() + (fa2 * up_a).sum()).backward()
    assert torch.equal(e[0].grad, c[0].grad) and torch.equal(e[1].grad, c[1].grad)


def _lbs_reference(v, R, T, s, K, tocam):
    """geom_utils.py:45-71 with torch ops (K-1 blended transforms, then the body transform); autograd gives the gradients the
    reference's autograd gives."""
    N = v.shape[0]
    Rk, Tk = R.view(N, K, 3, 3), T.view(N, K, 1, 3)
    vs = (s * (v[:, None] @ Rk[:, 1:] + Tk[:, 1:])).sum(1) if K > 1 else v
    return (vs @ Rk[:, 0] + Tk[:, 0]) if tocam else vs


@pytest.mark.parametrize('N,V,K,tocam', [(16, 642, 21, 1), (6, 1282, 36, 1), (2, 37, 2, 0), (4, 70, 1, 1), (3, 200, 66, 1),
                                         (2, 129, 17, 1), (1, 64, 65, 0)])
def test_lbs_backward_on_the_matrix_cores_matches_autograd_of_the_reference_formula(cuda, N, V, K, tocam):
    # K <= 65: the three contractions run on v_mfma_f32_16x16x4_f32 (blocks of 64 vertices; V = 37 / 70 / 129 / 200 / 642 leave
    # partial tiles); K = 66: the VALU kernel of rounds 1-4
    g = torch.Generator().manual_seed(N * K)
    leaves = [torch.randn(N, V, 3, generator=g), torch.randn(N * K, 3, 3, generator=g), torch.randn(N * K, 1, 3, generator=g),
              torch.softmax(torch.randn(N, max(K - 1, 1), V, 1, generator=g), 1)]
    up = torch.randn(N, V, 3, generator=g)
    a = [t.clone().double().requires_grad_(True) for t in leaves]
    (_lbs_reference(*a, K, tocam) * up.double()).sum().backward()
    b = [t.clone().to(cuda).requires_grad_(True) for t in leaves]
    (geom_utils.obj_to_cam(b[0], b[1], b[2], K, 1, b[3] if K > 1 else None, tocam=bool(tocam)) * up.to(cuda)).sum().backward()
    for name, x, y in zip(('verts', 'Rmat', 'Tmat', 'skin')[:4 if K > 1 else 3], b, a):
        if name in ('Rmat', 'Tmat') and not tocam and K == 1:
            continue
        assert rel(x.grad.cpu().double(), y.grad) <= 2e-5, name


def test_lbs_both_outputs_and_points_only_backward(cuda):
    # the two outputs of one blend (mesh_net.py:291, :298) and LASR's joint call, where only the points receive gradient
    N, V, K = 4, 300, 9
    g = torch.Generator().manual_seed(3)
    leaves = [torch.randn(N, V, 3, generator=g), torch.randn(N * K, 3, 3, generator=g), torch.randn(N * K, 1, 3, generator=g),
              torch.softmax(torch.randn(N, K - 1, V, 1, generator=g), 1)]
    up0, up1 = torch.randn(N, V, 3, generator=g), torch.randn(N, V, 3, generator=g)
    a = [t.clone().double().requires_grad_(True) for t in leaves]
    ((_lbs_reference(*a, K, 1) * up0.double()).sum() + (_lbs_reference(*a, K, 0) * up1.double()).sum()).backward()
    b = [t.clone().to(cuda).requires_grad_(True) for t in leaves]
    cam, blend = geom_utils.obj_to_cam_both(b[0], b[1], b[2], K, 1, b[3])
    ((cam * up0.to(cuda)).sum() + (blend * up1.to(cuda)).sum()).backward()
    for name, x, y in zip(('verts', 'Rmat', 'Tmat', 'skin'), b, a):
        assert rel(x.grad.cpu().double(), y.grad) <= 2e-5, name
    c = leaves[0].clone().to(cuda).requires_grad_(True)
    out = geom_utils.obj_to_cam(c, b[1].detach(), b[2].detach(), K, 1, b[3].detach())
    (out * up0.to(cuda)).sum().backward()
    a2 = [t.clone().double() for t in leaves]
    a2[0].requires_grad_(True)
    (_lbs_reference(*a2, K, 1) * up0.double()).sum().backward()
    assert rel(c.grad.cpu().double(), a2[0].grad) <= 2e-5


@pytest.mark.parametrize('level,N,NA', [(3, 16, 8), (2, 4, 2), (1, 2, 1)])
def test_mesh_regularisers_in_one_launch_equal_the_three_criteria(cuda, level, N, NA):
    v, f = synth.geodesic_sphere(2 ** level)
    V = v.shape[0]
    g = torch.Generator().manual_seed(level)
    faces = torch.from_numpy(np.asarray(f, np.int64))
    lap = loss_utils.LaplacianLoss(torch.from_numpy(v), faces).to(cuda)
    arap = loss_utils.ARAPLoss(torch.from_numpy(v), faces).to(cuda)
    flat = loss_utils.FlattenLoss(faces).to(cuda)
    x = (torch.from_numpy(v)[None] + 0.05 * torch.randn(N, V, 3, generator=g)).to(cuda)
    d0 = (torch.from_numpy(v)[None] + 0.05 * torch.randn(NA, V, 3, generator=g)).to(cuda)
    d1 = (torch.from_numpy(v)[None] + 0.05 * torch.randn(NA, V, 3, generator=g)).to(cuda)
    ups = [torch.randn(n, generator=g).to(cuda) for n in (N, N, NA)]

    a = [t.clone().requires_grad_(True) for t in (x, d0, d1)]
    l0, f0, a0 = lap(a[0]), flat(a[0]), arap(a[1], a[2])
    ((l0 * ups[0]).sum() + (f0 * ups[1]).sum() + (a0 * ups[2]).sum()).backward()
    b = [t.clone().requires_grad_(True) for t in (x, d0, d1)]
    l1, f1, a1 = fused_ops.mesh_regularisers(b[0], b[1], b[2], lap, flat, arap)
    ((l1 * ups[0]).sum() + (f1 * ups[1]).sum() + (a1 * ups[2]).sum()).backward()
    assert torch.equal(l1.detach(), l0.detach()) and torch.equal(f1.detach(), f0.detach()) and torch.equal(a1.detach(), a0.detach())
    assert torch.equal(b[1].grad, a[1].grad) and torch.equal(b[2].grad, a[2].grad)
    assert rel(b[0].grad, a[0].grad) <= 1e-6            # Laplacian + flatten parts added inside the kernel


@pytest.mark.parametrize('n2,H,K', [(2, 8, 21), (4, 1, 36), (2, 2, 2)])
def test_project_points_equals_the_identity_skin_lbs_and_pinhole_calls(cuda, n2, H, K):
    # nnutils/mesh_net.py:285-288 + :302 as the round-1..4 product ran them: cat / repeat of the points, obj_to_cam with a one-hot skin,
    # homogeneous cat, pinhole_cam
    g = torch.Generator().manual_seed(n2 * 100 + K)
    M, nb = n2 * H, K - 1
    rest = (0.3 * torch.randn(H * nb, 3, generator=g)).to(cuda)
    ctl = (0.3 * torch.randn(H * nb, 3, generator=g)).to(cuda)
    R = torch.randn(M * K, 3, 3, generator=g).to(cuda)
    T = torch.randn(M * K, 3, generator=g).to(cuda)
    T.view(M, K, 3)[:, 0, 2] += 8                                   # in front of the camera
    ppoint = (0.1 * torch.randn(n2, 2, generator=g)).to(cuda)
    scale = (torch.rand(n2, H, generator=g) + 8).to(cuda)
    up = torch.randn(M, 2 * nb, 4, generator=g).to(cuda)

    a = [t.clone().requires_grad_(True) for t in (rest, ctl)]
    eye = torch.eye(nb, device=cuda)[None, :, :, None]
    eye = torch.cat([eye, eye], 2)
    pts = torch.cat([a[0].view(H, nb, 3), a[1].view(H, nb, 3)], 1).repeat(n2, 1, 1)
    jc = geom_utils.obj_to_cam(pts, R, T[:, None], K, H, eye)
    want = geom_utils.pinhole_cam(torch.cat([jc, torch.ones_like(jc[:, :, :1])], -1), ppoint, scale)
    (want * up).sum().backward()
    b = [t.clone().requires_grad_(True) for t in (rest, ctl)]
    got = fused_ops.project_points(b[0], b[1], R, T, ppoint, scale, H, K)
    (got * up).sum().backward()
    assert got.shape == want.shape and rel(got.detach(), want.detach()) <= 1e-6
    for x, y, name in zip(b, a, ('rest_ts', 'ctl_ts')):
        assert rel(x.grad, y.grad) <= 2e-5, name


@pytest.mark.parametrize('N,level,C,shared', [(1, 2, 3, True), (5, 3, 3, True), (4, 2, 9, False), (3, 1, 1, False), (16, 3, 3, True)])
def test_face_gather_backward_over_the_incidence_lists_has_the_scan_kernel_s_bits(cuda, N, level, C, shared):
    # lasr_face_gather_backward_csr: same ascending corner order per vertex as the scanning kernel -> torch.equal
    v, f = synth.geodesic_sphere(2 ** level)
    f = torch.from_numpy(np.ascontiguousarray(f)).long()
    V, F = v.shape[0], f.shape[0]
    gen = torch.Generator(device='cpu').manual_seed(7 + N)
    if shared:
        faces = f[None].expand(N, F, 3).contiguous().to(cuda)
        inc_ptr, inc = fused_ops.face_incidence(f[None].to(cuda), V)
    else:
        faces = torch.stack([f[torch.randperm(F, generator=gen)] for _ in range(N)]).to(cuda)     # another corner order per mesh
        inc_ptr, inc = fused_ops.face_incidence(faces, V)
    g = torch.randn(N, F, 3, C, generator=gen).to(cuda)
    h = _lib.lib()
    want, got = torch.empty(N, V, C, device=cuda), torch.full((N, V, C), float('nan'), device=cuda)
    st = torch.cuda.current_stream(cuda).cuda_stream
    _lib.check(h.lasr_face_gather_backward(g.data_ptr(), faces.data_ptr(), want.data_ptr(), N, V, F, C, st), 'scan')
    _lib.check(h.lasr_face_gather_backward_csr(g.data_ptr(), inc_ptr.data_ptr(), inc.data_ptr(), 1 if shared else 0, got.data_ptr(),
                                               N, V, F, C, st), 'csr')
    assert torch.equal(got, want)
    # and it is the gradient of the gather
    ref = torch.zeros(N, V, C, dtype=torch.float64, device=cuda)
    ref.scatter_add_(1, faces.reshape(N, -1, 1).expand(N, 3 * F, C), g.reshape(N, 3 * F, C).double())
    assert (got.double() - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_face_vertices_caches_the_incidence_of_a_face_tensor_it_sees_again(cuda):
    from lasr_amd.soft_renderer.functional import geometry
    v, f = synth.geodesic_sphere(4)
    f = torch.from_numpy(np.ascontiguousarray(f)).long()
    V = v.shape[0]
    faces = f[None].repeat(3, 1, 1).to(cuda)
    verts = torch.randn(3, V, 3, device=cuda)
    w = torch.randn(3, f.shape[0], 3, 3, device=cuda)

    def grad(fc):
        x = verts.clone().requires_grad_(True)
        (geometry.face_vertices(x, fc) * w).sum().backward()
        return x.grad

    geometry._INC_CACHE.clear()
    g1 = grad(faces)                                   # first sighting: scanning kernel, entry without a structure
    ent = geometry._INC_CACHE[id(faces)]                # [weak reference, signature, structure, last seen]
    assert ent[2] is None
    g2 = grad(faces)                                   # second sighting: the structure is built and used
    assert geometry._INC_CACHE[id(faces)][2] is not None and torch.equal(g1, g2)
    faces[0, 0] = faces[0, 0].flip(0)                  # an in-place edit bumps the version: the entry is dropped, not reused
    g3 = grad(faces)
    assert geometry._INC_CACHE[id(faces)][2] is None
    assert torch.equal(g3, grad(faces)) and torch.equal(g3, grad(faces.clone()))
    tmp = faces.clone()
    key = id(tmp)
    grad(tmp)
    del tmp                                            # the entry of a dead tensor never matches a new one at the same id
    other = f[None].repeat(3, 1, 1).flip(1).to(cuda)
    if id(other) == key:
        assert geometry._INC_CACHE[key][0]() is None
    assert torch.equal(grad(other), grad(other.clone()))


def _odd_meshes():
    """Connectivities the sphere tests never produce: a tetrahedron with two vertices no face uses (valence 0) and a fan whose hub has
    valence 37 (more than the 16 lanes / 4-corner batches the vertex-centric backward kernels work in)."""
    tet = (np.array([[0, 0, 1], [1, 0, -0.5], [-0.5, 0.8, -0.5], [-0.5, -0.8, -0.5], [3, 3, 3], [-3, 3, 2]], np.float32) * 0.4,
           np.array([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]], np.int64))
    n = 37
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    rim = np.stack([np.cos(ang), np.sin(ang), 0.1 * np.sin(3 * ang)], 1)
    fan_v = np.concatenate([[[0, 0, 0.3]], rim]).astype(np.float32) * 0.6
    fan_f = np.array([[0, 1 + i, 1 + (i + 1) % n] for i in range(n)], np.int64)
    return {'tetrahedron+isolated': tet, 'fan37': (fan_v, fan_f)}


@pytest.mark.parametrize('name', ['tetrahedron+isolated', 'fan37'])
def test_vertex_centric_backwards_on_odd_connectivities(cuda, name):
    v, f = _odd_meshes()[name]
    V, F, N = v.shape[0], f.shape[0], 4                               # (N even: image n pairs with (n + N/2) % N)
    g = torch.Generator().manual_seed(V)
    faces = torch.from_numpy(f)[None].repeat(N, 1, 1).to(cuda)
    inc = fused_ops.face_incidence(faces[:1], V)
    # ---- face gather over the incidence lists vs the scanning kernel; guard words either side of the output stay untouched
    h = _lib.lib()
    st = torch.cuda.current_stream(cuda).cuda_stream
    gr = torch.randn(N, F, 3, 5, generator=g).to(cuda)
    want = torch.empty(N, V, 5, device=cuda)
    buf = torch.full((N * V * 5 + 64,), 12345.0, device=cuda)
    _lib.check(h.lasr_face_gather_backward(gr.data_ptr(), faces.data_ptr(), want.data_ptr(), N, V, F, 5, st), 'scan')
    _lib.check(h.lasr_face_gather_backward_csr(gr.data_ptr(), inc[0].data_ptr(), inc[1].data_ptr(), 1, buf[32:].data_ptr(), N, V, F, 5, st), 'csr')
    assert torch.equal(buf[32:32 + N * V * 5].view(N, V, 5), want) and bool((buf[:32] == 12345.0).all()) and bool((buf[-32:] == 12345.0).all())
    # ---- raster_faces vs raster_inputs + gathers
    cam = torch.from_numpy(v)[None] + 0.05 * torch.randn(N, V, 3, generator=g)
    cam[:, :, 2] += 8
    tex, pp, fl = torch.rand(N, V, 3, generator=g), 0.1 * torch.randn(N, 2, generator=g), torch.rand(N, generator=g) + 8
    eye = [0.0, 0.0, -2.732]
    up_v, up_a = torch.randn(N, F, 3, 3, generator=g).to(cuda), torch.randn(N, F, 3, 9, generator=g).to(cuda)
    a = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    pre, attrs, nf = fused_ops.raster_inputs(*a, eye)
    ((srf.face_vertices(srf.look_at(pre, eye), faces) * up_v).sum() + (srf.face_vertices(attrs, faces) * up_a).sum()).backward()
    b = [t.clone().to(cuda).requires_grad_(True) for t in (cam, tex, pp, fl)]
    fv1, fa1, nf1 = fused_ops.raster_faces(*b, eye, faces[:1].contiguous(), inc)
    ((fv1 * up_v).sum() + (fa1 * up_a).sum()).backward()
    assert torch.equal(b[0].grad, a[0].grad) and torch.equal(b[1].grad, a[1].grad)
    assert rel(b[2].grad, a[2].grad) <= 2e-5 and rel(b[3].grad, a[3].grad) <= 2e-5
    if name.startswith('tet'):
        assert float(b[0].grad[:, 4:].abs().max()) == 0.0                       # vertices no face uses get a zero gradient
    # ---- mesh regularisers (closed mesh only: the flatten term needs two faces per edge)
    if name.startswith('tet'):
        vt, ft = torch.from_numpy(v[:4]), torch.from_numpy(f)
        lap, arap, flat = (loss_utils.LaplacianLoss(vt, ft).to(cuda), loss_utils.ARAPLoss(vt, ft).to(cuda), loss_utils.FlattenLoss(ft).to(cuda))
        x = (vt[None] + 0.05 * torch.randn(2, 4, 3, generator=g)).to(cuda)
        d0, d1 = (vt[None] + 0.05 * torch.randn(1, 4, 3, generator=g)).to(cuda), (vt[None] + 0.05 * torch.randn(1, 4, 3, generator=g)).to(cuda)
        p = [t.clone().requires_grad_(True) for t in (x, d0, d1)]
        (lap(p[0]).sum() + 2 * flat(p[0]).sum() + 3 * arap(p[1], p[2]).sum()).backward()
        q = [t.clone().requires_grad_(True) for t in (x, d0, d1)]
        l1, f1, a1 = fused_ops.mesh_regularisers(q[0], q[1], q[2], lap, flat, arap)
        (l1.sum() + 2 * f1.sum() + 3 * a1.sum()).backward()
        assert torch.equal(q[1].grad, p[1].grad) and torch.equal(q[2].grad, p[2].grad) and rel(q[0].grad, p[0].grad) <= 1e-6


@pytest.mark.parametrize('M,H,K', [(16, 8, 21), (4, 1, 36), (2, 1, 1), (6, 3, 2)])
def test_bone_fixup_with_the_pair_angle_equals_fixup_plus_rotation_distance(cuda, M, H, K):
    # lasr_bone_fixup_pair_*: Rmat / Tmat of the plain fix-up and the angles of geodesic_distance(quat[:half], quat[half:]), with the
    # gradient of quat = the sum autograd forms of the two parts -- torch.equal
    g = torch.Generator().manual_seed(M * 10 + K)
    q, _ = torch.linalg.qr(torch.randn(M * K, 3, 3, generator=g))
    q[M * K // 2] = q[0]                                            # one identical pair: cos = 1, zero gradient branch
    leaves = [q.reshape(M * K, 9), torch.randn(M * K, 2, generator=g), torch.randn(M * K, 1, generator=g) + 8,
              0.3 * torch.randn(H, max(K - 1, 1) * 3, generator=g)]
    upR, upT, upA = torch.randn(M * K, 3, 3, generator=g).to(cuda), torch.randn(M * K, 3, generator=g).to(cuda), torch.randn(M * K // 2, generator=g).to(cuda)
    a = [t.clone().to(cuda).requires_grad_(True) for t in leaves]
    R0, T0 = fused_ops.bone_fixup(a[0], a[1], a[2], a[3], H, K)
    q0, q1 = a[0].view(2, -1, 3, 3).unbind(0)
    A0 = fused_ops.geodesic_distance(q0, q1)
    ((R0 * upR).sum() + (T0 * upT).sum() + (A0 * upA).sum()).backward()
    b = [t.clone().to(cuda).requires_grad_(True) for t in leaves]
    R1, T1, A1 = fused_ops.bone_fixup(b[0], b[1], b[2], b[3], H, K, pair_angle=True)
    ((R1 * upR).sum() + (T1 * upT).sum() + (A1 * upA).sum()).backward()
    assert torch.equal(R1, R0) and torch.equal(T1, T0) and torch.equal(A1, A0)
    for x, y, name in zip(b, a, ('quat', 'trans', 'depth', 'rest_ts')):
        if name == 'rest_ts' and K == 1:
            continue
        assert torch.equal(x.grad, y.grad), name
    # only the angle used: the other outputs' gradients arrive as None
    c = [t.clone().to(cuda).requires_grad_(True) for t in leaves]
    (fused_ops.bone_fixup(c[0], c[1], c[2], c[3], H, K, pair_angle=True)[2] * upA).sum().backward()
    d = leaves[0].clone().to(cuda).requires_grad_(True)
    d0, d1 = d.view(2, -1, 3, 3).unbind(0)
    (fused_ops.geodesic_distance(d0, d1) * upA).sum().backward()
    assert torch.equal(c[0].grad, d.grad)


@pytest.mark.parametrize('level,N,NA,NC,P', [(3, 16, 8, 8, 20), (2, 4, 2, 1, 35), (1, 2, 0, 3, 1)])
def test_step_regularisers_with_the_chamfer_pair_equal_the_four_criteria(cuda, level, N, NA, NC, P):
    v, f = synth.geodesic_sphere(2 ** level)
    V = v.shape[0]
    g = torch.Generator().manual_seed(level + NC)
    faces = torch.from_numpy(np.asarray(f, np.int64))
    lap = loss_utils.LaplacianLoss(torch.from_numpy(v), faces).to(cuda)
    arap = loss_utils.ARAPLoss(torch.from_numpy(v), faces).to(cuda)
    flat = loss_utils.FlattenLoss(faces).to(cuda)
    x = (torch.from_numpy(v)[None] + 0.05 * torch.randn(N, V, 3, generator=g)).to(cuda)
    d0 = (torch.from_numpy(v)[None] + 0.05 * torch.randn(NA, V, 3, generator=g)).to(cuda) if NA else x.new_empty(0, V, 3)
    d1 = (torch.from_numpy(v)[None] + 0.05 * torch.randn(NA, V, 3, generator=g)).to(cuda) if NA else x.new_empty(0, V, 3)
    ctl = (0.3 * torch.randn(NC, P, 3, generator=g)).to(cuda)
    flip = torch.tensor([-1., 1., 1.], device=cuda)
    ups = [torch.randn(n, generator=g).to(cuda) for n in (N, N, NA, NC)]

    a = [t.clone().requires_grad_(True) for t in (x, d0, d1, ctl)]
    l0, f0 = lap(a[0]), flat(a[0])
    a0 = arap(a[1], a[2]) if NA else x.new_zeros(0)
    c0 = fused_ops.chamfer(a[3], a[3] * flip)
    ((l0 * ups[0]).sum() + (f0 * ups[1]).sum() + (a0 * ups[2]).sum() + (c0 * ups[3]).sum()).backward()
    b = [t.clone().requires_grad_(True) for t in (x, d0, d1, ctl)]
    l1, f1, a1, c1 = fused_ops.mesh_regularisers(b[0], b[1], b[2], lap, flat, arap, (b[3], b[3] * flip))
    ((l1 * ups[0]).sum() + (f1 * ups[1]).sum() + (a1 * ups[2]).sum() + (c1 * ups[3]).sum()).backward()
    assert torch.equal(l1.detach(), l0.detach()) and torch.equal(f1.detach(), f0.detach()) and torch.equal(c1.detach(), c0.detach())
    assert torch.equal(b[3].grad, a[3].grad)
    if NA:
        assert torch.equal(a1.detach(), a0.detach()) and torch.equal(b[1].grad, a[1].grad) and torch.equal(b[2].grad, a[2].grad)
    assert rel(b[0].grad, a[0].grad) <= 1e-6
    # the Chamfer result unused: its gradient arrives as None, the control points get zeros
    e = [t.clone().requires_grad_(True) for t in (x, d0, d1, ctl)]
    res = fused_ops.mesh_regularisers(e[0], e[1], e[2], lap, flat, arap, (e[3], e[3] * flip))
    (res[0] * ups[0]).sum().backward()
    assert e[3].grad is None or float(e[3].grad.abs().max()) == 0.0


@pytest.mark.parametrize('B,H,K', [(1, 8, 21), (2, 1, 36), (1, 3, 1)])
def test_pose_chain_is_its_four_operators_in_one_launch(cuda, B, H, K):
    # fused_ops.pose_chain against intrinsics -> quat_to_rotmat -> repeat over the hypotheses -> bone_fixup(pair_angle) ->
    # project_points: the same values (every phase is its stand-alone kernel's expression sequence) and the same gradients of one
    # scalar that touches every output
    from lasr_amd.nnutils import fused_ops
    g = torch.Generator(device='cpu').manual_seed(B * 100 + H * 10 + K)
    r = lambda *s: torch.randn(*s, generator=g).to(cuda)                                            # noqa: E731
    n2, M = 2 * B, 2 * B * H
    cams = torch.cat([1 + 0.3 * torch.rand(n2, 1, generator=g), torch.randn(n2, 6, generator=g)], 1).to(cuda)
    pp, IS = 20 * r(n2, 2), 256
    leaf = lambda t: t.clone().requires_grad_(True)                                                 # noqa: E731
    base = dict(scale=1 + 0.1 * r(n2, H).abs(), depth=8 + r(n2, K).abs(), ppoint=0.1 * r(n2, 2),
                quat=torch.nn.functional.normalize(r(M * K, 4) + torch.tensor([0, 0, 0, 3.], device=cuda), dim=1),
                trans=0.1 * r(n2 * K, 2), rest=0.2 * r(H, max(K - 1, 0) * 3), ctl=0.2 * r(H, max(K - 1, 0) * 3))
    wts = {}

    def loss(outs):
        tot = 0
        for i, o in enumerate(outs):
            if o is None:
                continue
            if i not in wts:
                wts[i] = torch.randn(o.shape, generator=g).to(cuda)
            tot = tot + (o * wts[i]).sum()
        return tot

    def separate(v):
        scale, depth, ppoint = fused_ops.intrinsics(cams, pp, v['scale'], v['depth'], v['ppoint'], IS)
        rot = fused_ops.quat_to_rotmat(v['quat']).reshape(-1, 9)
        depth = depth.reshape(-1, 1).view(n2, 1, K, 1).repeat(1, H, 1, 1).view(-1, 1)
        trans = v['trans'].view(n2, 1, K, 2).repeat(1, H, 1, 1).view(-1, 2)
        Rmat, Tmat, ang = fused_ops.bone_fixup(rot, trans, depth, v['rest'] if K > 1 else None, H, K, pair_angle=True)
        proj = fused_ops.project_points(v['rest'], v['ctl'], Rmat, Tmat, ppoint, scale, H, K) if K > 1 else None
        return scale, ppoint, Rmat, Tmat, trans, depth, ang, proj

    def fused(v):
        return fused_ops.pose_chain(cams, pp, v['scale'], v['depth'], v['ppoint'], v['quat'], v['trans'], v['rest'], v['ctl'], H, K, IS)

    res = {}
    for name, fn in (('separate', separate), ('fused', fused)):
        v = {k: leaf(t) for k, t in base.items()}
        outs = fn(v)
        loss(outs).backward()
        res[name] = (outs, {k: t.grad for k, t in v.items()})
    for a, b in zip(*(res[n][0] for n in ('separate', 'fused'))):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == b.shape and torch.equal(a, b)
    for k in base:
        ga, gb = res['separate'][1][k], res['fused'][1][k]
        if K == 1 and k in ('rest', 'ctl'):
            assert gb is None or not gb.any()
            continue
        assert ga is not None and gb is not None and ga.shape == gb.shape
        assert (ga - gb).abs().max() <= 1e-6 * max(1., float(ga.abs().max())), k
