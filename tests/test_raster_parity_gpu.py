"""HIP soft-rasteriser vs the CPU oracle, through the C ABI (via the autograd Function).

Bars (BASELINE.json north_star): rendered image max-abs <= 1e-4 against the fp32
op-order-faithful oracle; hard-mode face-index map bit-exact; gradients within
1e-3 of the largest gradient magnitude (the HIP backward sums in a different,
but deterministic, order than the reference's atomics).
"""
import itertools
import os

import numpy as np
import pytest
import torch

from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMG_TOL = 1e-4
GRAD_REL = 1e-3


def run_hip(dev, fv, ft, IS, g=None, **kw):
    tfv = torch.from_numpy(fv).to(dev).requires_grad_(g is not None)
    tft = torch.from_numpy(ft).to(dev).requires_grad_(g is not None)
    if g is None:
        img, aggr = srf.soft_rasterize_raw(tfv, tft, IS, kw['background_color'], kw['near'], kw['far'],
                                           kw['fill_back'], kw['eps'], kw['sigma_val'], kw['dist_func'],
                                           kw['dist_eps'], kw['gamma_val'], kw['aggr_func_rgb'],
                                           kw['aggr_func_alpha'], kw['texture_type'])
        return img.cpu().numpy(), aggr.cpu().numpy()
    img = srf.soft_rasterize(tfv, tft, IS, **kw)
    img.backward(torch.from_numpy(g).to(dev))
    return img.detach().cpu().numpy(), tfv.grad.cpu().numpy(), tft.grad.cpu().numpy()


def compare(oracle, dev, fv, ft, IS, kw, seed=2, check_grad=True):
    ref = oracle.forward(fv, ft, IS, **kw)
    img, aggr = run_hip(dev, fv, ft, IS, **kw)
    err = np.abs(img - ref['soft_colors']).max()
    assert err <= IMG_TOL, 'image max-abs %.3e' % err
    if kw['aggr_func_rgb'] == 'hard':
        assert np.array_equal(aggr[:, 1], ref['aggrs_info'][:, 1]), 'face-index map differs'
        assert np.array_equal(aggr[:, 0], ref['aggrs_info'][:, 0]), 'z-buffer differs'
    else:
        assert np.abs(aggr[:, 1] - ref['aggrs_info'][:, 1]).max() <= 1e-6
        np.testing.assert_allclose(aggr[:, 0], ref['aggrs_info'][:, 0], rtol=1e-4, atol=1e-30)
    if not check_grad:
        return err
    g = synth.upstream_grad(fv.shape[0], IS, seed)
    gf_ref, gt_ref = oracle.backward(ref, g, IS, **kw)
    img2, gf, gt = run_hip(dev, fv, ft, IS, g=g, **kw)
    assert np.abs(img2 - ref['soft_colors']).max() <= IMG_TOL
    for name, a, b in (('grad_faces', gf, gf_ref), ('grad_textures', gt, gt_ref)):
        assert np.isfinite(a).all() == np.isfinite(b).all()
        scale = max(np.abs(b[np.isfinite(b)]).max() if np.isfinite(b).any() else 0.0, 1e-20)
        d = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max()
        assert d <= GRAD_REL * scale, '%s: max diff %.3e vs scale %.3e' % (name, d, scale)
    return err


def small_batch(nu, IS, n=2):
    fv, ft, near, far = synth.raster_batch(nu, 3, count=n)
    return fv, ft, near, far


@pytest.mark.parametrize('dist,rgb,alpha', list(itertools.product(
    ['hard', 'barycentric', 'euclidean'], ['hard', 'softmax'], ['hard', 'sum', 'prod'])))
def test_all_mode_combinations_vertex(oracle, cuda, dist, rgb, alpha):
    fv, ft, near, far = small_batch(4, 64)
    kw = dict(synth.LASR_MODES, near=near, far=far, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha)
    compare(oracle, cuda, fv, ft, 64, kw)


@pytest.mark.parametrize('res', [1, 2, 3])
@pytest.mark.parametrize('rgb', ['hard', 'softmax'])
def test_surface_textures(oracle, cuda, res, rgb):
    fv, _, near, far = small_batch(4, 48)
    rng = np.random.default_rng(5)
    ft = rng.uniform(0, 1, (fv.shape[0], fv.shape[1], res * res, 3)).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=near, far=far, texture_type='surface', aggr_func_rgb=rgb)
    compare(oracle, cuda, fv, ft, 48, kw)


@pytest.mark.parametrize('IS', [1, 7, 16, 20, 33])
def test_ragged_image_sizes(oracle, cuda, IS):
    fv, ft, near, far = small_batch(2, IS)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    compare(oracle, cuda, fv, ft, IS, kw)


def test_single_sided_and_sigma(oracle, cuda):
    fv, ft, near, far = small_batch(4, 64)
    for fill_back, sigma in ((False, 1e-4), (True, 1e-5), (False, 1e-5)):
        kw = dict(synth.LASR_MODES, near=near, far=far, fill_back=fill_back, sigma_val=sigma)
        compare(oracle, cuda, fv, ft, 64, kw)


def test_depth_culling_keeps_alpha_but_not_gradient(oracle, cuda):
    # near plane in the middle of the object: faces in front still count for alpha (K.cu:409-424)
    fv, ft, near, far = small_batch(4, 64)
    kw = dict(synth.LASR_MODES, near=10.0, far=far)
    compare(oracle, cuda, fv, ft, 64, kw)


def test_large_faces_and_more_than_one_list_round(oracle, cuda):
    # two screen-filling triangles + 1500 small ones: exercises big bboxes in the face-major
    # backward and a face count above the 1024-entry LDS list of the forward
    rng = np.random.default_rng(7)
    F = 1500
    c = rng.uniform(-0.9, 0.9, (1, F, 1, 2))
    tri = c + rng.uniform(-0.06, 0.06, (1, F, 3, 2))
    z = rng.uniform(2, 4, (1, F, 3, 1))
    fv = np.concatenate([tri, z], -1).astype(np.float32)
    fv[0, 0] = [[-1.5, -1.2, 3], [1.4, -1.1, 3.5], [0.1, 1.6, 2.5]]
    fv[0, 700] = [[-0.9, 0.8, 2.2], [0.95, 0.9, 3.9], [0.0, -0.97, 3.0]]
    ft = rng.uniform(0, 1, fv.shape).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=1.0, far=5.0)
    compare(oracle, cuda, fv, ft, 64, kw)
    kw = dict(kw, aggr_func_rgb='hard', dist_func='hard', aggr_func_alpha='hard')
    compare(oracle, cuda, fv, ft, 64, kw)


def test_empty_and_degenerate_inputs(oracle, cuda):
    kw = dict(synth.LASR_MODES, near=1.0, far=5.0)
    # no faces at all: background image, alpha 0
    fv = np.zeros((2, 0, 3, 3), np.float32)
    img, _ = run_hip(cuda, fv, fv.copy(), 16, **kw)
    assert np.array_equal(img[:, :3], np.ones_like(img[:, :3])) and np.array_equal(img[:, 3], np.zeros_like(img[:, 3]))
    # zero-area and duplicated faces (det clamp path, K.cu:282)
    fv = np.array([[[[0.1, 0.1, 2], [0.1, 0.1, 2], [0.1, 0.1, 2]],
                    [[-0.5, -0.5, 3], [0.5, -0.5, 3], [0.0, 0.5, 3]],
                    [[-0.5, -0.5, 3], [0.5, -0.5, 3], [0.0, 0.5, 3]],
                    [[-0.5, 0.0, 2.5], [0.0, 0.0, 2.5], [0.5, 0.0, 2.5]]]], np.float32)
    ft = np.random.default_rng(3).uniform(0, 1, fv.shape).astype(np.float32)
    for modes in (dict(), dict(aggr_func_rgb='hard', dist_func='hard', aggr_func_alpha='hard')):
        compare(oracle, cuda, fv, ft, 32, dict(kw, **modes), check_grad=False)


def test_lasr_config_m1_256(oracle, cuda):
    # spot3 stage-0 sized mesh at the full 256x256 (BASELINE configs[1])
    fv, ft, near, far = synth.raster_batch(8, 3, count=2)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    err = compare(oracle, cuda, fv, ft, 256, kw)
    print('M1 256x256 image max-abs err %.3e' % err)


def test_lasr_config_m2_256(oracle, cuda):
    # the ~1.2k vertex / 2.3k face mesh of the headline metric
    fv, ft, near, far = synth.raster_batch(11, 3, count=1)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    err = compare(oracle, cuda, fv, ft, 256, kw)
    print('M2 256x256 image max-abs err %.3e' % err)


def test_backward_is_deterministic(cuda):
    fv, ft, near, far = synth.raster_batch(8, 3, count=2)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    g = synth.upstream_grad(2, 128)
    a = run_hip(cuda, fv, ft, 128, g=g, **kw)
    b = run_hip(cuda, fv, ft, 128, g=g, **kw)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_cpu_tensor_is_rejected_like_the_reference():
    t = torch.zeros(1, 1, 3, 3)
    with pytest.raises(TypeError):
        srf.soft_rasterize(t, t, 8)


def test_device_resident_near_far_match_host_floats(cuda):
    # LASR keeps near/far as 0-dim device tensors (mesh_net.py:306-311): the *_dev entry points must give the
    # same bits as passing the floats, forward and backward, without a host sync
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    g = synth.upstream_grad(2, 64)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    a = run_hip(cuda, fv, ft, 64, g=g, **kw)
    kw_dev = dict(kw, near=torch.tensor(near, device=cuda), far=torch.tensor(far, device=cuda))
    b = run_hip(cuda, fv, ft, 64, g=g, **kw_dev)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_division_by_reciprocal_is_bit_exact(cuda):
    # the forward kernel divides by per-face / per-launch constants through RN(1/b) + two FMA corrections
    # (sr_device.h: div_by_recip); it must return the IEEE quotient bit for bit
    from lasr_amd import _lib
    h = _lib.lib()
    g = torch.Generator(device='cpu').manual_seed(3)
    n = 1 << 24
    total = 0
    for scale_a, scale_b in ((1.0, 1.0), (1e-3, 30.0), (50.0, 1e-6), (1e-12, 1e3)):
        a = (torch.randn(n, generator=g) * scale_a).to(cuda)
        b = (torch.randn(n, generator=g) * scale_b).to(cuda)
        # structured divisors too: values the kernels really divide by (gamma, sigma, depths, squared edge lengths)
        b[:8] = torch.tensor([1e-2, 1e-4, 1e-5, 10.0, 9.5, 3.0, 1.0, 0.75], device=cuda)
        bad = torch.zeros(1, dtype=torch.int32, device=cuda)
        rc = h.lasr_selftest_div(a.data_ptr(), b.data_ptr(), bad.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'lasr_selftest_div')
        total += int(bad.item())
    assert total <= 8, '%d of %d quotients differ' % (total, 4 * n)      # 2^-23 exceptional divisors at most


def test_shared_divisor_division_is_bit_exact(cuda):
    # clip/normalise (K.cu:53-58) divides three clipped barycentrics by their sum: one reciprocal refinement shared by the
    # three quotients (sr_device.h: div3_shared) must give the IEEE quotients bit for bit over the values that occur
    from lasr_amd import _lib
    h = _lib.lib()
    g = torch.Generator(device='cpu').manual_seed(4)
    n = 1 << 23
    total = 0
    for mode in range(3):
        a = torch.rand(n, 3, generator=g)
        if mode == 1:
            a = a * (torch.rand(n, 3, generator=g) < 0.6)                 # exact zeros from the clamp
        if mode == 2:
            a = a * 10.0 ** (-8 * torch.rand(n, 3, generator=g))          # small weights
        b = a.sum(1).clamp_min(1e-5) if mode < 2 else torch.rand(n, generator=g) * 3 + 1e-5
        a[:4] = torch.tensor([[1., 0., 0.], [0., 1., 1.], [1., 1., 1.], [0., 0., 0.]])
        b[:4] = torch.tensor([1., 2., 3., 1e-5])
        bad = torch.zeros(1, dtype=torch.int32, device=cuda)
        a, b = a.to(cuda).contiguous(), b.to(cuda).contiguous()
        rc = h.lasr_selftest_div3(a.data_ptr(), b.data_ptr(), bad.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'lasr_selftest_div3')
        total += int(bad.item())
    assert total == 0, '%d of %d quotients differ' % (total, 9 * n)


def test_lasr_config_m2_512(oracle, cuda):
    # BASELINE configs[2] renders at 512x512 (--img_size 512)
    fv, ft, near, far = synth.raster_batch(11, 3, count=1)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    err = compare(oracle, cuda, fv, ft, 512, kw)
    print('M2 512x512 image max-abs err %.3e' % err)


def test_six_channel_pass_equals_two_three_channel_renders(oracle, cuda):
    # two per-vertex attribute triples blended in one pass over the geometry
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    rng = np.random.default_rng(9)
    ft2 = rng.uniform(-2, 2, ft.shape).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    IS = 64
    g = np.concatenate([synth.upstream_grad(2, IS, 2)[:, :3], synth.upstream_grad(2, IS, 3)], 1)   # [2,7,IS,IS]
    tfv = torch.from_numpy(fv).to(cuda).requires_grad_(True)
    t6 = torch.from_numpy(np.concatenate([ft, ft2], -1)).to(cuda).requires_grad_(True)
    img6 = srf.soft_rasterize(tfv, t6, IS, **kw)
    assert img6.shape == (2, 7, IS, IS)
    img6.backward(torch.from_numpy(g).to(cuda))
    # the same through two ordinary renders; the alpha gradient is counted once
    ga = np.concatenate([g[:, 0:3], g[:, 6:7]], 1)
    gb = np.concatenate([g[:, 3:6], np.zeros_like(g[:, 6:7])], 1)
    ia, gfa, gta = run_hip(cuda, fv, ft, IS, g=ga, **kw)
    ib, gfb, gtb = run_hip(cuda, fv, ft2, IS, g=gb, **kw)
    out = img6.detach().cpu().numpy()
    assert np.array_equal(out[:, 0:3], ia[:, :3]) and np.array_equal(out[:, 3:6], ib[:, :3])
    assert np.array_equal(out[:, 6], ia[:, 3])
    gf = tfv.grad.cpu().numpy()
    scale = np.abs(gfa + gfb).max()
    assert np.abs(gf - (gfa + gfb)).max() <= 1e-5 * scale
    gt6 = t6.grad.cpu().numpy()
    assert np.abs(gt6[..., 0:3] - gta).max() <= 1e-5 * np.abs(gta).max()
    assert np.abs(gt6[..., 3:6] - gtb).max() <= 1e-5 * np.abs(gtb).max()
    # and against the oracle
    ref = oracle.forward(fv, ft2, IS, **kw)
    assert np.abs(out[:, 3:6] - ref['soft_colors'][:, :3]).max() <= IMG_TOL
    with pytest.raises(Exception):          # 6 channels exist for LASR's mode combination only
        srf.soft_rasterize(tfv, t6, IS, **dict(kw, aggr_func_rgb='hard'))


def test_nine_channel_pass_equals_three_three_channel_renders(oracle, cuda):
    # the three renders of a LASR step (texture, flow t->t', flow t'->t attributes) share their geometry: one pass, per-channel
    # background (white for the texture triple, black for the positions)
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    rng = np.random.default_rng(11)
    ft2 = rng.uniform(-2, 2, ft.shape).astype(np.float32)
    ft3 = rng.uniform(-2, 2, ft.shape).astype(np.float32)
    IS = 64
    kw = dict(synth.LASR_MODES, near=near, far=far)
    bgs = ([1., 1., 1.], [0., 0., 0.], [0.25, 0.5, 0.75])
    g = np.concatenate([synth.upstream_grad(2, IS, 2)[:, :3], synth.upstream_grad(2, IS, 3)[:, :3], synth.upstream_grad(2, IS, 4)], 1)
    assert g.shape == (2, 10, IS, IS)
    tfv = torch.from_numpy(fv).to(cuda).requires_grad_(True)
    t9 = torch.from_numpy(np.concatenate([ft, ft2, ft3], -1)).to(cuda).requires_grad_(True)
    img9 = srf.soft_rasterize(tfv, t9, IS, **dict(kw, background_color=bgs[0] + bgs[1] + bgs[2]))
    assert img9.shape == (2, 10, IS, IS)
    img9.backward(torch.from_numpy(g).to(cuda))
    out = img9.detach().cpu().numpy()
    gf_sum, zero_a = 0, np.zeros_like(g[:, 9:10])
    for k, (tex, bg) in enumerate(zip((ft, ft2, ft3), bgs)):
        gk = np.concatenate([g[:, 3 * k:3 * k + 3], g[:, 9:10] if k == 0 else zero_a], 1)      # the alpha gradient counted once
        ik, gfk, gtk = run_hip(cuda, fv, tex, IS, g=gk, **dict(kw, background_color=bg))
        assert np.array_equal(out[:, 3 * k:3 * k + 3], ik[:, :3]), 'triple %d' % k
        assert np.array_equal(out[:, 9], ik[:, 3])
        gt9 = t9.grad.cpu().numpy()[..., 3 * k:3 * k + 3]
        assert np.abs(gt9 - gtk).max() <= 1e-5 * np.abs(gtk).max()
        gf_sum = gf_sum + gfk
    assert np.abs(tfv.grad.cpu().numpy() - gf_sum).max() <= 1e-5 * np.abs(gf_sum).max()
    ref = oracle.forward(fv, ft3, IS, **dict(kw, background_color=bgs[2]))
    assert np.abs(out[:, 6:9] - ref['soft_colors'][:, :3]).max() <= IMG_TOL
    with pytest.raises(Exception):          # 9 channels exist for LASR's mode combination only
        srf.soft_rasterize(tfv, t9, IS, **dict(kw, aggr_func_rgb='hard'))


def test_nine_channel_pass_vs_oracle_at_lasr_size(oracle, cuda):
    # the render LASR.forward issues (spot3 stage-0 sized mesh, 256x256, nine attributes) straight against the oracle: each
    # attribute triple is an oracle render of its own; the face gradient is the sum of the three oracle gradients
    fv, ft, near, far = synth.raster_batch(8, 3, count=2)
    rng = np.random.default_rng(21)
    tex = [ft, rng.uniform(-1, 3, ft.shape).astype(np.float32), rng.uniform(-1, 3, ft.shape).astype(np.float32)]
    bgs = ([1., 1., 1.], [0., 0., 0.], [0., 0., 0.])
    IS = 256
    kw = dict(synth.LASR_MODES, near=near, far=far)
    g = np.concatenate([synth.upstream_grad(2, IS, 5)[:, :3], synth.upstream_grad(2, IS, 6)[:, :3], synth.upstream_grad(2, IS, 7)], 1)
    tfv = torch.from_numpy(fv).to(cuda).requires_grad_(True)
    t9 = torch.from_numpy(np.concatenate(tex, -1)).to(cuda).requires_grad_(True)
    img = srf.soft_rasterize(tfv, t9, IS, **dict(kw, background_color=bgs[0] + bgs[1] + bgs[2]))
    img.backward(torch.from_numpy(g).to(cuda))
    out, gf_ref = img.detach().cpu().numpy(), 0
    for k in range(3):
        okw = dict(kw, background_color=bgs[k])
        ref = oracle.forward(fv, tex[k], IS, **okw)
        assert np.abs(out[:, 3 * k:3 * k + 3] - ref['soft_colors'][:, :3]).max() <= IMG_TOL
        assert np.abs(out[:, 9] - ref['soft_colors'][:, 3]).max() <= IMG_TOL
        gk = np.concatenate([g[:, 3 * k:3 * k + 3], g[:, 9:10] if k == 0 else np.zeros_like(g[:, 9:10])], 1)
        gf_k, gt_k = oracle.backward(ref, gk, IS, **okw)
        gf_ref = gf_ref + gf_k
        gt9 = t9.grad.cpu().numpy()[..., 3 * k:3 * k + 3]
        assert np.abs(gt9 - gt_k).max() <= GRAD_REL * np.abs(gt_k).max(), 'attribute gradient of triple %d' % k
    assert np.abs(tfv.grad.cpu().numpy() - gf_ref).max() <= GRAD_REL * np.abs(gf_ref).max()


def test_integration_stub_binds_like_the_reference_extension(oracle, cuda):
    # INTEGRATION.md "Option B": the ctypes module a reference maintainer would drop in for
    # soft_renderer.cuda.soft_rasterize -- same two entry points, same argument order as the pybind functions
    # (soft_rasterize_cuda.cpp:59-76, 94-114), called here exactly as functional/soft_rasterize.py:47-62, 86-100 does
    import math
    import re
    import types
    from lasr_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    code = re.search(r'```python\n(# soft_renderer/cuda/soft_rasterize.py.*?)```', text, re.S).group(1)
    code = code.replace("'/path/to/lasr_amd/csrc/liblasr_hip.so'", repr(_lib.LIB_PATH))
    stub = types.ModuleType('soft_rasterize_stub')
    exec(compile(code, 'INTEGRATION.md', 'exec'), stub.__dict__)

    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    IS, N, F = 48, fv.shape[0], fv.shape[1]
    kw = dict(synth.LASR_MODES, near=near, far=far)
    faces = torch.from_numpy(fv).to(cuda).reshape(N, F, 3, 3).contiguous()
    textures = torch.from_numpy(ft).to(cuda).reshape(N, F, 3, 3).contiguous()
    faces_info = torch.zeros(N, F, 27, device=cuda)
    aggrs_info = torch.zeros(N, 2, IS, IS, device=cuda)
    soft_colors = torch.ones(N, 4, IS, IS, device=cuda)
    scalars = (IS, near, far, kw['eps'], kw['sigma_val'], 2, math.log(1. / kw['dist_eps'] - 1.), kw['gamma_val'], 1, 2, 1, True)
    out = stub.forward_soft_rasterize(faces, textures, faces_info, aggrs_info, soft_colors, *scalars)
    assert out[2] is soft_colors
    ref = oracle.forward(fv, ft, IS, **kw)
    assert np.abs(soft_colors.cpu().numpy() - ref['soft_colors']).max() <= IMG_TOL
    g = synth.upstream_grad(N, IS)
    grad_faces, grad_textures = torch.zeros_like(faces), torch.zeros_like(textures)
    stub.backward_soft_rasterize(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                                 torch.from_numpy(g).to(cuda), *scalars)
    rgf, rgt = oracle.backward(ref, g, IS, **kw)
    scale = np.abs(rgf).max()
    assert np.abs(grad_faces.cpu().numpy().reshape(rgf.shape) - rgf).max() <= 1e-3 * scale
    with pytest.raises(RuntimeError):                       # CHECK_INPUT of the pybind layer (soft_rasterize_cuda.cpp:54-56)
        stub.forward_soft_rasterize(faces.cpu(), textures, faces_info, aggrs_info, soft_colors, *scalars)
    # float64 tensors through the same two functions (the reference dispatches on faces.type(), K.cu:701,716,780)
    d = [t.double() for t in (faces, textures)]
    info64, aggr64 = torch.zeros(N, F, 27, dtype=torch.float64, device=cuda), torch.zeros(N, 2, IS, IS, dtype=torch.float64, device=cuda)
    col64 = torch.ones(N, 4, IS, IS, dtype=torch.float64, device=cuda)
    stub.forward_soft_rasterize(d[0], d[1], info64, aggr64, col64, *scalars)
    ref64 = oracle.forward(fv.astype(np.float64), ft.astype(np.float64), IS, dtype=np.float64, **kw)
    assert np.abs(col64.cpu().numpy() - ref64['soft_colors']).max() <= 1e-9
    gf64, gt64 = torch.zeros_like(d[0]), torch.zeros_like(d[1])
    stub.backward_soft_rasterize(d[0], d[1], col64, info64, aggr64, gf64, gt64, torch.from_numpy(g).to(cuda).double(), *scalars)
    rgf64, _ = oracle.backward(ref64, g.astype(np.float64), IS, dtype=np.float64, **kw)
    assert np.abs(gf64.cpu().numpy().reshape(rgf64.shape) - rgf64).max() <= 1e-9 * np.abs(rgf64).max()


def test_relaxed_forward_math_stays_inside_the_tolerance(oracle, cuda):
    # LASR_SR_RELAXED_MATH (a per-call flag): the distance and threshold decision stay bit-faithful, the rest of the pixel
    # pipeline is fp32 rcp/exp arithmetic; the image must still meet the north-star bar against the op-faithful oracle
    from lasr_amd import _lib
    from lasr_amd.soft_renderer import functional as srf
    worst = 0.0
    try:
        for nu, IS, count in ((4, 64, 2), (8, 128, 2), (11, 256, 1)):
            fv, ft, near, far = synth.raster_batch(nu, 26, count=count)
            kw = dict(synth.LASR_MODES, near=near, far=far)
            ref = oracle.forward(fv, ft, IS, **kw)
            srf.set_forward_flags(0)
            exact, _ = run_hip(cuda, fv, ft, IS, **kw)
            srf.set_forward_flags(_lib.SR_RELAXED_MATH)
            relaxed, gf, _ = run_hip(cuda, fv, ft, IS, g=synth.upstream_grad(count, IS), **kw)
            err = np.abs(relaxed - ref['soft_colors']).max()
            worst = max(worst, err)
            assert err <= IMG_TOL and np.abs(relaxed - exact).max() > 0          # a different (cheaper) rounding sequence
            rgf, _ = oracle.backward(ref, synth.upstream_grad(count, IS), IS, **kw)
            assert np.abs(gf - rgf).max() <= GRAD_REL * np.abs(rgf).max()       # backward reads the relaxed image / aggregates
    finally:
        srf.set_forward_flags(_lib.SR_DEFAULT_FLAGS)
    print('relaxed forward: worst image max-abs error %.2e' % worst)


def test_faces_info_tensor_matches_the_reference_layout_bit_for_bit(oracle, cuda):
    # the optional [N,F,27] `faces_info` output (K.cu:245-305: inverse matrix, vertex Gram matrix + 1, first-obtuse-corner
    # flags, 6 unused slots) is written in the reference's layout for callers that still want it
    for nu, n in ((4, 2), (11, 1)):
        fv, ft, near, far = synth.raster_batch(nu, 7, count=n)
        kw = dict(synth.LASR_MODES, near=near, far=far)
        ref = oracle.forward(fv, ft, 32, **kw)
        _, _, info = srf.soft_rasterize_raw(torch.from_numpy(fv).to(cuda), torch.from_numpy(ft).to(cuda), 32,
                                            kw['background_color'], near, far, kw['fill_back'], kw['eps'], kw['sigma_val'],
                                            kw['dist_func'], kw['dist_eps'], kw['gamma_val'], kw['aggr_func_rgb'],
                                            kw['aggr_func_alpha'], kw['texture_type'], want_faces_info=True)
        info = info.cpu().numpy()
        assert info.shape == ref['faces_info'].shape == (n, fv.shape[1], 27)
        assert np.array_equal(info[..., 0:9], ref['faces_info'][..., 0:9]), 'inverse matrix'
        assert np.array_equal(info[..., 9:18], ref['faces_info'][..., 9:18]), 'Gram matrix'
        assert np.array_equal(info[..., 18:21], ref['faces_info'][..., 18:21]), 'obtuse flags'
        assert not info[..., 21:].any()


def test_backward_on_the_forwards_records_equals_a_rebuild(cuda):
    # LASR_SR_RECORDS_VALID: the backward may skip its face setup when the forward's workspace is untouched -- same bits
    from lasr_amd import _lib
    import math
    h = _lib.lib()
    fv, ft, near, far = synth.raster_batch(8, 5, count=3)
    N, F, IS = fv.shape[0], fv.shape[1], 96
    m = synth.LASR_MODES
    tail = (float(m['eps']), float(m['sigma_val']), 2, float(math.log(1. / m['dist_eps'] - 1.)), float(m['gamma_val']), 1, 2, 1, 1)
    tfv = torch.from_numpy(fv).to(cuda).reshape(N, F, 9).contiguous()
    tft = torch.from_numpy(ft).to(cuda).reshape(N, F, 9).contiguous()
    g = torch.from_numpy(synth.upstream_grad(N, IS, 5)).to(cuda)
    colors = torch.ones(N, 4, IS, IS, device=cuda)
    aggrs = torch.empty(N, 2, IS, IS, device=cuda)
    ws = torch.empty(h.lasr_sr_workspace_bytes(N, F, 3, IS), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(h.lasr_sr_forward_ex(tfv.data_ptr(), tft.data_ptr(), None, aggrs.data_ptr(), colors.data_ptr(), ws.data_ptr(),
                                    ws.numel(), N, F, 3, 3, IS, float(near), float(far), None, *tail, 0, st), 'forward_ex')
    out = []
    for flags in (_lib.SR_RECORDS_VALID, 0):
        gf, gt = torch.zeros(N, F, 9, device=cuda), torch.zeros(N, F, 9, device=cuda)
        _lib.check(h.lasr_sr_backward_ex(tfv.data_ptr(), tft.data_ptr(), colors.data_ptr(), aggrs.data_ptr(), gf.data_ptr(),
                                         gt.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), N, F, 3, 3, IS, float(near),
                                         float(far), None, *tail, flags, st), 'backward_ex')
        out.append((gf.cpu().numpy(), gt.cpu().numpy()))
    assert np.abs(out[0][0]).max() > 0
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # the backward reads the records and rects only -- a workspace of the size WITHOUT the forward's tile-order table
    # (lasr_sr_workspace_bytes(N, F, T, 0), what backward-only callers of ABI versions 1-2 allocated) is accepted; smaller is not
    small = h.lasr_sr_workspace_bytes(N, F, 3, 0)
    assert small < ws.numel()
    ws2 = torch.empty(small, dtype=torch.uint8, device=cuda)
    gf, gt = torch.zeros(N, F, 9, device=cuda), torch.zeros(N, F, 9, device=cuda)
    _lib.check(h.lasr_sr_backward_ex(tfv.data_ptr(), tft.data_ptr(), colors.data_ptr(), aggrs.data_ptr(), gf.data_ptr(),
                                     gt.data_ptr(), g.data_ptr(), ws2.data_ptr(), ws2.numel(), N, F, 3, 3, IS, float(near),
                                     float(far), None, *tail, 0, st), 'backward_ex on a records-only workspace')
    assert np.array_equal(gf.cpu().numpy(), out[1][0]) and np.array_equal(gt.cpu().numpy(), out[1][1])
    assert h.lasr_sr_backward_ex(tfv.data_ptr(), tft.data_ptr(), colors.data_ptr(), aggrs.data_ptr(), gf.data_ptr(), gt.data_ptr(),
                                 g.data_ptr(), ws2.data_ptr(), small - 1, N, F, 3, 3, IS, float(near), float(far), None, *tail, 0,
                                 st) == -3


def test_invalidate_records_accepts_a_device_without_an_index(cuda):
    # torch.device('cuda').index is None; the helper used to match nothing and leave a pending eager backward on
    # records a graph replay had overwritten
    import importlib
    sr_mod = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    img = srf.soft_rasterize(torch.from_numpy(fv).to(cuda), torch.from_numpy(ft).to(cuda), 32, **dict(synth.LASR_MODES, near=near, far=far))
    key = (cuda.index, torch.cuda.current_stream(cuda).cuda_stream)
    for dev in ('cuda', torch.device('cuda'), cuda, 'cuda:0'):
        before = sr_mod._records_of[key]
        srf.invalidate_records(dev)
        assert sr_mod._records_of[key] == before + 1, dev
    with pytest.raises(ValueError):
        srf.invalidate_records('cpu')
    del img


@pytest.mark.parametrize('channels', [3, 6, 9])
def test_backward_may_overwrite_uninitialised_gradient_buffers(cuda, channels):
    # LASR_SR_GRADS_OVERWRITE (vertex textures): every gradient element is stored by the wavefront that owns its face, so the
    # caller need not zero the buffers -- same bits as accumulating into zeros, including faces that receive no gradient at all
    from lasr_amd import _lib
    import math
    h = _lib.lib()
    fv, ft, near, far = synth.raster_batch(8, 5, count=3)
    fv[1, :40] += 50.                                       # a block of faces far outside the view: empty rects, zero gradient
    N, F, IS, C = fv.shape[0], fv.shape[1], 96, channels
    rng = np.random.default_rng(C)
    attrs = rng.uniform(0, 1, (N, F, 3, C)).astype(np.float32)
    m = synth.LASR_MODES
    tail = (float(m['eps']), float(m['sigma_val']), 2, float(math.log(1. / m['dist_eps'] - 1.)), float(m['gamma_val']), 1, 2, 1, 1)
    tfv = torch.from_numpy(fv).to(cuda).reshape(N, F, 9).contiguous()
    tft = torch.from_numpy(attrs).to(cuda).contiguous()
    g = torch.from_numpy(rng.standard_normal((N, C + 1, IS, IS)).astype(np.float32) / (IS * IS)).to(cuda)
    colors = torch.ones(N, C + 1, IS, IS, device=cuda)
    aggrs = torch.empty(N, 2, IS, IS, device=cuda)
    ws = torch.empty(h.lasr_sr_workspace_bytes(N, F, 3, IS), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(h.lasr_sr_forward_ex(tfv.data_ptr(), tft.data_ptr(), None, aggrs.data_ptr(), colors.data_ptr(), ws.data_ptr(),
                                    ws.numel(), N, F, 3, C, IS, float(near), float(far), None, *tail, 0, st), 'forward_ex')
    out = []
    for flags, init in ((0, 0.0), (_lib.SR_GRADS_OVERWRITE, float('nan'))):
        gf = torch.full((N, F, 9), init, device=cuda)
        gt = torch.full((N, F, 3, C), init, device=cuda)
        _lib.check(h.lasr_sr_backward_ex(tfv.data_ptr(), tft.data_ptr(), colors.data_ptr(), aggrs.data_ptr(), gf.data_ptr(),
                                         gt.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), N, F, 3, C, IS, float(near),
                                         float(far), None, *tail, flags, st), 'backward_ex')
        out.append((gf.cpu().numpy(), gt.cpu().numpy()))
    assert np.abs(out[0][0]).max() > 0 and not out[0][0][1, :40].any()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('rgb,alpha,channels', [(1, 2, 3), (1, 2, 9), (0, 2, 3), (0, 0, 3), (1, 1, 3)])
def test_forward_with_the_background_as_an_argument(cuda, rgb, alpha, channels):
    # lasr_sr_forward_bg: same image and aggregates as the pre-fill + lasr_sr_forward_ex sequence of the reference's caller, written
    # into an UNINITIALISED soft_colors buffer (also where no face lands: hard mode leaves the pre-filled background there)
    from lasr_amd import _lib
    import ctypes
    import math
    h = _lib.lib()
    fv, ft, near, far = synth.raster_batch(6, 4, count=2)
    N, F, IS, C = fv.shape[0], fv.shape[1], 80, channels
    rng = np.random.default_rng(C + rgb)
    attrs = rng.uniform(0, 1, (N, F, 3, C)).astype(np.float32)
    bg = [0.3, 0.6, 0.9, 0., 0.25, 1., 0.5, 0.7, 0.1][:C]
    m = synth.LASR_MODES
    tail = (float(m['eps']), float(m['sigma_val']), 2, float(math.log(1. / m['dist_eps'] - 1.)), float(m['gamma_val']), rgb, alpha, 1, 1)
    tfv = torch.from_numpy(fv).to(cuda).reshape(N, F, 9).contiguous()
    tft = torch.from_numpy(attrs).to(cuda).contiguous()
    ws = torch.empty(h.lasr_sr_workspace_bytes(N, F, 3, IS), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    ref_c = torch.tensor(bg + [1.0], device=cuda).view(1, C + 1, 1, 1).repeat(N, 1, IS, IS).contiguous()
    ref_a = torch.empty(N, 2, IS, IS, device=cuda)
    _lib.check(h.lasr_sr_forward_ex(tfv.data_ptr(), tft.data_ptr(), None, ref_a.data_ptr(), ref_c.data_ptr(), ws.data_ptr(), ws.numel(),
                                    N, F, 3, C, IS, float(near), float(far), None, *tail, 0, st), 'forward_ex')
    out_c = torch.full((N, C + 1, IS, IS), float('nan'), device=cuda)
    out_a = torch.empty(N, 2, IS, IS, device=cuda)
    _lib.check(h.lasr_sr_forward_bg(tfv.data_ptr(), tft.data_ptr(), None, out_a.data_ptr(), out_c.data_ptr(), ws.data_ptr(), ws.numel(),
                                    N, F, 3, C, IS, float(near), float(far), None, *tail, (ctypes.c_float * C)(*bg), 0, st), 'forward_bg')
    assert torch.equal(out_c, ref_c) and torch.equal(out_a, ref_a)
    assert float((out_c[:, 0] == out_c[0, 0, 0, 0]).float().mean()) > 0.2          # a good part of the image is background


def test_autograd_backward_reuses_the_forwards_records_only_while_they_are_there(cuda):
    # small launches skip the backward's setup launch when the workspace still holds the records their own forward wrote
    # (soft_rasterize.py: _records_of); a second render in between, or a backward that rebuilt its records, takes that away
    import importlib
    sr = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    fa, ta, near, far = synth.raster_batch(4, 3, count=2)
    fb = fa[::-1].copy() * np.float32(0.9)
    fb[..., 2] = fa[::-1][..., 2]
    kw = dict(synth.LASR_MODES, near=near, far=far)
    g = torch.from_numpy(synth.upstream_grad(2, 64, 2)).to(cuda)
    key = (cuda.index, torch.cuda.current_stream(cuda).cuda_stream)

    def leaf(x):
        return torch.from_numpy(x).to(cuda).requires_grad_(True)

    def alone(fv):
        v, t = leaf(fv), leaf(ta)
        img = srf.soft_rasterize(v, t, 64, **kw)
        before = sr._records_of[key]
        img.backward(g)
        assert sr._records_of[key] == before                  # reused: the backward wrote no records
        return v.grad.clone(), t.grad.clone()
    want_a, want_b = alone(fa), alone(fb)
    va, tta, vb, ttb = leaf(fa), leaf(ta), leaf(fb), leaf(ta)
    ia = srf.soft_rasterize(va, tta, 64, **kw)
    ib = srf.soft_rasterize(vb, ttb, 64, **kw)               # the workspace now holds B's records
    before = sr._records_of[key]
    ia.backward(g)                                            # must rebuild A's
    assert sr._records_of[key] == before + 1
    ib.backward(g)                                            # ... which took B's away: rebuild again
    assert sr._records_of[key] == before + 2
    for got, want in ((va.grad, want_a[0]), (tta.grad, want_a[1]), (vb.grad, want_b[0]), (ttb.grad, want_b[1])):
        assert torch.equal(got, want)
    old, sr.REUSE_RECORDS_MAX_FACES = sr.REUSE_RECORDS_MAX_FACES, 0     # large launches always rebuild
    try:
        v, t = leaf(fa), leaf(ta)
        img = srf.soft_rasterize(v, t, 64, **kw)
        before = sr._records_of[key]
        img.backward(g)
        assert sr._records_of[key] == before + 1 and torch.equal(v.grad, want_a[0])
    finally:
        sr.REUSE_RECORDS_MAX_FACES = old
    # records written BEHIND the operator's back -- a HIP-graph replay, a direct C-ABI call on the same workspace -- must be
    # announced (invalidate_records; LASRTrainer does so after every replay): the pending backward then rebuilds its records
    from lasr_amd import _lib
    v, t = leaf(fa), leaf(ta)
    img = srf.soft_rasterize(v, t, 64, **kw)
    ws = sr._workspaces[key]
    fbt, tat = torch.from_numpy(fb).to(cuda).reshape(2, -1, 9).contiguous(), torch.from_numpy(ta).to(cuda).reshape(2, -1, 9).contiguous()
    scratch_img, scratch_aggr = torch.ones(2, 4, 64, 64, device=cuda), torch.empty(2, 2, 64, 64, device=cuda)
    _lib.check(_lib.lib().lasr_sr_forward(fbt.data_ptr(), tat.data_ptr(), None, scratch_aggr.data_ptr(), scratch_img.data_ptr(),
                                          ws.data_ptr(), ws.numel(), 2, fbt.shape[1], 3, 64, float(near), float(far), 1e-3, 1e-4, 2,
                                          float(np.log(1. / 1e-4 - 1.)), 1e-2, 1, 2, 1, 1, key[1]), 'direct forward')
    before = sr._records_of[key]
    srf.invalidate_records(cuda)
    assert sr._records_of[key] == before + 1
    img.backward(g)
    assert sr._records_of[key] == before + 2 and torch.equal(v.grad, want_a[0]) and torch.equal(t.grad, want_a[1])
