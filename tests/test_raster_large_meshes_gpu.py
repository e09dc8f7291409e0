"""Meshes with MORE THAN 4096 FACES through every forward kernel, against the reference kernels.

The forward kernels bin faces in groups of 64 (sr_raster.hip `groups_of`): the first 64 group rects are tested up front, the
rest in refills of 64 (`g_next`), u16 list ids are rebased when a tile's groups span more than 65536 faces, and a face
numbering that is not patch-coherent degrades the group test to a full scan.  Until round 4 no collected test had more than
2880 faces (45 groups).  Here:

  (a) geodesic nu = 16 (5120 faces, 80 groups) and nu = 19 (7220 faces, 113 groups) at 256^2 with 1 / 4 / 16 / 64 frames --
      with the default thresholds these launches take the eight-wave, the four-wave, the device-chosen and the one-wave
      kernel -- LASR modes and hard modes (the generic 16x16 kernel), live against oracle/_ref/sr_ref_nofma.so (the
      reference's soft_rasterize_cuda_kernel.cu:370-453 built for gfx950) and, for one frame, against the C oracle;
      every kernel forced in turn on the same input must give the same bits;
  (b) nu = 64 (81920 faces) at 64^2: tiles whose groups span more than 65536 face ids (asserted on the host) -- u16 rebasing;
  (c) the nu = 16 mesh with a shuffled face order (group rects cover the whole object: full scan);
  (d) the reference's own demo asset, database/misc/spot/spot_triangulated.obj (5856 faces, scripts/render_syn.py:71), posed
      as render_syn.py poses it: tests/golden/spot_reference_kernels.npz holds the posed inputs and what the reference
      kernels returned for them (oracle/gen_ref_vectors_large.py).

Bars: image max-abs <= 1e-4 (measured ~3e-7), hard-mode face-index map and z-buffer equal, gradients within 1e-3 of the largest entry.
"""
import os

import numpy as np
import pytest
import torch

from lasr_amd import _lib, synth
from lasr_amd.soft_renderer import functional as srf
from oracle import sr_ref

pytestmark = pytest.mark.gpu

BIG = 10 ** 12
VARIANTS = {'default thresholds': (2200, 14336, 49152),
            'eight waves per 8x8 tile': (BIG, BIG, BIG),
            'four waves per 8x8 tile': (0, BIG, BIG),
            'one wave per 8x8 tile': (0, 0, 0)}
HARD = dict(dist_func='hard', aggr_func_rgb='hard', aggr_func_alpha='hard')
HAVE_REF = sr_ref.available('sr_ref_nofma')
SPOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'spot_reference_kernels.npz')


@pytest.fixture
def thresholds():
    yield lambda name: srf.set_launch_thresholds(*VARIANTS[name])
    srf.set_launch_thresholds()


def hip_forward(dev, fv, ft, IS, kw):
    a = torch.from_numpy(fv).to(dev) if isinstance(fv, np.ndarray) else fv
    b = torch.from_numpy(ft).to(dev) if isinstance(ft, np.ndarray) else ft
    return srf.soft_rasterize_raw(a, b, IS, kw['background_color'], kw['near'], kw['far'], kw['fill_back'], kw['eps'],
                                  kw['sigma_val'], kw['dist_func'], kw['dist_eps'], kw['gamma_val'], kw['aggr_func_rgb'],
                                  kw['aggr_func_alpha'], kw['texture_type'])


def hip_fwd_bwd(dev, fv, ft, IS, kw, g):
    a = torch.from_numpy(fv).to(dev).requires_grad_(True)
    b = torch.from_numpy(ft).to(dev).requires_grad_(True)
    img = srf.soft_rasterize(a, b, IS, **kw)
    img.backward(g)
    return img.detach(), a.grad, b.grad


def check_against(img, aggr, gf, gt, ref_img, ref_aggr, ref_gf, ref_gt, hard):
    """HIP outputs against reference-kernel outputs (torch tensors or numpy arrays of the same shapes)."""
    t = lambda x: torch.as_tensor(x).to(img.device)
    assert float((img - t(ref_img)).abs().max()) <= 1e-4
    if hard:
        assert torch.equal(aggr, t(ref_aggr)), 'face-index map / z-buffer differs from the reference kernel'
    for mine, theirs in ((gf, ref_gf), (gt, ref_gt)):
        if mine is None:
            continue
        theirs = t(theirs)
        scale = max(float(theirs.abs().max()), 1e-30)
        assert float((mine.reshape(theirs.shape) - theirs).abs().max()) <= 1e-3 * scale


def reference_live(dev, fv, ft, IS, kw, g):
    tfv, tft = torch.from_numpy(fv).to(dev), torch.from_numpy(ft).to(dev)
    s = sr_ref.forward(tfv, tft, IS, variant='sr_ref_nofma', **kw)
    gf, gt = sr_ref.backward(s, g, IS, variant='sr_ref_nofma', **kw)
    return s['soft_colors'], s['aggrs_info'], gf, gt


def reference_oracle(oracle, fv, ft, IS, kw, g):
    ref = oracle.forward(fv, ft, IS, **kw)
    gf, gt = oracle.backward(ref, g.cpu().numpy(), IS, **kw)
    return ref['soft_colors'], ref['aggrs_info'], gf, gt


# ---- (a) more than 64 groups, every launch-size class, both references ---------------------------------------------------

@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref/sr_ref_nofma.so not in this snapshot (oracle/build_ref.py)')
@pytest.mark.parametrize('hard', [False, True], ids=['lasr_modes', 'hard_modes'])
@pytest.mark.parametrize('count', [1, 4, 16, 64])
@pytest.mark.parametrize('nu', [16, 19])
def test_more_than_64_face_groups_against_the_reference_build(cuda, nu, count, hard):
    fv, ft, near, far = synth.raster_batch(nu, 26, count=count)
    assert fv.shape[1] == 20 * nu * nu > 4096
    kw = dict(synth.LASR_MODES, near=near, far=far)
    if hard:
        kw.update(HARD)
    IS = 256
    g = torch.from_numpy(synth.upstream_grad(count, IS)).to(cuda)
    ref = reference_live(cuda, fv, ft, IS, kw, g)
    img, aggr = hip_forward(cuda, fv, ft, IS, kw)
    img2, gf, gt = hip_fwd_bwd(cuda, fv, ft, IS, kw, g)
    assert torch.equal(img, img2)                          # lasr_sr_forward (pre-filled background) == lasr_sr_forward_bg
    check_against(img, aggr, gf, gt, *ref, hard)
    if hard:
        assert int((aggr[:, 1] >= 0).sum()) > 1000 * count
        assert float(aggr[:, 1].max()) > 4096              # faces beyond the first 64 groups are visible


@pytest.mark.parametrize('hard', [False, True], ids=['lasr_modes', 'hard_modes'])
@pytest.mark.parametrize('nu', [16, 19])
def test_more_than_64_face_groups_against_the_c_oracle(oracle, cuda, nu, hard):
    fv, ft, near, far = synth.raster_batch(nu, 26, count=1, first=3)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    if hard:
        kw.update(HARD)
    IS = 256
    g = torch.from_numpy(synth.upstream_grad(1, IS)).to(cuda)
    ref = reference_oracle(oracle, fv, ft, IS, kw, g)
    img, aggr = hip_forward(cuda, fv, ft, IS, kw)
    _, gf, gt = hip_fwd_bwd(cuda, fv, ft, IS, kw, g)
    check_against(img, aggr, gf, gt, *ref, hard)
    assert float((img - torch.from_numpy(ref[0]).to(cuda)).abs().max()) <= 1e-6


@pytest.mark.parametrize('count', [1, 5])
@pytest.mark.parametrize('nu', [16, 19])
def test_every_forward_kernel_gives_the_same_bits_with_more_than_64_groups(thresholds, cuda, nu, count):
    fv, ft, near, far = synth.raster_batch(nu, 26, count=count, first=7)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    tfv, tft = torch.from_numpy(fv).to(cuda), torch.from_numpy(ft).to(cuda)
    out = {}
    for name in VARIANTS:
        thresholds(name)
        out[name] = srf.soft_rasterize(tfv, tft, 256, **kw).cpu().numpy()
    first = out['default thresholds']
    for name, img in out.items():
        assert np.array_equal(img.view(np.uint32), first.view(np.uint32)), \
            '%s differs from the default choice: max %.3e' % (name, np.abs(img - first).max())


# ---- (b) more than 65536 faces: u16 ids are rebased inside a tile's scan ---------------------------------------------------

def tile_spans(fv, IS, margin):
    """Per 8x8 tile: (min, max) face id among the faces whose padded bounding box meets the tile (host estimate)."""
    x, y = fv[0, :, :, 0], fv[0, :, :, 1]
    to_px = lambda v: (v * IS + IS - 1) * 0.5
    x0 = np.floor(to_px(x.min(1) - margin) / 8).clip(0, IS // 8 - 1).astype(int)
    x1 = np.floor(to_px(x.max(1) + margin) / 8).clip(0, IS // 8 - 1).astype(int)
    r0 = np.floor((IS - 1 - to_px(y.max(1) + margin)) / 8).clip(0, IS // 8 - 1).astype(int)
    r1 = np.floor((IS - 1 - to_px(y.min(1) - margin)) / 8).clip(0, IS // 8 - 1).astype(int)
    lo = np.full((IS // 8, IS // 8), 1 << 30)
    hi = np.full((IS // 8, IS // 8), -1)
    for f in range(fv.shape[1]):
        lo[r0[f]:r1[f] + 1, x0[f]:x1[f] + 1] = np.minimum(lo[r0[f]:r1[f] + 1, x0[f]:x1[f] + 1], f)
        hi[r0[f]:r1[f] + 1, x0[f]:x1[f] + 1] = f
    return lo, hi


@pytest.mark.parametrize('hard', [False, True], ids=['lasr_modes', 'hard_modes'])
def test_more_than_65536_faces_rebase_the_list_ids(thresholds, oracle, cuda, hard):
    nu, IS = 64, 64
    fv, ft, near, far = synth.raster_batch(nu, 26, count=1, first=2)
    assert fv.shape[1] == 81920
    kw = dict(synth.LASR_MODES, near=near, far=far)
    if hard:
        kw.update(HARD)
    lo, hi = tile_spans(fv, IS, np.sqrt(kw['sigma_val'] * np.log(1. / kw['dist_eps'] - 1.)))
    assert int(((hi - lo) > 65536 + 64).sum()) >= 4, 'no tile spans more than 65536 face ids: the case would prove nothing'
    g = torch.from_numpy(synth.upstream_grad(1, IS)).to(cuda)
    ref = reference_oracle(oracle, fv, ft, IS, kw, g)
    for name in (['default thresholds'] if hard else list(VARIANTS)):
        thresholds(name)
        img, aggr = hip_forward(cuda, fv, ft, IS, kw)
        _, gf, gt = hip_fwd_bwd(cuda, fv, ft, IS, kw, g)
        check_against(img, aggr, gf, gt, *ref, hard)
        assert float((img - torch.from_numpy(ref[0]).to(cuda)).abs().max()) <= 1e-6, name
    if HAVE_REF:
        check_against(img, aggr, gf, gt, *reference_live(cuda, fv, ft, IS, kw, g), hard)


# ---- (c) a face numbering that is not patch-coherent ---------------------------------------------------------------------------

@pytest.mark.parametrize('hard', [False, True], ids=['lasr_modes', 'hard_modes'])
def test_shuffled_face_order(thresholds, oracle, cuda, hard):
    fv, ft, near, far = synth.raster_batch(16, 26, count=2, first=11)
    perm = np.random.default_rng(16).permutation(fv.shape[1])
    fv, ft = np.ascontiguousarray(fv[:, perm]), np.ascontiguousarray(ft[:, perm])
    kw = dict(synth.LASR_MODES, near=near, far=far)
    if hard:
        kw.update(HARD)
    IS = 256
    g = torch.from_numpy(synth.upstream_grad(2, IS)).to(cuda)
    ref = reference_live(cuda, fv, ft, IS, kw, g) if HAVE_REF else reference_oracle(oracle, fv, ft, IS, kw, g)
    bits = None
    for name in (['default thresholds'] if hard else list(VARIANTS)):
        thresholds(name)
        img, aggr = hip_forward(cuda, fv, ft, IS, kw)
        _, gf, gt = hip_fwd_bwd(cuda, fv, ft, IS, kw, g)
        check_against(img, aggr, gf, gt, *ref, hard)
        if bits is not None:
            assert torch.equal(img, bits), name
        bits = img


# ---- (d) the reference's demo asset ------------------------------------------------------------------------------------------

@pytest.fixture(scope='module')
def spot():
    if not os.path.exists(SPOT):
        pytest.fail('tests/golden/spot_reference_kernels.npz is missing (oracle/gen_ref_vectors_large.py)')
    with np.load(SPOT) as z:
        return {k: z[k] for k in z.files}


def spot_kwargs(spot):
    near, far = float(spot['near_far'][0]), float(spot['near_far'][1])
    soft = dict(synth.LASR_MODES, near=near, far=far)
    # scripts/render_syn.py:135-137: hard colours on a soft silhouette with sigma 1e-12, surface textures
    hard = dict(background_color=(0.2, 0.3, 0.4), near=near, far=far, fill_back=True, eps=1e-3, sigma_val=1e-12,
                dist_func='hard', dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='hard', aggr_func_alpha='hard',
                texture_type='surface')
    return soft, hard


def test_spot_asset_soft_modes_against_the_stored_reference_outputs(thresholds, spot, cuda):
    fv = spot['face_vertices']                            # [3, 5856, 3, 3]: the three poses of render_syn.py's default sweep
    assert fv.shape[1] == 5856
    k = int(spot['stored_frame'])
    col = np.ascontiguousarray(spot['vertex_colours'][None])
    soft, _ = spot_kwargs(spot)
    g = torch.from_numpy(synth.upstream_grad(1, 256, seed=3)).to(cuda)
    for name in VARIANTS:
        thresholds(name)
        img, aggr = hip_forward(cuda, np.ascontiguousarray(fv[k:k + 1]), col, 256, soft)
        _, gf, gt = hip_fwd_bwd(cuda, np.ascontiguousarray(fv[k:k + 1]), col, 256, soft, g)
        check_against(img, aggr, gf, gt, spot['soft/soft_colors'], None, spot['soft/grad_faces'], spot['soft/grad_textures'],
                      False)
        assert float((img - torch.from_numpy(spot['soft/soft_colors']).to(cuda)).abs().max()) <= 1e-6, name


def test_spot_asset_data_generation_modes_against_the_stored_reference_outputs(spot, cuda):
    fv = spot['face_vertices']
    k = int(spot['stored_frame'])
    tex = np.ascontiguousarray(spot['surface_textures_f16'].astype(np.float32)[None])      # [1, 5856, 25, 3]
    _, hard = spot_kwargs(spot)
    img, aggr = hip_forward(cuda, np.ascontiguousarray(fv[k:k + 1]), tex, 256, hard)
    assert torch.equal(img, torch.from_numpy(spot['hard/soft_colors']).to(cuda))
    assert torch.equal(aggr, torch.from_numpy(spot['hard/aggrs_info']).to(cuda)), 'face-index map / z-buffer differs'
    assert len(np.unique(spot['hard/aggrs_info'][:, 1])) > 1500


@pytest.mark.skipif(not HAVE_REF, reason='oracle/_ref/sr_ref_nofma.so not in this snapshot (oracle/build_ref.py)')
def test_spot_asset_all_poses_live_against_the_reference_build(spot, cuda):
    fv = np.ascontiguousarray(spot['face_vertices'])
    N = fv.shape[0]
    soft, hard = spot_kwargs(spot)
    col = np.ascontiguousarray(np.broadcast_to(spot['vertex_colours'][None], (N,) + spot['vertex_colours'].shape))
    g = torch.from_numpy(synth.upstream_grad(N, 256, seed=3)).to(cuda)
    ref = reference_live(cuda, fv, col, 256, soft, g)
    img, aggr = hip_forward(cuda, fv, col, 256, soft)
    _, gf, gt = hip_fwd_bwd(cuda, fv, col, 256, soft, g)
    check_against(img, aggr, gf, gt, *ref, False)
    tex = spot['surface_textures_f16'].astype(np.float32)
    tex = np.ascontiguousarray(np.broadcast_to(tex[None], (N,) + tex.shape))
    ref = reference_live(cuda, fv, tex, 256, hard, g)
    img, aggr = hip_forward(cuda, fv, tex, 256, hard)
    _, gf, gt = hip_fwd_bwd(cuda, fv, tex, 256, hard, g)
    check_against(img, aggr, gf, gt, *ref, True)
