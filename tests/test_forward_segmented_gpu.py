"""LASR_SR_SEGMENTED (include/lasr_sr.h, opt-in): for launches small enough for the several-waves-per-tile kernels, a tile's face
list is split into index-ordered segments that 4 or 8 waves fold in parallel; the partial alpha-product / depth-softmax states are
merged at the end.  In exact arithmetic nothing changes (both aggregates are symmetric in the fragments, nnutils/mesh_net.py:318-363
renders N = 2 / 4 / 16 meshes per call); the rounding sequence does.  Bars: image within 1e-6 of the default (reference-order) path
and within 1e-4 of the oracle, gradients of a backward pass that reads the segmented forward's aggregates within 1e-3 of the oracle's
largest entry; launches outside the small-launch range and other mode combinations ignore the flag bit for bit."""
import numpy as np
import pytest
import torch

from lasr_amd import _lib, synth
from lasr_amd.soft_renderer import functional as srf

pytestmark = pytest.mark.gpu
BIG = 10 ** 12


@pytest.fixture
def flags():
    yield srf.set_forward_flags
    srf.set_forward_flags(_lib.SR_DEFAULT_FLAGS)
    srf.set_launch_thresholds()


def render(dev, fv, ft, IS, kw, g=None):
    a = torch.from_numpy(fv).to(dev).requires_grad_(g is not None)
    b = torch.from_numpy(ft).to(dev).requires_grad_(g is not None)
    img = srf.soft_rasterize(a, b, IS, **kw)
    if g is None:
        return img.detach().cpu().numpy()
    img.backward(torch.from_numpy(g).to(dev))
    return img.detach().cpu().numpy(), a.grad.cpu().numpy(), b.grad.cpu().numpy()


@pytest.mark.parametrize('waves', [8, 4])
@pytest.mark.parametrize('nu,count,IS', [(11, 1, 256), (11, 4, 256), (11, 16, 256), (8, 3, 100), (19, 2, 256)])
def test_segmented_forward_stays_within_1e6_of_the_default_order(flags, oracle, cuda, nu, count, IS, waves):
    fv, ft, near, far = synth.raster_batch(nu, 26, count=count)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    tiles = count * ((IS + 7) // 8) ** 2
    # the flag applies up to twice the eight-wave bound: at or below the bound eight waves fold a tile, above it four
    srf.set_launch_thresholds(BIG if waves == 8 else tiles - 1, BIG, BIG)
    g = synth.upstream_grad(count, IS)
    flags(0)
    want = render(cuda, fv, ft, IS, kw)
    flags(_lib.SR_SEGMENTED)
    got, gf, gt = render(cuda, fv, ft, IS, kw, g)
    assert np.abs(got - want).max() <= 1e-6
    assert not np.array_equal(got.view(np.uint32), want.view(np.uint32)) or count * nu < 20    # it IS another rounding sequence
    if count <= 4:
        ref = oracle.forward(fv, ft, IS, **kw)
        assert np.abs(got - ref['soft_colors']).max() <= 1e-4
        rgf, rgt = oracle.backward(ref, g, IS, **kw)
        assert np.abs(gf.reshape(rgf.shape) - rgf).max() <= 1e-3 * np.abs(rgf).max()
        assert np.abs(gt.reshape(rgt.shape) - rgt).max() <= 1e-3 * np.abs(rgt).max()


@pytest.mark.parametrize('channels', [6, 9])
def test_segmented_forward_with_six_and_nine_channels(flags, cuda, channels):
    fv, ft, near, far = synth.raster_batch(8, 3, count=2)
    rng = np.random.default_rng(channels)
    tex = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(channels // 3 - 1)], -1)
    kw = dict(synth.LASR_MODES, near=near, far=far, background_color=[0.25 * k for k in range(channels)])
    flags(0)
    want = render(cuda, fv, tex, 128, kw)
    flags(_lib.SR_SEGMENTED)
    got = render(cuda, fv, tex, 128, kw)
    assert got.shape == (2, channels + 1, 128, 128) and np.abs(got - want).max() <= 2e-6      # attributes up to |2|


def test_segmented_forward_when_a_tile_needs_more_than_one_list_round(flags, cuda):
    rng = np.random.default_rng(7)
    F = 2500
    c = rng.uniform(-0.15, 0.15, (2, F, 1, 2))
    tri = c + rng.uniform(-0.06, 0.06, (2, F, 3, 2))
    fv = np.concatenate([tri, rng.uniform(2, 4, (2, F, 3, 1))], -1).astype(np.float32)
    ft = rng.uniform(0, 1, fv.shape).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=1.0, far=5.0)
    flags(0)
    want = render(cuda, fv, ft, 64, kw)
    flags(_lib.SR_SEGMENTED)
    got = render(cuda, fv, ft, 64, kw)
    assert np.abs(got - want).max() <= 2e-6


def test_the_flag_is_ignored_where_it_does_not_apply(flags, cuda):
    fv, ft, near, far = synth.raster_batch(8, 3, count=3)
    # other mode combinations: bit for bit the default
    hard = dict(synth.LASR_MODES, near=near, far=far, aggr_func_rgb='hard', dist_func='hard', aggr_func_alpha='hard')
    flags(0)
    want = render(cuda, fv, ft, 96, hard)
    flags(_lib.SR_SEGMENTED)
    assert np.array_equal(render(cuda, fv, ft, 96, hard).view(np.uint32), want.view(np.uint32))
    # launches beyond twice the eight-wave range keep the reference-order kernels whatever the flag says
    kw = dict(synth.LASR_MODES, near=near, far=far)
    srf.set_launch_thresholds(100, BIG, BIG)                                  # 3 x 144 tiles > 200
    flags(0)
    want = render(cuda, fv, ft, 96, kw)
    flags(_lib.SR_SEGMENTED)
    assert np.array_equal(render(cuda, fv, ft, 96, kw).view(np.uint32), want.view(np.uint32))
