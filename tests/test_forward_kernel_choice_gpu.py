"""The forward pass of LASR's mode combination picks one of four kernels by launch size (lasr_amd/csrc/sr_raster.hip
forward_impl): eight or four waves sharing an 8x8 tile for launches that cannot fill the chip (sr_forward_coop.h), one wave per
8x8 tile, and -- from 3 frames of 256x256 up -- the pair-walk kernel (sr_forward_pairs.h; two teams of four waves per tile up to 12 frames), whose lanes walk the (pixel, face)
pairs of their own pixel.  The first three evaluate every pair with the same instruction sequence and visit the faces of a pixel
in index order, so their outputs must be IDENTICAL bit for bit; the pair-walk kernel evaluates every pair with that same
sequence but folds a pixel's fragments in another order (inside fragments first, a stolen run merged at the end of a chunk):
alpha product and depth softmax are symmetric in the fragments, so its image agrees to rounding (<= 1e-6 asked, ~5e-7 measured),
and the running depth maximum exactly.  This file forces each kernel in turn (lasr_sr_options through the operator's
set_launch_thresholds, include/lasr_sr.h) on the same inputs -- ragged image sizes, 3 / 6 / 9 channels, a tile whose list needs
more than one round, device-resident near/far, an empty mesh, degenerate faces -- and compares; the kernels are also held against
the oracle.  The device-side choice is checked against both of its outcomes."""
import numpy as np
import pytest
import torch

from lasr_amd import _lib, synth
from lasr_amd.soft_renderer import functional as srf

pytestmark = pytest.mark.gpu

BIG = 10 ** 12
#            coop8_max  coop_max  choose_max   (8x8-pixel tiles)   order_max  pair_min_tiles
VARIANTS = {'eight waves per 8x8 tile': (BIG, BIG, BIG, -1, BIG),
            'four waves per 8x8 tile': (0, BIG, BIG, -1, BIG),
            'one wave per 8x8 tile': (0, 0, 0, -1, BIG)}
PAIR_WALK = (0, 0, 0, -1, 0)              # round 6: every launch through the pair-walk kernel
PAIR_TOL = 1e-6
DEFAULTS = (2200, 14336, 49152)


@pytest.fixture
def thresholds():
    yield lambda name: srf.set_launch_thresholds(*VARIANTS[name])
    srf.set_launch_thresholds()


def render(dev, fv, ft, IS, kw):
    tfv, tft = torch.from_numpy(fv).to(dev), torch.from_numpy(ft).to(dev)
    img = srf.soft_rasterize(tfv, tft, IS, **kw)
    torch.cuda.synchronize()
    return img.cpu().numpy()


def all_variants(thresholds, dev, fv, ft, IS, kw):
    out = {}
    for name in VARIANTS:
        thresholds(name)
        out[name] = render(dev, fv, ft, IS, kw)
    first = next(iter(out))
    for name, img in out.items():
        assert img.shape == out[first].shape
        assert np.array_equal(img.view(np.uint32), out[first].view(np.uint32)), \
            '%s differs from %s: max %.3e' % (name, first, np.abs(img - out[first]).max())
    srf.set_launch_thresholds(*PAIR_WALK)
    pw = render(dev, fv, ft, IS, kw)
    assert pw.shape == out[first].shape and np.isfinite(pw).all()
    assert np.abs(pw - out[first]).max() <= PAIR_TOL, 'pair walk: max %.3e' % np.abs(pw - out[first]).max()
    return out[first]


@pytest.mark.parametrize('IS', [1, 7, 8, 20, 33, 64, 100])
def test_every_kernel_gives_the_same_bits_on_ragged_image_sizes(thresholds, oracle, cuda, IS):
    fv, ft, near, far = synth.raster_batch(4, 3, count=3)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    img = all_variants(thresholds, cuda, fv, ft, IS, kw)
    ref = oracle.forward(fv, ft, IS, **kw)
    assert np.abs(img - ref['soft_colors']).max() <= 1e-6
    thresholds('one wave per 8x8 tile')
    srf.set_launch_thresholds(*PAIR_WALK)
    assert np.abs(render(cuda, fv, ft, IS, kw) - ref['soft_colors']).max() <= 1e-6      # the pair walk against the oracle itself


@pytest.mark.parametrize('channels', [6, 9])
def test_every_kernel_gives_the_same_bits_with_six_and_nine_channels(thresholds, cuda, channels):
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    rng = np.random.default_rng(channels)
    tex = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(channels // 3 - 1)], -1)
    kw = dict(synth.LASR_MODES, near=near, far=far, background_color=[0.25 * k for k in range(channels)])
    img = all_variants(thresholds, cuda, fv, tex, 72, kw)
    assert img.shape == (2, channels + 1, 72, 72) and np.isfinite(img).all()
    # the first triple is the three-channel render of the same geometry
    thresholds('four waves per 8x8 tile')
    img3 = render(cuda, fv, ft, 72, dict(kw, background_color=[0., 0.25, 0.5]))
    assert np.array_equal(img[:, :3], img3[:, :3]) and np.array_equal(img[:, channels], img3[:, 3])


def test_every_kernel_gives_the_same_bits_when_a_tile_needs_more_than_one_list_round(thresholds, oracle, cuda):
    # 2500 faces in the centre of the image: the tiles there meet more faces than one LDS list holds (1024 / 2048 entries)
    rng = np.random.default_rng(7)
    F = 2500
    c = rng.uniform(-0.15, 0.15, (2, F, 1, 2))
    tri = c + rng.uniform(-0.08, 0.08, (2, F, 3, 2))
    z = rng.uniform(2, 4, (2, F, 3, 1))
    fv = np.concatenate([tri, z], -1).astype(np.float32)
    fv[0, 0] = [[-1.5, -1.2, 3], [1.4, -1.1, 3.5], [0.1, 1.6, 2.5]]
    ft = rng.uniform(0, 1, fv.shape).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=1.0, far=5.0)
    img = all_variants(thresholds, cuda, fv, ft, 64, kw)
    ref = oracle.forward(fv, ft, 64, **kw)
    assert np.abs(img - ref['soft_colors']).max() <= 1e-5


def test_every_kernel_takes_device_resident_near_far_and_an_empty_mesh(thresholds, cuda):
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    nf = torch.tensor([near, far], dtype=torch.float32, device=cuda)
    kw = dict(synth.LASR_MODES, near=nf[0], far=nf[1])
    a = all_variants(thresholds, cuda, fv, ft, 48, kw)
    b = all_variants(thresholds, cuda, fv, ft, 48, dict(kw, near=near, far=far))
    assert np.array_equal(a, b)
    empty = np.zeros((2, 0, 3, 3), np.float32)
    img = all_variants(thresholds, cuda, empty, empty.copy(), 24, dict(synth.LASR_MODES, near=1.0, far=5.0))
    assert np.array_equal(img[:, :3], np.ones_like(img[:, :3])) and not img[:, 3].any()


def test_every_kernel_at_the_launch_sizes_lasr_uses(thresholds, cuda):
    # 16 meshes of 1280 faces at 256x256 (spot3 stage 0: 2 images x 8 hypotheses) and 4 of 2420 faces
    for nu, count in ((8, 16), (11, 4)):
        fv, ft, near, far = synth.raster_batch(nu, 3, count=count)
        all_variants(thresholds, cuda, fv, ft, 256, dict(synth.LASR_MODES, near=near, far=far))


def test_the_device_side_choice_takes_either_kernel_and_the_bits_do_not_change(thresholds, cuda):
    # 6 frames of 64x64 = 384 tiles; the object's bounding boxes cover about half of them.  coop_max between the estimate and
    # the launch size: the device picks four waves per tile; below the estimate: one wave per tile.
    import importlib
    sr_mod = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    fv, ft, near, far = synth.raster_batch(4, 3, count=6)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    h = _lib.lib()
    thresholds('one wave per 8x8 tile')
    want = render(cuda, fv, ft, 64, kw)
    import ctypes
    seen = {}
    for coop_max in (300, 40):
        srf.set_launch_thresholds(0, coop_max, BIG, 0, BIG)      # fixed tile order: sr_choose_kernel's bounding-box estimate decides
        got = render(cuda, fv, ft, 64, kw)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ws = sr_mod._workspaces[(cuda.index, torch.cuda.current_stream(cuda).cuda_stream)]
        c = ctypes.c_int(-1)
        _lib.check(h.lasr_sr_peek_choice(ws.data_ptr(), fv.shape[0], fv.shape[1], ctypes.byref(c), None), 'peek')
        seen[coop_max] = c.value
    assert seen == {300: 1, 40: 0}, seen


def test_default_thresholds_pick_by_launch_size(cuda):
    # the defaults are in force outside this file's fixture: launches below the pair-walk threshold (3 frames of 256x256) keep the
    # reference order -- a frame rendered alone or in a batch of two gives the same bits; from there the pair walk takes over
    # whatever the face size (1280 faces = 51 pixels per face, 2420 faces = 27), with two teams of four waves per tile up to 12
    # frames and one beyond: same image to rounding
    for nu in (8, 11):
        fv, ft, near, far = synth.raster_batch(nu, 3, count=2)
        kw = dict(synth.LASR_MODES, near=near, far=far)
        batch = render(cuda, fv, ft, 256, kw)
        one = render(cuda, fv[1:2], ft[1:2], 256, kw)
        assert np.array_equal(batch[1:2], one)
    for nu, cnt in ((8, 16), (11, 16), (8, 6), (11, 6)):
        fv, ft, near, far = synth.raster_batch(nu, 3, count=cnt)
        kw = dict(synth.LASR_MODES, near=near, far=far)
        batch = render(cuda, fv, ft, 256, kw)
        one = render(cuda, fv[5:6], ft[5:6], 256, kw)
        assert np.abs(batch[5:6] - one).max() <= PAIR_TOL and not np.array_equal(batch[5:6], one)


def test_one_and_two_teams_per_tile_agree(cuda):
    # lasr_amd/csrc/sr_forward_pairs.h: SPLIT teams of four waves share a 16x16 tile, team t walks the chunks t, t + SPLIT, ... of the
    # tile's list and the teams' partial states meet at the end -- forced either way by the per-call flags (three and nine channels)
    from lasr_amd import _lib
    fv, ft, near, far = synth.raster_batch(11, 3, count=5)
    rng = np.random.default_rng(5)
    tex9 = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(2)], -1)
    try:
        srf.set_launch_thresholds(*PAIR_WALK)
        for tex, kw in ((ft, dict(synth.LASR_MODES, near=near, far=far)),
                        (tex9, dict(synth.LASR_MODES, near=near, far=far, background_color=[0.1 * k for k in range(9)]))):
            srf.set_forward_flags(_lib.SR_PAIR_ONE_TEAM)
            a = render(cuda, fv, tex, 160, kw)
            srf.set_forward_flags(_lib.SR_PAIR_TWO_TEAMS)
            b = render(cuda, fv, tex, 160, kw)
            assert np.abs(a - b).max() <= PAIR_TOL and not np.array_equal(a, b)
    finally:
        srf.set_forward_flags(_lib.SR_DEFAULT_FLAGS)
        srf.set_launch_thresholds()


def test_the_pair_walk_handles_faces_that_are_not_tame(thresholds, oracle, cuda):
    # a sliver, a zero-area face, a face in front of the near plane, huge coordinates: records without the tame flag go through
    # the generic arithmetic at the end of each chunk (sr_forward_pairs.h: the slow mask)
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    fv = fv.copy()
    fv[0, 0] = [[0, 0, 3], [0.5, 0.5, 3], [1e-7, 0, 3]]
    fv[0, 1] = [[0.1, 0.1, 3], [0.1, 0.1, 3], [0.1, 0.1, 3]]
    fv[1, 2, :, 2] = 1e-9
    fv[1, 3] = [[-3e4, -2e4, 3], [4e4, -1e4, 3], [0, 5e4, 3]]
    kw = dict(synth.LASR_MODES, near=near, far=far)
    img = all_variants(thresholds, cuda, fv, ft, 64, kw)
    ref = oracle.forward(fv, ft, 64, **kw)
    assert np.abs(img - ref['soft_colors']).max() <= 1e-5


# ---- this launch's own tile order (sr_order_kernel): five frames and more issue their 8x8 tiles heaviest first

def _order_table(cuda, N, F, IS):
    """The block -> tile table the last forward call left in the operator's workspace, and the setup kernel's pixel rects."""
    import importlib
    sr_mod = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    ws = sr_mod._workspaces[(cuda.index, torch.cuda.current_stream(cuda).cuda_stream)]
    up = lambda v: (v + 255) // 256 * 256
    base = (-ws.data_ptr()) % 256
    o_rects = base + up(N * F * 48 * 4)
    o_grects = o_rects + up(N * F * 8)
    o_order = o_grects + up(N * ((F + 63) // 64) * 8) + 256
    t8 = (IS + 7) // 8
    rects = ws[o_rects:o_rects + N * F * 8].view(torch.int16).reshape(N, F, 4).cpu().numpy().astype(np.int64)
    table = ws[o_order:o_order + N * t8 * t8 * 4].view(torch.int32).cpu().numpy()
    return table, rects, t8


@pytest.mark.parametrize('count,IS,channels', [(8, 64, 3), (16, 100, 3), (8, 256, 3), (16, 72, 9), (24, 48, 6), (12, 64, 3), (7, 64, 9)])
def test_heaviest_first_tile_order_gives_the_same_bits_in_every_kernel(cuda, count, IS, channels):
    fv, ft, near, far = synth.raster_batch(4 if IS < 256 else 8, 5, count=count)
    rng = np.random.default_rng(count)
    if channels > 3:
        ft = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(channels // 3 - 1)], -1)
    kw = dict(synth.LASR_MODES, near=near, far=far, background_color=[0.125 * k for k in range(channels)])
    try:
        out = {}
        for name, th in VARIANTS.items():
            for order_max in (0, BIG):
                srf.set_launch_thresholds(*th[:3], order_max, th[4])
                out[name, order_max] = render(cuda, fv, ft, IS, kw)
        first = out[next(iter(out))]
        for key, img in out.items():
            assert np.array_equal(img.view(np.uint32), first.view(np.uint32)), key
    finally:
        srf.set_launch_thresholds()


@pytest.mark.parametrize('N,IS,nu', [(16, 104, 4), (12, 96, 4), (5, 64, 4), (64, 64, 19)])
def test_the_tile_order_is_a_permutation_sorted_by_the_faces_that_touch_each_tile(cuda, N, IS, nu):
    # 104: 13 x 13 tiles per frame, the last column / row cut by the image edge; 12 and 5 frames: an image split between two XCDs;
    # 64 frames of 7220 faces: an XCD's 8 images hold 11 MB of records, sorted and issued in two groups of 4 images
    fv, ft, near, far = synth.raster_batch(nu, 7, count=N)
    fv[3] += np.array([0.45, -0.3, 0.], np.float32)            # one object off-centre: the fixed spiral would start in its empty middle
    F = fv.shape[1]
    try:
        srf.set_launch_thresholds(-1, -1, -1, 0, BIG)           # (the 8x8-tile kernels: the pair walk orders 16x16 tiles)
        want = render(cuda, fv, ft, IS, dict(synth.LASR_MODES, near=near, far=far))
        srf.set_launch_thresholds(-1, -1, -1, BIG, BIG)
        got = render(cuda, fv, ft, IS, dict(synth.LASR_MODES, near=near, far=far))
    finally:
        srf.set_launch_thresholds()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    table, rects, t8 = _order_table(cuda, N, F, IS)
    bn, ty, tx = table >> 16, (table >> 8) & 255, table & 255
    assert sorted(zip(bn.tolist(), ty.tolist(), tx.tolist())) == [(n, y, x) for n in range(N) for y in range(t8) for x in range(t8)]
    w = np.zeros((N, t8, t8), np.int64)
    for n in range(N):
        for x0, x1, y0, y1 in rects[n]:
            if x1 >= x0 and y1 >= y0:
                w[n, y0 >> 3:(y1 >> 3) + 1, x0 >> 3:(x1 >> 3) + 1] += 1
    assert w.max() > 20
    groups = 2 if nu == 19 else 1                               # per XCD
    per = N * t8 * t8 // (8 * groups)
    entry = (bn * t8 + ty) * t8 + tx                            # position in the image-major (image, row, column) list
    for k in range(8 * groups):                                 # block b runs on XCD b % 8 and takes entry b // 8 of its list:
        mine = slice(k * per, (k + 1) * per)                    # the XCD's share of that list (an image possibly split with a neighbour),
        assert ((entry[mine] >= k * per) & (entry[mine] < (k + 1) * per)).all()                         # group after group
        key = np.minimum(w[bn[mine], ty[mine], tx[mine]], 255)
        assert (np.diff(key) <= 0).all(), 'slice %d: tiles not in descending weight' % k


def test_launches_the_tile_order_does_not_cover_fall_back_to_the_fixed_order(cuda):
    # 4 frames (below the 5 the order pays from), 12 frames of 5 x 5 tiles (300 tiles do not divide over the 8 XCDs), and
    # order_max_tiles below the launch size: no table, same bits as with it switched off
    for count, IS, order_max in ((4, 64, BIG), (12, 40, BIG), (16, 64, 100)):
        fv, ft, near, far = synth.raster_batch(4, 5, count=count)
        kw = dict(synth.LASR_MODES, near=near, far=far)
        try:
            srf.set_launch_thresholds(-1, -1, -1, order_max)
            a = render(cuda, fv, ft, IS, kw)
            srf.set_launch_thresholds(-1, -1, -1, 0)
            b = render(cuda, fv, ft, IS, kw)
        finally:
            srf.set_launch_thresholds()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_ordered_launches_choose_on_the_count_of_non_empty_tiles(cuda):
    # 8 frames of 64x64 = 512 tiles.  coop_max x 4/7 below the launch size and choose_max x 7/16 above it: both kernels are launched
    # and decide on the device from sr_order_kernel's count of tiles that meet at least one face (<= coop_max x 3/8: four waves).
    import ctypes
    import importlib
    sr_mod = importlib.import_module('lasr_amd.soft_renderer.functional.soft_rasterize')
    N, IS = 8, 64
    fv, ft, near, far = synth.raster_batch(4, 5, count=N)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    h = _lib.lib()
    try:
        srf.set_launch_thresholds(0, 0, 0, 0, BIG)
        want = render(cuda, fv, ft, IS, kw)
        seen = {}
        for coop_max in (880, 160):                                     # x 3/8 = 330 and 60 non-empty tiles
            srf.set_launch_thresholds(0, coop_max, BIG, BIG, BIG)    # pair walk off: the device-side choice of round 4
            got = render(cuda, fv, ft, IS, kw)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            ws = sr_mod._workspaces[(cuda.index, torch.cuda.current_stream(cuda).cuda_stream)]
            c = ctypes.c_int(-1)
            _lib.check(h.lasr_sr_peek_choice(ws.data_ptr(), N, fv.shape[1], ctypes.byref(c), None), 'peek')
            seen[coop_max] = c.value
    finally:
        srf.set_launch_thresholds()
    table, rects, t8 = _order_table(cuda, N, fv.shape[1], IS)
    w = np.zeros((N, t8, t8), np.int64)
    for n in range(N):
        for x0, x1, y0, y1 in rects[n]:
            if x1 >= x0 and y1 >= y0:
                w[n, y0 >> 3:(y1 >> 3) + 1, x0 >> 3:(x1 >> 3) + 1] += 1
    busy = int((w > 0).sum())
    assert 60 < busy <= 330, busy                                       # so the two settings took different kernels
    assert seen == {880: busy, 160: busy}, (seen, busy)


@pytest.mark.parametrize('count,IS,channels', [(8, 64, 3), (16, 96, 3), (16, 256, 3), (8, 128, 9), (7, 64, 6)])
def test_the_pair_walk_gives_the_same_bits_in_any_tile_order_and_run_after_run(cuda, count, IS, channels):
    # which workgroup renders which 16x16 tile changes no tile's arithmetic, and nothing in the kernel depends on timing (the
    # lanes' pairing is a function of the pair counts): fixed order, the launch's own order, and a second run are identical
    fv, ft, near, far = synth.raster_batch(4 if IS < 256 else 11, 5, count=count)
    rng = np.random.default_rng(count)
    if channels > 3:
        ft = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(channels // 3 - 1)], -1)
    kw = dict(synth.LASR_MODES, near=near, far=far, background_color=[0.125 * k for k in range(channels)])
    try:
        srf.set_launch_thresholds(0, 0, 0, 0, 0)
        a = render(cuda, fv, ft, IS, kw)
        srf.set_launch_thresholds(0, 0, 0, BIG, 0)
        b = render(cuda, fv, ft, IS, kw)
        c = render(cuda, fv, ft, IS, kw)
    finally:
        srf.set_launch_thresholds()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(b.view(np.uint32), c.view(np.uint32))


@pytest.mark.parametrize('seed,nu,IS', [(0, 4, 64), (1, 8, 128), (2, 11, 256), (3, 16, 128)])
def test_one_projection_agrees_with_three_on_slivers_and_edge_on_faces(cuda, seed, nu, IS):
    # the pair walk evaluates ONE edge projection per pixel where the nearest edge line is clear (sr_device.h: near_tie -- a relative
    # and an absolute margin scaled by the face's smallest height), the one-wave kernel the reference's three for every inside pixel:
    # on objects squashed to a fraction of their width (edge-on faces at the silhouette, slivers inside) the images still agree to
    # rounding.  With the relative margin alone LASR's own meshes differed by up to 5e-5 (tools/prof/tie_stress.py runs more cases).
    rng = np.random.default_rng(seed)
    fv, ft, near, far = synth.raster_batch(nu, 3, count=4)
    fv = fv.copy()
    for n in range(fv.shape[0]):
        a = rng.uniform(0, np.pi)
        d = np.array([np.cos(a), np.sin(a)], np.float32)
        k = rng.uniform(0.05, 0.6)
        xy = fv[n, :, :, :2]
        c = xy.reshape(-1, 2).mean(0)
        rel = xy - c
        fv[n, :, :, :2] = c + rel - (1 - k) * (rel @ d)[..., None] * d
    kw = dict(synth.LASR_MODES, near=near, far=far)
    try:
        srf.set_launch_thresholds(*VARIANTS['one wave per 8x8 tile'])
        want = render(cuda, fv, ft, IS, kw)
        srf.set_launch_thresholds(*PAIR_WALK)
        got = render(cuda, fv, ft, IS, kw)
    finally:
        srf.set_launch_thresholds()
    assert np.abs(got - want).max() <= PAIR_TOL, np.abs(got - want).max()
