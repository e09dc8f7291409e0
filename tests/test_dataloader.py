"""Video loader on a tiny sequence written to disk in the reference's layout: PFM round trip,
the OpenCV-equivalent crop / resize helpers on hand-computable cases, crop-box and flow-rebasing semantics of
dataloader/vidbase.py:98-147, epoch padding of dataloader/vid.py:60-80, and the trainer's batch dictionary."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lasr_amd.dataloader import vid
from lasr_amd.ext_utils import image as iu
from lasr_amd.ext_utils import util_flow


def test_pfm_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for shape in ((5, 7), (5, 7, 3), (4, 4, 1)):
        a = rng.standard_normal(shape).astype(np.float32)
        p = str(tmp_path / 'a.pfm')
        util_flow.write_pfm(p, a)
        b, scale = util_flow.readPFM(p)
        assert scale == 1.0 and np.array_equal(b, a.reshape(b.shape))
    raw = open(p, 'rb').read()
    assert raw.startswith(b'Pf\n4 4\n-1.000000\n')              # little-endian marker, bottom-up rows follow
    assert np.array_equal(np.frombuffer(raw[-16:], '<f4'), a[0, :, 0])


def test_resize_and_crop_match_opencv_conventions():
    row = np.array([[0., 1.]])
    assert np.allclose(iu.resize_linear(row, 4, 1), [[0., 0.25, 0.75, 1.]])        # half-pixel centres, clamped border
    assert np.allclose(iu.resize_linear(np.arange(8.)[None], 4, 1), [[0.5, 2.5, 4.5, 6.5]])   # no anti-aliasing
    assert np.array_equal(iu.resize_nearest(np.arange(5)[None], 3, 1), [[0, 1, 3]])         # floor(i * 5/3)
    img = np.arange(12.).reshape(3, 4)
    c = iu.crop_pad(img, -1, 1, 3, border=-7.)
    assert np.array_equal(c, [[-7., 4., 5.], [-7., 8., 9.], [-7., -7., -7.]])
    rgb = iu.crop_pad(np.ones((2, 2, 3)), 1, 1, 2, border=np.array([.1, .2, .3]))
    assert np.allclose(rgb[0, 0], 1) and np.allclose(rgb[1, 1], [.1, .2, .3])


def test_distance_transforms_and_contour():
    m = np.zeros((32, 32)); m[10:20, 12:22] = 1
    assert iu.compute_dt(m, iters=0)[15, 15] == 0 and abs(iu.compute_dt(m, iters=0)[15, 5] - 7 / 32) < 1e-9
    assert iu.compute_dt(m, iters=10)[15, 2] == 0                                  # inside the 10-pixel dilation
    c = iu.sample_contour(m, seed=0)
    assert c.shape == (1000, 2) and np.abs(c).max() <= 1
    # every sample lies within the reference's 2-pixel band of a contour vertex; (x, y) order, [-1, 1] normalisation
    rc = (c[:, ::-1] + 1) / 2 * 32
    cv = iu.contour_vertices(m)
    assert np.abs(rc[:, None] - cv[None]).max(-1).min(1).max() <= 2 + 1e-9
    assert len(np.unique(np.round(rc).astype(int), axis=0)) > 100                 # drawn without replacement from a band of > 1000 points


def _marching_squares_vertices(mask, level=0.0):
    """The vertices find_contours would emit, enumerated edge by edge from the published algorithm (Lorensen & Cline marching squares
    as in skimage/measure/_find_contours_cy.pyx): an edge of a 2x2 cell carries a vertex when exactly one of its end pixels is > level."""
    m = np.asarray(mask, np.float64)
    out = set()
    H, W = m.shape

    def vertex(a, b):
        va, vb = m[a], m[b]
        frac = 0.0 if vb == va else (level - va) / (vb - va)
        return (a[0] + frac * (b[0] - a[0]), a[1] + frac * (b[1] - a[1]))
    for r in range(H - 1):
        for c in range(W - 1):
            ul, ur, ll, lr = (r, c), (r, c + 1), (r + 1, c), (r + 1, c + 1)
            for a, b in ((ul, ur), (ur, lr), (ll, lr), (ul, ll)):
                if (m[a] > level) != (m[b] > level):
                    out.add(vertex(a, b))
    return out


def test_contour_vertices_are_the_marching_squares_vertices_at_level_zero():
    yy, xx = np.mgrid[:40, :40]
    shapes = [((xx - 19.3) ** 2 + (yy - 17.8) ** 2 <= 11 ** 2), np.zeros((40, 40), bool), (xx + yy) % 7 == 0]
    shapes[1][5:30, 8:9] = True; shapes[1][0:6, 20:30] = True                      # a one-pixel line and a block touching the border
    for s in shapes:
        verts = iu.contour_vertices(s.astype(np.float64))
        got = {tuple(v) for v in verts.tolist()}
        assert got == _marching_squares_vertices(s.astype(np.float64))
        # one vertex per crossed cell EDGE (the list find_contours' contours add up to, minus each closed contour's repeated first
        # point): a background pixel appears once per foreground 4-neighbour -- counted here edge by edge
        edges = {}
        H, W = s.shape
        for r in range(H):
            for c in range(W):
                for rr, cc in ((r, c + 1), (r + 1, c)):
                    if rr < H and cc < W and s[r, c] != s[rr, cc]:
                        bg = (r, c) if not s[r, c] else (rr, cc)
                        edges[(float(bg[0]), float(bg[1]))] = edges.get((float(bg[0]), float(bg[1])), 0) + 1
        mult = {}
        for v in verts.tolist():
            mult[tuple(v)] = mult.get(tuple(v), 0) + 1
        assert mult == edges
    assert max(np.unique(iu.contour_vertices(shapes[0].astype(np.float64)), axis=0, return_counts=True)[1]) >= 2    # the disc has concave pixel corners


def write_sequence(root, name='toy', n=4, W=96, H=80, step=(5, -3)):
    """An ellipse translating by `step` pixels per frame, with exact flow."""
    from PIL import Image
    dirs = {k: os.path.join(root, 'database', 'DAVIS', k, 'Full-Resolution', name)
            for k in ('JPEGImages', 'Annotations', 'FlowFW', 'FlowBW', 'Camera')}
    for d in dirs.values():
        os.makedirs(d)
    yy, xx = np.mgrid[:H, :W]
    for i in range(n):
        cx, cy = 30 + step[0] * i, 45 + step[1] * i
        mask = ((xx - cx) / 14.) ** 2 + ((yy - cy) / 9.) ** 2 <= 1
        img = np.zeros((H, W, 3), np.uint8)
        img[mask] = (200, 120, 40)
        img[~mask] = (10, 30, 60)
        Image.fromarray(img).save(os.path.join(dirs['JPEGImages'], '%05d.jpg' % i), quality=100, subsampling=0)
        Image.fromarray((128 * mask).astype(np.uint8)).save(os.path.join(dirs['Annotations'], '%05d.png' % i))
        np.savetxt(os.path.join(dirs['Camera'], '%05d.txt' % i), [10., 0., 0., 1., 0., 0., 0., 10.])
        for kind, sgn, ok in (('FlowFW', 1, i < n - 1), ('FlowBW', -1, i > 0)):
            if ok:
                fl = np.zeros((H, W, 3), np.float32)
                fl[mask] = (sgn * step[0], sgn * step[1], 1)
                util_flow.write_pfm(os.path.join(dirs[kind], 'flo-%05d.pfm' % i), fl)
                util_flow.write_pfm(os.path.join(dirs[kind], 'occ-%05d.pfm' % i), -np.ones((H, W), np.float32))
    os.makedirs(os.path.join(root, 'configs'), exist_ok=True)
    with open(os.path.join(root, 'configs', '%s.config' % name), 'w') as fh:
        fh.write('[data]\ndatapath = database/DAVIS/JPEGImages/Full-Resolution/%s/\ndframe = 1\ninit_frame = 0\n'
                 'end_frame = -1\ncan_frame = 1\n' % name)


def make_opts(**kw):
    d = dict(dataname='toy', sil_path='none', batch_size=2, ngpu=1, local_rank=0, img_size=64, n_data_workers=0)
    d.update(kw)
    return SimpleNamespace(**d)


def test_video_loader_elements(tmp_path):
    write_sequence(str(tmp_path))
    loader, length = vid.data_loader(make_opts(), shuffle=False, root=str(tmp_path))
    ds = loader.dataset
    assert length == 4
    # 3 forward + 3 backward pairs, first and last duplicated (vid.py:75-76), repeated to ~200 iterations x batch
    assert ds.baselist[:8] == [0, 0, 1, 2, 1, 2, 3, 3] and ds.directlist[:8] == [1, 1, 1, 1, 0, 0, 0, 0]
    assert len(ds) == 8 * ((2 * 1 * 200) // 8)
    e = ds[2]                                                     # frames 1 -> 2
    assert (e['id0'], e['id1']) == (1, 2) and e['is_canonical'] and not e['is_canonicaln']
    assert e['img'].shape == (3, 64, 64) and e['mask'].shape == (2, 64, 64) and e['flow'].shape == (3, 64, 64)
    assert e['mask_dts'].shape == (2, 64, 64) and e['mask_contour'].shape == (2, 1000, 2) and e['occ'].shape == (64, 64)
    # crop box: centre of the silhouette, half-size int(1.2 * 14) = 16 -> 32 px crop, upscaled x2
    assert np.allclose(e['pps'], [[35 - 16, 42 - 16], [40 - 16, 39 - 16]]) and tuple(e['shape']) == (96, 80)
    assert np.allclose(e['cam'], [2., 0, 0, 1, 0, 0, 0]) and float(e['depth'][0]) == 10.
    # the object fills the same part of both crops: the rebased flow vanishes on it, validity = silhouette
    from scipy.ndimage import binary_erosion
    fg = binary_erosion(e['mask'][0] > 0, iterations=3)           # away from the bilinear blend with the background's 0
    assert fg.sum() > 500 and np.abs(e['flow'][:2][:, fg]).max() < 1e-6
    inner = e['mask_dts'][0] == 0
    assert e['flow'][2][inner].min() == 1 and e['flow'][2][e['mask_dts'][0] > 2 / 64].max() == 0
    # background painted with the complement of the foreground colour
    assert np.allclose(e['img'][:, 0, 0], 1 - np.array([200, 120, 40]) / 255., atol=0.02)       # JPEG
    assert ds[2]['img'] is e['img']                               # decoded once, served from the cache afterwards
    back = ds[4]                                                  # frames 1 -> 0 use the backward flow of frame 1
    assert (back['id0'], back['id1']) == (1, 0)


def test_flow_rebasing_between_unequal_crops(tmp_path):
    # a point fixed on the object must map to itself: target-crop coordinate of x0 + (u + 0.5) * alp - 0.5 + flow
    write_sequence(str(tmp_path), step=(4, 0))
    loader, _ = vid.data_loader(make_opts(batch_size=1), shuffle=False, root=str(tmp_path))
    e = loader.dataset[2]
    size = 64
    u = np.arange(size)
    fx = e['flow'][0][size // 2]                                  # middle row, NDC units of the target crop
    x_target = u + fx * size / 2                                   # same pixel grid when the two crops are congruent
    from scipy.ndimage import binary_erosion
    fg = binary_erosion(e['mask'][0] > 0, iterations=3)[size // 2]
    assert fg.sum() > 10 and np.abs(x_target[fg] - u[fg]).max() < 1e-6


def test_trainer_batch_dictionary_from_loader(tmp_path):
    from lasr_amd import synth_data
    from lasr_amd.nnutils import train_utils
    write_sequence(str(tmp_path))
    opts = make_opts(checkpoint_dir='', name='t')
    loader, _ = vid.data_loader(opts, shuffle=False, root=str(tmp_path))
    tr = train_utils.LASRTrainer.__new__(train_utils.LASRTrainer)
    tr.opts, tr.device = opts, torch.device('cpu')
    batch = tr.set_input(next(iter(loader)))
    assert list(batch.keys()) == synth_data.KEYS
    B, IS = 2, 64
    shapes = {'input_imgs  ': (2 * B, 3, IS, IS), 'masks       ': (2 * B, IS, IS), 'cams        ': (2 * B, 7),
              'flow        ': (2 * B, 3, IS, IS), 'ddts_barrier': (2 * B, 1, IS, IS), 'pp          ': (2 * B, 2),
              'occ         ': (2 * B, IS, IS), 'mask_contour': (2 * B, 1, 1000, 2), 'depth_gt    ': (2 * B, 1)}
    for k, s in shapes.items():
        assert tuple(batch[k].shape) == s, k
    # interleaved pairs (train_utils.py:179-180): rows 0,1 = (frame t, frame t') of the first pair
    undo = batch['frameid'].view(B, 2).t().reshape(-1)
    assert undo.tolist() == [0., 0., 1., 1.]                      # elements 0 and 1 of the list are both the pair 0 -> 1


def test_resident_loader_serves_the_same_batches(tmp_path):
    # dataloader/resident.py: every pair prepared once and gathered by row must equal collate -> set_input per iteration
    from lasr_amd.dataloader import resident
    from lasr_amd.nnutils import train_utils
    write_sequence(str(tmp_path))
    opts = make_opts(batch_size=2)
    loader, _ = vid.data_loader(opts, shuffle=True, root=str(tmp_path))
    tr = train_utils.LASRTrainer.__new__(train_utils.LASRTrainer)
    tr.opts, tr.device = opts, torch.device('cpu')
    one = train_utils.LASRTrainer.__new__(train_utils.LASRTrainer)
    one.opts, one.device = SimpleNamespace(batch_size=1), torch.device('cpu')
    res = resident.ResidentLoader(loader, one._set_input_from_loader, torch.device('cpu'))
    assert res.n_pairs == 6 and len(res) == len(loader)
    for k, (a, b) in enumerate(zip(loader, res)):
        ref = tr.set_input(a)
        assert tr.set_input(b) is b and list(b.keys()) == list(ref.keys())
        for key in ref:
            assert torch.equal(ref[key], b[key]), key
        if k == 4:
            break


def test_pair_lists_for_frame_skipping_and_flow_directories(tmp_path):
    # vid.py:54-76: with dframe = 2 the flow lives in <seq>_02/, pairs are (i, i+2) forward and (i+2, i) backward, every
    # second start frame is kept, first and last pair are duplicated
    names = ['/data/DAVIS/JPEGImages/Full-Resolution/cat/%05d.jpg' % i for i in range(7)]
    ds = vid.VidDataset(make_opts(batch_size=1), imglist=names, can_frame=3, dframe=2, init_frame=1)
    assert ds.flowfwlist[0] == '/data/DAVIS/FlowFW/Full-Resolution/cat_02/flo-00000.pfm'
    assert ds.flowbwlist[4] == '/data/DAVIS/FlowBW/Full-Resolution/cat_02/flo-00004.pfm'
    assert ds.masklist[2] == '/data/DAVIS/Annotations/Full-Resolution/cat/00002.png'
    assert ds.camlist[2] == '/data/DAVIS/Camera/Full-Resolution/cat/00002.txt'
    unit = 6                                              # [1, 1, 3 | 3, 5, 5]: starts 1 and 3 forward, 3 and 5 backward
    assert ds.baselist[:unit] == [1, 1, 3, 3, 5, 5] and ds.directlist[:unit] == [1, 1, 1, 0, 0, 0]
    assert len(ds) == unit * (200 // unit)
    sil = vid.VidDataset(make_opts(batch_size=1, sil_path='/masks'), imglist=names, dframe=1)
    assert sil.masklist[1] == '/masks/cat/00001.png'


def test_image_ops_on_analytic_cases_of_the_opencv_definitions():
    # OpenCV is not installed, so the restatements of cv2.resize / cv2.remap in ext_utils/image.py are
    # pinned to cases whose OpenCV result follows from the documented sampling rule (INTER_LINEAR: src = (dst + 0.5) * scale - 0.5
    # with the border replicated; INTER_NEAREST: src = floor(dst * scale); remap with integer maps: a copy with constant border).
    # 2x2 -> 4x4 bilinear upsample: separable, weights (1, .75, .25, 0) along each axis
    src = np.array([[0., 10.], [20., 30.]])
    w = np.array([1., .75, .25, 0.])
    exp = (np.outer(w, w) * 0 + np.outer(w, 1 - w) * 10 + np.outer(1 - w, w) * 20 + np.outer(1 - w, 1 - w) * 30)
    assert np.allclose(iu.resize_linear(src, 4, 4), exp)
    # a linear ramp is reproduced exactly away from the clamped border, for a non-integer scale too
    ramp = np.arange(10.)[None].repeat(3, 0)
    out = iu.resize_linear(ramp, 7, 3)
    pos = (np.arange(7) + 0.5) * (10 / 7) - 0.5
    assert np.allclose(out[1], np.clip(pos, 0, 9))
    # 3 channels resize like three planes; an identity resize is the identity
    rgb = np.random.default_rng(0).uniform(0, 1, (5, 6, 3))
    assert np.allclose(iu.resize_linear(rgb, 6, 5), rgb)
    assert np.allclose(iu.resize_linear(rgb, 9, 4)[..., 1], iu.resize_linear(rgb[..., 1], 9, 4))
    # nearest: floor(i * scale), never rounds up (7 -> 3: indices 0, 2, 4), upsampling repeats
    assert np.array_equal(iu.resize_nearest(np.arange(7)[None], 3, 1), [[0, 2, 4]])
    assert np.array_equal(iu.resize_nearest(np.arange(3)[None], 6, 1), [[0, 0, 1, 1, 2, 2]])
    # remap with integer maps = crop with constant padding, any side
    img = np.arange(20.).reshape(4, 5)
    assert np.array_equal(iu.crop_pad(img, 3, 2, 3, border=-1.), [[13., 14., -1.], [18., 19., -1.], [-1., -1., -1.]])
    assert np.array_equal(iu.crop_pad(img, -2, -1, 3, border=0.), [[0., 0., 0.], [0., 0., 0.], [0., 0., 5.]])
    assert np.array_equal(iu.crop_pad(img, 10, 10, 2, border=7.), np.full((2, 2), 7.))         # entirely outside
    # distance transforms: a single foreground pixel gives Euclidean distances; the barrier is a sigmoid of the signed distance
    m = np.zeros((9, 9)); m[4, 4] = 1
    dt = iu.compute_dt(m, iters=0) * 9
    assert abs(dt[4, 7] - 3) < 1e-9 and abs(dt[1, 0] - 5) < 1e-9 and dt[4, 4] == 0
    b = iu.compute_dt_barrier(m, k=50)
    assert b[4, 4] < 0.5 < b[4, 5] and abs(b[0, 0] - 1 / (1 + np.exp(-50 * np.hypot(4, 4) / 9))) < 1e-9


def test_resize_linear_against_an_independent_bilinear_sampler():
    # a second opinion on the cv2.resize(INTER_LINEAR) restatement: scipy's own bilinear sampler (map_coordinates, order=1,
    # mode='nearest' = replicated border) evaluated at OpenCV's documented source positions (dst + 0.5) * scale - 0.5
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(3)
    for (h, w), (H, W) in (((17, 23), (40, 31)), ((64, 48), (25, 25)), ((9, 9), (9, 20)), ((5, 7), (256, 256))):
        src = rng.uniform(-1, 1, (h, w))
        yy = (np.arange(H) + 0.5) * (h / H) - 0.5
        xx = (np.arange(W) + 0.5) * (w / W) - 0.5
        ref = map_coordinates(src, np.meshgrid(yy, xx, indexing='ij'), order=1, mode='nearest')
        assert np.abs(iu.resize_linear(src, W, H) - ref).max() < 1e-12
        near = iu.resize_nearest(src, W, H)
        assert np.array_equal(near, src[np.minimum((np.arange(H) * (h / H)).astype(int), h - 1)][:, np.minimum((np.arange(W) * (w / W)).astype(int), w - 1)])
