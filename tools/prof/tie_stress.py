"""Stress of the near-tie logic (sr_device.h: near_tie): meshes with slivers and edge-on faces at several sizes, LASR's modes; the
pair-walk forward (one edge projection where the choice is clear) against the one-wave kernel (the reference's three projections for
every inside pixel): the images must agree to rounding.   python tools/prof/tie_stress.py   (one MI355X)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf
dev = torch.device('cuda', 0)
BIG = 10 ** 12
worst = 0.0
for seed in range(12):
    rng = np.random.default_rng(seed)
    nu = [4, 8, 11, 16][seed % 4]
    fv, ft, near, far = synth.raster_batch(nu, 3, count=6)
    fv = fv.copy()
    # squash the object along a random direction per frame (edge-on faces at the silhouette, slivers in the interior)
    for n in range(fv.shape[0]):
        a = rng.uniform(0, np.pi)
        d = np.array([np.cos(a), np.sin(a)], np.float32)
        k = rng.uniform(0.05, 0.6)
        xy = fv[n, :, :, :2]
        c = xy.reshape(-1, 2).mean(0)
        rel = xy - c
        fv[n, :, :, :2] = c + rel - (1 - k) * (rel @ d)[..., None] * d
    for IS in (64, 128, 256):
        kw = dict(synth.LASR_MODES, near=near, far=far)
        t = lambda x: torch.from_numpy(x).to(dev)
        srf.set_launch_thresholds(0, 0, 0, -1, BIG)          # one wave per 8x8 tile
        a = srf.soft_rasterize(t(fv), t(ft), IS, **kw).cpu().numpy()
        srf.set_launch_thresholds(0, 0, 0, -1, 0)            # pair walk
        b = srf.soft_rasterize(t(fv), t(ft), IS, **kw).cpu().numpy()
        srf.set_launch_thresholds()
        dmax = float(np.abs(a - b).max())
        worst = max(worst, dmax)
        print('seed %2d nu %2d IS %3d  max |pair walk - one wave| = %.3e  (pixels > 1e-6: %d)' % (seed, nu, IS, dmax, int((np.abs(a - b) > 1e-6).sum())), flush=True)
print('worst', worst)
