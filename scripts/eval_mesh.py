#!/usr/bin/env python3
"""Chamfer + normal-consistency evaluation of reconstructed meshes against ground truth (reference: /root/reference/scripts/eval_mesh.py).

    python scripts/eval_mesh.py --testdir <dir with pred*.obj> --gtdir <dir with *.obj>

Protocol of the reference (:116-160): both meshes are centred, scaled so that their largest point-to-point distance is
10, 10 000 points are sampled uniformly by area from each, the prediction is aligned to the ground truth with rigid
ICP, and the symmetric Chamfer distance (mean squared nearest-neighbour distance, both directions summed --
pytorch3d.loss.chamfer_distance) of two fresh samples is reported, together with the normal term of the same call
(:165-167, :197-198: `chamfer_distance(X, Y, x_normals=nx, y_normals=ny)` returns as its second value
mean_x(1 - |cos(n_x, n_nn(x))|) + mean_y(1 - |cos(n_y, n_nn(y))|) over the sampled faces' normals; the script prints 1 minus
it as the normal consistency).  pytorch3d / trimesh are not available here: area sampling with face normals, Kabsch ICP,
the Chamfer sum and the normal term are written out; the nearest-neighbour searches run on lasr_nearest_point.
(The reference also re-meshes the prediction with the external Manifold binary and renders error images: not done.)
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasr_amd.nnutils import fused_ops                     # noqa: E402
from lasr_amd.soft_renderer.functional import load_obj      # noqa: E402


def sample_points(verts, faces, n, gen, return_normals=False):
    """n points uniformly distributed over the surface (area-weighted faces, uniform barycentrics); with return_normals also the
    unit normal of the face each point was drawn from (pytorch3d.ops.sample_points_from_meshes(..., return_normals=True))."""
    tri = verts[faces]
    cr = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = cr.norm(dim=1)
    pick = torch.multinomial(area / area.sum(), n, replacement=True, generator=gen)
    u = torch.rand(n, 2, device=verts.device, generator=gen)
    su = u[:, :1].sqrt()
    w = torch.cat([1 - su, su * (1 - u[:, 1:]), su * u[:, 1:]], 1)
    pts = (tri[pick] * w[:, :, None]).sum(1)
    if not return_normals:
        return pts
    return pts, cr[pick] / area[pick, None].clamp_min(1e-12)


def diameter(x):
    return float(torch.cdist(x, x).max())


def icp(x, y, iters=100, tol=1e-7):
    """Rigid (R, t) minimising sum |x R + t - nn_y(x R + t)|^2 by alternating nearest neighbours and Kabsch."""
    R = torch.eye(3, device=x.device)
    t = torch.zeros(3, device=x.device)
    prev = None
    for _ in range(iters):
        xt = x @ R + t
        d2, idx = fused_ops.nearest_point(xt[None], y[None])
        err = float(d2.mean())
        if prev is not None and abs(prev - err) < tol * max(prev, 1e-12):
            break
        prev = err
        tgt = y[idx[0]]
        mx, my = x.mean(0), tgt.mean(0)
        U, _, Vt = torch.linalg.svd((x - mx).t() @ (tgt - my))
        D = torch.diag(torch.tensor([1., 1., float(torch.sign(torch.linalg.det(U @ Vt)))], device=x.device))
        R = U @ D @ Vt
        t = my - mx @ R
    return R, t


def chamfer(x, y):
    dx = fused_ops.nearest_point(x[None], y[None])[0].mean()
    dy = fused_ops.nearest_point(y[None], x[None])[0].mean()
    return float(dx + dy)


def chamfer_with_normals(x, nx, y, ny):
    """(Chamfer distance, normal term) of two point sets with unit normals: the two return values of
    pytorch3d.loss.chamfer_distance(x, y, x_normals=nx, y_normals=ny) with its defaults (mean over points, the two directions
    summed, abs_cosine: a flipped face orientation does not count)."""
    dx, ix = fused_ops.nearest_point(x[None], y[None])
    dy, iy = fused_ops.nearest_point(y[None], x[None])
    cos_x = torch.nn.functional.cosine_similarity(nx, ny[ix[0]], dim=1, eps=1e-6).abs()
    cos_y = torch.nn.functional.cosine_similarity(ny, nx[iy[0]], dim=1, eps=1e-6).abs()
    return float(dx.mean() + dy.mean()), float((1 - cos_x).mean() + (1 - cos_y).mean())


def evaluate_pair(pred, gt, n=10000, seed=0, with_normals=False):
    """pred, gt: (verts [V,3], faces [F,3]) on the GPU -> Chamfer distance after normalisation and ICP; with_normals: the pair
    (Chamfer distance, normal consistency = 1 - normal term) the reference prints per frame (:197)."""
    gen = torch.Generator(device=pred[0].device).manual_seed(seed)
    (xv, xf), (yv, yf) = pred, gt
    yv = yv - yv.mean(0, keepdim=True)
    yv = 10 * yv / diameter(sample_points(yv, yf, 4000, gen))
    xv = xv - xv.mean(0, keepdim=True)
    xv = 10 * xv / diameter(sample_points(xv, xf, 4000, gen))
    R, t = icp(sample_points(xv, xf, n, gen), sample_points(yv, yf, n, gen))
    xv = xv @ R + t
    if not with_normals:
        return chamfer(sample_points(xv, xf, n, gen), sample_points(yv, yf, n, gen))
    x, nx = sample_points(xv, xf, n, gen, True)
    y, ny = sample_points(yv, yf, n, gen, True)
    cd, norm = chamfer_with_normals(x, nx, y, ny)
    return cd, 1. - norm


def main(argv=None):
    ap = argparse.ArgumentParser(description='mesh evaluation')
    ap.add_argument('--testdir', required=True)
    ap.add_argument('--gtdir', required=True)
    args = ap.parse_args(argv)
    dev = torch.device('cuda', 0)
    gts = sorted(glob.glob('%s/*.obj' % args.gtdir))
    preds = sorted(glob.glob('%s/pred*.obj' % args.testdir)) or sorted(glob.glob('%s/*.obj' % args.testdir))
    assert len(gts) == len(preds) and gts, 'need the same number of predicted and ground-truth meshes'
    cds, ncs = [], []
    for i, (p, g) in enumerate(zip(preds, gts)):
        pm, gm = load_obj(p, device=dev), load_obj(g, device=dev)
        cd, nc = evaluate_pair((pm[0].float(), pm[1].long()), (gm[0].float(), gm[1].long()), with_normals=True)
        cds.append(cd); ncs.append(nc)
        print('%04d: %.2f, %.2f' % (i, cd, nc))                       # the reference's line (:197): Chamfer, normal consistency
    print('ALL: %.2f, %.2f' % (np.mean(cds), np.mean(ncs)))
    main.normal_consistency = ncs
    return cds


if __name__ == '__main__':
    main()
