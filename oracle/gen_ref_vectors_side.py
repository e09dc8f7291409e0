#!/usr/bin/env python3
"""Golden vectors of the reference's two side kernels, run on an MI355X:
  load_textures_cuda_kernel.cu:8-66 (texture atlas -> per-face surface texels; scripts/render_syn.py:71 via load_obj.py:93)
  chamfer3D.cu:12-134 NmDistanceKernel (nearest neighbour, dist1 / idx1 as used at nnutils/mesh_net.py:477)

    gpurun -- python oracle/gen_ref_vectors_side.py      # writes gpurun_out/ref_vectors/side_reference_kernels.npz
    cp gpurun_out/ref_vectors/side_reference_kernels.npz tests/golden/

TEST INFRASTRUCTURE ONLY (inputs + the reference's outputs, no source).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sr_ref                        # noqa: E402


def main():
    dev = torch.device('cuda:0')
    out = {}
    lt = sr_ref.side_module('load_textures')
    rng = np.random.default_rng(3)
    for k, (H, W, F, R) in enumerate(((9, 17, 5, 3), (64, 48, 300, 5), (7, 7, 2, 1), (96, 128, 400, 5))):
        img = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
        uv = rng.uniform(0, 0.999, (F, 3, 2)).astype(np.float32)
        uv[0] = [[0, 0], [1, 0], [1, 0.5]]                 # u == 1: the kernel's x+1 neighbour carries weight 0
        upd = (rng.uniform(0, 1, F) > 0.2).astype(np.int32)
        upd[0] = 1
        tex = torch.full((F, R * R, 3), 0.25, dtype=torch.float32, device=dev)     # untouched where is_update == 0
        res = lt.load_textures(torch.from_numpy(img).to(dev), torch.from_numpy(uv).to(dev), tex, torch.from_numpy(upd).to(dev))
        torch.cuda.synchronize()
        for n, a in (('image', img), ('faces_uv', uv), ('is_update', upd), ('textures', res.cpu().numpy())):
            out['load_textures/%d/%s' % (k, n)] = a
    ch = sr_ref.side_module('chamfer_3D')
    rng = np.random.default_rng(4)
    for k, (B, n, m) in enumerate(((1, 642, 642), (2, 802, 802), (1, 1282, 500), (3, 100, 37))):
        a = rng.standard_normal((B, n, 3)).astype(np.float32)
        b = (a[:, rng.permutation(n)[:m] if m <= n else rng.integers(0, n, m)] * np.float32([-1, 1, 1])
             + 0.05 * rng.standard_normal((B, m, 3))).astype(np.float32)          # the mirrored set of mesh_net.py:474-477
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        d1, d2 = torch.zeros(B, n, device=dev), torch.zeros(B, m, device=dev)
        i1, i2 = torch.zeros(B, n, dtype=torch.int32, device=dev), torch.zeros(B, m, dtype=torch.int32, device=dev)
        ch.forward(ta, tb, d1, d2, i1, i2)
        torch.cuda.synchronize()
        for nme, arr in (('xyz1', a), ('xyz2', b), ('dist1', d1.cpu().numpy()), ('dist2', d2.cpu().numpy()),
                         ('idx1', i1.cpu().numpy()), ('idx2', i2.cpu().numpy())):
            out['chamfer/%d/%s' % (k, nme)] = arr
    outdir = os.path.join(ROOT, 'gpurun_out', 'ref_vectors')
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, 'side_reference_kernels.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
