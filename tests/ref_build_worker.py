#!/usr/bin/env python3
"""Child process that runs ONE build of the reference soft-rasteriser (oracle/_ref/<variant>.so) on given inputs.

    python tests/ref_build_worker.py <variant> <in.npz> <out.npz>

TEST INFRASTRUCTURE (it uses oracle/).  The two builds of oracle/build_ref.py -- `sr_ref` (compiler defaults: FMA contraction
on, what the reference's setup.py produces) and `sr_ref_nofma` (-ffp-contract=off) -- export the same kernel symbols and cannot
share a process (oracle/sr_ref.py), so tests that need the other build, or both, go through this worker.

in.npz : face_vertices [N,F,3,3] f32, textures [N,F,T,3] f32, image_size, kwargs (JSON bytes: the keyword arguments of
         soft_rasterize), optional grad_soft_colors [N,4,IS,IS], optional flags `fp64` (also run the same kernels in double,
         AT_DISPATCH_FLOATING_TYPES, K.cu:701), `reps` (time forward + backward over that many repetitions).
out.npz: soft_colors, aggrs_info [, grad_faces, grad_textures] [, soft_colors_fp64] [, ms_per_step]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sr_ref                                   # noqa: E402


def main():
    variant, src, dst = sys.argv[1:4]
    dev = torch.device('cuda:0')
    with np.load(src) as z:
        fv = torch.from_numpy(z['face_vertices']).to(dev)
        ft = torch.from_numpy(z['textures']).to(dev)
        IS = int(z['image_size'])
        kw = json.loads(bytes(z['kwargs']).decode())
        g = torch.from_numpy(z['grad_soft_colors']).to(dev) if 'grad_soft_colors' in z.files else None
        fp64 = bool(z['fp64']) if 'fp64' in z.files else False
        reps = int(z['reps']) if 'reps' in z.files else 0
    kw['background_color'] = tuple(kw['background_color'])
    out = {}
    s = sr_ref.forward(fv, ft, IS, variant=variant, **kw)
    out['soft_colors'] = s['soft_colors'].cpu().numpy()
    out['aggrs_info'] = s['aggrs_info'].cpu().numpy()
    if g is not None:
        gf, gt = sr_ref.backward(s, g, IS, variant=variant, **kw)
        out['grad_faces'], out['grad_textures'] = gf.cpu().numpy(), gt.cpu().numpy()
    if fp64:
        s64 = sr_ref.forward(fv, ft, IS, variant=variant, dtype=torch.float64, **kw)
        out['soft_colors_fp64'] = s64['soft_colors'].cpu().numpy()
    if reps:
        gg = g if g is not None else torch.zeros(fv.shape[0], 4, IS, IS, device=dev)

        def step():
            st = sr_ref.forward(fv, ft, IS, variant=variant, **kw)
            sr_ref.backward(st, gg, IS, variant=variant, **kw)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        out['ms_per_step'] = np.float64((time.perf_counter() - t0) / reps * 1e3)
    np.savez(dst, **out)


if __name__ == '__main__':
    main()
