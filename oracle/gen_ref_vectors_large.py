#!/usr/bin/env python3
"""Reference-kernel outputs for the reference's own demo asset (5856 faces: more than 64 groups of 64 faces).

    python oracle/build_ref.py                                                   # here (needs /root/reference)
    mkdir -p scratch/spot_asset && cp /root/reference/database/misc/spot/* scratch/spot_asset/   # DATA files, not kept
    gpurun -- python oracle/gen_ref_vectors_large.py                             # writes gpurun_out/ref_vectors/
    cp gpurun_out/ref_vectors/spot_reference_kernels.npz tests/golden/ && rm -r scratch/spot_asset

TEST INFRASTRUCTURE ONLY.  The model is posed as scripts/render_syn.py:70-75,145-160 poses it (--model spot: y flip, +0.1,
/1.2; yaw sweep 3*1.57 + 6.28 i / 3; depth 10; orthographic).  The fixture holds the posed face vertices of the three default
poses, the vertex colours / 5x5 surface texels used as textures (texels rounded to fp16 so that the stored input is exactly
what the kernels saw) and, for one pose, what the reference's forward_soft_rasterize / backward_soft_rasterize
(soft_rasterize_cuda.cpp:59-138 -> soft_rasterize_cuda_kernel.cu:245-668, oracle/_ref/sr_ref_nofma.so) returned:
  soft/*  LASR's training modes (nnutils/mesh_net.py:136-145), vertex colours = normalised positions
  hard/*  the data-generation modes of render_syn.py:135-137 with the model's surface textures.
tests/test_raster_large_meshes_gpu.py checks the HIP path against it.
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasr_amd import synth                                          # noqa: E402
from lasr_amd.soft_renderer import functional as srf                # noqa: E402
from oracle import sr_ref                                           # noqa: E402

STORED = 1          # which pose's outputs are stored
IS = 256


def main():
    dev = torch.device('cuda:0')
    obj = os.path.join(ROOT, 'scratch', 'spot_asset', 'spot_triangulated.obj')
    v, f, tex = srf.load_obj(obj, load_texture=True, texture_res=5, texture_type='surface', device=dev)
    v = v.clone()
    v[:, 1] *= -1; v[:, 1] += 0.1; v /= 1.2
    frames = []
    for i in range(3):
        ry = 3 * 1.57 + 6.28 * i / 3
        R = torch.tensor([[math.cos(ry), 0, math.sin(ry)], [0, 1, 0], [-math.sin(ry), 0, math.cos(ry)]],
                         dtype=torch.float32, device=dev)
        p = v @ R.t()
        p = torch.stack([p[:, 0], -p[:, 1], p[:, 2] + 10.], 1)
        frames.append(p[f.long()])
    fv = torch.stack(frames).contiguous()
    near, far = float(fv[..., 2].min() - 1), float(fv[..., 2].max() + 1)
    col = ((v - v.min(0)[0]) / (v.max(0)[0] - v.min(0)[0]))[f.long()].contiguous()
    tex16 = tex.to(torch.float16)
    out = dict(face_vertices=fv.cpu().numpy(), vertex_colours=col.cpu().numpy(), surface_textures_f16=tex16.cpu().numpy(),
               near_far=np.array([near, far], np.float32), stored_frame=np.int32(STORED))
    g = torch.from_numpy(synth.upstream_grad(1, IS, seed=3)).to(dev)
    one = fv[STORED:STORED + 1].contiguous()
    soft = dict(synth.LASR_MODES, near=near, far=far)
    s = sr_ref.forward(one, col[None].contiguous(), IS, variant='sr_ref_nofma', **soft)
    gf, gt = sr_ref.backward(s, g, IS, variant='sr_ref_nofma', **soft)
    out.update({'soft/soft_colors': s['soft_colors'].cpu().numpy(), 'soft/grad_faces': gf.cpu().numpy(),
                'soft/grad_textures': gt.cpu().numpy()})
    hard = dict(background_color=(0.2, 0.3, 0.4), near=near, far=far, fill_back=True, eps=1e-3, sigma_val=1e-12, dist_func='hard',
                dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='hard', aggr_func_alpha='hard', texture_type='surface')
    s = sr_ref.forward(one, tex16.float()[None].contiguous(), IS, variant='sr_ref_nofma', **hard)
    out.update({'hard/soft_colors': s['soft_colors'].cpu().numpy(), 'hard/aggrs_info': s['aggrs_info'].cpu().numpy()})
    outdir = os.path.join(ROOT, 'gpurun_out', 'ref_vectors')
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, 'spot_reference_kernels.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes; faces', int(f.shape[0]), 'vertices', int(v.shape[0]),
          'distinct faces visible', len(np.unique(out['hard/aggrs_info'][:, 1])) - 1, torch.cuda.get_device_name(0))


if __name__ == '__main__':
    main()
