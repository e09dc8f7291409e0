#!/usr/bin/env python3
"""HIP rasteriser vs the CPU oracle on a user-supplied mesh -- meant for the reference's textured spot model (BASELINE configs[1]:
"spot3 256x256 ... HIP soft-rasterizer fwd/bwd vs reference tolerance check"), which is read from a reference checkout and not shipped.

    python tests/spot_parity.py --obj <reference>/database/misc/spot/spot_triangulated.obj [--frames 3] [--out profiles/x.json]

Frames are posed exactly as scripts/render_syn.py poses them (--model spot placement, yaw sweep, orthographic look_at).  Checked per frame:
  soft LASR modes (euclidean / softmax / prod, vertex colours = normalised positions): image max-abs vs oracle, gradients vs oracle;
  hard data-generation modes with the model's own 5x5 surface textures: image max-abs, face-index map and z-buffer equality.
The oracle is test infrastructure (oracle/), so this checker lives under tests/ (pytest does not collect it: it needs the model file)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasr_amd import synth                                          # noqa: E402
from lasr_amd.soft_renderer import functional as srf                # noqa: E402
from oracle import sr_oracle                                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--obj', required=True)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--img_size', type=int, default=256)
    ap.add_argument('--out', default='')
    ap.add_argument('--bench', type=int, default=0, help='also time forward+backward of this many posed frames (soft LASR modes)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    v, f, tex = srf.load_obj(a.obj, load_texture=True, texture_res=5, texture_type='surface', device=dev)
    v = v.clone()
    v[:, 1] *= -1; v[:, 1] += 0.1; v /= 1.2                         # render_syn.py:72-75 of the reference
    F, IS = f.shape[0], a.img_size
    fv_frames = []
    for i in range(a.frames):                                       # the yaw sweep of render_syn.py, depth 10, focal 10 (orthographic)
        ry = 3 * 1.57 + 6.28 * i / a.frames
        R = torch.tensor([[math.cos(ry), 0, math.sin(ry)], [0, 1, 0], [-math.sin(ry), 0, math.cos(ry)]], dtype=torch.float32, device=dev)
        p = v @ R.t()
        p = torch.stack([p[:, 0], -p[:, 1], p[:, 2] + 10.], 1)      # image-space y flip of the look_at pipeline, object at depth 10
        fv_frames.append(p[f.long()])
    fv = torch.stack(fv_frames).contiguous()                        # [N,F,3,3]
    N = fv.shape[0]
    near, far = float(fv[..., 2].min() - 1), float(fv[..., 2].max() + 1)
    out = {'mesh': os.path.basename(a.obj), 'vertices': int(v.shape[0]), 'faces': int(F), 'frames': N, 'image_size': IS}
    sr_oracle.lib()

    # ---- soft LASR modes, vertex colours
    col = ((v - v.min(0)[0]) / (v.max(0)[0] - v.min(0)[0]))[f.long()][None].repeat(N, 1, 1, 1).contiguous()
    kw = dict(synth.LASR_MODES, near=near, far=far)
    tfv, tft = fv.clone().requires_grad_(True), col.clone().requires_grad_(True)
    img = srf.soft_rasterize(tfv, tft, IS, **kw)
    g = torch.from_numpy(synth.upstream_grad(N, IS, 3)).to(dev)
    img.backward(g)
    ref = sr_oracle.forward(fv.cpu().numpy(), col.cpu().numpy(), IS, **kw)
    gf_ref, gt_ref = sr_oracle.backward(ref, g.cpu().numpy(), IS, **kw)
    out['soft'] = {'image_max_abs': float(np.abs(img.detach().cpu().numpy() - ref['soft_colors']).max()),
                   'grad_faces_rel': float(np.abs(tfv.grad.cpu().numpy().reshape(gf_ref.shape) - gf_ref).max() / np.abs(gf_ref).max()),
                   'grad_textures_rel': float(np.abs(tft.grad.cpu().numpy().reshape(gt_ref.shape) - gt_ref).max() / np.abs(gt_ref).max()),
                   'covered_fraction': float((ref['soft_colors'][:, 3] > 0.5).mean())}

    # ---- hard data-generation modes, the model's surface textures (scripts/render_syn.py:135-137 of the reference)
    hard = dict(background_color=[0.2, 0.3, 0.4], near=near, far=far, fill_back=True, eps=1e-3, sigma_val=1e-12, dist_func='hard',
                dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='hard', aggr_func_alpha='hard', texture_type='surface')
    texN = tex[None].repeat(N, 1, 1, 1).contiguous()
    himg, haggr = srf.soft_rasterize_raw(fv, texN, IS, hard['background_color'], near, far, True, 1e-3, 1e-12, 'hard', 1e-4, 1e-4,
                                         'hard', 'hard', 'surface')
    href = sr_oracle.forward(fv.cpu().numpy(), texN.cpu().numpy(), IS, **hard)
    out['hard_surface'] = {'image_max_abs': float(np.abs(himg.cpu().numpy() - href['soft_colors']).max()),
                           'face_index_map_equal': bool(np.array_equal(haggr[:, 1].cpu().numpy(), href['aggrs_info'][:, 1])),
                           'z_buffer_equal': bool(np.array_equal(haggr[:, 0].cpu().numpy(), href['aggrs_info'][:, 0])),
                           'distinct_faces_visible': int(len(np.unique(href['aggrs_info'][:, 1])) - 1)}
    if a.bench:
        import time
        B = a.bench
        frames = []
        for i in range(B):                                          # 26 yaw positions, repeated (the "~26 frames" of spot3)
            ry = 3 * 1.57 + 6.28 * (i % 26) / 26
            R = torch.tensor([[math.cos(ry), 0, math.sin(ry)], [0, 1, 0], [-math.sin(ry), 0, math.cos(ry)]], dtype=torch.float32, device=dev)
            p = v @ R.t()
            frames.append(torch.stack([p[:, 0], -p[:, 1], p[:, 2] + 10.], 1)[f.long()])
        bfv = torch.stack(frames).contiguous().requires_grad_(True)
        bft = col[:1].repeat(B, 1, 1, 1).contiguous().requires_grad_(True)
        bg = torch.from_numpy(synth.upstream_grad(B, IS, 1)).to(dev)

        def step():
            bfv.grad = bft.grad = None
            srf.soft_rasterize(bfv, bft, IS, **kw).backward(bg)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out['bench'] = {'frames_per_step': B, 'ms_per_step': dt * 1e3, 'frames_per_s': B / dt,
                        'note': 'soft_rasterize forward + backward through the autograd operator, soft LASR modes, vertex colours'}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
