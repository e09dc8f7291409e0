#!/usr/bin/env python3
"""Throughput of the REFERENCE kernels (oracle/_ref, built by oracle/build_ref.py) on this GPU, next to the HIP
path, on bench.py's workload (mesh M2 at 256x256, LASR modes).  A tool, not a test: run on the GPU box,
    python tests/ref_gpu_timing.py [frames] > gpurun_out/ref_gpu_timing.json
It lives under tests/ because it uses oracle/ (test infrastructure).  north_star's target is stated against the
reference's single-GPU soft-rasteriser forward+backward throughput; this is that number on an MI355X.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasr_amd import synth                                     # noqa: E402
from lasr_amd.soft_renderer import functional as srf           # noqa: E402
from oracle import sr_ref                                      # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    IS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device('cuda:0')
    fv, ft, near, far = synth.raster_batch(11, 26, count=n)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    g = torch.from_numpy(synth.upstream_grad(n, IS)).to(dev)
    tfv, tft = torch.from_numpy(fv).to(dev), torch.from_numpy(ft).to(dev)
    out = dict(frames=n, image_size=IS, faces=int(fv.shape[1]), device=torch.cuda.get_device_name(0))
    variant = sys.argv[3] if len(sys.argv) > 3 else 'sr_ref_nofma'        # one build per process (oracle/sr_ref.py)

    def step():
        s = sr_ref.forward(tfv, tft, IS, variant=variant, **kw)
        sr_ref.backward(s, g, IS, variant=variant, **kw)
    dt = timed(step, 3)
    out['reference_build'] = dict(variant=variant, ms_per_step=dt * 1e3, frames_per_s=n / dt)
    a = tfv.clone().requires_grad_(True)
    b = tft.clone().requires_grad_(True)

    def hip_step():
        a.grad = b.grad = None
        srf.soft_rasterize(a, b, IS, **kw).backward(g)
    dt = timed(hip_step, 10)
    out['lasr_hip_operator'] = dict(ms_per_step=dt * 1e3, frames_per_s=n / dt)
    out['speedup_vs_reference_build'] = out['lasr_hip_operator']['frames_per_s'] / out['reference_build']['frames_per_s']
    # parity at this size, image and gradients
    s = sr_ref.forward(tfv, tft, IS, variant=variant, **kw)
    gf, gt = sr_ref.backward(s, g, IS, variant=variant, **kw)
    img = srf.soft_rasterize(a, b, IS, **kw)
    a.grad = b.grad = None
    img.backward(g)
    d = (img.detach() - s['soft_colors']).abs()
    out['image_max_abs'] = float(d.max())
    out['pixels_over_1e-4'] = int((d > 1e-4).sum())
    out['grad_faces_rel'] = float((a.grad - gf).abs().max() / gf.abs().max())
    out['grad_textures_rel'] = float((b.grad - gt).abs().max() / gt.abs().max())
    print(json.dumps(out))


if __name__ == '__main__':
    main()
