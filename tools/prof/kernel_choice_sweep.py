#!/usr/bin/env python3
"""Forward kernel time of LASR's mode combination by launch size and kernel form (eight / four waves per 8x8 tile, one wave per
tile, and what the default thresholds pick): the measurement behind the built-in thresholds of lasr_sr_options.
    python tools/prof/kernel_choice_sweep.py [channels]   (on an MI355X; mesh M2 at 256x256)"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasr_amd import _lib, synth                                    # noqa: E402
from lasr_amd.soft_renderer import functional as srf                # noqa: E402

BIG = 10 ** 12
FORMS = {'eight': (BIG, BIG, BIG), 'four': (0, BIG, BIG), 'one': (0, 0, 0), 'default': (-1, -1, -1)}
C = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FOCAL = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0      # 9: the bench object (a third of the tiles busy); 16: the object fills the frame, as LASR's crops do
dev = torch.device('cuda:0')
h = _lib.lib()
st = torch.cuda.current_stream(dev).cuda_stream
out = {}
for n in (1, 2, 4, 6, 8, 12, 16, 24, 32, 40, 48, 64, 96):
    v, f, tex = synth.blobby_mesh(11)
    pv = synth.frame_vertices(v, 26, focal=FOCAL, count=n)
    near, far = synth.near_far(pv[:, :, 2])
    fv = pv[:, f]
    import numpy as np
    ft = np.broadcast_to(tex[f][None], fv.shape).copy()
    kw = dict(synth.LASR_MODES, near=near, far=far)
    a = torch.from_numpy(fv).to(dev)
    b = torch.from_numpy(ft).to(dev)
    if C > 3:
        b = torch.cat([b] * (C // 3), -1).contiguous()
        kw['background_color'] = [1.0] * C
    row = {}
    for name, th in FORMS.items():
        srf.set_launch_thresholds(*th)
        for _ in range(3):
            srf.soft_rasterize(a, b, 256, **kw)
        torch.cuda.synchronize()
        h.lasr_prof_enable(st, 1)
        for _ in range(20):
            srf.soft_rasterize(a, b, 256, **kw)
        torch.cuda.synchronize()
        h.lasr_prof_enable(st, 0)
        ms, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
        tot = 0.0
        for k in range(h.lasr_prof_kernel_count()):
            h.lasr_prof_collect(st, k, ctypes.byref(ms), ctypes.byref(cnt))
            if h.lasr_prof_kernel_name(k).decode() in ('sr_forward_kernel', 'sr_order_kernel'):   # multiples of 8 frames: + the tile order
                tot += ms.value / 20
        row[name] = round(tot, 5)
    out[n] = row
    print(n, row, flush=True)
srf.set_launch_thresholds()
print(json.dumps(out))
