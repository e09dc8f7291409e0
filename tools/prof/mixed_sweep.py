#!/usr/bin/env python3
"""Forward time (forward kernel + the two order kernels, library HIP events) of ordered launches under the per-range kernel choice
of round 5 (sr_forward_mixed_kernel): weight threshold from which a tile gets the four-wave body, against one wave per tile
(threshold 0) and four waves for every tile.  The measurement behind the default of lasr_sr_options.mixed_min_weight.
    python tools/prof/mixed_sweep.py        (on an MI355X)
Cases: the bench object (mesh M2, a third of the tiles busy) with 3 channels; LASR's own render (1280 faces, nine channels, the
object filling the crop: focal 16)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lasr_amd import _lib, synth                                    # noqa: E402
from lasr_amd.soft_renderer import functional as srf                # noqa: E402

BIG = 10 ** 12
dev = torch.device('cuda:0')
h = _lib.lib()
st = torch.cuda.current_stream(dev).cuda_stream
CASES = [('M2 3ch focal 9', 11, 3, 9.0, 256, (8, 16, 24, 32, 64, 128)),
         ('M1 9ch focal 16 (LASR crop)', 8, 9, 16.0, 256, (8, 16, 32, 96)),
         ('M1 9ch focal 9', 8, 9, 9.0, 256, (16, 96)),
         ('M2 3ch focal 9 512', 11, 3, 9.0, 512, (16, 64))]
if len(sys.argv) > 1 and sys.argv[1] == 'short':
    CASES = [('M2 3ch focal 9', 11, 3, 9.0, 256, (16, 24, 64)), ('M1 9ch focal 16 (LASR crop)', 8, 9, 16.0, 256, (16, 96))]
FORMS = [('one wave', (0, 0, 0, BIG, 0)), ('four waves', (0, BIG, BIG, BIG, 0))] + \
        [('mixed %d' % t, (0, 0, 0, BIG, t)) for t in (8, 16, 24, 32, 48, 64, 96)]
out = {}
for label, nu, C, focal, IS, counts in CASES:
    v, f, tex = synth.blobby_mesh(nu)
    for n in counts:
        pv = synth.frame_vertices(v, 26, focal=focal, count=n)
        near, far = synth.near_far(pv[:, :, 2])
        fv = pv[:, f]
        ft = np.broadcast_to(tex[f][None], fv.shape).copy()
        kw = dict(synth.LASR_MODES, near=near, far=far)
        a = torch.from_numpy(fv).to(dev)
        b = torch.from_numpy(ft).to(dev)
        if C > 3:
            b = torch.cat([b] * (C // 3), -1).contiguous()
            kw['background_color'] = [1.0] * C
        row = {}
        for name, th in FORMS:
            srf.set_launch_thresholds(*th)
            for _ in range(3):
                srf.soft_rasterize(a, b, IS, **kw)
            torch.cuda.synchronize()
            reps = 20 if n * IS <= 64 * 256 else 8
            h.lasr_prof_enable(st, 1)
            for _ in range(reps):
                srf.soft_rasterize(a, b, IS, **kw)
            torch.cuda.synchronize()
            h.lasr_prof_enable(st, 0)
            ms, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
            fwd = order = 0.0
            for k in range(h.lasr_prof_kernel_count()):
                h.lasr_prof_collect(st, k, ctypes.byref(ms), ctypes.byref(cnt))
                nm = h.lasr_prof_kernel_name(k).decode()
                if nm == 'sr_forward_kernel':
                    fwd = ms.value / reps
                elif nm == 'sr_order_kernel':
                    order = ms.value / reps
            row[name] = [round(fwd, 5), round(order, 5)]
        out['%s, %d frames' % (label, n)] = row
        print('%-44s %s' % ('%s, %d frames' % (label, n), '  '.join('%s %.4f' % (k, v[0]) for k, v in row.items())), flush=True)
srf.set_launch_thresholds()
print(json.dumps(out))
