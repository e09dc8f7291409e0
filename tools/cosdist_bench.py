#!/usr/bin/env python3
"""Per-layer timing of the perceptual-distance reduction kernels (lasr_cosdist_forward / _backward) at the five AlexNet
feature shapes of a 256x256 render, for the spot3 stage-0 step (N = 32 rendered images, 8 hypotheses per observed image).
Algorithmic bytes: forward 2*N*C*P*4, backward 3*N*C*P*4; the HBM fractions count the observed-side features once per image
(their 8 re-reads per hypothesis come from L2).  Prints one JSON object; run on an MI355X."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasr_amd import _lib                                             # noqa: E402
from lasr_amd.nnutils import fused_ops                                # noqa: E402

dev = torch.device('cuda:0')
h = _lib.lib()
st = torch.cuda.current_stream(dev).cuda_stream      # the stream fused_ops launches on (lasr_prof_* is scoped to a stream)
N, rep = 32, 8
out = {'N': N, 'repeat': rep, 'hbm_peak_GBs': 8000.0, 'layers': {}}
for name, (C, hw) in {'conv1': (64, 63), 'conv2': (192, 31), 'conv3': (384, 15), 'conv4': (256, 15), 'conv5': (256, 15)}.items():
    g = torch.Generator().manual_seed(0)
    fa = torch.randn(N // rep, C, hw, hw, generator=g).to(dev)
    fb = torch.randn(N, C, hw, hw, generator=g).to(dev).requires_grad_(True)
    up = torch.randn(N, generator=g).to(dev)
    for _ in range(3):
        d = fused_ops.cosine_distance(fa, fb, rep)
        d.backward(up)
    torch.cuda.synchronize()
    # kernel-only times from the library's own HIP events around each launch (lasr_prof_*): forward = reduction + fold
    h.lasr_prof_enable(st, 1)
    reps = 20
    for _ in range(reps):
        fb.grad = None
        fused_ops.cosine_distance(fa, fb, rep).backward(up)
    torch.cuda.synchronize()
    h.lasr_prof_enable(st, 0)
    t = {}
    for k in range(h.lasr_prof_kernel_count()):
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        h.lasr_prof_collect(st, k, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            t[h.lasr_prof_kernel_name(k).decode()] = ms.value / reps * 1e3
    fwd, bwd = t['cosdist_forward_kernel'], t['cosdist_backward_kernel']
    P = hw * hw
    bf, bb = 2 * N * C * P * 4, 3 * N * C * P * 4
    hbm_f, hbm_b = (N + N // rep) * C * P * 4, (2 * N + N // rep) * C * P * 4      # the observed side comes from L2 after its first read
    out['layers'][name] = {'C': C, 'P': P, 'blocks': N * -(-P // 32), 'fwd_us': round(fwd, 2), 'bwd_us': round(bwd, 2),
                           'fwd_algorithmic_GBs': round(bf / fwd / 1e3, 1), 'bwd_algorithmic_GBs': round(bb / bwd / 1e3, 1),
                           'fwd_frac_hbm': round(hbm_f / fwd / 1e3 / 8000, 3), 'bwd_frac_hbm': round(hbm_b / bwd / 1e3 / 8000, 3)}
print(json.dumps(out))
