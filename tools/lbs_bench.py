#!/usr/bin/env python3
"""Micro-benchmark of the LBS kernels (the one MFMA user on the path) at the sizes of SURVEY.md section 8:
S0 (N=16, V=642, K=21), C4 (N=4, V=1212, K=36), dog15 stage 4 (N=6, V=1282, K=36), and a large batch.
Prints per-call times from HIP events; run under `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 ...` for MFMA counters."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasr_amd.nnutils import geom_utils

dev = torch.device('cuda:0')
out = {}
for name, (N, V, K) in {'S0': (16, 642, 21), 'C4': (4, 1212, 36), 'dog15': (6, 1282, 36), 'big': (256, 1212, 36)}.items():
    g = torch.Generator(device='cpu').manual_seed(0)
    v = torch.randn(N, V, 3, generator=g).to(dev).requires_grad_(True)
    R = torch.randn(N * K, 3, 3, generator=g).to(dev).requires_grad_(True)
    T = torch.randn(N * K, 1, 3, generator=g).to(dev).requires_grad_(True)
    s = torch.softmax(torch.randn(N, K - 1, V, 1, generator=g), 1).to(dev).requires_grad_(True)
    up = torch.randn(N, V, 3, generator=g).to(dev)
    for _ in range(3):
        o = geom_utils.obj_to_cam(v, R, T, K, 1, s); o.backward(up)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 50
    torch.cuda.synchronize()
    e[0].record()
    with torch.no_grad():
        for _ in range(reps):
            o = geom_utils.obj_to_cam(v, R, T, K, 1, s)
    e[1].record()
    for _ in range(reps):
        o = geom_utils.obj_to_cam(v, R, T, K, 1, s); o.backward(up)
    e[2].record()
    torch.cuda.synchronize()
    fwd = e[0].elapsed_time(e[1]) / reps * 1e3
    both = e[1].elapsed_time(e[2]) / reps * 1e3
    flops = 2 * N * V * (K - 1) * 12 + 2 * N * V * 12
    byts = N * (12 * V + 4 * V * (K - 1) + 48 * K + 12 * V)
    out[name] = dict(N=N, V=V, K=K, fwd_us=round(fwd, 2), fwd_bwd_us=round(both, 2), fwd_gflops=round(flops / fwd / 1e3, 2),
                     fwd_GBs=round(byts / fwd / 1e3, 2))
print(json.dumps(out))
