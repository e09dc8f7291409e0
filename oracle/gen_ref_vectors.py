#!/usr/bin/env python3
"""Golden vectors of the REFERENCE soft-rasteriser kernels, produced by running them on an MI355X.

    python oracle/build_ref.py                                  # here (needs /root/reference)
    gpurun -- python oracle/gen_ref_vectors.py                  # on the GPU box: writes gpurun_out/ref_vectors/
    cp gpurun_out/ref_vectors/sr_reference_kernels.npz tests/golden/

TEST INFRASTRUCTURE ONLY.  The fixture holds data: seeded inputs, the call's scalar arguments and what the
reference's `forward_soft_rasterize` / `backward_soft_rasterize` (soft_rasterize_cuda.cpp:59-138 ->
soft_rasterize_cuda_kernel.cu:245-668) returned for them, for both builds of oracle/build_ref.py
("fma": compiler defaults; "nofma": -ffp-contract=off).  tests/test_oracle_vs_reference_vectors.py checks
oracle/sr_oracle.c against it on the CPU; tests/test_raster_vs_reference_gpu.py checks the HIP path.
"""
import hashlib
import itertools
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lasr_amd import synth                       # noqa: E402   (input generator only: numpy)
from oracle import sr_ref                        # noqa: E402


def cases():
    """(name, face_vertices, textures, image_size, kwargs, dtype) -- the reference tests nothing itself;
    the list covers every mode id of soft_rasterize.py:22-25, both texture types, the depth-cull / winding /
    sigma edge cases of SURVEY App. A and the sizes BASELINE names."""
    out = []
    fv, ft, near, far = synth.raster_batch(3, 3, count=1)
    base = dict(synth.LASR_MODES, near=near, far=far)
    for dist, rgb, alpha in itertools.product(['hard', 'barycentric', 'euclidean'], ['hard', 'softmax'],
                                              ['hard', 'sum', 'prod']):
        out.append(('modes_%s_%s_%s' % (dist, rgb, alpha), fv, ft, 48,
                    dict(base, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha), np.float32))
    rng = np.random.default_rng(5)
    for res, rgb in itertools.product([1, 2, 3], ['hard', 'softmax']):
        tx = rng.uniform(0, 1, (1, fv.shape[1], res * res, 3)).astype(np.float32)
        out.append(('surface_r%d_%s' % (res, rgb), fv, tx, 48, dict(base, texture_type='surface', aggr_func_rgb=rgb),
                    np.float32))
    out.append(('single_sided', fv, ft, 48, dict(base, fill_back=False), np.float32))
    out.append(('sigma_1e-5', fv, ft, 48, dict(base, sigma_val=1e-5), np.float32))
    out.append(('single_sided_hard', fv, ft, 48, dict(base, fill_back=False, aggr_func_rgb='hard', dist_func='hard',
                                                     aggr_func_alpha='hard'), np.float32))
    out.append(('near_plane_inside', fv, ft, 48, dict(base, near=10.0), np.float32))
    out.append(('render_syn_modes', fv, ft, 48, dict(base, sigma_val=1e-12, aggr_func_rgb='hard'), np.float32))
    for IS in (1, 7, 33):
        out.append(('ragged_%d' % IS, fv, ft, IS, base, np.float32))
    out.append(('fp64', fv.astype(np.float64), ft.astype(np.float64), 48, base, np.float64))
    # screen-filling faces among many small ones
    rng = np.random.default_rng(7)
    F = 600
    c = rng.uniform(-0.9, 0.9, (1, F, 1, 2))
    tri = c + rng.uniform(-0.08, 0.08, (1, F, 3, 2))
    z = rng.uniform(2, 4, (1, F, 3, 1))
    big = np.concatenate([tri, z], -1).astype(np.float32)
    big[0, 0] = [[-1.5, -1.2, 3], [1.4, -1.1, 3.5], [0.1, 1.6, 2.5]]
    big[0, 300] = [[-0.9, 0.8, 2.2], [0.95, 0.9, 3.9], [0.0, -0.97, 3.0]]
    bt = rng.uniform(0, 1, big.shape).astype(np.float32)
    out.append(('big_faces', big, bt, 64, dict(synth.LASR_MODES, near=1.0, far=5.0), np.float32))
    out.append(('big_faces_hard', big, bt, 64, dict(synth.LASR_MODES, near=1.0, far=5.0, aggr_func_rgb='hard',
                                                    dist_func='hard', aggr_func_alpha='hard'), np.float32))
    # the meshes BASELINE names: M1 (spot3 stage 0 size) at 128^2, M2 (~1.2k / 2.3k) at 256^2
    fv1, ft1, n1, f1 = synth.raster_batch(8, 3, count=1, first=1)
    out.append(('M1_128', fv1, ft1, 128, dict(synth.LASR_MODES, near=n1, far=f1), np.float32))
    fv2, ft2, n2, f2 = synth.raster_batch(11, 26, count=1, first=5)
    out.append(('M2_256', fv2, ft2, 256, dict(synth.LASR_MODES, near=n2, far=f2), np.float32))
    out.append(('M2_256_hard', fv2, ft2, 256, dict(synth.LASR_MODES, near=n2, far=f2, aggr_func_rgb='hard',
                                                   dist_func='hard', aggr_func_alpha='hard'), np.float32))
    return out


def run_variant(variant, tag, path):
    """Child process: one build of the reference per process (oracle/sr_ref.py explains why)."""
    dev = torch.device('cuda:0')
    arrays = {}
    for name, fv, ft, IS, kw, dt in cases():
        tdt = torch.float32 if dt == np.float32 else torch.float64
        g = synth.upstream_grad(fv.shape[0], IS, seed=2).astype(dt)
        s = sr_ref.forward(torch.from_numpy(fv).to(dev), torch.from_numpy(ft).to(dev), IS, variant=variant,
                           dtype=tdt, **kw)
        gf, gt = sr_ref.backward(s, torch.from_numpy(g).to(dev), IS, variant=variant, **kw)
        torch.cuda.synchronize()
        for k, v in (('soft_colors', s['soft_colors']), ('aggrs_info', s['aggrs_info']),
                     ('faces_info', s['faces_info']), ('grad_faces', gf), ('grad_textures', gt)):
            if tag == 'fma' and k == 'faces_info':
                continue
            arrays['%s/%s/%s' % (name, tag, k)] = v.cpu().numpy()
    np.savez(path, **arrays)


def main():
    assert torch.cuda.is_available(), 'run on the GPU box'
    if len(sys.argv) == 4:
        return run_variant(*sys.argv[1:])
    import subprocess
    outdir = os.path.join(ROOT, 'gpurun_out', 'ref_vectors')
    os.makedirs(outdir, exist_ok=True)
    arrays, manifest = {}, {}
    for variant, tag in (('sr_ref_nofma', 'nofma'), ('sr_ref', 'fma')):
        part = os.path.join(outdir, 'part_%s.npz' % tag)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), variant, tag, part])
        with np.load(part) as z:
            arrays.update({k: z[k] for k in z.files})
        os.remove(part)
    for name, fv, ft, IS, kw, dt in cases():
        g = synth.upstream_grad(fv.shape[0], IS, seed=2).astype(dt)
        inputs = {}
        for field, arr in (('face_vertices', fv), ('textures', ft), ('grad_soft_colors', g)):
            key = 'input/' + hashlib.sha1(np.ascontiguousarray(arr).tobytes() + str(arr.dtype).encode()).hexdigest()[:12]
            arrays[key] = arr                   # cases that share an input array store it once
            inputs[field] = key
        manifest[name] = dict(image_size=IS, dtype=np.dtype(dt).name, inputs=inputs,
                              kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()})
        # The default-flags build is NOT kept as a golden output: with hipcc's default FMA contraction the reference's
        # ill-conditioned edge arithmetic lands on other roundings and a few hundred pixels per 256^2 frame move by up to
        # ~1 (and gradients overflow); only summary statistics of that build are recorded.
        st = {}
        for k in ('soft_colors', 'grad_faces', 'grad_textures'):
            a, b = arrays.pop('%s/fma/%s' % (name, k)).astype(np.float64), arrays['%s/nofma/%s' % (name, k)].astype(np.float64)
            d = np.abs(np.nan_to_num(a, posinf=1e30, neginf=-1e30) - b)
            st[k] = dict(max_abs=float(d.max()) if d.size else 0.0, over_1e_4=int((d > 1e-4).sum()),
                         scale=float(np.abs(b).max()) if b.size else 0.0)
        arrays.pop('%s/fma/aggrs_info' % name)
        manifest[name]['default_flags_build_vs_nofma'] = st
        print('%-34s IS=%3d F=%4d  default-flags build vs nofma: image %.2e (%d px), grad_faces %.2e of %.2e'
              % (name, IS, fv.shape[1], st['soft_colors']['max_abs'], st['soft_colors']['over_1e_4'],
                 st['grad_faces']['max_abs'], st['grad_faces']['scale']))
    arrays['manifest'] = np.frombuffer(json.dumps(manifest).encode(), np.uint8)
    path = os.path.join(outdir, 'sr_reference_kernels.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path), 'bytes;', torch.cuda.get_device_name(0))


if __name__ == '__main__':
    main()
