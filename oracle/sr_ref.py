"""Loader of the reference soft-rasteriser extension built by oracle/build_ref.py (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: used by tests/ and oracle/gen_ref_vectors.py to obtain outputs of the REFERENCE kernels
(third_party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu:245-668, run on the MI355X through the entry
points of soft_rasterize_cuda.cpp:59-138).  The call convention below is that of the reference's autograd Function
(soft_renderer/functional/soft_rasterize.py:12-102): mode strings -> ids (:22-25), dist_eps -> logit (:35),
faces_info / aggrs_info zero-filled (:47-48), soft_colors = background with alpha 1 (:50-53), gradients zero-filled
(:88-89; the kernel accumulates with atomics).
"""
import importlib.machinery
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODS = {}
_SIDE = {}

DIST = {'hard': 0, 'barycentric': 1, 'euclidean': 2}
RGB = {'hard': 0, 'softmax': 1}
ALPHA = {'hard': 0, 'sum': 1, 'prod': 2}
TEX = {'surface': 0, 'vertex': 1}


def path(variant='sr_ref'):
    return os.path.join(_HERE, '_ref', variant + '.so')


def available(variant='sr_ref'):
    return os.path.exists(path(variant))


def module(variant='sr_ref'):
    """The pybind module (its init function is named after the reference's module, `soft_rasterize`)."""
    if variant not in _MODS:
        if _MODS:
            # both builds carry the same kernel and host symbol names; HIP's code-object registration of the second one
            # makes the first one's launches produce garbage (observed on the MI355X) -- one build per process
            raise RuntimeError('oracle.sr_ref: %s is already loaded in this process; load %s in another process'
                               % (next(iter(_MODS)), variant))
        import torch                                    # noqa: F401  (libtorch must be loaded first)
        loader = importlib.machinery.ExtensionFileLoader('soft_rasterize', path(variant))
        spec = importlib.util.spec_from_loader('soft_rasterize', loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        _MODS[variant] = mod
    return _MODS[variant]


def side_module(which):
    """The reference's side kernels (rows f3 / f4): 'load_textures' (load_textures_cuda.cpp: `load_textures(image, faces,
    textures, is_update)`) or 'chamfer_3D' (chamfer_cuda.cpp: `forward(xyz1, xyz2, dist1, dist2, idx1, idx2)`)."""
    so, init = {'load_textures': ('load_textures_ref', 'load_textures'), 'chamfer_3D': ('chamfer_3D_ref', 'chamfer_3D_ref')}[which]
    if so not in _SIDE:
        import torch                                    # noqa: F401
        loader = importlib.machinery.ExtensionFileLoader(init, path(so))
        mod = importlib.util.module_from_spec(importlib.util.spec_from_loader(init, loader))
        loader.exec_module(mod)
        _SIDE[so] = mod
    return _SIDE[so]


def _scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb, aggr_func_alpha,
             texture_type, fill_back):
    return (float(near), float(far), float(eps), float(sigma_val), DIST[dist_func],
            float(np.log(1. / dist_eps - 1.)), float(gamma_val), RGB[aggr_func_rgb], ALPHA[aggr_func_alpha],
            TEX[texture_type], bool(fill_back))


def forward(face_vertices, textures, image_size=256, background_color=(0, 0, 0), near=1, far=100,
            fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
            gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
            variant='sr_ref', dtype=None):
    """Device tensors in, dict of device tensors out (same keys as oracle.sr_oracle.forward)."""
    import torch
    dt = dtype or face_vertices.dtype
    N = face_vertices.shape[0]
    fv = face_vertices.detach().to(dt).reshape(N, -1, 3, 3).contiguous().clone()
    F = fv.shape[1]
    tx = textures.detach().to(dt).reshape(N, F, -1, 3).contiguous().clone()
    IS = int(image_size)
    infos = torch.zeros(N, F, 27, dtype=dt, device=fv.device)
    aggrs = torch.zeros(N, 2, IS, IS, dtype=dt, device=fv.device)
    colors = torch.ones(N, 4, IS, IS, dtype=dt, device=fv.device)
    for k in range(3):
        colors[:, k] *= background_color[k]
    # the reference launches on the legacy default stream (K.cu:702,717: no stream argument) whatever torch's current stream
    # is: fence both sides so that a caller on a side stream (the trainer's graph-capture stream) sees ordered results
    torch.cuda.synchronize()
    module(variant).forward_soft_rasterize(fv, tx, infos, aggrs, colors, IS,
                                           *_scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                                     aggr_func_rgb, aggr_func_alpha, texture_type, fill_back))
    torch.cuda.synchronize()
    return dict(soft_colors=colors, aggrs_info=aggrs, faces_info=infos, face_vertices=fv, textures=tx)


def backward(saved, grad_soft_colors, image_size=256, background_color=(0, 0, 0), near=1, far=100,
             fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
             gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
             variant='sr_ref'):
    import torch
    fv, tx = saved['face_vertices'], saved['textures']
    g = grad_soft_colors.detach().to(fv.dtype).contiguous()
    gf = torch.zeros_like(fv)
    gt = torch.zeros_like(tx)
    torch.cuda.synchronize()
    module(variant).backward_soft_rasterize(fv, tx, saved['soft_colors'], saved['faces_info'], saved['aggrs_info'],
                                            gf, gt, g, int(image_size),
                                            *_scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                                      aggr_func_rgb, aggr_func_alpha, texture_type, fill_back))
    torch.cuda.synchronize()
    return gf, gt
