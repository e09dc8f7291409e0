"""ctypes front-end of the CPU oracle (oracle/sr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under lasr_amd/ may import this module.

The call signatures mirror the reference autograd Function
(/root/reference/third_party/softras/soft_renderer/functional/soft_rasterize.py:9-102):
mode strings are mapped to ids exactly as :22-25 does, `dist_eps` is turned into
the logit log(1/dist_eps - 1) as :35 does, soft_colors is pre-filled with the
background colour and alpha = 1 as :50-53 does, faces_info is pre-zeroed (:47).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DIST = {'hard': 0, 'barycentric': 1, 'euclidean': 2}
RGB = {'hard': 0, 'softmax': 1}
ALPHA = {'hard': 0, 'sum': 1, 'prod': 2}
TEX = {'surface': 0, 'vertex': 1}


def build(force=False):
    so = os.path.join(_HERE, 'libsr_oracle.so')
    srcs = [os.path.join(_HERE, n) for n in ('sr_oracle.c', 'sr_oracle_body.inc')]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libsr_oracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb, aggr_func_alpha,
             texture_type, fill_back):
    f = ctypes.c_float
    return (f(float(near)), f(float(far)), f(float(eps)), f(float(sigma_val)), ctypes.c_int(DIST[dist_func]),
            f(float(np.log(1. / dist_eps - 1.))), f(float(gamma_val)), ctypes.c_int(RGB[aggr_func_rgb]),
            ctypes.c_int(ALPHA[aggr_func_alpha]), ctypes.c_int(TEX[texture_type]), ctypes.c_int(1 if fill_back else 0))


def forward(face_vertices, textures, image_size=256, background_color=(0, 0, 0), near=1, far=100,
            fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
            gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
            dtype=np.float32):
    """Returns dict(soft_colors [N,4,IS,IS], aggrs_info [N,2,IS,IS], faces_info [N,F,27])."""
    dt = np.dtype(dtype)
    fv = np.ascontiguousarray(face_vertices, dtype=dt).reshape(face_vertices.shape[0], -1, 9)
    N, F = fv.shape[:2]
    tx = np.ascontiguousarray(textures, dtype=dt).reshape(N, F, -1, 3)
    T = tx.shape[2]
    IS = int(image_size)
    infos = np.zeros((N, F, 27), dt)
    aggrs = np.zeros((N, 2, IS, IS), dt)
    colors = np.ones((N, 4, IS, IS), dt)
    for k in range(3):
        colors[:, k] *= dt.type(background_color[k])
    fn = lib().oracle_sr_forward_f32 if dt == np.float32 else lib().oracle_sr_forward_f64
    rc = fn(_ptr(fv), _ptr(tx), _ptr(infos), _ptr(aggrs), _ptr(colors),
            ctypes.c_int(N), ctypes.c_int(F), ctypes.c_int(T), ctypes.c_int(IS),
            *_scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb,
                      aggr_func_alpha, texture_type, fill_back))
    if rc != 0:
        raise RuntimeError('oracle_sr_forward failed: %d' % rc)
    return dict(soft_colors=colors, aggrs_info=aggrs, faces_info=infos, face_vertices=fv, textures=tx)


def backward(saved, grad_soft_colors, image_size=256, background_color=(0, 0, 0), near=1, far=100,
             fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
             gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface',
             dtype=np.float32):
    """`saved` is forward()'s return value.  Returns (grad_faces [N,F,3,3], grad_textures [N,F,T,3])."""
    dt = np.dtype(dtype)
    fv, tx = saved['face_vertices'], saved['textures']
    N, F = fv.shape[:2]
    T = tx.shape[2]
    IS = int(image_size)
    g = np.ascontiguousarray(grad_soft_colors, dtype=dt)
    assert g.shape == (N, 4, IS, IS)
    gf = np.zeros((N, F, 3, 3), dt)
    gt = np.zeros((N, F, T, 3), dt)
    fn = lib().oracle_sr_backward_f32 if dt == np.float32 else lib().oracle_sr_backward_f64
    rc = fn(_ptr(fv), _ptr(tx), _ptr(saved['soft_colors']), _ptr(saved['faces_info']), _ptr(saved['aggrs_info']),
            _ptr(gf), _ptr(gt), _ptr(g),
            ctypes.c_int(N), ctypes.c_int(F), ctypes.c_int(T), ctypes.c_int(IS),
            *_scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb,
                      aggr_func_alpha, texture_type, fill_back))
    if rc != 0:
        raise RuntimeError('oracle_sr_backward failed: %d' % rc)
    return gf, gt


def set_threads(n):
    """OpenMP thread count of the oracle (bench.py's cpu_baseline leg); returns the count now in force."""
    lib().oracle_sr_set_threads(ctypes.c_int(int(n)))
    return int(lib().oracle_sr_max_threads())


def backward_banded(saved, grad_soft_colors, bands, image_size=256, background_color=(0, 0, 0), near=1, far=100,
                    fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
                    gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='surface'):
    """fp32 backward with (image, row band) parallelism: per-band gradient slabs folded in band order (deterministic;
    same arithmetic per pixel-face pair as backward()).  Used by bench.py's cpu_baseline leg so that the backward keeps
    every host core busy, not one core per image."""
    dt = np.dtype(np.float32)
    fv, tx = saved['face_vertices'], saved['textures']
    N, F = fv.shape[:2]
    T = tx.shape[2]
    IS = int(image_size)
    g = np.ascontiguousarray(grad_soft_colors, dtype=dt)
    assert g.shape == (N, 4, IS, IS)
    gf = np.zeros((N, F, 3, 3), dt)
    gt = np.zeros((N, F, T, 3), dt)
    slabs = np.zeros((int(bands), gf.size + gt.size), dt)
    rc = lib().oracle_sr_backward_banded_f32(
        _ptr(fv), _ptr(tx), _ptr(saved['soft_colors']), _ptr(saved['faces_info']), _ptr(saved['aggrs_info']),
        _ptr(gf), _ptr(gt), _ptr(g), _ptr(slabs), ctypes.c_int(int(bands)),
        ctypes.c_int(N), ctypes.c_int(F), ctypes.c_int(T), ctypes.c_int(IS),
        *_scalars(near, far, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb,
                  aggr_func_alpha, texture_type, fill_back))
    if rc != 0:
        raise RuntimeError('oracle_sr_backward_banded failed: %d' % rc)
    return gf, gt
