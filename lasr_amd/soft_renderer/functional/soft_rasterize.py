"""Soft rasterisation operator backed by the gfx950 HIP kernels.

Mirror of the reference operator
  /root/reference/third_party/softras/soft_renderer/functional/soft_rasterize.py:9-119
(same positional/keyword arguments, same mode strings, same error for CPU
tensors) with the native call going through the C ABI of include/lasr_sr.h
instead of the `soft_renderer.cuda.soft_rasterize` pybind module.
"""
import ctypes
import math
import os

import torch
from torch.autograd import Function

from ... import _lib
from .cameras import const_tensor

_DIST = {'hard': 0, 'barycentric': 1, 'euclidean': 2}
_RGB = {'hard': 0, 'softmax': 1}
_ALPHA = {'hard': 0, 'sum': 1, 'prod': 2}
_TEX = {'surface': 0, 'vertex': 1}

_workspaces = {}
_forward_flags = _lib.SR_DEFAULT_FLAGS


_launch_options = None


def set_forward_flags(flags):
    """Forward-kernel flags the autograd operator passes per call: 0 / _lib.SR_DEFAULT_FLAGS (default arithmetic) or
    _lib.SR_RELAXED_MATH (include/lasr_sr.h).  This is the CALLER's setting (a module variable of this Python operator); the
    native library itself keeps no state.  Returns the previous value."""
    global _forward_flags
    old, _forward_flags = _forward_flags, int(flags)
    return old


def set_launch_thresholds(coop8_max_tiles=-1, coop_max_tiles=-1, choose_max_tiles=-1, order_max_tiles=-1, pair_min_tiles=-1):
    """Launch options this operator passes with every forward call (lasr_sr_options; negative = library default, no argument =
    all defaults): the kernel-choice thresholds and the size limit of the heaviest-first tile order (0 = the fixed centre-out
    order), and the launch size from which the pair-walk kernel takes over (pair_min_tiles; 0 = always, huge = never).  The output of
    the other kernels is bit-identical whichever runs in whichever order; the pair-walk kernel's agrees with theirs to ~1e-6
    (another accumulation order per pixel).  Tests force each one through here."""
    global _launch_options
    if coop8_max_tiles < 0 and coop_max_tiles < 0 and choose_max_tiles < 0 and order_max_tiles < 0 and pair_min_tiles < 0:
        _launch_options = None
    else:
        _launch_options = _lib.SrOptions(int(coop8_max_tiles), int(coop_max_tiles), int(choose_max_tiles), int(order_max_tiles),
                                         int(pair_min_tiles))


def _options_ref():
    return ctypes.byref(_launch_options) if _launch_options is not None else None


def forward_flags():
    return _forward_flags


def _texels(textures):
    """T of a [N,F,T,3] (or flattened [N,F,T*3]) texture tensor; well defined for F == 0 too."""
    n = 1
    for d in textures.shape[2:]:
        n *= int(d)
    return max(n // 3, 1)


# Whose per-face records does a workspace hold?  Every call that writes records into it (a forward, or a backward that rebuilds
# them) takes the next number; a backward may skip its setup launch (LASR_SR_RECORDS_VALID) exactly when the number its forward
# took is still the current one.  Measured on an MI355X, mesh M2 at 256x256 (profiles/r04_flag_sweep.txt), step ms with rebuilt ->
# reused records: 64 frames 0.961 -> 0.952, 256 frames 3.353 -> 3.343 (the forward's records are colder in L2 / Infinity Cache when
# the face-major backward reads them: +19 us, the second setup launch: 33 us); in round 2 the sign at 256 frames was the other way
# (profiles/r02e_records_reuse.txt) and launches above 200k faces rebuilt.  The limit stays as a knob.
REUSE_RECORDS_MAX_FACES = int(os.environ.get('LASR_SR_REUSE_RECORDS_MAX_FACES', 1 << 40))   # 0 = the backward always rebuilds
_records_of = {}


def _workspace(device, stream, nbytes):
    key = (device.index, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
        _records_of[key] = _records_of.get(key, 0) + 1           # whatever records the old buffer held are gone
    return ws


def invalidate_records(device=None, stream=None):
    """Tell the operator that something OUTSIDE its eager calls wrote face records into the raster workspace of (device, stream)
    -- a HIP-graph replay of captured forward / backward calls, or direct C-ABI calls on the same buffer: pending eager backward
    passes then rebuild their records instead of trusting the forward's (LASR_SR_RECORDS_VALID).  No arguments: every
    workspace.  LASRTrainer calls this after each graph replay; other users of the workspace must do the same."""
    index = None
    if device is not None:
        d = torch.device(device)
        if d.type != 'cuda':
            raise ValueError('invalidate_records: %r is not a HIP device' % (device,))
        index = d.index if d.index is not None else torch.cuda.current_device()      # 'cuda' = the current device
    for key in list(_records_of):
        if (device is None or key[0] == index) and (stream is None or key[1] == stream):
            _records_of[key] = _records_of.get(key, 0) + 1


def _new_records(device, stream):
    key = (device.index, stream)
    _records_of[key] = _records_of.get(key, 0) + 1
    return key, _records_of[key]


def _as_float(v):
    return float(v.item()) if torch.is_tensor(v) else float(v)


def _near_far_dev(near, far, dev):
    """LASR stores 0-dim DEVICE tensors in rasterizer.near/far (mesh_net.py:306-311).  Returns a 2-float device tensor
    for the *_dev entry points (no host sync), or None when both are plain numbers."""
    if (torch.is_tensor(near) and torch.is_tensor(far) and near.device.type == 'cuda' and near._base is not None
            and near._base is far._base and near._base.shape == (2,) and near._base.dtype == torch.float32
            and near.data_ptr() == near._base.data_ptr() and far.data_ptr() == near.data_ptr() + 4):
        return near._base.detach()                       # already {near, far} side by side (fused_ops.raster_inputs): no launch
    if torch.is_tensor(near) and near.device.type == 'cuda' or torch.is_tensor(far) and far.device.type == 'cuda':
        return torch.stack([torch.as_tensor(near, dtype=torch.float32, device=dev).reshape(()),
                            torch.as_tensor(far, dtype=torch.float32, device=dev).reshape(())]).detach()
    return None


class SoftRasterizeFunction(Function):
    """face_vertices [N,F,3,3], textures [N,F,T,3] (T = 3 vertex colours or R*R surface texels) -> [N,4,IS,IS].

    Extension over the reference: vertex textures may carry 6 or 9 channels, [N,F,3,6] -> [N,7,IS,IS] (alpha last): two
    or three attribute triples depth-blended in ONE pass over the geometry (lasr_sr_*_attr), equal to separate 3-channel
    renders; `background_color` may then hold one value per channel.
    """

    @staticmethod
    def forward(ctx, face_vertices, textures, image_size=256,
                background_color=[0, 0, 0], near=1, far=100,
                fill_back=True, eps=1e-3,
                sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
                gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod',
                texture_type='surface'):
        if face_vertices.dtype == torch.float64 and textures.dtype == torch.float64:
            return _forward_f64(ctx, face_vertices, textures, image_size, background_color, near, far, fill_back, eps, sigma_val,
                                dist_func, dist_eps, gamma_val, aggr_func_rgb, aggr_func_alpha, texture_type)
        if face_vertices.dtype != torch.float32 or textures.dtype != torch.float32:
            raise TypeError('lasr_amd soft_rasterize takes float32 tensors (or float64 for both, as the reference dispatches)')
        dev = face_vertices.device
        N, F = face_vertices.shape[:2]
        fv = face_vertices.detach().reshape(N, F, 9).contiguous()
        C = int(textures.shape[-1]) if textures.ndimension() == 4 else 3
        if C == 3:
            T = _texels(textures)
            tx = textures.detach().reshape(N, F, T, 3).contiguous()
        elif C in (6, 9) and texture_type == 'vertex' and textures.shape[2] == 3:
            T = 3
            tx = textures.detach().contiguous()
        else:
            raise ValueError('textures must be [N,F,T,3], or [N,F,3,6] / [N,F,3,9] vertex attributes')
        IS = int(image_size)
        nf = _near_far_dev(near, far, dev)
        tail = (float(eps), float(sigma_val), _DIST[dist_func], float(math.log(1. / dist_eps - 1.)), float(gamma_val),
                _RGB[aggr_func_rgb], _ALPHA[aggr_func_alpha], _TEX[texture_type], 1 if fill_back else 0)
        ctx.geom, ctx.nf, ctx.tail, ctx.C = (N, F, T, IS), nf, tail, C
        ctx.near_far = (0.0, 0.0) if nf is not None else (_as_float(near), _as_float(far))
        ctx.in_shapes = (face_vertices.shape, textures.shape)

        aggrs_info = torch.empty(N, 2, IS, IS, dtype=torch.float32, device=dev)
        # background per channel: 3 values repeat for every attribute triple, or one value per channel; alpha plane starts at 1
        nb = len(background_color)
        bg = [float(background_color[k if nb == C else k % 3]) for k in range(C)] + [1.0]
        # no pre-fill: lasr_sr_forward_bg takes the background as an argument and writes every element of soft_colors
        soft_colors = torch.empty(N, C + 1, IS, IS, dtype=torch.float32, device=dev)

        h = _lib.lib()
        # One scratch buffer per (device, stream) serves every call; large launches rebuild the per-face records in the backward
        # (29 us for 620k faces), small ones reuse the forward's when nothing else has written into the buffer since (_records_of).
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = _workspace(dev, stream, h.lasr_sr_workspace_bytes(N, F, T, IS))
            rc = h.lasr_sr_forward_opt(fv.data_ptr(), tx.data_ptr(), None, aggrs_info.data_ptr(), soft_colors.data_ptr(),
                                       ws.data_ptr(), ws.numel(), N, F, T, C, IS, *ctx.near_far,
                                       nf.data_ptr() if nf is not None else None, *tail, (ctypes.c_float * C)(*bg[:C]),
                                       forward_flags(), _options_ref(), stream)
        _lib.check(rc, 'lasr_sr_forward')
        ctx.records = _new_records(dev, stream)
        ctx.save_for_backward(fv, tx, soft_colors, aggrs_info)
        ctx.mark_non_differentiable(aggrs_info)
        return soft_colors

    @staticmethod
    def backward(ctx, grad_soft_colors):
        if getattr(ctx, 'f64', False):
            return _backward_f64(ctx, grad_soft_colors)
        fv, tx, soft_colors, aggrs_info = ctx.saved_tensors
        N, F, T, IS = ctx.geom
        C, nf, tail = ctx.C, ctx.nf, ctx.tail
        dev = fv.device
        # vertex attributes: every gradient element is stored by the wavefront that owns its face (LASR_SR_GRADS_OVERWRITE), no
        # zero fill needed; surface texels are credited with atomics and need zeroed buffers like the reference's (:88-89)
        vertex = tail[7] == _TEX['vertex']
        grads = (torch.empty if vertex else torch.zeros)(N * F * 9 + tx.numel(), dtype=torch.float32, device=dev)
        grad_faces, grad_textures = grads[:N * F * 9].view(N, F, 9), grads[N * F * 9:].view(tx.shape)
        if vertex and (N == 0 or F == 0 or IS == 0):
            grads.zero_()                                                   # nothing is launched for an empty problem
        g = grad_soft_colors.contiguous().float()
        h = _lib.lib()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = _workspace(dev, stream, h.lasr_sr_workspace_bytes(N, F, T, IS))
            key = (dev.index, stream)
            reuse = N * F <= REUSE_RECORDS_MAX_FACES and ctx.records == (key, _records_of.get(key))
            if not reuse:
                _new_records(dev, stream)                                   # this call's setup launch overwrites the workspace
            rc = h.lasr_sr_backward_ex(fv.data_ptr(), tx.data_ptr(), soft_colors.data_ptr(), aggrs_info.data_ptr(),
                                       grad_faces.data_ptr(), grad_textures.data_ptr(), g.data_ptr(), ws.data_ptr(),
                                       ws.numel(), N, F, T, C, IS, *ctx.near_far,
                                       nf.data_ptr() if nf is not None else None, *tail,
                                       (_lib.SR_GRADS_OVERWRITE if vertex else 0) | (_lib.SR_RECORDS_VALID if reuse else 0), stream)
        _lib.check(rc, 'lasr_sr_backward')
        fshape, tshape = ctx.in_shapes
        return (grad_faces.reshape(fshape), grad_textures.reshape(tshape),
                None, None, None, None, None, None, None, None, None, None, None, None, None)


def _forward_f64(ctx, face_vertices, textures, image_size, background_color, near, far, fill_back, eps, sigma_val, dist_func,
                 dist_eps, gamma_val, aggr_func_rgb, aggr_func_alpha, texture_type):
    """float64 tensors (the reference dispatches on the tensor type, soft_rasterize_cuda_kernel.cu:701,716,780): the
    brute-force double-precision path of csrc/sr_fp64.hip, same call convention as the reference's Function -- faces_info zeroed,
    soft_colors = background with alpha 1 (soft_rasterize.py:47-53), gradients accumulated into zeroed buffers (:88-89)."""
    dev = face_vertices.device
    N, F = face_vertices.shape[:2]
    fv = face_vertices.detach().reshape(N, F, 9).contiguous()
    T = _texels(textures)
    tx = textures.detach().reshape(N, F, T, 3).contiguous()
    IS = int(image_size)
    scalars = (_as_float(near), _as_float(far), float(eps), float(sigma_val), _DIST[dist_func], float(math.log(1. / dist_eps - 1.)),
               float(gamma_val), _RGB[aggr_func_rgb], _ALPHA[aggr_func_alpha], _TEX[texture_type], 1 if fill_back else 0)
    faces_info = torch.zeros(N, F, 27, dtype=torch.float64, device=dev)
    aggrs_info = torch.zeros(N, 2, IS, IS, dtype=torch.float64, device=dev)
    soft_colors = torch.ones(N, 4, IS, IS, dtype=torch.float64, device=dev)
    for k in range(3):
        soft_colors[:, k] = float(background_color[k])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.lib().lasr_sr_forward_f64(fv.data_ptr(), tx.data_ptr(), faces_info.data_ptr(), aggrs_info.data_ptr(),
                                            soft_colors.data_ptr(), None, 0, N, F, T, IS, *scalars, stream)
    _lib.check(rc, 'lasr_sr_forward_f64')
    ctx.f64, ctx.geom, ctx.scalars = True, (N, F, T, IS), scalars
    ctx.in_shapes = (face_vertices.shape, textures.shape)
    ctx.save_for_backward(fv, tx, soft_colors, aggrs_info, faces_info)
    ctx.mark_non_differentiable(aggrs_info)
    return soft_colors


def _backward_f64(ctx, grad_soft_colors):
    fv, tx, soft_colors, aggrs_info, faces_info = ctx.saved_tensors
    N, F, T, IS = ctx.geom
    grad_faces, grad_textures = torch.zeros_like(fv), torch.zeros_like(tx)
    g = grad_soft_colors.contiguous().double()
    with torch.cuda.device(fv.device):
        stream = torch.cuda.current_stream(fv.device).cuda_stream
        rc = _lib.lib().lasr_sr_backward_f64(fv.data_ptr(), tx.data_ptr(), soft_colors.data_ptr(), faces_info.data_ptr(),
                                             aggrs_info.data_ptr(), grad_faces.data_ptr(), grad_textures.data_ptr(), g.data_ptr(),
                                             None, 0, N, F, T, IS, *ctx.scalars, stream)
    _lib.check(rc, 'lasr_sr_backward_f64')
    fshape, tshape = ctx.in_shapes
    return (grad_faces.reshape(fshape), grad_textures.reshape(tshape)) + (None,) * 13


def soft_rasterize(face_vertices, textures, image_size=256,
                   background_color=[0, 0, 0], near=1, far=100,
                   fill_back=True, eps=1e-3,
                   sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
                   gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod',
                   texture_type='surface'):
    if face_vertices.device.type != 'cuda':
        raise TypeError('Rasterize module supports only cuda Tensors')
    return SoftRasterizeFunction.apply(face_vertices, textures, image_size,
                                       background_color, near, far,
                                       fill_back, eps,
                                       sigma_val, dist_func, dist_eps,
                                       gamma_val, aggr_func_rgb, aggr_func_alpha,
                                       texture_type)


def soft_rasterize_raw(face_vertices, textures, image_size, background_color, near, far, fill_back, eps,
                       sigma_val, dist_func, dist_eps, gamma_val, aggr_func_rgb, aggr_func_alpha, texture_type,
                       want_faces_info=False):
    """Forward only, returning (soft_colors, aggrs_info[, faces_info]) like the raw extension call
    (soft_rasterize_cuda.cpp:59-76); the autograd Function hides aggrs_info."""
    dev = face_vertices.device
    N, F = face_vertices.shape[:2]
    fv = face_vertices.detach().reshape(N, F, 9).contiguous()
    T, IS = _texels(textures), int(image_size)
    tx = textures.detach().reshape(N, F, T, 3).contiguous()
    aggrs_info = torch.empty(N, 2, IS, IS, dtype=torch.float32, device=dev)
    soft_colors = torch.ones(N, 4, IS, IS, dtype=torch.float32, device=dev)
    for k in range(3):
        soft_colors[:, k] = float(background_color[k])
    faces_info = torch.zeros(N, F, 27, dtype=torch.float32, device=dev) if want_faces_info else None
    scalars = (_as_float(near), _as_float(far), float(eps), float(sigma_val), _DIST[dist_func],
               float(math.log(1. / dist_eps - 1.)), float(gamma_val), _RGB[aggr_func_rgb],
               _ALPHA[aggr_func_alpha], _TEX[texture_type], 1 if fill_back else 0)
    h = _lib.lib()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _workspace(dev, stream, h.lasr_sr_workspace_bytes(N, F, T, IS))
        # (lasr_sr_forward is this call with flags 0 and no options; the raw path takes the operator's settings as well)
        rc = h.lasr_sr_forward_opt(fv.data_ptr(), tx.data_ptr(), faces_info.data_ptr() if want_faces_info else None,
                                   aggrs_info.data_ptr(), soft_colors.data_ptr(), ws.data_ptr(), ws.numel(),
                                   N, F, T, 3, IS, scalars[0], scalars[1], None, *scalars[2:], None, forward_flags(),
                                   _options_ref(), stream)
    _lib.check(rc, 'lasr_sr_forward')
    _new_records(dev, stream)
    return (soft_colors, aggrs_info, faces_info) if want_faces_info else (soft_colors, aggrs_info)
