"""ctypes binding of liblasr_hip.so (the C ABI declared in include/lasr_sr.h).

There is deliberately NO fallback: if the library is missing or an entry point
fails, the caller gets an exception.  The CPU oracle under oracle/ is test
infrastructure and is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LASR_HIP_LIB') or os.path.join(_HERE, 'csrc', 'liblasr_hip.so')   # env override: kernel A/B builds

# every symbol include/lasr_sr.h and include/lasr_ops.h declare (checked by tests/test_abi.py)
_f, _i, _p, _sz = ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
_RASTER_SCALARS = [_i, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]
_RASTER_SCALARS_DEV = [_i, _i, _i, _i, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]
SIGNATURES = {
    'lasr_abi_version': (_i, []),
    'lasr_strerror': (ctypes.c_char_p, [_i]),
    'lasr_last_hip_error': (_i, []),
    'lasr_sr_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'lasr_sr_forward': (_i, [_p, _p, _p, _p, _p, _p, _sz] + _RASTER_SCALARS),
    'lasr_sr_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _sz] + _RASTER_SCALARS),
    # include/lasr_ops.h
    'lasr_lbs_forward': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_lbs_forward_both': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_lbs_backward_both': (_i, [_p] * 11 + [_i, _i, _i, _p]),
    'lasr_lbs_backward_scratch_floats': (_sz, [_i, _i, _i]),
    'lasr_lbs_backward': (_i, [_p] * 10 + [_i, _i, _i, _i, _p]),
    'lasr_project_points_forward': (_i, [_p] * 7 + [_i, _i, _i, _p]),
    'lasr_project_points_backward': (_i, [_p] * 8 + [_i, _i, _i, _p]),
    'lasr_pose_chain_forward': (_i, [_p, _i] + [_p] * 17 + [_i, _i, _i, _f, _p]),
    'lasr_pose_chain_backward': (_i, [_p, _i] + [_p] * 22 + [_i, _i, _i, _p]),
    'lasr_pinhole_forward': (_i, [_p, _p, _p, _p, _i, _i, _p]),
    'lasr_pinhole_backward': (_i, [_p] * 7 + [_i, _i, _p]),
    'lasr_loss_scratch_floats': (_sz, [_i, _i, _i]),
    'lasr_mask_loss_forward': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_mask_loss_backward': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_flow_loss_forward': (_i, [_p] * 8 + [_i, _i, _i, _i, _p]),
    'lasr_flow_loss_backward': (_i, [_p] * 8 + [_i, _i, _i, _i, _p]),
    'lasr_flow_loss_forward_vis': (_i, [_p] * 9 + [_i, _i, _i, _i, _p]),
    'lasr_tex_loss_forward': (_i, [_p] * 7 + [_f, _i, _i, _i, _p]),
    'lasr_tex_loss_backward': (_i, [_p] * 9 + [_f, _i, _i, _i, _p]),
    'lasr_arap_forward': (_i, [_p] * 5 + [_i, _i, _p]),
    'lasr_arap_backward': (_i, [_p] * 7 + [_i, _i, _p]),
    'lasr_laplacian_forward': (_i, [_p] * 4 + [_i, _i, _p]),
    'lasr_laplacian_backward': (_i, [_p] * 6 + [_i, _i, _p]),
    'lasr_mesh_regularisers_forward': (_i, [_p] * 12 + [_i, _i, _i, _i, _p]),
    'lasr_mesh_regularisers_backward': (_i, [_p] * 17 + [_i, _i, _i, _i, _p]),
    'lasr_step_regularisers_forward': (_i, [_p] * 12 + [_i, _i, _i, _i] + [_p] * 5 + [_i, _i, _i, _p]),
    'lasr_step_regularisers_backward': (_i, [_p] * 17 + [_i, _i, _i, _i] + [_p] * 7 + [_i, _i, _i, _p]),
    'lasr_flow_reproject_scratch_floats': (_sz, [_i, _i]),
    'lasr_flow_reproject_forward': (_i, [_p] * 7 + [_i, _i, _p]),
    'lasr_flow_reproject_backward': (_i, [_p] * 7 + [_i, _i, _p]),
    'lasr_flow_reproject_planes_forward': (_i, [_p, ctypes.c_longlong] + [_p] * 6 + [_i, _i, _p]),
    'lasr_flow_reproject_planes_backward': (_i, [_p, ctypes.c_longlong] + [_p] * 6 + [_i, _i, _p]),
    'lasr_quat_to_rotmat_forward': (_i, [_p, _p, _i, _p]),
    'lasr_quat_to_rotmat_backward': (_i, [_p, _p, _p, _i, _p]),
    'lasr_skin_weights_forward': (_i, [_p] * 5 + [_i, _i, _i, _p]),
    'lasr_skin_weights_backward': (_i, [_p] * 10 + [_i, _i, _i, _p]),
    'lasr_flatten_forward': (_i, [_p, _p, _p, _i, _i, _i, _p]),
    'lasr_flatten_backward': (_i, [_p] * 7 + [_i, _i, _i, _p]),
    'lasr_face_gather_forward': (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_face_gather_backward': (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_face_gather_backward_csr': (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
    'lasr_nearest_point': (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_point_mesh_scratch_floats': (_sz, [_i, _i, _i]),
    'lasr_point_mesh_forward': (_i, [_p] * 8 + [_i, _i, _i, _i, _p]),
    'lasr_point_mesh_backward': (_i, [_p] * 5 + [_f, _f, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_cosdist_scratch_floats': (_sz, [_i, _i]),
    'lasr_cosdist_forward': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_cosdist_backward': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_cosdist_multi_scratch_floats': (_sz, [_p, _i, _i]),
    'lasr_cosdist_multi_forward': (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _i, _p]),
    'lasr_cosdist_multi_backward': (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _i, _p]),
    'lasr_load_textures': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    # lasr_amd/csrc/glue.hip
    'lasr_geodesic_forward': (_i, [_p, _p, _p, _i, _p]),
    'lasr_geodesic_backward': (_i, [_p, _p, _p, _p, _p, _i, _p]),
    'lasr_weighted_means_forward': (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    'lasr_weighted_means_backward': (_i, [_p, _p, _i, _p, _p, _p]),
    'lasr_intrinsics_forward': (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    'lasr_intrinsics_backward': (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_bone_fixup_forward': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_bone_fixup_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_bone_fixup_pair_forward': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_bone_fixup_pair_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_chamfer_forward': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_chamfer_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'lasr_fill_planes': (_i, [_p, _p, _i, _i, ctypes.c_longlong, _p]),
    'lasr_tail_chunk_elems': (_i, []),
    'lasr_tail_step': (_i, [_p, _p, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'lasr_obs_pair': (_i, [_p, _p, _p, _i, _i, _p]),
    'lasr_render_tables_scratch_floats': (_sz, [_i, _i, _i]),
    'lasr_render_tables_forward': (_i, [_p, _p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p, _f] + [_p] * 9 + [_i, _i, _i, _p]),
    'lasr_render_tables_forward_imgs': (_i, [_p, _p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p, _f] + [_p] * 9 + [_i, _i, _i, _p]),
    'lasr_render_tables_backward': (_i, [_p, _p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p, _f] + [_p] * 8 + [_i, _i, _i, _p]),
    'lasr_raster_inputs_forward': (_i, [_p] * 9 + [_i, _i, _p]),
    'lasr_raster_inputs_backward': (_i, [_p] * 8 + [_i, _i, _p]),
    'lasr_raster_faces_scratch_floats': (_sz, [_i, _i, _i]),
    'lasr_raster_faces_forward': (_i, [_p] * 6 + [_i] + [_p] * 4 + [_i, _i, _i, _p]),
    'lasr_raster_faces_backward': (_i, [_p] * 4 + [_i] + [_p] * 7 + [_i, _i, _i, _p]),
    'lasr_gather_rows': (_i, [_p, ctypes.c_longlong, _i, _p, _i, _i, _p, _p, _p, _p, _p]),
    'lasr_mean_shape_forward': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_mean_shape_backward': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'lasr_sr_forward_dev': (_i, [_p, _p, _p, _p, _p, _p, _sz] + _RASTER_SCALARS_DEV),
    'lasr_sr_backward_dev': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _sz] + _RASTER_SCALARS_DEV),
    'lasr_sr_forward_attr': (_i, [_p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]),
    'lasr_sr_backward_attr': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]),
    'lasr_sr_forward_ex': (_i, [_p] * 6 + [_sz] + [_i] * 5 + [_f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i, _p]),
    'lasr_sr_forward_bg': (_i, [_p] * 6 + [_sz] + [_i] * 5 + [_f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p, _i, _p]),
    'lasr_sr_backward_ex': (_i, [_p] * 8 + [_sz] + [_i] * 5 + [_f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _i, _p]),
    'lasr_sr_forward_opt': (_i, [_p] * 6 + [_sz] + [_i] * 5 + [_f, _f, _p, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p, _i, _p, _p]),
    'lasr_sr_workspace_bytes_f64': (_sz, [_i, _i]),
    'lasr_sr_forward_f64': (_i, [_p] * 6 + [_sz] + [_i] * 4 + [_f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]),
    'lasr_sr_backward_f64': (_i, [_p] * 9 + [_sz] + [_i] * 4 + [_f, _f, _f, _f, _i, _f, _f, _i, _i, _i, _i, _p]),
    'lasr_sr_peek_choice': (_i, [_p, _i, _i, ctypes.POINTER(ctypes.c_int), _p]),
    'lasr_selftest_div': (_i, [_p, _p, _p, _i, _p]),
    'lasr_selftest_div3': (_i, [_p, _p, _p, _i, _p]),
    'lasr_prof_enable': (_i, [_p, _i]),
    'lasr_prof_kernel_count': (_i, []),
    'lasr_prof_kernel_name': (ctypes.c_char_p, [_i]),
    'lasr_prof_collect': (_i, [_p, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
}

ABI_VERSION = 4                                     # LASR_ABI_VERSION of include/lasr_sr.h (tests/test_abi.py compares them)
# flags of the *_ex entry points (include/lasr_sr.h)
SR_DEFAULT_FLAGS, SR_RELAXED_MATH, SR_SEGMENTED, SR_RECORDS_VALID, SR_GRADS_OVERWRITE = -1, 1, 2, 4, 8
SR_PAIR_ONE_TEAM, SR_PAIR_TWO_TEAMS = 16, 32          # forward: teams of four waves per tile of the pair-walk kernel (default: by launch size)
MEANS_MAX_TERMS, TAIL_MAX_GROUPS = 24, 16          # LASR_MEANS_MAX_TERMS / LASR_TAIL_MAX_GROUPS of include/lasr_ops.h



class SrOptions(ctypes.Structure):
    """lasr_sr_options (include/lasr_sr.h): per-call kernel-choice thresholds of the forward pass and the size limit of the
    heaviest-first tile order; a negative field = default."""
    _fields_ = [('coop8_max_tiles', ctypes.c_longlong), ('coop_max_tiles', ctypes.c_longlong), ('choose_max_tiles', ctypes.c_longlong),
                ('order_max_tiles', ctypes.c_longlong), ('pair_min_tiles', ctypes.c_longlong)]


_lib = None


class LasrNativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LasrNativeError(
                'liblasr_hip.so is not built (%s). Run `python -m lasr_amd.build` '
                '(hipcc --offload-arch=gfx950); there is no CPU fallback.' % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; load it FIRST so that this library
        # binds to the same HIP runtime instance (stream handles are only valid
        # inside the runtime that created them).
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)       # AttributeError here == ABI mismatch: let it propagate
            fn.restype = res
            fn.argtypes = args
        if h.lasr_abi_version() != ABI_VERSION:
            raise LasrNativeError('%s speaks ABI version %d, this package expects %d (LASR_ABI_VERSION of include/lasr_sr.h): rebuild '
                                  'with `python -m lasr_amd.build`' % (LIB_PATH, h.lasr_abi_version(), ABI_VERSION))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        h = lib()
        msg = h.lasr_strerror(rc).decode()
        if rc == -4:
            msg += ' (hipError_t %d)' % h.lasr_last_hip_error()
        raise LasrNativeError('%s failed: %s' % (what, msg))


def stream_of(t):
    """(device-guard context, raw hipStream_t) for the tensor's device: kernels go on torch's current stream."""
    import torch
    return torch.cuda.device(t.device), torch.cuda.current_stream(t.device).cuda_stream


def need_cuda(*tensors):
    for t in tensors:
        if t is not None and t.device.type != 'cuda':
            raise TypeError('lasr_amd kernels support only cuda (HIP) tensors; there is no CPU fallback')
