"""Reader of tests/golden/sr_reference_kernels.npz: outputs of the REFERENCE soft-rasteriser kernels
(third_party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu:245-668 behind soft_rasterize_cuda.cpp:59-138),
built by oracle/build_ref.py with -ffp-contract=off and run on an MI355X by oracle/gen_ref_vectors.py."""
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sr_reference_kernels.npz')
_Z = None


def _z():
    global _Z
    if _Z is None:
        _Z = np.load(PATH)
    return _Z


def manifest():
    return json.loads(bytes(_z()['manifest']).decode())


def names():
    return sorted(manifest())


def case(name):
    """dict(face_vertices, textures, grad_soft_colors, image_size, dtype, kwargs, soft_colors, aggrs_info, faces_info,
    grad_faces, grad_textures)"""
    m = manifest()[name]
    kw = dict(m['kwargs'])
    kw['background_color'] = tuple(kw['background_color'])
    out = dict(image_size=m['image_size'], dtype=np.dtype(m['dtype']), kwargs=kw)
    for field, key in m['inputs'].items():
        out[field] = _z()[key]
    for k in ('soft_colors', 'aggrs_info', 'faces_info', 'grad_faces', 'grad_textures'):
        out[k] = _z()['%s/nofma/%s' % (name, k)]
    return out
