"""Rows f3 / f4 either side of the path against outputs of the reference's own kernels.

tests/golden/side_reference_kernels.npz: load_textures_cuda_kernel.cu:8-66 and chamfer3D.cu:12-134 built for gfx950 by
oracle/build_ref.py (device code unmodified) and run on an MI355X by oracle/gen_ref_vectors_side.py.
CPU part: the numpy restatement the product is tested against elsewhere (oracle/path_oracle.py:load_textures) reproduces
the reference's texels bit for bit; a brute-force nearest-neighbour search reproduces chamfer3D's indices.
GPU part: lasr_load_textures / lasr_nearest_point through the C ABI against the same vectors.
"""
import os

import numpy as np
import pytest
import torch

from oracle import path_oracle as po

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'side_reference_kernels.npz'))
N_LT = len({k.split('/')[1] for k in Z.files if k.startswith('load_textures/')})
N_CH = len({k.split('/')[1] for k in Z.files if k.startswith('chamfer/')})


@pytest.mark.parametrize('k', range(N_LT))
def test_load_textures_restatement_equals_the_reference_kernel(k):
    g = lambda n: Z['load_textures/%d/%s' % (k, n)]                       # noqa: E731
    R = int(round(np.sqrt(g('textures').shape[1])))
    upd = g('is_update').astype(bool)
    mine = po.load_textures(g('image'), g('faces_uv'), R, g('is_update'))
    assert np.array_equal(mine[upd], g('textures')[upd])
    assert (g('textures')[~upd] == 0.25).all()                             # the reference leaves those slices untouched


@pytest.mark.parametrize('k', range(N_CH))
def test_brute_force_nearest_neighbour_equals_chamfer3d(k):
    g = lambda n: Z['chamfer/%d/%s' % (k, n)]                             # noqa: E731
    a, b = g('xyz1').astype(np.float64), g('xyz2').astype(np.float64)
    d = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    assert np.array_equal(d.argmin(2), g('idx1')) and np.array_equal(d.argmin(1), g('idx2'))
    assert np.abs(d.min(2) - g('dist1')).max() <= 2e-6 and np.abs(d.min(1) - g('dist2')).max() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('k', range(N_LT))
def test_hip_load_textures_equals_the_reference_kernel(cuda, k):
    from lasr_amd.soft_renderer import functional as srf
    g = lambda n: Z['load_textures/%d/%s' % (k, n)]                       # noqa: E731
    R = int(round(np.sqrt(g('textures').shape[1])))
    upd = g('is_update').astype(bool)
    got = srf.load_textures(torch.from_numpy(g('image')).to(cuda), torch.from_numpy(g('faces_uv')).to(cuda), R,
                            torch.from_numpy(g('is_update'))).cpu().numpy()
    assert np.abs(got[upd] - g('textures')[upd]).max() <= 1e-6
    assert (got[~upd] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('k', range(N_CH))
def test_hip_nearest_point_equals_chamfer3d(cuda, k):
    from lasr_amd.nnutils import fused_ops
    g = lambda n: Z['chamfer/%d/%s' % (k, n)]                             # noqa: E731
    a, b = torch.from_numpy(g('xyz1')).to(cuda), torch.from_numpy(g('xyz2')).to(cuda)
    d1, i1 = fused_ops.nearest_point(a, b)
    d2, i2 = fused_ops.nearest_point(b, a)
    assert np.array_equal(i1.cpu().numpy(), g('idx1')) and np.array_equal(i2.cpu().numpy(), g('idx2'))
    assert np.abs(d1.cpu().numpy() - g('dist1')).max() <= 2e-6 and np.abs(d2.cpu().numpy() - g('dist2')).max() <= 2e-6
