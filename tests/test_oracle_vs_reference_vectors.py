"""The CPU oracle (oracle/sr_oracle.c) against outputs of the REFERENCE kernels themselves.

tests/golden/sr_reference_kernels.npz was produced by the reference's own soft_rasterize_cuda_kernel.cu, built for
gfx950 by oracle/build_ref.py (torch's hipify + hipcc, -ffp-contract=off; the device code is unmodified) and run on
an MI355X by oracle/gen_ref_vectors.py: 38 calls covering all 18 mode combinations of soft_rasterize.py:22-25,
surface and vertex textures, single-sided faces, the near-plane quirk, sigma 1e-5 / 1e-12, ragged image sizes, fp64,
screen-filling faces, and the meshes BASELINE names (M1 at 128^2, M2 at 256^2, soft and hard).

This is the pin of SURVEY section 8(c): the oracle is checked against the reference run here, not against itself.
Bars: faces_info (face setup, K.cu:245-305) and the hard-mode face-index map / z-buffer bit for bit; image within
1e-6 (only expf differs: device vs glibc, <= 1 ulp); gradients within 1e-5 of the largest entry (the reference
accumulates with float atomics in launch order, the oracle in pixel order).
"""
import numpy as np
import pytest

from tests import refvec


@pytest.mark.parametrize('name', refvec.names())
def test_oracle_reproduces_reference_kernel_outputs(oracle, name):
    c = refvec.case(name)
    dt, IS, kw = c['dtype'], c['image_size'], c['kwargs']
    ref = oracle.forward(c['face_vertices'], c['textures'], IS, dtype=dt, **kw)
    assert np.array_equal(ref['faces_info'], c['faces_info'].reshape(ref['faces_info'].shape)), 'face setup differs'
    tol = 1e-6 if dt == np.float32 else 1e-14
    assert np.abs(ref['soft_colors'] - c['soft_colors']).max() <= tol
    if kw['aggr_func_rgb'] == 'hard':
        assert np.array_equal(ref['aggrs_info'], c['aggrs_info']), 'hard-mode z-buffer / face-index map differs'
    else:
        assert np.array_equal(ref['aggrs_info'][:, 1] > 0, c['aggrs_info'][:, 1] > 0)
        assert np.abs(ref['aggrs_info'][:, 1] - c['aggrs_info'][:, 1]).max() <= tol          # softmax maximum
        np.testing.assert_allclose(ref['aggrs_info'][:, 0], c['aggrs_info'][:, 0], rtol=5e-6 if dt == np.float32 else 1e-13)
    gf, gt = oracle.backward(ref, c['grad_soft_colors'], IS, dtype=dt, **kw)
    for mine, theirs in ((gf, c['grad_faces']), (gt, c['grad_textures'])):
        scale = max(float(np.abs(theirs).max()), 1e-30)
        assert np.abs(mine.reshape(theirs.shape) - theirs).max() <= (1e-5 if dt == np.float32 else 1e-12) * scale


def test_fixture_covers_every_mode_and_the_baseline_meshes():
    m = refvec.manifest()
    seen = {(v['kwargs']['dist_func'], v['kwargs']['aggr_func_rgb'], v['kwargs']['aggr_func_alpha']) for v in m.values()}
    assert len(seen) == 18
    assert {v['kwargs']['texture_type'] for v in m.values()} == {'surface', 'vertex'}
    sizes = {(v['image_size'], refvec.case(k)['face_vertices'].shape[1]) for k, v in m.items()}
    assert (256, 2420) in sizes and (128, 1280) in sizes
