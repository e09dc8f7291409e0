"""float64 tensors through the operator (csrc/sr_fp64.hip): the reference dispatches its kernels on the tensor type
(AT_DISPATCH_FLOATING_TYPES, soft_rasterize_cuda_kernel.cu:701,716,780).  Checked against
  * the fp64 golden vector of tests/golden/sr_reference_kernels.npz (the reference's own kernels run in double on an MI355X),
  * the C oracle's double instantiation for all 18 mode combinations, both texture types, ragged sizes,
  * the reference build live, in double (oracle/_ref/sr_ref_nofma.so, when the snapshot carries it).
Bars: hard-mode face-index map / z-buffer equal; image max-abs <= 1e-9 (double arithmetic in the same order; the library exp
differs in the last bits); gradients within 1e-9 of the largest entry (atomics order)."""
import itertools

import numpy as np
import pytest
import torch

from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf
from oracle import sr_ref
from tests import refvec

pytestmark = pytest.mark.gpu


def run64(dev, fv, ft, IS, kw, g):
    a = torch.from_numpy(fv.astype(np.float64)).to(dev).requires_grad_(True)
    b = torch.from_numpy(ft.astype(np.float64)).to(dev).requires_grad_(True)
    img = srf.soft_rasterize(a, b, IS, **kw)
    assert img.dtype == torch.float64
    img.backward(torch.from_numpy(g.astype(np.float64)).to(dev))
    return img.detach().cpu().numpy(), a.grad.cpu().numpy(), b.grad.cpu().numpy()


def close(mine, theirs, rel):
    scale = max(float(np.abs(theirs).max()), 1e-300)
    return float(np.abs(mine.reshape(theirs.shape) - theirs).max()) <= rel * scale


def test_the_fp64_golden_vector_of_the_reference_kernels(cuda):
    c = refvec.case('fp64')
    assert c['dtype'] == np.float64
    img, gf, gt = run64(cuda, c['face_vertices'], c['textures'], c['image_size'], c['kwargs'], c['grad_soft_colors'])
    assert np.abs(img - c['soft_colors']).max() <= 1e-9
    assert close(gf, c['grad_faces'], 1e-9) and close(gt, c['grad_textures'], 1e-9)


@pytest.mark.parametrize('dist,rgb,alpha', list(itertools.product(['hard', 'barycentric', 'euclidean'], ['hard', 'softmax'],
                                                                  ['hard', 'sum', 'prod'])))
def test_every_mode_combination_against_the_double_oracle(oracle, cuda, dist, rgb, alpha):
    fv, ft, near, far = synth.raster_batch(3, 3, count=2)
    kw = dict(synth.LASR_MODES, near=near, far=far, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha)
    IS = 40
    g = synth.upstream_grad(2, IS).astype(np.float64)
    fv64, ft64 = fv.astype(np.float64), ft.astype(np.float64)
    ref = oracle.forward(fv64, ft64, IS, dtype=np.float64, **kw)
    assert ref['soft_colors'].dtype == np.float64
    rgf, rgt = oracle.backward(ref, g, IS, dtype=np.float64, **kw)
    img, gf, gt = run64(cuda, fv, ft, IS, kw, g)
    assert np.abs(img - ref['soft_colors']).max() <= 1e-9
    assert close(gf, rgf, 1e-9) and close(gt, rgt, 1e-9)


@pytest.mark.parametrize('res,rgb', [(1, 'softmax'), (2, 'hard'), (3, 'softmax')])
def test_surface_textures_and_ragged_sizes_in_double(oracle, cuda, res, rgb):
    fv, _, near, far = synth.raster_batch(3, 3, count=1)
    tx = np.random.default_rng(res).uniform(0, 1, (1, fv.shape[1], res * res, 3))
    kw = dict(synth.LASR_MODES, near=near, far=far, texture_type='surface', aggr_func_rgb=rgb)
    IS = 33
    g = synth.upstream_grad(1, IS).astype(np.float64)
    ref = oracle.forward(fv.astype(np.float64), tx, IS, dtype=np.float64, **kw)
    rgf, rgt = oracle.backward(ref, g, IS, dtype=np.float64, **kw)
    img, gf, gt = run64(cuda, fv, tx, IS, kw, g)
    assert np.abs(img - ref['soft_colors']).max() <= 1e-9
    assert close(gf, rgf, 1e-9) and close(gt, rgt, 1e-9)


@pytest.mark.skipif(not sr_ref.available('sr_ref_nofma'), reason='oracle/_ref/sr_ref_nofma.so not in this snapshot (oracle/build_ref.py)')
@pytest.mark.parametrize('hard', [False, True])
def test_live_against_the_reference_build_in_double(cuda, hard):
    fv, ft, near, far = synth.raster_batch(11, 26, count=2)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    if hard:
        kw.update(dist_func='hard', aggr_func_rgb='hard', aggr_func_alpha='hard')
    IS = 128
    g = synth.upstream_grad(2, IS).astype(np.float64)
    tfv, tft = torch.from_numpy(fv.astype(np.float64)).to(cuda), torch.from_numpy(ft.astype(np.float64)).to(cuda)
    s = sr_ref.forward(tfv, tft, IS, variant='sr_ref_nofma', dtype=torch.float64, **kw)
    rgf, rgt = sr_ref.backward(s, torch.from_numpy(g).to(cuda), IS, variant='sr_ref_nofma', **kw)
    img, gf, gt = run64(cuda, fv, ft, IS, kw, g)
    assert np.abs(img - s['soft_colors'].cpu().numpy()).max() <= 1e-9
    assert close(gf, rgf.cpu().numpy(), 1e-9) and close(gt, rgt.cpu().numpy(), 1e-9)
    if hard:                                                 # the index map rides in aggrs_info: check it through the raw call
        from lasr_amd import _lib
        import math
        N, F = fv.shape[:2]
        info = torch.zeros(N, F, 27, dtype=torch.float64, device=cuda)
        aggr = torch.zeros(N, 2, IS, IS, dtype=torch.float64, device=cuda)
        col = torch.ones(N, 4, IS, IS, dtype=torch.float64, device=cuda)
        _lib.check(_lib.lib().lasr_sr_forward_f64(tfv.reshape(N, F, 9).contiguous().data_ptr(), tft.contiguous().data_ptr(),
                                                  info.data_ptr(), aggr.data_ptr(), col.data_ptr(), None, 0, N, F, 3, IS,
                                                  float(near), float(far), 1e-3, 1e-4, 0, float(math.log(1. / 1e-4 - 1.)), 1e-2,
                                                  0, 0, 1, 1, torch.cuda.current_stream(cuda).cuda_stream), 'forward_f64')
        assert torch.equal(aggr, s['aggrs_info']) and torch.equal(info, s['faces_info'])
