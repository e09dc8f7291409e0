"""The C oracle (fp64 instance) against an independently written numpy restatement (oracle/sr_numpy.py) that uses
plain Euclidean geometry instead of the reference's barycentric-space edge projection: all 18 mode combinations."""
import itertools

import numpy as np
import pytest

from lasr_amd import synth
from oracle import sr_numpy


@pytest.mark.parametrize('dist,rgb,alpha', list(itertools.product(
    ['hard', 'barycentric', 'euclidean'], ['hard', 'softmax'], ['hard', 'sum', 'prod'])))
def test_c_oracle_fp64_equals_numpy_restatement(oracle, dist, rgb, alpha):
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    kw = dict(synth.LASR_MODES, near=near, far=far, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha)
    a = oracle.forward(fv, ft, 40, dtype=np.float64, **kw)
    b = sr_numpy.forward(fv, ft, 40, **kw)
    assert np.abs(a['soft_colors'] - b['soft_colors']).max() < 1e-9
    if rgb == 'hard':
        assert np.array_equal(a['aggrs_info'][:, 1], b['aggrs_info'][:, 1])
    else:
        np.testing.assert_allclose(a['aggrs_info'], b['aggrs_info'], rtol=1e-9, atol=1e-12)


def test_single_sided_and_other_sigma(oracle):
    fv, ft, near, far = synth.raster_batch(2, 3, count=1)
    for fill_back, sigma in ((False, 1e-4), (True, 1e-5)):
        kw = dict(synth.LASR_MODES, near=near, far=far, fill_back=fill_back, sigma_val=sigma)
        a = oracle.forward(fv, ft, 32, dtype=np.float64, **kw)
        b = sr_numpy.forward(fv, ft, 32, **kw)
        assert np.abs(a['soft_colors'] - b['soft_colors']).max() < 1e-9
