"""The HIP path against the reference as it would SHIP -- built with the compiler's default flags (FMA contraction on) --
outside the pixels where the reference's own fp32 arithmetic is ill-conditioned.

The parity target of every other test is the un-contracted evaluation of the reference (`sr_ref_nofma`, and the C oracle that
restates it).  The default-flags build (`sr_ref.so`: what the reference's setup.py gives under nvcc or hipcc) rounds the
edge-projection arithmetic of K.cu:62-151 differently, and on edge-on / sliver faces that arithmetic is ill-conditioned (SURVEY
App. D): the two builds of the SAME source differ from each other by up to ~1 on ~1 % of the pixels.  So "within 1e-4 of the
reference" can only hold where the reference agrees with itself.  This test states where that is:

  mask = pixels where EITHER fp32 build of the reference differs from the SAME kernels run in fp64
         (AT_DISPATCH_FLOATING_TYPES, K.cu:701) by more than MASK_TOL = 3e-5 in any channel -- the reference's own
         precision-limited pixels, found without looking at the HIP output;
  bar  = HIP vs the default-flags build: image max-abs <= 1e-4 on every pixel OUTSIDE the mask (with HIP within 1e-6 of the
         un-contracted build this follows from the triangle inequality: what the test establishes is that the HIP path sits on
         one of the reference's own roundings, and HOW MANY pixels the statement has to exclude).
  gradients: the default-flags build's backward is not a usable yardstick as a whole -- on the mask's pixels its coefficients
         overflow, and zeroing the upstream gradient there does not help (0 x inf = NaN poisons the face's sum, K.cu:657-666) --
         so the comparison runs over the gradient entries that build leaves FINITE: at least 99.9 % of them must lie within
         GRAD_TOL of the HIP gradient's largest entry; the number of non-finite entries is recorded.
The JSON line also sweeps the PREDICTIVE form of the mask (un-contracted fp32 vs fp64 only, not looking at the default-flags
build): at 1e-5 it excludes 4 % of the pixels and still leaves ~100 pixels of 1 M beyond 1e-4 -- one fp32 evaluation landing
close to fp64 does not mean the pixel is well conditioned.

Mask sizes and residuals are written to gpurun_out/default_flags_parity.json (DESIGN section 2 quotes them).
Each build runs in its own child process (tests/ref_build_worker.py): they cannot share one.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf
from oracle import sr_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK_TOL = 3e-5
GRAD_TOL = 1e-2
needs_both = pytest.mark.skipif(not (sr_ref.available('sr_ref') and sr_ref.available('sr_ref_nofma')),
                                reason='oracle/_ref/sr_ref{,_nofma}.so not in this snapshot (oracle/build_ref.py)')


def run_worker(variant, tmp, tag, **arrays):
    src, dst = os.path.join(tmp, tag + '_in.npz'), os.path.join(tmp, tag + '_out.npz')
    np.savez(src, **arrays)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tests', 'ref_build_worker.py'), variant, src, dst], cwd=ROOT)
    with np.load(dst) as z:
        return {k: z[k] for k in z.files}


def record(entry):
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'default_flags_parity.json'), 'a') as f:
            f.write(json.dumps(entry) + '\n')
    except OSError:
        pass
    print(json.dumps(entry))


@needs_both
@pytest.mark.parametrize('nu,n_frames,count,IS', [(8, 16, 16, 256), (11, 26, 26, 256), (11, 26, 8, 512)],
                         ids=['M1_256', 'M2_256', 'M2_512'])
def test_within_1e4_of_the_default_flags_build_outside_the_ill_conditioned_pixels(cuda, tmp_path, nu, n_frames, count, IS):
    fv, ft, near, far = synth.raster_batch(nu, n_frames, count=count)
    kw = dict(synth.LASR_MODES, near=near, far=far)
    kwj = np.frombuffer(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}).encode(), np.uint8)
    g = synth.upstream_grad(count, IS)

    # the reference against itself: fp32 vs fp64 of the un-contracted build -> the mask
    a = run_worker('sr_ref_nofma', str(tmp_path), 'nofma', face_vertices=fv, textures=ft, image_size=IS, kwargs=kwj, fp64=True)
    dev64 = np.abs(a['soft_colors'].astype(np.float64) - a['soft_colors_fp64']).max(1)        # [N, IS, IS]
    # the reference as shipped (compiler defaults) against ITS fp64 run
    b = run_worker('sr_ref', str(tmp_path), 'fma', face_vertices=fv, textures=ft, image_size=IS, kwargs=kwj, fp64=True)
    dev64_fma = np.abs(np.nan_to_num(b['soft_colors']).astype(np.float64) - b['soft_colors_fp64']).max(1)
    mask = (dev64 > MASK_TOL) | (dev64_fma > MASK_TOL)
    covered = a['soft_colors_fp64'][:, 3] > 1e-3
    gm = g * (~mask)[:, None]
    # its backward with the upstream gradient zeroed on the mask
    b.update(run_worker('sr_ref', str(tmp_path), 'fma_bwd', face_vertices=fv, textures=ft, image_size=IS, kwargs=kwj,
                        grad_soft_colors=gm.astype(np.float32)))

    tfv = torch.from_numpy(fv).to(cuda).requires_grad_(True)
    tft = torch.from_numpy(ft).to(cuda).requires_grad_(True)
    img = srf.soft_rasterize(tfv, tft, IS, **kw)
    img.backward(torch.from_numpy(gm.astype(np.float32)).to(cuda))
    mine = img.detach().cpu().numpy()

    d_fma = np.abs(mine - b['soft_colors']).max(1)
    d_nofma = np.abs(mine - a['soft_colors']).max(1)
    d_builds = np.abs(np.nan_to_num(b['soft_colors']) - a['soft_colors']).max(1)
    entry = dict(case='nu%d_%dx%d_%dframes' % (nu, IS, IS, count), mask_tol=MASK_TOL,
                 mask_pixels=int(mask.sum()), mask_pct_of_pixels=100. * float(mask.mean()),
                 mask_pct_of_covered_pixels=100. * float((mask & covered).sum()) / max(int(covered.sum()), 1),
                 reference_fp32_vs_fp64_max=float(dev64.max()), default_build_fp32_vs_fp64_max=float(dev64_fma.max()),
                 fp64_runs_of_the_two_builds_max=float(np.abs(a['soft_colors_fp64'] - b['soft_colors_fp64']).max()),
                 default_vs_uncontracted_build_max=float(d_builds.max()),
                 default_vs_uncontracted_build_px_over_1e4=int((d_builds > 1e-4).sum()),
                 default_vs_uncontracted_px_over_1e4_outside_mask=int(((d_builds > 1e-4) & ~mask).sum()),
                 hip_vs_uncontracted_max=float(d_nofma.max()),
                 hip_vs_default_max_all_pixels=float(d_fma.max()),
                 hip_vs_default_px_over_1e4_all_pixels=int((d_fma > 1e-4).sum()),
                 hip_vs_default_max_outside_mask=float(d_fma[~mask].max()),
                 hip_vs_default_px_over_1e4_outside_mask=int((d_fma[~mask] > 1e-4).sum()))
    for tol in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4):
        m = dev64 > tol
        entry['predictive_mask_tol_%g' % tol] = dict(mask_pct=100. * float(m.mean()), hip_vs_default_max_outside=float(d_fma[~m].max()),
                                           px_over_1e4_outside=int((d_fma[~m] > 1e-4).sum()))
    for name, minegrad, theirs in (('grad_faces', tfv.grad, b['grad_faces']), ('grad_textures', tft.grad, b['grad_textures'])):
        mg = minegrad.cpu().numpy().reshape(theirs.shape).astype(np.float64)
        fin = np.isfinite(theirs)
        scale = max(float(np.abs(mg).max()), 1e-30)
        d = np.abs(mg[fin] - theirs[fin].astype(np.float64)) / scale
        entry[name + '_default_build_nonfinite_entries'] = int((~fin).sum())
        entry[name + '_finite_entries_within_tol'] = float((d <= GRAD_TOL).mean()) if d.size else 1.0
        entry[name + '_finite_entries_median_rel'] = float(np.median(d)) if d.size else 0.0
        entry[name + '_finite_entries_max_rel'] = float(d.max()) if d.size else 0.0
    record(entry)

    assert entry['hip_vs_uncontracted_max'] <= 1e-6
    assert entry['hip_vs_default_max_outside_mask'] <= 1e-4, entry
    assert entry['mask_pct_of_pixels'] < 5.0, entry
    assert entry['grad_faces_finite_entries_within_tol'] >= 0.999 and entry['grad_textures_finite_entries_within_tol'] >= 0.999, entry
