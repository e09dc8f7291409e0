import sys, numpy as np, torch
sys.path.insert(0, '.')
from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf
from oracle import sr_oracle as o
dev = torch.device('cuda:0')
fv, _, near, far = synth.raster_batch(4, 3, count=2)
for res in (1, 2):
    rng = np.random.default_rng(5)
    ft = rng.uniform(0, 1, (fv.shape[0], fv.shape[1], res * res, 3)).astype(np.float32)
    kw = dict(synth.LASR_MODES, near=near, far=far, texture_type='surface', aggr_func_rgb='softmax')
    IS = 48
    ref = o.forward(fv, ft, IS, **kw)
    g = synth.upstream_grad(2, IS)
    gf_ref, gt_ref = o.backward(ref, g, IS, **kw)
    tfv = torch.from_numpy(fv).to(dev).requires_grad_(True); tft = torch.from_numpy(ft).to(dev).requires_grad_(True)
    img = srf.soft_rasterize(tfv, tft, IS, **kw); img.backward(torch.from_numpy(g).to(dev))
    gt = tft.grad.cpu().numpy()
    d = np.abs(gt - gt_ref)
    idx = np.unravel_index(d.argmax(), d.shape)
    print('res', res, 'max', d.max(), 'at', idx, gt[idx], gt_ref[idx], 'sum', gt.sum(), gt_ref.sum(), 'nbad', (d > 1e-6).sum(), d.size)
    bad = np.argwhere(d > 1e-6)[:10]
    for b in bad: print(tuple(b), gt[tuple(b)], gt_ref[tuple(b)])
