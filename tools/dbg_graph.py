import sys, faulthandler; faulthandler.enable()
sys.path.insert(0, '.')
import numpy as np, torch
from lasr_amd import synth
from lasr_amd.soft_renderer import functional as srf
from lasr_amd.nnutils import geom_utils, image_losses
dev = torch.device('cuda:0')
fv, ft, near, far = synth.raster_batch(4, 3, count=2)
kw = dict(synth.LASR_MODES, near=torch.tensor(near, device=dev), far=torch.tensor(far, device=dev))
tfv = torch.from_numpy(fv).to(dev).requires_grad_(True); tft = torch.from_numpy(ft).to(dev).requires_grad_(True)
def step():
    img = srf.soft_rasterize(tfv, tft, 64, **kw)
    img.sum().backward()
    return img
which = sys.argv[1]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tfv.grad = None; tft.grad = None
        if which == 'raster': step()
torch.cuda.current_stream().wait_stream(s)
tfv.grad = None; tft.grad = None
g = torch.cuda.CUDAGraph()
print('capturing', which, flush=True)
with torch.cuda.graph(g):
    if which == 'raster': out = step()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('replayed', float(out.sum()), float(tfv.grad.abs().sum()), flush=True)
