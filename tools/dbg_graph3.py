import sys, subprocess
if len(sys.argv) == 1:
    for w in ['lbs', 'grid', 'percept', 'codepred', 'misc']:
        r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True)
        print(w, 'rc', r.returncode, r.stdout.strip().split('\n')[-1][:100], flush=True)
    sys.exit(0)
sys.path.insert(0, '.')
import numpy as np, torch
import torch.nn.functional as F
from lasr_amd import synth
import lasr_amd.soft_renderer as sr
from lasr_amd.nnutils import geom_utils, image_losses, loss_utils, mesh_net
dev = torch.device('cuda:0'); w = sys.argv[1]
torch.manual_seed(0)
def P(*shape): return torch.randn(*shape, device=dev, requires_grad=True)
N, V, K, IS, H = 4, 162, 5, 64, 2
v, f, _ = synth.blobby_mesh(4)
faces = torch.from_numpy(f)
if w == 'encoder':
    enc = mesh_net.Encoder((64, 64), nz_feat=32).to(dev); x = torch.randn(2, 3, 64, 64, device=dev)
    for m in enc.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)): m.eval()
    fn = lambda: enc(x).sum()
    params = list(enc.parameters())
elif w == 'lbs':
    a, R, T = P(N, V, 3), P(N * K, 3, 3), P(N * K, 1, 3); s = torch.softmax(torch.randn(N, K - 1, V, 1, device=dev), 1).requires_grad_(True)
    fn = lambda: geom_utils.obj_to_cam(a, R, T, K, 1, s).sum(); params = [a, R, T, s]
elif w == 'pinhole':
    a = torch.rand(N, V, 4, device=dev) + 1; a.requires_grad_(True); pp, fl = P(2, 2), P(2, 2)
    fn = lambda: geom_utils.pinhole_cam(a, pp, fl).sum(); params = [a, pp, fl]
elif w in ('mask', 'flow', 'tex'):
    occ = torch.ones(2, IS, IS, device=dev); masks = (torch.rand(2, IS, IS, device=dev) > 0.5).float()
    if w == 'mask':
        a = P(2, H, IS, IS); fn = lambda: image_losses.mask_loss_table(a, masks, occ).sum(); params = [a]
    elif w == 'flow':
        a = P(2, H, IS, IS, 2); obs = torch.randn(2, 3, IS, IS, device=dev); bg = torch.rand(2, H, IS, IS, device=dev) > 0.5
        fn = lambda: image_losses.flow_loss_table(a, obs, bg, occ, masks)[0].sum(); params = [a]
    else:
        a, b = P(2, H, 3, IS, IS), P(2, H, IS, IS); o1, o2 = torch.rand(2, 3, IS, IS, device=dev), torch.rand(2, 3, IS, IS, device=dev)
        fn = lambda: image_losses.tex_loss_table(o1, o2, a, b, occ, 1.0).sum(); params = [a, b]
elif w in ('arap', 'lap', 'flatten'):
    x = torch.from_numpy(v)[None].repeat(3, 1, 1).to(dev).requires_grad_(True); dx = (x.detach() + 0.01).requires_grad_(True)
    if w == 'arap': L = loss_utils.ARAPLoss(torch.from_numpy(v), faces).to(dev); fn = lambda: L(dx, x).sum(); params = [x, dx]
    elif w == 'lap': L = loss_utils.LaplacianLoss(torch.from_numpy(v), faces).to(dev); fn = lambda: L(x).sum(); params = [x]
    else: L = loss_utils.FlattenLoss(faces).to(dev); fn = lambda: L(x).sum(); params = [x]
elif w == 'grid':
    img = torch.rand(2, 1, IS, IS, device=dev); g = P(2, 5, 1, 2)
    fn = lambda: F.grid_sample(img, g, padding_mode='border', align_corners=False).mean(); params = [g]
elif w == 'percept':
    pn = mesh_net.PerceptualDistance().to(dev); a = P(4, 3, IS, IS); b = torch.rand(4, 3, IS, IS, device=dev)
    fn = lambda: pn.forward_pair(a, b).sum(); params = [a]
elif w == 'codepred':
    cp = mesh_net.CodePredictor(nz_feat=32, n_bones=K, n_hypo=H).to(dev); feat = P(2, 32)
    fn = lambda: sum(t.sum() for t in cp(feat)); params = [feat]
elif w == 'misc':
    q = P(6, 4); a = P(3, 10, 3)
    def fn():
        R = mesh_net.quaternion_to_rotation_matrix(q)
        e = torch.eye(4, device=dev)[None, :, :, None]
        d = mesh_net.chamfer_distance(a, a * sr.functional.const_tensor([-1, 1, 1], dev))
        sm = (-10 * a.pow(2).sum(2)).softmax(1)
        return R.sum() + e.sum() + d + sm.sum() + mesh_net.geodesic_distance(R[:3], R[3:]).mean() + torch.where(a > 0, a, a.detach()).sum()
    params = [q, a]
elif w == 'render':
    r = sr.SoftRenderer(image_size=IS, sigma_val=1e-4, gamma_val=1e-2, camera_mode='look_at', perspective=False,
                        light_mode='vertex', light_intensity_ambient=1., light_intensity_directionals=0.)
    pv = torch.from_numpy(synth.frame_vertices(v, 3)).to(dev).requires_grad_(True); tx = P(3, V, 3)
    fc = faces.to(dev)[None].repeat(3, 1, 1)
    def fn():
        r.rasterizer.near = pv[:, :, 2].min().detach() - 1; r.rasterizer.far = pv[:, :, 2].max().detach() + 1
        return r.render_mesh(sr.Mesh(pv, fc, textures=tx, texture_type='vertex')).sum()
    params = [pv, tx]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        for p in params: p.grad = None
        fn().backward()
torch.cuda.current_stream().wait_stream(side)
for p in params: p.grad = None
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = fn(); out.backward()
gr.replay(); torch.cuda.synchronize()
print('ok', float(out))
